/*
 * oracle_impl.h -- body of the CPU oracle, compiled twice by oracle.c:
 * REAL = float (FN(x) = x_f32, the arithmetic the device path uses) and
 * REAL = double (FN(x) = x_f64, the accuracy yardstick).
 *
 * TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: the reference (TensorFlow 1.14 +
 * TFP 0.7.0) cannot be imported or run in the build container and ships no
 * golden vectors for this path, so this restatement is
 * pinned against analytic answers and a float64 autograd restatement of the
 * reference's model programs (oracle/ed2_ref.py), not against reference output.
 *
 * Plain loops over the flattened state, one chain at a time; nothing here knows
 * about lanes except the RNG stream partition (`lanes`), which is part of the
 * sampler's specification ("Randomness").
 */

/* ------------------------------------------------------------------------
 * log joint + gradient, additive constants dropped.
 * x is the state in the (a,b)-parameterised coordinates of
 * program_transformations.py:555-600:  xt ~ N(a mu, sigma^b),
 * x = mu + sigma^(1-b) (xt - a mu).
 * ---------------------------------------------------------------------- */

/* radon, reference models.py:826-837 (sigma_y = 1, models.py:839).
 * Parts in trace order: mua, b1, b2, m[J]. */
static REAL FN(radon_logp_grad)(const orc_model* M, const float* a, const float* b,
                                const REAL* x, REAL* g) {
  (void)b; /* every latent scale is 1: sigma^b = 1 */
  const int J = M->J;
  const REAL mua = x[0], b1 = x[1], b2 = x[2];
  REAL lp = -(REAL)0.5 * (mua * mua + b1 * b1 + b2 * b2);
  REAL g_mua = -mua, g_b1 = -b1, g_b2 = -b2;
  /* likelihood terms that do not involve a county effect */
  lp += b2 * ((REAL)M->sxy - (REAL)0.5 * b2 * (REAL)M->sxx);
  g_b2 += (REAL)M->sxy - b2 * (REAL)M->sxx;
  for (int j = 0; j < J; ++j) {
    const REAL aj = a[3 + j];
    const REAL mu = mua + (REAL)M->u[j] * b1;
    const REAL r = x[3 + j] - aj * mu;          /* xt - a mu */
    const REAL m = r + mu;                      /* centred county effect */
    const REAL t = (REAL)M->sy[j] - b2 * (REAL)M->sx[j];
    const REAL l = t - (REAL)M->n[j] * m;       /* d loglik / d m_j */
    lp += -(REAL)0.5 * r * r - (REAL)0.5 * m * ((REAL)M->n[j] * m - 2 * t);
    const REAL gm = l - r;
    g[3 + j] = gm;
    const REAL h = l - aj * gm;                 /* d / d mu_j */
    g_mua += h;
    g_b1 += (REAL)M->u[j] * h;
    g_b2 -= m * (REAL)M->sx[j];
  }
  g[0] = g_mua; g[1] = g_b1; g[2] = g_b2;
  return lp;
}

static void FN(radon_to_centered)(const orc_model* M, const float* a, const float* b,
                                  const REAL* x, REAL* out) {
  (void)b;
  out[0] = x[0]; out[1] = x[1]; out[2] = x[2];
  for (int j = 0; j < M->J; ++j) {
    REAL mu = x[0] + (REAL)M->u[j] * x[1];
    out[3 + j] = x[3 + j] + ((REAL)1 - (REAL)a[3 + j]) * mu;
  }
}
static void FN(radon_from_centered)(const orc_model* M, const float* a, const float* b,
                                    const REAL* x, REAL* out) {
  (void)b;
  out[0] = x[0]; out[1] = x[1]; out[2] = x[2];
  for (int j = 0; j < M->J; ++j) {
    REAL mu = x[0] + (REAL)M->u[j] * x[1];
    out[3 + j] = x[3 + j] - ((REAL)1 - (REAL)a[3 + j]) * mu;
  }
}


/* A group of Normal latents with a shared location parent `loc` and a shared
 * log-scale parent `ls` (8 schools' theta, election's a):
 *   xt_k ~ N(a_k loc, exp(b_k ls)),  x_k = loc + exp(ls) z_k,  z_k = (xt_k - a_k loc) exp(-b_k ls).
 * Given w_k = d loglik / d x_k it adds the prior + chain-rule terms:
 *   d/dxt_k = e_k (sigma w_k - z_k),  d/dloc += w_k - a_k d/dxt_k,
 *   d/dls  += b_k z_k^2 - b_k + w_k sigma z_k (1 - b_k),  logp += -z_k^2/2 - b_k ls.   */

/* eight schools, reference models.py:139-147.  Parts: mu, log_tau, theta[8]. */
static REAL FN(schools_logp_grad)(const orc_model* M, const float* a, const float* b,
                                  const REAL* x, REAL* g) {
  const REAL s0 = (REAL)pow(5.0, (double)b[0]), s1 = (REAL)pow(5.0, (double)b[1]);
  const REAL c0 = 5 / s0, c1 = 5 / s1;
  const REAL mu = c0 * x[0], lt = c1 * x[1];
  const REAL tau = (REAL)exp((double)lt);
  REAL lp = -(REAL)0.5 * (x[0] / s0) * (x[0] / s0) - (REAL)0.5 * (x[1] / s1) * (x[1] / s1);
  REAL g_mu = 0, g_lt = 0;
  for (int k = 0; k < 8; ++k) {
    const REAL ak = a[2 + k], bk = b[2 + k];
    const REAL e = (REAL)exp((double)(-bk * lt));
    const REAL z = (x[2 + k] - ak * mu) * e;
    const REAL th = mu + tau * z;
    const REAL sg = M->u[k];                       /* treatment stddev */
    const REAL w = ((REAL)M->y[k] - th) / (sg * sg);
    lp += -(REAL)0.5 * z * z - bk * lt - (REAL)0.5 * ((REAL)M->y[k] - th) * ((REAL)M->y[k] - th) / (sg * sg);
    const REAL gk = e * (tau * w - z);
    g[2 + k] = gk;
    g_mu += w - ak * gk;
    g_lt += bk * z * z - bk + w * tau * z * (1 - bk);
  }
  g[0] = -x[0] / (s0 * s0) + c0 * g_mu;
  g[1] = -x[1] / (s1 * s1) + c1 * g_lt;
  return lp;
}
static void FN(schools_convert)(const orc_model* M, const float* a, const float* b, const REAL* x,
                                REAL* out, int to_centered) {
  (void)M;
  const REAL c0 = (REAL)pow(5.0, 1.0 - (double)b[0]), c1 = (REAL)pow(5.0, 1.0 - (double)b[1]);
  if (to_centered) {
    const REAL mu = c0 * x[0], lt = c1 * x[1];
    out[0] = mu; out[1] = lt;
    for (int k = 0; k < 8; ++k)
      out[2 + k] = mu + (REAL)exp((double)((1 - b[2 + k]) * lt)) * (x[2 + k] - a[2 + k] * mu);
  } else {
    const REAL mu = x[0], lt = x[1];
    out[0] = mu / c0; out[1] = lt / c1;
    for (int k = 0; k < 8; ++k)
      out[2 + k] = a[2 + k] * mu + (x[2 + k] - mu) * (REAL)exp((double)(-(1 - b[2 + k]) * lt));
  }
}

/* election, reference models.py:969-982.  Parts: mua, log_sigma_a, a[S], b1, b2.
 * Likelihood collapsed to (state, female, black) cells: cell (t, f, k) holds
 * n = #observations and y = #ones; t is the reference's 1-based state index used
 * as a 0-based one-hot column, so t = S has no `a` term (tf.one_hot zero row). */
static REAL FN(election_logp_grad)(const orc_model* M, const float* a, const float* b,
                                   const REAL* x, REAL* g) {
  const int S = M->S;
  const int iB1 = 2 + S, iB2 = 3 + S;
  const REAL s0 = (REAL)pow(100.0, (double)b[0]), s1 = (REAL)pow(10.0, (double)b[1]);
  const REAL sb1 = (REAL)pow(100.0, (double)b[iB1]), sb2 = (REAL)pow(100.0, (double)b[iB2]);
  const REAL c0 = 100 / s0, c1 = 10 / s1, cb1 = 100 / sb1, cb2 = 100 / sb2;
  const REAL mua = c0 * x[0], ls = c1 * x[1], b1 = cb1 * x[iB1], b2 = cb2 * x[iB2];
  const REAL sig = (REAL)exp((double)ls);
  /* the log density is a sum of ~260 terms of size 10 .. 10^4 whose DIFFERENCES between states decide the Metropolis
   * test: the terms are formed in REAL, the running sum is kept in double (a float running sum of a value ~10^4 .. 10^5
   * is off by 5e-3 .. 7e-2, several times the device path's error, which sums per lane and per state pair) */
  double lp = -0.5 * (double)((x[0] / s0) * (x[0] / s0) + (x[1] / s1) * (x[1] / s1) +
                              (x[iB1] / sb1) * (x[iB1] / sb1) + (x[iB2] / sb2) * (x[iB2] / sb2));
  REAL g_mua = 0, g_ls = 0, g_b1 = 0, g_b2 = 0;
  for (int t = 0; t <= S; ++t) {               /* t == S: observations without a state effect */
    REAL as = 0, z = 0, e = 0, at = 0, bt = 0;
    if (t < S) {
      at = a[2 + t]; bt = b[2 + t];
      e = (REAL)exp((double)(-bt * ls));
      z = (x[2 + t] - at * mua) * e;
      as = mua + sig * z;
    }
    REAL W = 0;
    for (int fk = 0; fk < 4; ++fk) {
      const REAL n = M->cell_n[t * 4 + fk], y = M->cell_y[t * 4 + fk];
      if (n == 0) continue;
      const REAL f = (REAL)(fk & 1), k = (REAL)(fk >> 1);
      const REAL eta = as + f * b2 + k * b1;
      const REAL sp = (eta > 0 ? eta : 0) + (REAL)log1p(exp(-fabs((double)eta)));
      const REAL sgm = (REAL)(1.0 / (1.0 + exp(-(double)eta)));
      lp += (double)(y * eta - n * sp);
      const REAL w = y - n * sgm;
      W += w; g_b2 += f * w; g_b1 += k * w;
    }
    if (t < S) {
      lp += (double)(-(REAL)0.5 * z * z - bt * ls);
      const REAL gt = e * (sig * W - z);
      g[2 + t] = gt;
      g_mua += W - at * gt;
      g_ls += bt * z * z - bt + W * sig * z * (1 - bt);
    }
  }
  g[0] = -x[0] / (s0 * s0) + c0 * g_mua;
  g[1] = -x[1] / (s1 * s1) + c1 * g_ls;
  g[iB1] = -x[iB1] / (sb1 * sb1) + cb1 * g_b1;
  g[iB2] = -x[iB2] / (sb2 * sb2) + cb2 * g_b2;
  return (REAL)lp;
}
static void FN(election_convert)(const orc_model* M, const float* a, const float* b, const REAL* x,
                                 REAL* out, int to_centered) {
  const int S = M->S, iB1 = 2 + S, iB2 = 3 + S;
  const REAL c0 = (REAL)pow(100.0, 1.0 - (double)b[0]), c1 = (REAL)pow(10.0, 1.0 - (double)b[1]);
  const REAL cb1 = (REAL)pow(100.0, 1.0 - (double)b[iB1]), cb2 = (REAL)pow(100.0, 1.0 - (double)b[iB2]);
  if (to_centered) {
    const REAL mua = c0 * x[0], ls = c1 * x[1];
    out[0] = mua; out[1] = ls; out[iB1] = cb1 * x[iB1]; out[iB2] = cb2 * x[iB2];
    for (int t = 0; t < S; ++t)
      out[2 + t] = mua + (REAL)exp((double)((1 - b[2 + t]) * ls)) * (x[2 + t] - a[2 + t] * mua);
  } else {
    const REAL mua = x[0], ls = x[1];
    out[0] = mua / c0; out[1] = ls / c1; out[iB1] = x[iB1] / cb1; out[iB2] = x[iB2] / cb2;
    for (int t = 0; t < S; ++t)
      out[2 + t] = a[2 + t] * mua + (x[2 + t] - mua) * (REAL)exp((double)(-(1 - b[2 + t]) * ls));
  }
}

/* german_credit_lognormalcentered, reference models.py:888-904.
 * Parts: overall_log_scale, beta_log_scales[F], beta[F]; logits = X beta. */
static REAL FN(german_logp_grad)(const orc_model* M, const float* a, const float* b,
                                 const REAL* x, REAL* g) {
  const int F = M->F, N = M->N;
  const REAL s0 = (REAL)pow(10.0, (double)b[0]), c0 = 10 / s0;
  const REAL ols = c0 * x[0];
  REAL lp = -(REAL)0.5 * (x[0] / s0) * (x[0] / s0);
  REAL* beta = (REAL*)malloc(sizeof(REAL) * 2 * (size_t)F);
  REAL* v = beta + F;
  for (int d = 0; d < F; ++d) {
    const REAL ad = a[1 + d];
    const REAL r = x[1 + d] - ad * ols;
    const REAL bls = r + ols;
    const REAL bd = b[1 + F + d];
    beta[d] = (REAL)exp((double)((1 - bd) * bls)) * x[1 + F + d];
    v[d] = 0;
    lp += -(REAL)0.5 * r * r;
  }
  for (int n = 0; n < N; ++n) {
    const float* row = M->X + (size_t)n * F;
    REAL eta = 0;
    for (int d = 0; d < F; ++d) eta += (REAL)row[d] * beta[d];
    const REAL sp = (eta > 0 ? eta : 0) + (REAL)log1p(exp(-fabs((double)eta)));
    const REAL sgm = (REAL)(1.0 / (1.0 + exp(-(double)eta)));
    const REAL yn = M->y[n];
    lp += yn * eta - sp;
    const REAL w = yn - sgm;
    for (int d = 0; d < F; ++d) v[d] += (REAL)row[d] * w;
  }
  REAL g_ols = 0;
  for (int d = 0; d < F; ++d) {
    const REAL ad = a[1 + d], bd = b[1 + F + d];
    const REAL r = x[1 + d] - ad * ols;
    const REAL bls = r + ols;
    const REAL e = (REAL)exp((double)(-bd * bls));
    const REAL zb = x[1 + F + d] * e;
    lp += -(REAL)0.5 * zb * zb - bd * bls;
    g[1 + F + d] = -zb * e + v[d] * (REAL)exp((double)((1 - bd) * bls));
    const REAL hb = bd * zb * zb - bd + v[d] * (1 - bd) * beta[d];
    g[1 + d] = hb - r;
    g_ols += ad * r + (1 - ad) * hb;
  }
  g[0] = -x[0] / (s0 * s0) + c0 * g_ols;
  free(beta);
  return lp;
}
static void FN(german_convert)(const orc_model* M, const float* a, const float* b, const REAL* x,
                               REAL* out, int to_centered) {
  const int F = M->F;
  const REAL c0 = (REAL)pow(10.0, 1.0 - (double)b[0]);
  if (to_centered) {
    const REAL ols = c0 * x[0];
    out[0] = ols;
    for (int d = 0; d < F; ++d) {
      const REAL bls = x[1 + d] + (1 - a[1 + d]) * ols;
      out[1 + d] = bls;
      out[1 + F + d] = (REAL)exp((double)((1 - b[1 + F + d]) * bls)) * x[1 + F + d];
    }
  } else {
    const REAL ols = x[0];
    out[0] = ols / c0;
    for (int d = 0; d < F; ++d) {
      out[1 + d] = x[1 + d] - (1 - a[1 + d]) * ols;
      out[1 + F + d] = x[1 + F + d] * (REAL)exp((double)(-(1 - b[1 + F + d]) * x[1 + d]));
    }
  }
}

/* radon_stddvs, reference models.py:763-806.  Parts: mua, b1, b2, m[J], log_m_stddv[J]. */
static REAL FN(radon_sd_logp_grad)(const orc_model* M, const float* a, const float* b,
                                   const REAL* x, REAL* g) {
  (void)b;
  const int J = M->J;
  const REAL mua = x[0], b1 = x[1], b2 = x[2];
  REAL lp = -(REAL)0.5 * (mua * mua + b1 * b1 + b2 * b2);
  REAL g_mua = -mua, g_b1 = -b1, g_b2 = -b2;
  for (int j = 0; j < J; ++j) {
    const REAL aj = a[3 + j], s = x[3 + J + j];
    const REAL mu = mua + (REAL)M->u[j] * b1;
    const REAL r = x[3 + j] - aj * mu;
    const REAL m = r + mu;
    const REAL w = (REAL)exp(-2.0 * (double)s);
    /* sum over the county of (y - m - b2 x)^2 and of (y - m - b2 x) */
    const REAL resid = (REAL)M->sy[j] - b2 * (REAL)M->sx[j] - (REAL)M->n[j] * m;
    const REAL Q = (REAL)M->syy_j[j] - 2 * m * (REAL)M->sy[j] - 2 * b2 * (REAL)M->sxy_j[j] + (REAL)M->n[j] * m * m +
                   2 * m * b2 * (REAL)M->sx[j] + b2 * b2 * (REAL)M->sxx_j[j];
    lp += -(REAL)0.5 * r * r - (REAL)0.5 * s * s - (REAL)M->n[j] * s - (REAL)0.5 * w * Q;
    const REAL l = w * resid;
    const REAL gm = l - r;
    g[3 + j] = gm;
    g[3 + J + j] = -s - (REAL)M->n[j] + w * Q;
    const REAL h = l - aj * gm;
    g_mua += h; g_b1 += (REAL)M->u[j] * h;
    g_b2 += w * ((REAL)M->sxy_j[j] - m * (REAL)M->sx[j] - b2 * (REAL)M->sxx_j[j]);
  }
  g[0] = g_mua; g[1] = g_b1; g[2] = g_b2;
  return lp;
}
static void FN(radon_sd_convert)(const orc_model* M, const float* a, const float* b, const REAL* x,
                                 REAL* out, int to_centered) {
  (void)b;
  const int J = M->J;
  out[0] = x[0]; out[1] = x[1]; out[2] = x[2];
  for (int j = 0; j < J; ++j) {
    REAL mu = x[0] + (REAL)M->u[j] * x[1];
    REAL d = ((REAL)1 - (REAL)a[3 + j]) * mu;
    out[3 + j] = to_centered ? x[3 + j] + d : x[3 + j] - d;
    out[3 + J + j] = x[3 + J + j];
  }
}

/* Neal's funnel, reference models.py:674-677.  Parts: x1, x2. */
static REAL FN(funnel_logp_grad)(const orc_model* M, const float* a, const float* b, const REAL* x, REAL* g) {
  (void)M; (void)a;
  const REAL s1 = (REAL)pow(3.0, (double)b[0]), c1 = 3 / s1, b2 = b[1];
  const REAL x1 = c1 * x[0];
  const REAL e = (REAL)exp((double)(-(REAL)0.5 * b2 * x1));
  const REAL z = x[1] * e, u1 = x[0] / s1;
  g[1] = -z * e;
  g[0] = -u1 / s1 + c1 * (REAL)0.5 * b2 * (z * z - 1);
  return -(REAL)0.5 * u1 * u1 - (REAL)0.5 * z * z - (REAL)0.5 * b2 * x1;
}
static void FN(funnel_convert)(const orc_model* M, const float* a, const float* b, const REAL* x, REAL* out, int to_centered) {
  (void)M; (void)a;
  const REAL c1 = (REAL)pow(3.0, 1.0 - (double)b[0]);
  if (to_centered) { out[0] = c1 * x[0]; out[1] = x[1] * (REAL)exp(0.5 * (1.0 - (double)b[1]) * (double)out[0]); }
  else { out[0] = x[0] / c1; out[1] = x[1] * (REAL)exp(-0.5 * (1.0 - (double)b[1]) * (double)x[0]); }
}

/* electric company, reference models.py:1013-1035.  Parts: mua[G], sigma_y[G], a[P], b[G].
 *   mua_k ~ N(0,1), sigma_y_k ~ N(0,1)   (unit scale: every (a,b) is the identity)
 *   a_j ~ N(mu_j, 1), mu_j = 100 mua[grade_pair_j]:  at_j ~ N(al_j mu_j, 1), a_j = at_j + (1 - al_j) mu_j
 *   b_k ~ N(0,100):  bt_k ~ N(0, 100^be_k), b_k = 100^(1-be_k) bt_k
 *   y_i ~ N(a[pair_i] + b[grade_i] treatment_i, exp(sigma_y[grade_i]))
 * All three index vectors are 1-based in the data and used as 0-based one-hot columns, so index
 * n hits an all-zero row: pair 96 has no pair effect, grade 4 has b = 0 and scale exp(0) = 1,
 * grade_pair 4 gives mu_j = 0; a[0], mua[0], sigma_y[0] and b[0] are informed by the prior only. */
static REAL FN(electric_logp_grad)(const orc_model* M, const float* a, const float* b,
                                   const REAL* x, REAL* g) {
  const int P = M->P, G = M->G, N = M->N;
  const int iS = G, iA = 2 * G, iB = 2 * G + P;
  REAL mu[P], dA[P], bb[4], cb[4], sb[4], dB[4] = {0, 0, 0, 0}, dS[4] = {0, 0, 0, 0}, dM[4] = {0, 0, 0, 0};
  REAL lp = 0;
  for (int k = 0; k < G; ++k) {
    sb[k] = (REAL)pow(100.0, (double)b[iB + k]);
    cb[k] = 100 / sb[k];
    bb[k] = cb[k] * x[iB + k];
    lp += -(REAL)0.5 * (x[k] * x[k] + x[iS + k] * x[iS + k] + (x[iB + k] / sb[k]) * (x[iB + k] / sb[k]));
  }
  for (int j = 0; j < P; ++j) {
    const int k = M->grade_pair[j];
    mu[j] = (k >= 0 && k < G) ? 100 * x[k] : 0;
    dA[j] = 0;
  }
  for (int i = 0; i < N; ++i) {
    const int j = M->pair[i], k = M->grade[i];
    const int hasj = j >= 0 && j < P, hask = k >= 0 && k < G;
    const REAL aj = hasj ? mu[j] + x[iA + j] - (REAL)a[iA + j] * mu[j] : 0;
    const REAL bk = hask ? bb[k] : 0, sk = hask ? x[iS + k] : 0;
    const REAL t = M->treat[i];
    const REAL w = (REAL)exp(-2.0 * (double)sk);
    const REAL r = (REAL)M->y[i] - aj - bk * t;
    lp += -sk - (REAL)0.5 * w * r * r;
    if (hasj) dA[j] += w * r;
    if (hask) { dB[k] += t * w * r; dS[k] += w * r * r - 1; }
  }
  for (int j = 0; j < P; ++j) {
    const REAL al = a[iA + j];
    const REAL r = x[iA + j] - al * mu[j];
    lp += -(REAL)0.5 * r * r;
    const REAL ga = dA[j] - r;
    g[iA + j] = ga;
    const int k = M->grade_pair[j];
    if (k >= 0 && k < G) dM[k] += 100 * (dA[j] - al * ga);
  }
  for (int k = 0; k < G; ++k) {
    g[k] = -x[k] + dM[k];
    g[iS + k] = -x[iS + k] + dS[k];
    g[iB + k] = -x[iB + k] / (sb[k] * sb[k]) + cb[k] * dB[k];
  }
  return lp;
}
static void FN(electric_convert)(const orc_model* M, const float* a, const float* b, const REAL* x,
                                 REAL* out, int to_centered) {
  const int P = M->P, G = M->G, iA = 2 * G, iB = 2 * G + P;
  for (int k = 0; k < 2 * G; ++k) out[k] = x[k];
  for (int k = 0; k < G; ++k) {
    const REAL c = (REAL)pow(100.0, 1.0 - (double)b[iB + k]);
    out[iB + k] = to_centered ? c * x[iB + k] : x[iB + k] / c;
  }
  for (int j = 0; j < P; ++j) {
    const int k = M->grade_pair[j];
    const REAL mu = (k >= 0 && k < G) ? 100 * x[k] : 0;
    const REAL sh = (1 - (REAL)a[iA + j]) * mu;
    out[iA + j] = to_centered ? x[iA + j] + sh : x[iA + j] - sh;
  }
}

/* local linear trend ("time_series"), reference models.py:1071-1094.  x = [sa, sm, (at_t, mt_t) t < T, beta].
 *   sa, sm, beta ~ N(0,1);  Sa = softplus(sa), Sm = softplus(sm)
 *   alpha_t ~ N(m_t, Sa), m_t = alpha_{t-1} + mu_{t-1} (m_0 = 0):  at_t ~ N(a m_t, Sa^b), alpha_t = m_t + Sa^(1-b) (at_t - a m_t)
 *   mu_t ~ N(mu_{t-1}, Sm) (mu_{-1} = 0):                          mt_t ~ N(a' mu_{t-1}, Sm^b'), likewise
 *   y_t ~ N(alpha_t + beta x_t, 0.12)
 * Forward recurrence in t, then the adjoint recurrence backwards: with e_t = d loglik / d alpha_t,
 * z_t = (at_t - a m_t) / Sa^b and the messages G_t = d logp / d m_t, H_t = d logp / d mu_{t-1} (through mu_t's prior),
 *   Abar_t = e_t + G_{t+1},  Mbar_t = G_{t+1} + H_{t+1},
 *   G_t = (1 - a c) Abar_t + a z / Sa^b,  H_t = (1 - a' c') Mbar_t + a' z' / Sm^b',  c = Sa^(1-b), c' = Sm^(1-b'). */
static REAL FN(time_series_run)(const orc_model* M, const float* a, const float* b, const REAL* x, REAL* g,
                                REAL* centred, REAL* mloc) {
  const int T = M->N, iB = 2 + 2 * T;
  const REAL s2i = (REAL)(1.0 / (0.12 * 0.12));
  const REAL sa = x[0], sm = x[1], beta = x[iB];
  const REAL Sa = (REAL)log1p(exp((double)sa)), Sm = (REAL)log1p(exp((double)sm));
  const REAL lSa = (REAL)log((double)Sa), lSm = (REAL)log((double)Sm);
  REAL al[T], mu[T], zA[T], zM[T], e[T];
  REAL lp = -(REAL)0.5 * (sa * sa + sm * sm + beta * beta);
  REAL ap = 0, mp = 0;
  for (int t = 0; t < T; ++t) {
    const int iA = 2 + 2 * t, iM = 3 + 2 * t;
    const REAL mA = ap + mp, mM = mp;                 /* locations (centred parents) */
    const REAL cA = (REAL)exp((double)((1 - b[iA]) * lSa)), cM = (REAL)exp((double)((1 - b[iM]) * lSm));
    const REAL rA = x[iA] - a[iA] * mA, rM = x[iM] - a[iM] * mM;
    al[t] = mA + cA * rA; mu[t] = mM + cM * rM;
    zA[t] = rA * (REAL)exp((double)(-b[iA] * lSa)); zM[t] = rM * (REAL)exp((double)(-b[iM] * lSm));
    const REAL res = (REAL)M->y[t] - al[t] - beta * (REAL)M->u[t];
    e[t] = res * s2i;
    lp += -(REAL)0.5 * zA[t] * zA[t] - b[iA] * lSa - (REAL)0.5 * zM[t] * zM[t] - b[iM] * lSm - (REAL)0.5 * res * res * s2i;
    if (centred) { centred[iA] = al[t]; centred[iM] = mu[t]; }
    if (mloc) { mloc[iA] = mA; mloc[iM] = mM; }
    ap = al[t]; mp = mu[t];
  }
  if (!g) return lp;
  REAL G = 0, H = 0, g_lSa = 0, g_lSm = 0, g_beta = -beta;
  for (int t = T - 1; t >= 0; --t) {
    const int iA = 2 + 2 * t, iM = 3 + 2 * t;
    const REAL cA = (REAL)exp((double)((1 - b[iA]) * lSa)), cM = (REAL)exp((double)((1 - b[iM]) * lSm));
    const REAL eA = (REAL)exp((double)(-b[iA] * lSa)), eM = (REAL)exp((double)(-b[iM] * lSm));
    const REAL Ab = e[t] + G, Mb = G + H;
    g[iA] = cA * Ab - zA[t] * eA;
    g[iM] = cM * Mb - zM[t] * eM;
    g_lSa += Ab * (1 - b[iA]) * zA[t] * Sa + b[iA] * (zA[t] * zA[t] - 1);
    g_lSm += Mb * (1 - b[iM]) * zM[t] * Sm + b[iM] * (zM[t] * zM[t] - 1);
    g_beta += e[t] * (REAL)M->u[t];
    G = (1 - a[iA] * cA) * Ab + a[iA] * zA[t] * eA;
    H = (1 - a[iM] * cM) * Mb + a[iM] * zM[t] * eM;
  }
  g[0] = -sa + g_lSa * (REAL)(1.0 / (1.0 + exp(-(double)sa))) / Sa;
  g[1] = -sm + g_lSm * (REAL)(1.0 / (1.0 + exp(-(double)sm))) / Sm;
  g[iB] = g_beta;
  return lp;
}
static REAL FN(time_series_logp_grad)(const orc_model* M, const float* a, const float* b, const REAL* x, REAL* g) {
  return FN(time_series_run)(M, a, b, x, g, NULL, NULL);
}
static void FN(time_series_convert)(const orc_model* M, const float* a, const float* b, const REAL* x,
                                    REAL* out, int to_centered) {
  const int T = M->N, iB = 2 + 2 * T;
  out[0] = x[0]; out[1] = x[1]; out[iB] = x[iB];
  if (to_centered) { FN(time_series_run)(M, a, b, x, NULL, out, NULL); return; }
  const REAL lSa = (REAL)log(log1p(exp((double)x[0]))), lSm = (REAL)log(log1p(exp((double)x[1])));
  REAL ap = 0, mp = 0;
  for (int t = 0; t < T; ++t) {   /* x holds centred values: every location is known */
    const int iA = 2 + 2 * t, iM = 3 + 2 * t;
    const REAL mA = ap + mp, mM = mp;
    out[iA] = a[iA] * mA + (x[iA] - mA) * (REAL)exp((double)(-(1 - b[iA]) * lSa));
    out[iM] = a[iM] * mM + (x[iM] - mM) * (REAL)exp((double)(-(1 - b[iM]) * lSm));
    ap = x[iA]; mp = x[iM];
  }
}

/* dispatch */
static REAL FN(logp_grad)(const orc_model* M, const float* a, const float* b, const REAL* x, REAL* g) {
  switch (M->model) {
    case 0: return FN(schools_logp_grad)(M, a, b, x, g);
    case 1: return FN(radon_logp_grad)(M, a, b, x, g);
    case 2: return FN(german_logp_grad)(M, a, b, x, g);
    case 3: return FN(election_logp_grad)(M, a, b, x, g);
    case 4: return FN(radon_sd_logp_grad)(M, a, b, x, g);
    case 5: return FN(funnel_logp_grad)(M, a, b, x, g);
    case 6: return FN(electric_logp_grad)(M, a, b, x, g);
    case 7: return FN(time_series_logp_grad)(M, a, b, x, g);
    default: return (REAL)NAN;
  }
}
static void FN(to_centered)(const orc_model* M, const float* a, const float* b, const REAL* x, REAL* o) {
  switch (M->model) {
    case 0: FN(schools_convert)(M, a, b, x, o, 1); break;
    case 1: FN(radon_to_centered)(M, a, b, x, o); break;
    case 2: FN(german_convert)(M, a, b, x, o, 1); break;
    case 3: FN(election_convert)(M, a, b, x, o, 1); break;
    case 4: FN(radon_sd_convert)(M, a, b, x, o, 1); break;
    case 5: FN(funnel_convert)(M, a, b, x, o, 1); break;
    case 6: FN(electric_convert)(M, a, b, x, o, 1); break;
    case 7: FN(time_series_convert)(M, a, b, x, o, 1); break;
    default: break;
  }
}
static void FN(from_centered)(const orc_model* M, const float* a, const float* b, const REAL* x, REAL* o) {
  switch (M->model) {
    case 0: FN(schools_convert)(M, a, b, x, o, 0); break;
    case 1: FN(radon_from_centered)(M, a, b, x, o); break;
    case 2: FN(german_convert)(M, a, b, x, o, 0); break;
    case 3: FN(election_convert)(M, a, b, x, o, 0); break;
    case 4: FN(radon_sd_convert)(M, a, b, x, o, 0); break;
    case 5: FN(funnel_convert)(M, a, b, x, o, 0); break;
    case 6: FN(electric_convert)(M, a, b, x, o, 0); break;
    case 7: FN(time_series_convert)(M, a, b, x, o, 0); break;
    default: break;
  }
}

int FN(orc_logp_grad)(const orc_model* M, const float* a, const float* b, const REAL* x, int C,
                      REAL* logp, REAL* grad) {
  const int D = M->D;
#pragma omp parallel for schedule(static)
  for (int c = 0; c < C; ++c) logp[c] = FN(logp_grad)(M, a, b, x + (size_t)c * D, grad + (size_t)c * D);
  return 0;
}

int FN(orc_transform)(const orc_model* M, const float* a, const float* b, int dir, const REAL* in, int C,
                      REAL* out) {
  const int D = M->D;
  for (int c = 0; c < C; ++c) {
    if (dir == 0) FN(to_centered)(M, a, b, in + (size_t)c * D, out + (size_t)c * D);
    else FN(from_centered)(M, a, b, in + (size_t)c * D, out + (size_t)c * D);
  }
  return 0;
}

/* ------------------------------------------------------------------------
 * Momentum and Metropolis draws of one transition for one chain.
 * Stream partition (layout 0): slot s of `lanes` owns the replicated top-level scalars
 * (only slot 0's draw is used) followed, for each sliced latent part in trace
 * order, by its elements j = s + lanes*i.  Layout 1 (orc_model.mom_spec, radon): the
 * slices come first and the top-level scalars are dealt out over the slots after them.
 * ---------------------------------------------------------------------- */
static void FN(draw_momentum)(const orc_model* M, orc_rng* streams, int lanes, REAL* p, REAL* u_out) {
  const int NG = M->n_glob, G = M->n_groups, P = M->n_local_parts;
  const int per_lane = orc_per_lane(M, lanes);
  const int spec1 = M->mom_spec == 1;
  const int extra = spec1 ? (NG + lanes - 1) / lanes : 0;
  /* normals every slot draws.  layout 0: scalars, then part by part; layout 1: the slices, then `extra` scalars */
  const int nd = spec1 ? P * per_lane + extra : NG + P * per_lane;
  const int first_local = spec1 ? 0 : NG;
  for (int s = 0; s < lanes; ++s) {
    orc_rng* r = &streams[s];
    for (int i = 0; i < nd; i += 2) {
      float z0, z1;
      uint32_t w0 = orc_rng_next(r), w1 = orc_rng_next(r);
      orc_normal_pair(w0, w1, &z0, &z1);
      for (int k = 0; k < 2; ++k) {
        int ii = i + k;
        float z = k ? z1 : z0;
        if (ii >= nd) break;
        if (!spec1 && ii < NG) {
          if (s == 0) p[M->glob_idx[ii]] = (REAL)z;
        } else if (spec1 && ii >= P * per_lane) {
          const int gk = s + lanes * (ii - P * per_lane);
          if (gk < NG) p[M->glob_idx[gk]] = (REAL)z;
        } else {
          int part = (ii - first_local) / per_lane;
          int j = M->contig ? s * per_lane + (ii - first_local) % per_lane : s + lanes * ((ii - first_local) % per_lane);
          if (j < G && M->group_idx[part * G + j] >= 0) p[M->group_idx[part * G + j]] = (REAL)z;
        }
      }
    }
    uint32_t w = orc_rng_next(r);
    if (s == 0) *u_out = (REAL)((float)((w >> 8) + 1u) * 5.9604644775390625e-08f);
  }
}

/* One HMC transition (tfp.mcmc.HamiltonianMonteCarlo.one_step as wired at
 * inference.py:218-222; TFP internals restated from the published algorithm:
 * unit mass, element-wise step size, L leapfrog steps, Metropolis test with a
 * non-finite energy error rejecting).  The two half kicks of consecutive
 * leapfrog steps are merged into one full kick. */
static REAL FN(hmc_transition)(const orc_model* M, const float* a, const float* b, orc_rng* streams,
                               int lanes, int L, const REAL* eps, REAL* q, REAL* g, REAL* lp,
                               int* accepted, REAL* work, REAL* margin, REAL* escale) {
  const int D = M->D;
  REAL* p = work; REAL* q1 = work + D; REAL* g1 = work + 2 * D;
  REAL u = 1;
  for (int d = 0; d < D; ++d) p[d] = 0;
  FN(draw_momentum)(M, streams, lanes, p, &u);
  REAL ke0 = 0;
  for (int d = 0; d < D; ++d) ke0 += p[d] * p[d];
  ke0 *= (REAL)0.5;
  for (int d = 0; d < D; ++d) { q1[d] = q[d]; p[d] += (REAL)0.5 * eps[d] * g[d]; }
  REAL lp1 = 0;
  for (int l = 0; l < L; ++l) {
    for (int d = 0; d < D; ++d) q1[d] += eps[d] * p[d];
    lp1 = FN(logp_grad)(M, a, b, q1, g1);
    const REAL w = (l + 1 < L) ? (REAL)1 : (REAL)0.5;
    for (int d = 0; d < D; ++d) p[d] += w * eps[d] * g1[d];
  }
  REAL ke1 = 0;
  for (int d = 0; d < D; ++d) ke1 += p[d] * p[d];
  ke1 *= (REAL)0.5;
  REAL la = (lp1 - *lp) + (ke0 - ke1);
  if (!isfinite((double)la)) la = -(REAL)INFINITY;
  *accepted = (REAL)log((double)u) < la;
  if (margin) *margin = (REAL)log((double)u) - la;
  if (escale) {
    REAL m = (REAL)fabs((double)*lp);
    if ((REAL)fabs((double)lp1) > m) m = (REAL)fabs((double)lp1);
    if (ke0 > m) m = ke0;
    if (ke1 > m) m = ke1;
    *escale = m;
  }
  if (*accepted) {
    for (int d = 0; d < D; ++d) { q[d] = q1[d]; g[d] = g1[d]; }
    *lp = lp1;
  }
  return la;
}

/* Step-size multiplier update after transition n (1-based).
 * DUAL: tfp.mcmc.DualAveragingStepSizeAdaptation defaults (inference.py:224-226):
 *   target 0.75, exploration_shrinkage 0.05, step_count_smoothing 10,
 *   decay_rate 0.75, shrinkage target log(10 eps0); per-chain because the step
 *   size carries the chain axis.
 * SIMPLE: tfp.mcmc.SimpleStepSizeAdaptation(rate, target) (inference.py:288-306). */
static void FN(adapt_update)(int kind, long long n, int n_adapt, REAL target, REAL rate, REAL la,
                             REAL* kappa, REAL* esum, REAL* logavg) {
  if (kind == 0) return;
  const REAL lacc = la < 0 ? la : 0;
  if (kind == 1) {
    if (n <= n_adapt) {
      const REAL t = (REAL)n;
      *esum += target - (REAL)exp((double)lacc);
      const REAL ls = (REAL)2.302585092994046 - *esum * (REAL)sqrt((double)t) / ((t + 10) * (REAL)0.05);
      const REAL eta = (REAL)pow((double)t, -0.75);
      *logavg = eta * ls + (1 - eta) * *logavg;
      *kappa = (REAL)exp((double)ls);
    } else if (n_adapt > 0) {
      *kappa = (REAL)exp((double)*logavg);
    }
  } else {
    if (n <= n_adapt) {
      const REAL opr = 1 + rate;
      *kappa = (lacc > (REAL)log((double)target)) ? *kappa * opr : *kappa / opr;
    }
  }
}

/* test hooks: a bare leapfrog trajectory and one adaptation update */
int FN(orc_leapfrog)(const orc_model* M, const float* a, const float* b, int L, const REAL* eps,
                     REAL* q, REAL* p, REAL* logp_out) {
  const int D = M->D;
  REAL* g = (REAL*)malloc(sizeof(REAL) * D);
  REAL lp = FN(logp_grad)(M, a, b, q, g);
  for (int d = 0; d < D; ++d) p[d] += (REAL)0.5 * eps[d] * g[d];
  for (int l = 0; l < L; ++l) {
    for (int d = 0; d < D; ++d) q[d] += eps[d] * p[d];
    lp = FN(logp_grad)(M, a, b, q, g);
    const REAL w = (l + 1 < L) ? (REAL)1 : (REAL)0.5;
    for (int d = 0; d < D; ++d) p[d] += w * eps[d] * g[d];
  }
  *logp_out = lp;
  free(g);
  return 0;
}
void FN(orc_adapt_update)(int kind, long long n, int n_adapt, REAL target, REAL rate, REAL la, REAL* state3) {
  FN(adapt_update)(kind, n, n_adapt, target, rate, la, &state3[0], &state3[1], &state3[2]);
}

/* A run of n_steps transitions for C chains, same contract as arp_hmc_run
 * (include/autoreparam.h) with host buffers; `rng` holds [C][16] stream states
 * and is seeded here when step_base == 0. */
/* arp_hmc_io.stats: one recorded sample x[D] of chain c, sample index r (0-based over the whole run) */
static void FN(stats_update)(const orc_hmc_cfg* cfg, int C, int D, int c, long long r, const REAL* x) {
  REAL* S = (REAL*)cfg->stats;
  const size_t comp = (size_t)C * D, o = (size_t)c * D;
  const int batch = cfg->stats_batch > 0 ? cfg->stats_batch : 1;
  const int batch_end = (r + 1) % batch == 0;
  for (int d = 0; d < D; ++d) {
    if (r == 0) S[o + d] = x[d];
    const REAL dx = x[d] - S[o + d];
    S[comp + o + d] += dx;
    S[2 * comp + o + d] += dx * dx;
    if (batch_end) {   /* plane 3 holds s1 as it was when the batch began */
      const REAL bm = (S[comp + o + d] - S[3 * comp + o + d]) / (REAL)batch;
      S[4 * comp + o + d] += bm; S[5 * comp + o + d] += bm * bm; S[3 * comp + o + d] = S[comp + o + d];
    }
  }
}

int FN(orc_hmc_run)(const orc_model* M, const float* a, const float* b, const orc_hmc_cfg* cfg,
                    REAL* q, REAL* grad, REAL* logp, REAL* adapt, uint32_t* rng, uint32_t* accept_count,
                    const float* eps0, REAL* trace, uint8_t* trace_accept) {
  const int D = M->D, C = cfg->n_chains, lanes = cfg->lanes;
#pragma omp parallel
  {
    REAL* work = (REAL*)malloc(sizeof(REAL) * (size_t)D * 5);
    REAL* eps = work + 3 * D;
    REAL* xc = work + 4 * D;
#pragma omp for schedule(static)
    for (int c = 0; c < C; ++c) {
      REAL* qc = q + (size_t)c * D;
      REAL* gc = grad + (size_t)c * D;
      orc_rng* st = (orc_rng*)(rng + (size_t)c * 16 * 4);
      REAL lp, kappa, esum, logavg;
      uint32_t nacc;
      if (cfg->step_base == 0) {
        lp = FN(logp_grad)(M, a, b, qc, gc);
        kappa = 1; esum = 0; logavg = 0; nacc = 0;
        for (int s = 0; s < lanes; ++s)
          st[s] = orc_rng_seed(cfg->seed, (uint64_t)(cfg->chain_offset + c), (uint32_t)s, (uint32_t)lanes);
      } else {
        lp = logp[c];
        kappa = adapt[c * 4 + 0]; esum = adapt[c * 4 + 1]; logavg = adapt[c * 4 + 2];
        nacc = accept_count[c];
      }
      for (int s = 0; s < cfg->n_steps; ++s) {
        for (int d = 0; d < D; ++d) eps[d] = (REAL)eps0[d] * kappa;
        int acc;
        REAL la = FN(hmc_transition)(M, a, b, st, lanes, cfg->n_leapfrog, eps, qc, gc, &lp, &acc, work,
                                     cfg->margin ? (REAL*)cfg->margin + (size_t)s * C + c : NULL,
                                     cfg->escale ? (REAL*)cfg->escale + (size_t)s * C + c : NULL);
        nacc += (uint32_t)acc;
        if (cfg->log_alpha) ((REAL*)cfg->log_alpha)[(size_t)s * C + c] = la;
        const long long n = cfg->step_base + s + 1;
        FN(adapt_update)(cfg->adapt_kind, n, cfg->n_adapt, (REAL)cfg->adapt_target, (REAL)cfg->adapt_rate,
                         la, &kappa, &esum, &logavg);
        /* tfp.mcmc.sample_chain: result r after transition 1 + burnin + r*thin */
        const long long k = n - 1 - cfg->n_burnin;
        if (k >= 0 && k % cfg->thin == 0 && k / cfg->thin < cfg->n_samples) {
          const long long r = k / cfg->thin;
          const int tc = (cfg->trace_chains > 0 && cfg->trace_chains < C) ? cfg->trace_chains : C;
          const REAL* x = qc;
          if (cfg->trace_centered && (trace || cfg->stats)) { FN(to_centered)(M, a, b, qc, xc); x = xc; }
          if (trace && c < tc) memcpy(trace + ((size_t)r * tc + c) * D, x, sizeof(REAL) * D);
          if (cfg->stats) FN(stats_update)(cfg, C, D, c, r, x);
          if (trace_accept) trace_accept[(size_t)r * C + c] = (uint8_t)acc;
          if (cfg->rec_accept) cfg->rec_accept[c] += (uint32_t)acc;
        }
      }
      logp[c] = lp;
      adapt[c * 4 + 0] = kappa; adapt[c * 4 + 1] = esum; adapt[c * 4 + 2] = logavg;
      accept_count[c] = nacc;
    }
    free(work);
  }
  return 0;
}

/* Interleaved sampling, same contract as arp_interleaved_run: per step a
 * transition under parameterisation (a0,b0), the change of coordinates, a
 * transition under (a1,b1), and the change back (interleaved.py:113-155);
 * logp/grad are recomputed ("bootstrapped") after every change of coordinates
 * (interleaved.py:120-123, 136-139); each inner kernel carries its own
 * adaptation state (inference.py:288-306). */
int FN(orc_interleaved_run)(const orc_model* M, const float* a0, const float* b0, const float* a1,
                            const float* b1, const orc_hmc_cfg* cfg, int L1, REAL* q, REAL* adapt0,
                            REAL* adapt1, uint32_t* rng, uint32_t* acc0, uint32_t* acc1,
                            const float* eps0_0, const float* eps0_1, REAL* trace,
                            uint8_t* trace_acc0, uint8_t* trace_acc1) {
  const int D = M->D, C = cfg->n_chains, lanes = cfg->lanes;
#pragma omp parallel
  {
    REAL* work = (REAL*)malloc(sizeof(REAL) * (size_t)D * 6);
    REAL* eps = work + 3 * D; REAL* xc = work + 4 * D; REAL* g = work + 5 * D;
#pragma omp for schedule(static)
    for (int c = 0; c < C; ++c) {
      REAL* qc = q + (size_t)c * D;
      orc_rng* st = (orc_rng*)(rng + (size_t)c * 16 * 4);
      REAL k0, e0, l0, k1, e1, l1;
      uint32_t n0, n1;
      if (cfg->step_base == 0) {
        k0 = k1 = 1; e0 = e1 = 0; l0 = l1 = 0; n0 = n1 = 0;
        for (int s = 0; s < lanes; ++s)
          st[s] = orc_rng_seed(cfg->seed, (uint64_t)(cfg->chain_offset + c), (uint32_t)s, (uint32_t)lanes);
      } else {
        k0 = adapt0[c * 4]; e0 = adapt0[c * 4 + 1]; l0 = adapt0[c * 4 + 2];
        k1 = adapt1[c * 4]; e1 = adapt1[c * 4 + 1]; l1 = adapt1[c * 4 + 2];
        n0 = acc0[c]; n1 = acc1[c];
      }
      for (int s = 0; s < cfg->n_steps; ++s) {
        const long long n = cfg->step_base + s + 1;
        int a_0, a_1;
        REAL lp = FN(logp_grad)(M, a0, b0, qc, g);
        for (int d = 0; d < D; ++d) eps[d] = (REAL)eps0_0[d] * k0;
        REAL* mg = cfg->margin ? (REAL*)cfg->margin + (size_t)s * 2 * C + c : NULL;
        REAL* es = cfg->escale ? (REAL*)cfg->escale + (size_t)s * 2 * C + c : NULL;
        REAL la = FN(hmc_transition)(M, a0, b0, st, lanes, cfg->n_leapfrog, eps, qc, g, &lp, &a_0, work, mg, es);
        n0 += (uint32_t)a_0;
        if (cfg->log_alpha) ((REAL*)cfg->log_alpha)[(size_t)s * 2 * C + c] = la;
        FN(adapt_update)(cfg->adapt_kind, n, cfg->n_adapt, (REAL)cfg->adapt_target, (REAL)cfg->adapt_rate, la, &k0, &e0, &l0);
        FN(to_centered)(M, a0, b0, qc, xc);
        FN(from_centered)(M, a1, b1, xc, qc);
        lp = FN(logp_grad)(M, a1, b1, qc, g);
        for (int d = 0; d < D; ++d) eps[d] = (REAL)eps0_1[d] * k1;
        la = FN(hmc_transition)(M, a1, b1, st, lanes, L1, eps, qc, g, &lp, &a_1, work, mg ? mg + C : NULL, es ? es + C : NULL);
        n1 += (uint32_t)a_1;
        if (cfg->log_alpha) ((REAL*)cfg->log_alpha)[(size_t)s * 2 * C + C + c] = la;
        FN(adapt_update)(cfg->adapt_kind, n, cfg->n_adapt, (REAL)cfg->adapt_target, (REAL)cfg->adapt_rate, la, &k1, &e1, &l1);
        FN(to_centered)(M, a1, b1, qc, xc);
        FN(from_centered)(M, a0, b0, xc, qc);
        const long long k = n - 1 - cfg->n_burnin;
        if (k >= 0 && k % cfg->thin == 0 && k / cfg->thin < cfg->n_samples) {
          const long long r = k / cfg->thin;
          const int tc = (cfg->trace_chains > 0 && cfg->trace_chains < C) ? cfg->trace_chains : C;
          const REAL* x = cfg->trace_centered ? xc : qc;
          if (trace && c < tc) memcpy(trace + ((size_t)r * tc + c) * D, x, sizeof(REAL) * D);
          if (cfg->stats) FN(stats_update)(cfg, C, D, c, r, x);
          if (trace_acc0) trace_acc0[(size_t)r * C + c] = (uint8_t)a_0;
          if (trace_acc1) trace_acc1[(size_t)r * C + c] = (uint8_t)a_1;
          if (cfg->rec_accept) cfg->rec_accept[c] += (uint32_t)a_0;
          if (cfg->rec_accept1) cfg->rec_accept1[c] += (uint32_t)a_1;
        }
      }
      adapt0[c * 4] = k0; adapt0[c * 4 + 1] = e0; adapt0[c * 4 + 2] = l0;
      adapt1[c * 4] = k1; adapt1[c * 4 + 1] = e1; adapt1[c * 4 + 2] = l1;
      acc0[c] = n0; acc1[c] = n1;
    }
    free(work);
  }
  return 0;
}

/* d logp / d a_i, d logp / d b_i (cVIP learns a): with xt ~ N(a mu, sigma^b) and
 * g = d logp / d xt,  d/da = -mu g,  d/db = -log(sigma) (1 + (xt - a mu) g)
 * (derived from program_transformations.py:569-576; checked against autograd in tests). */
static void FN(dparam)(const orc_model* M, const float* a, const float* b, const REAL* x, const REAL* g,
                       REAL* da, REAL* db) {
  const int D = M->D;
  REAL* xc = (REAL*)malloc(sizeof(REAL) * D);
  REAL* mu = (REAL*)calloc(D, sizeof(REAL));
  REAL* ls = (REAL*)calloc(D, sizeof(REAL));
  FN(to_centered)(M, a, b, x, xc);
  switch (M->model) {
    case 0: ls[0] = ls[1] = (REAL)log(5.0); for (int k = 0; k < 8; ++k) { mu[2 + k] = xc[0]; ls[2 + k] = xc[1]; } break;
    case 1: case 4: for (int j = 0; j < M->J; ++j) mu[3 + j] = xc[0] + (REAL)M->u[j] * xc[1]; break;
    case 2: ls[0] = (REAL)log(10.0);
      for (int d = 0; d < M->F; ++d) { mu[1 + d] = xc[0]; ls[1 + M->F + d] = xc[1 + d]; } break;
    case 5: ls[0] = (REAL)log(3.0); ls[1] = xc[0] / 2; break;
    case 3: ls[0] = (REAL)log(100.0); ls[1] = (REAL)log(10.0);
      ls[2 + M->S] = ls[3 + M->S] = (REAL)log(100.0);
      for (int t = 0; t < M->S; ++t) { mu[2 + t] = xc[0]; ls[2 + t] = xc[1]; } break;
    case 7: {
      FN(time_series_run)(M, a, b, x, NULL, NULL, mu);   /* locations of every trend latent */
      const REAL lSa = (REAL)log(log1p(exp((double)x[0]))), lSm = (REAL)log(log1p(exp((double)x[1])));
      for (int t = 0; t < M->N; ++t) { ls[2 + 2 * t] = lSa; ls[3 + 2 * t] = lSm; }
      break;
    }
    case 6:
      for (int k = 0; k < M->G; ++k) ls[2 * M->G + M->P + k] = (REAL)log(100.0);
      for (int j = 0; j < M->P; ++j) {
        const int k = M->grade_pair[j];
        mu[2 * M->G + j] = (k >= 0 && k < M->G) ? 100 * xc[k] : 0;
      }
      break;
    default: break;
  }
  for (int d = 0; d < D; ++d) {
    da[d] = -mu[d] * g[d];
    db[d] = -ls[d] * (1 + (x[d] - (REAL)a[d] * mu[d]) * g[d]);
  }
  free(xc); free(mu); free(ls);
}
int FN(orc_dparam)(const orc_model* M, const float* a, const float* b, const REAL* x, REAL* da, REAL* db) {
  REAL* g = (REAL*)malloc(sizeof(REAL) * M->D);
  FN(logp_grad)(M, a, b, x, g);
  FN(dparam)(M, a, b, x, g, da, db);
  free(g);
  return 0;
}

/* Mean-field VI, same contract as arp_vi_run (inference.py:26-154, util.py:232-268,
 * program_transformations.py:192-241): q = prod N(loc, softplus(rho)); ELBO from
 * n_mc reparameterised draws; tf.train.AdamOptimizer defaults on -ELBO; NaN
 * gradients zeroed; learning rate base, base/5 after a third, base/20 after two
 * thirds of the steps.  Draw layout: `block/lanes` streams per slot, sample s uses
 * stream s mod (block/lanes). */
/* d/dx log Mixture(logits (0,5,0); Laplace(0,0.1), Uniform(0,1), Laplace(1,0.1))(x), 0 < x < 1 (main.py:244-253) */
static REAL FN(discrete_prior_dlogp)(REAL x) {
  const double l0 = 5.0 * exp(-10.0 * (double)x), l1 = 5.0 * exp(-10.0 * (1.0 - (double)x));
  return (REAL)(10.0 * (l1 - l0) / (l0 + exp(5.0) + l1));
}

/* learn_a: bit 0 = optimise a (cVIP), bit 1 = add the --discrete_prior term (inference.py:50-54) */
int FN(orc_vi_run)(const orc_model* M, const float* a_in, const float* b_in, int n_lr, int n_steps, int n_mc,
                   int learn_a_flags, int tied_b, uint64_t seed, int lanes, int block, const float* lr_in,
                   REAL* loc_io, REAL* rho_io, REAL* w_io, REAL* wb_io, REAL* elbo_out, double const_base, int n_top,
                   const int* top_idx, const double* top_logscale, const int* a_group, const int* b_group,
                   REAL* prior_out) {
  const int learn_a = learn_a_flags & 1, a_prior = (learn_a_flags >> 1) & 1;
  const int D = M->D, NG = M->n_glob, G = M->n_groups, P = M->n_local_parts;
  const int per_lane = orc_per_lane(M, lanes), nd = NG + P * per_lane;
  const int cpp = block / lanes, passes = (n_mc + cpp - 1) / cpp;
  for (int li = 0; li < n_lr; ++li) {
    REAL* loc = loc_io + (size_t)li * D; REAL* rho = rho_io + (size_t)li * D;
    REAL* w = learn_a ? w_io + (size_t)li * D : NULL;
    REAL* wb = (learn_a && wb_io) ? wb_io + (size_t)li * D : NULL;
    float* a = (float*)malloc(sizeof(float) * D); float* b = (float*)malloc(sizeof(float) * D);
    memcpy(a, a_in, sizeof(float) * D); memcpy(b, b_in, sizeof(float) * D);
    orc_rng* streams = (orc_rng*)malloc(sizeof(orc_rng) * (size_t)cpp * lanes);
    for (int c0 = 0; c0 < cpp; ++c0)
      for (int s = 0; s < lanes; ++s)
        streams[c0 * lanes + s] = orc_rng_seed(seed ^ 0x5649564956495649ull, ((uint64_t)li << 32) | (uint32_t)c0,
                                               (uint32_t)s, (uint32_t)lanes);
    REAL* m1 = (REAL*)calloc(4 * (size_t)D, sizeof(REAL)); REAL* m2 = (REAL*)calloc(4 * (size_t)D, sizeof(REAL));
    REAL* sig = (REAL*)malloc(sizeof(REAL) * D); REAL* eps = (REAL*)malloc(sizeof(REAL) * D);
    REAL* z = (REAL*)malloc(sizeof(REAL) * D); REAL* g = (REAL*)malloc(sizeof(REAL) * D);
    REAL* da = (REAL*)malloc(sizeof(REAL) * D); REAL* db = (REAL*)malloc(sizeof(REAL) * D);
    REAL* acc = (REAL*)malloc(sizeof(REAL) * 4 * (size_t)D);
    REAL b1t = 1, b2t = 1;
    for (int step = 0; step < n_steps; ++step) {
      for (int d = 0; d < D; ++d) {
        sig[d] = rho[d] > 20 ? rho[d] : (REAL)log(1.0 + exp((double)rho[d]));
        if (learn_a) { a[d] = (float)(1.0 / (1.0 + exp(-(double)w[d]))); if (tied_b) b[d] = a[d];
          if (wb) b[d] = (float)(1.0 / (1.0 + exp(-(double)wb[d]))); }
      }
      memset(acc, 0, sizeof(REAL) * 4 * (size_t)D);
      REAL elbo = 0;
      for (int pass = 0; pass < passes; ++pass) {
        for (int c0 = 0; c0 < cpp; ++c0) {
          for (int d = 0; d < D; ++d) eps[d] = 0;
          for (int s = 0; s < lanes; ++s) {   /* same layout as draw_momentum, without the uniform */
            orc_rng* r = &streams[c0 * lanes + s];
            for (int i = 0; i < nd; i += 2) {
              float z0, z1;
              uint32_t w0 = orc_rng_next(r), w1 = orc_rng_next(r);
              orc_normal_pair(w0, w1, &z0, &z1);
              for (int k = 0; k < 2; ++k) {
                int ii = i + k; float zz = k ? z1 : z0;
                if (ii >= nd) break;
                if (ii < NG) { if (s == 0) eps[M->glob_idx[ii]] = (REAL)zz; }
                else {
                  int part = (ii - NG) / per_lane;
                  int j = M->contig ? s * per_lane + (ii - NG) % per_lane : s + lanes * ((ii - NG) % per_lane);
                  if (j < G && M->group_idx[part * G + j] >= 0) eps[M->group_idx[part * G + j]] = (REAL)zz;
                }
              }
            }
          }
          if (c0 + pass * cpp >= n_mc) continue;
          REAL ent = 0;
          for (int d = 0; d < D; ++d) { z[d] = loc[d] + sig[d] * eps[d]; ent += (REAL)0.5 * eps[d] * eps[d] + (REAL)log((double)sig[d]); }
          REAL lp = FN(logp_grad)(M, a, b, z, g);
          elbo += lp + ent;
          for (int d = 0; d < D; ++d) { acc[d] += g[d]; acc[D + d] += g[d] * eps[d]; }
          if (learn_a) {
            FN(dparam)(M, a, b, z, g, da, db);
            for (int d = 0; d < D; ++d) { acc[2 * D + d] += da[d]; acc[3 * D + d] += db[d]; }
          }
        }
      }
      REAL lr = (REAL)lr_in[li];
      if (3 * step > 2 * n_steps) lr = lr / 20; else if (3 * step > n_steps) lr = lr / 5;
      b1t *= (REAL)0.9; b2t *= (REAL)0.999;
      const REAL lr_t = lr * (REAL)sqrt((double)(1 - b2t)) / (1 - b1t);
      double c = const_base + 0.9189385332046727 * D;
      for (int k = 0; k < n_top; ++k) c -= (double)b[top_idx[k]] * top_logscale[k];
      elbo_out[(size_t)li * n_steps + step] = elbo / n_mc + (REAL)c;
      /* untied parameterisation variables shared over a part (program_transformations.py:486-533: `a` has the shape of
       * the loc, `b` of the scale): the leader (first element of the group) collects its members' likelihood terms,
       * the prior counts once; members copy the leader's value after the update */
      if (learn_a && (a_group || (wb && b_group))) {
        for (int d = D - 1; d >= 0; --d) {
          if (a_group && a_group[d] != d) { acc[2 * D + a_group[d]] += acc[2 * D + d]; }
          if (wb && b_group && b_group[d] != d) { acc[3 * D + b_group[d]] += acc[3 * D + d]; }
        }
      }
      if (prior_out) {   /* log density of the --discrete_prior mixture on every learnable variable (inference.py:50-54) */
        double lpr = 0;
        for (int d = 0; d < D && learn_a; ++d) {
          const double norm = log(2.0 + exp(5.0));
          if (!a_group || a_group[d] == d)
            lpr += log(5.0 * exp(-10.0 * (double)a[d]) + exp(5.0) + 5.0 * exp(-10.0 * (1.0 - (double)a[d]))) - norm;
          if (wb && (!b_group || b_group[d] == d))
            lpr += log(5.0 * exp(-10.0 * (double)b[d]) + exp(5.0) + 5.0 * exp(-10.0 * (1.0 - (double)b[d]))) - norm;
        }
        prior_out[(size_t)li * n_steps + step] = (REAL)lpr;
      }
      for (int d = 0; d < D; ++d) {
        REAL gr[4];
        gr[0] = -acc[d] / n_mc;
        gr[1] = -(acc[D + d] / n_mc + 1 / sig[d]) * (REAL)(1.0 / (1.0 + exp(-(double)rho[d])));
        const REAL pa = a_prior ? FN(discrete_prior_dlogp)((REAL)a[d]) : 0, pb = a_prior ? FN(discrete_prior_dlogp)((REAL)b[d]) : 0;
        gr[2] = learn_a ? -((acc[2 * D + d] + (tied_b ? acc[3 * D + d] : 0)) / n_mc + pa) * a[d] * (1 - a[d]) : 0;
        gr[3] = wb ? -(acc[3 * D + d] / n_mc + pb) * b[d] * (1 - b[d]) : 0;
        REAL* par[4] = {&loc[d], &rho[d], learn_a ? &w[d] : NULL, wb ? &wb[d] : NULL};
        for (int k = 0; k < 4; ++k) {
          REAL gk = gr[k];
          if (gk != gk) gk = 0;
          m1[k * D + d] = (REAL)0.9 * m1[k * D + d] + (REAL)0.1 * gk;
          m2[k * D + d] = (REAL)0.999 * m2[k * D + d] + (REAL)0.001 * gk * gk;
          if (par[k]) *par[k] -= lr_t * m1[k * D + d] / ((REAL)sqrt((double)m2[k * D + d]) + (REAL)1e-8);
        }
      }
      for (int d = 0; d < D && learn_a; ++d) {
        if (a_group && a_group[d] != d) w[d] = w[a_group[d]];
        if (wb && b_group && b_group[d] != d) wb[d] = wb[b_group[d]];
      }
    }
    free(a); free(b); free(streams); free(m1); free(m2); free(sig); free(eps); free(z); free(g); free(da); free(db); free(acc);
  }
  return 0;
}
