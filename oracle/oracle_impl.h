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

/* dispatch */
static REAL FN(logp_grad)(const orc_model* M, const float* a, const float* b, const REAL* x, REAL* g) {
  switch (M->model) {
    case 1: return FN(radon_logp_grad)(M, a, b, x, g);
    default: return (REAL)NAN;
  }
}
static void FN(to_centered)(const orc_model* M, const float* a, const float* b, const REAL* x, REAL* o) {
  switch (M->model) {
    case 1: FN(radon_to_centered)(M, a, b, x, o); break;
    default: break;
  }
}
static void FN(from_centered)(const orc_model* M, const float* a, const float* b, const REAL* x, REAL* o) {
  switch (M->model) {
    case 1: FN(radon_from_centered)(M, a, b, x, o); break;
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
 * Stream partition: slot s of `lanes` owns the replicated top-level scalars
 * (only slot 0's draw is used) followed by the groups j = s + lanes*i.
 * ---------------------------------------------------------------------- */
static void FN(draw_momentum)(const orc_model* M, orc_rng* streams, int lanes, REAL* p, REAL* u_out) {
  const int NG = M->n_glob, G = M->n_groups;
  const int per_lane = (G + lanes - 1) / lanes;
  const int nd = NG + per_lane;
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
        if (ii < NG) {
          if (s == 0) p[M->glob_idx[ii]] = (REAL)z;
        } else {
          int j = s + lanes * (ii - NG);
          if (j < G && M->group_idx[j] >= 0) p[M->group_idx[j]] = (REAL)z;
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
                               int* accepted, REAL* work) {
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

/* A run of n_steps transitions for C chains, same contract as arp_hmc_run
 * (include/autoreparam.h) with host buffers; `rng` holds [C][16] stream states
 * and is seeded here when step_base == 0. */
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
        REAL la = FN(hmc_transition)(M, a, b, st, lanes, cfg->n_leapfrog, eps, qc, gc, &lp, &acc, work);
        nacc += (uint32_t)acc;
        const long long n = cfg->step_base + s + 1;
        FN(adapt_update)(cfg->adapt_kind, n, cfg->n_adapt, (REAL)cfg->adapt_target, (REAL)cfg->adapt_rate,
                         la, &kappa, &esum, &logavg);
        /* tfp.mcmc.sample_chain: result r after transition 1 + burnin + r*thin */
        const long long k = n - 1 - cfg->n_burnin;
        if (k >= 0 && k % cfg->thin == 0 && k / cfg->thin < cfg->n_samples) {
          const long long r = k / cfg->thin;
          if (trace) {
            REAL* row = trace + ((size_t)r * C + c) * D;
            if (cfg->trace_centered) { FN(to_centered)(M, a, b, qc, xc); memcpy(row, xc, sizeof(REAL) * D); }
            else memcpy(row, qc, sizeof(REAL) * D);
          }
          if (trace_accept) trace_accept[(size_t)r * C + c] = (uint8_t)acc;
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
