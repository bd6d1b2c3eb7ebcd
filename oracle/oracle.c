/*
 * oracle.c -- CPU restatement (plain C99 + OpenMP over chains) of the hot path:
 * per-model log joint + gradient under the general VIP parameterisation, the
 * HMC transition, dual-averaging / simple step-size adaptation and the
 * sample_chain trace schedule.
 *
 * TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg; the product path (autoreparam_amd/) never links
 * or calls it.  PARITY UNPINNED (see oracle_impl.h).
 *
 * Reference files restated: models.py:131-166, 671-696, 763-857, 884-923, 967-1141
 * (densities), program_transformations.py:262-279, 555-600 (NCP / VIP algebra),
 * inference.py:198-242 (HMC wiring, step scaling, thinning),
 * interleaved.py:113-155 (interleaving order).  TFP internals (leapfrog,
 * Metropolis, adaptation recurrences) are restated from the published
 * algorithms; they are not under /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct orc_model {
  int model;       /* same ids as include/autoreparam.h */
  int D;
  int n_glob;      /* replicated top-level scalars (RNG stream layout) */
  int n_groups;    /* sliced axis length (RNG stream layout) */
  int n_local_parts; /* latent parts sliced along that axis (german: beta_log_scales and beta) */
  int contig;      /* 1: slot s owns consecutive elements s*per_lane + i (german); 0: s + lanes*i */
  int lane_unit;   /* elements per lane come in whole units of this many (0 or 1: none; time_series: 2, a lane owns whole
                    * (alpha_t, mu_t) time steps, so per_lane = ceil(n_groups / lanes) rounded up to even) */
  int mom_spec;    /* momentum stream layout: 0 = top-level scalars first, drawn by every slot, slot 0's used;
                    * 1 = sliced elements first, then ceil(n_glob / lanes) extra normals per slot of which extra x of
                    * slot s is top-level scalar s + lanes*x (radon: no slot draws a normal it discards) */
  int glob_idx[16]; /* flattened index of each top-level scalar */
  int* group_idx;  /* [n_local_parts][n_groups] flattened index of element j, -1 if it has no latent */
  /* radon sufficient statistics */
  int J;
  float *n, *sx, *sy, *u;
  float *sxx_j, *sxy_j, *syy_j;   /* radon_stddvs: per-county second moments */
  float sxy, sxx;
  /* schools: u = treatment stddevs, y = effects.  german: X [N][F], y [N].
   * election: cell tables [(S+1)][4] indexed (state, female + 2*black). */
  int S, F, N;
  float *y, *X, *cell_n, *cell_y;
  /* electric: raw observations; pair / grade / grade_pair are the reference's 1-based values, used
   * as 0-based one-hot columns (out of range = all-zero row).  P = n_pair, G = n_grade. */
  int P, G;
  int *pair, *grade, *grade_pair;
  float* treat;
  double logp_const;   /* value dropped from logp under CP (tests add the (a,b)-dependent part) */
} orc_model;

typedef struct orc_hmc_cfg {
  int n_chains, n_leapfrog, n_steps;
  long long step_base, chain_offset;
  uint64_t seed;
  int adapt_kind, n_adapt;
  float adapt_target, adapt_rate;
  int n_burnin, thin, n_samples, trace_centered;
  int lanes;
  /* streaming statistics of the recorded samples (arp_hmc_io.stats: [6][C][D] in the run's REAL type, zeroed by the
   * caller), samples per batch, chains a trace row holds (0 = all), accepted-among-recorded counters */
  int stats_batch, trace_chains;
  void* stats;
  uint32_t* rec_accept;
  uint32_t* rec_accept1;
  /* test diagnostics, in the run's REAL type, indexed by the in-launch step s (or NULL):
   * margin [n_steps][kernels][C] = log u - log alpha of every Metropolis test (its sign IS the decision, its size how
   * close the test sat to its threshold), escale the same shape = max(|logp|, |logp'|, K, K') of that test (the size
   * of the terms whose float32 rounding can flip it).  kernels = 1 (orc_hmc_run) or 2 (orc_interleaved_run). */
  void* margin;
  void* escale;
  void* log_alpha;   /* same shape: log alpha itself (what the step-size recurrences consume) */
} orc_hmc_cfg;

/* ---- RNG: MWC64X streams seeded by Philox4x32-10 (DESIGN.md "Randomness").  The stream record
 * keeps the 16-byte layout of the state buffers: s[0] = x, s[1] = carry, s[2] = s[3] = 0. ---- */
typedef struct { uint32_t s[4]; } orc_rng;

static uint32_t orc_rng_next(orc_rng* r) {
  uint32_t* s = r->s;
  uint32_t result = s[0] ^ s[1];
  uint64_t t = (uint64_t)s[0] * 4294883355u + s[1];   /* x' = lo(A x + c), c' = hi(A x + c) */
  s[0] = (uint32_t)t; s[1] = (uint32_t)(t >> 32);
  return result;
}

static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

static orc_rng orc_rng_seed(uint64_t seed, uint64_t chain, uint32_t slot, uint32_t lanes) {
  uint32_t c[4] = {(uint32_t)chain, (uint32_t)(chain >> 32), slot, lanes};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  orc_rng r = {{c[0], c[1] >> 1, 0u, 0u}};
  if ((r.s[0] | r.s[1]) == 0u) r.s[0] = 1u;
  return r;
}

/* Box-Muller: u = (float)w0 2^-32 + 2^-33 in (0,1] (uint32 -> float conversions round to nearest even, as
 * v_cvt_f32_u32 does); angle = the low 23 bits of w1 as the fraction of a revolution (the device builds the float
 * 1.fraction and lets sin/cos drop the leading 1) */
static void orc_normal_pair(uint32_t w0, uint32_t w1, float* z0, float* z1) {
  float u = fmaf((float)w0, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
  float rev = (float)(w1 & 0x007fffffu) * 1.1920928955078125e-07f;   /* exact: 23 bits x 2^-23 */
  float r = sqrtf(-1.3862943611198906f * log2f(u));
  float ang = 6.283185307179586f * rev;
  *z0 = r * cosf(ang);
  *z1 = r * sinf(ang);
}

/* exported for the RNG known-answer tests */
void orc_philox(uint32_t ctr[4], uint32_t k0, uint32_t k1) { philox4x32_10(ctr, k0, k1); }
void orc_stream(uint64_t seed, uint64_t chain, uint32_t slot, uint32_t lanes, int n, uint32_t* out) {
  orc_rng r = orc_rng_seed(seed, chain, slot, lanes);
  for (int i = 0; i < n; ++i) out[i] = orc_rng_next(&r);
}
void orc_normals(uint64_t seed, uint64_t chain, uint32_t slot, uint32_t lanes, int n_pairs, float* out) {
  orc_rng r = orc_rng_seed(seed, chain, slot, lanes);
  for (int i = 0; i < n_pairs; ++i) {
    uint32_t w0 = orc_rng_next(&r), w1 = orc_rng_next(&r);
    orc_normal_pair(w0, w1, &out[2 * i], &out[2 * i + 1]);
  }
}

/* ---- model construction from the reference's raw inputs ---- */
static const double HALF_LOG_2PI = 0.9189385332046727;

orc_model* orc_radon_create(int N, int J, const int32_t* county, const float* u, const float* x,
                            const float* y) {
  orc_model* M = (orc_model*)calloc(1, sizeof(orc_model));
  M->model = 1; M->J = J; M->D = 3 + J;
  M->n_glob = 3; M->n_groups = J; M->n_local_parts = 1; M->mom_spec = 1;
  M->glob_idx[0] = 0; M->glob_idx[1] = 1; M->glob_idx[2] = 2;
  M->group_idx = (int*)malloc(sizeof(int) * J);
  M->n = (float*)calloc(J, sizeof(float)); M->sx = (float*)calloc(J, sizeof(float));
  M->sy = (float*)calloc(J, sizeof(float)); M->u = (float*)calloc(J, sizeof(float));
  double* n = (double*)calloc(J, sizeof(double));
  double* sx = (double*)calloc(J, sizeof(double));
  double* sy = (double*)calloc(J, sizeof(double));
  double sxy = 0, sxx = 0, syy = 0;
  for (int i = 0; i < N; ++i) {
    int j = county[i];
    sxy += (double)x[i] * y[i]; sxx += (double)x[i] * x[i]; syy += (double)y[i] * y[i];
    if (j < 0 || j >= J) continue; /* tf.one_hot: all-zero row, models.py:834 */
    n[j] += 1; sx[j] += x[i]; sy[j] += y[i];
  }
  for (int j = 0; j < J; ++j) {
    M->group_idx[j] = 3 + j;
    M->n[j] = (float)n[j]; M->sx[j] = (float)sx[j]; M->sy[j] = (float)sy[j]; M->u[j] = u[j];
  }
  M->sxy = (float)sxy; M->sxx = (float)sxx;
  M->logp_const = -(3.0 + J + N) * HALF_LOG_2PI - 0.5 * syy;
  free(n); free(sx); free(sy);
  return M;
}

orc_model* orc_radon_sd_create(int N, int J, const int32_t* county, const float* u, const float* x,
                               const float* y) {
  orc_model* M = (orc_model*)calloc(1, sizeof(orc_model));
  M->model = 4; M->J = J; M->D = 3 + 2 * J;
  M->n_glob = 3; M->n_groups = J; M->n_local_parts = 2;
  M->glob_idx[0] = 0; M->glob_idx[1] = 1; M->glob_idx[2] = 2;
  M->group_idx = (int*)malloc(sizeof(int) * 2 * J);
  float** tabs[7] = {&M->n, &M->sx, &M->sy, &M->u, &M->sxx_j, &M->sxy_j, &M->syy_j};
  for (int k = 0; k < 7; ++k) *tabs[k] = (float*)calloc(J, sizeof(float));
  double* st = (double*)calloc(6 * (size_t)J, sizeof(double));
  for (int i = 0; i < N; ++i) {
    int j = county[i];
    if (j < 0 || j >= J) continue;
    st[j] += 1; st[J + j] += x[i]; st[2 * J + j] += y[i];
    st[3 * J + j] += (double)x[i] * x[i]; st[4 * J + j] += (double)x[i] * y[i]; st[5 * J + j] += (double)y[i] * y[i];
  }
  for (int j = 0; j < J; ++j) {
    M->group_idx[j] = 3 + j; M->group_idx[J + j] = 3 + J + j;
    M->n[j] = (float)st[j]; M->sx[j] = (float)st[J + j]; M->sy[j] = (float)st[2 * J + j];
    M->sxx_j[j] = (float)st[3 * J + j]; M->sxy_j[j] = (float)st[4 * J + j]; M->syy_j[j] = (float)st[5 * J + j];
    M->u[j] = u[j];
  }
  M->logp_const = -(3.0 + 2.0 * J + N) * HALF_LOG_2PI;
  free(st);
  return M;
}

orc_model* orc_funnel_create(void) {
  orc_model* M = (orc_model*)calloc(1, sizeof(orc_model));
  M->model = 5; M->D = 2; M->n_glob = 1; M->n_groups = 1; M->n_local_parts = 1;
  M->glob_idx[0] = 0;
  M->group_idx = (int*)malloc(sizeof(int));
  M->group_idx[0] = 1;
  M->logp_const = -2.0 * HALF_LOG_2PI - log(3.0);
  return M;
}

orc_model* orc_schools_create(const float* y, const float* sigma) {
  orc_model* M = (orc_model*)calloc(1, sizeof(orc_model));
  M->model = 0; M->D = 10; M->n_glob = 2; M->n_groups = 8; M->n_local_parts = 1;
  M->glob_idx[0] = 0; M->glob_idx[1] = 1;
  M->group_idx = (int*)malloc(sizeof(int) * 8);
  M->u = (float*)malloc(sizeof(float) * 8); M->y = (float*)malloc(sizeof(float) * 8);
  double c = -18.0 * HALF_LOG_2PI - 2.0 * log(5.0);
  for (int k = 0; k < 8; ++k) { M->group_idx[k] = 2 + k; M->u[k] = sigma[k]; M->y[k] = y[k]; c -= log((double)sigma[k]); }
  M->logp_const = c;
  return M;
}

orc_model* orc_election_create(int N, int S, const int32_t* state, const float* female,
                               const float* black, const float* y) {
  orc_model* M = (orc_model*)calloc(1, sizeof(orc_model));
  M->model = 3; M->S = S; M->N = N; M->D = S + 4;
  M->n_glob = 4; M->n_groups = S + 1; M->n_local_parts = 1; M->mom_spec = 1;
  M->glob_idx[0] = 0; M->glob_idx[1] = 1; M->glob_idx[2] = 2 + S; M->glob_idx[3] = 3 + S;
  M->group_idx = (int*)malloc(sizeof(int) * (S + 1));
  for (int t = 0; t < S; ++t) M->group_idx[t] = 2 + t;
  M->group_idx[S] = -1;
  M->cell_n = (float*)calloc((size_t)(S + 1) * 4, sizeof(float));
  M->cell_y = (float*)calloc((size_t)(S + 1) * 4, sizeof(float));
  for (int i = 0; i < N; ++i) {
    int t = state[i];
    if (t < 0 || t >= S) t = S;                 /* tf.one_hot zero row, models.py:978 */
    int fk = (female[i] != 0.0f ? 1 : 0) + (black[i] != 0.0f ? 2 : 0);
    M->cell_n[t * 4 + fk] += 1.0f;
    M->cell_y[t * 4 + fk] += y[i];
  }
  M->logp_const = -(4.0 + S) * HALF_LOG_2PI - 3.0 * log(100.0) - log(10.0);
  return M;
}

/* reference models.py:1011-1046.  Parts: mua[G], sigma_y[G], a[P], b[G]. */
orc_model* orc_electric_create(int N, int P, int G, const int32_t* pair, const int32_t* grade,
                               const int32_t* grade_pair, const float* treatment, const float* y) {
  if (G > 4) return NULL;
  orc_model* M = (orc_model*)calloc(1, sizeof(orc_model));
  M->model = 6; M->N = N; M->P = P; M->G = G; M->D = 3 * G + P;
  /* RNG stream layout: the 3G grade-level scalars are replicated, `a` is sliced; the extra group P
   * stands for the observations whose pair index falls outside the one-hot (no latent) */
  M->n_glob = 3 * G; M->n_groups = P + 1; M->n_local_parts = 1;
  for (int k = 0; k < G; ++k) { M->glob_idx[k] = k; M->glob_idx[G + k] = G + k; M->glob_idx[2 * G + k] = 2 * G + P + k; }
  M->group_idx = (int*)malloc(sizeof(int) * (P + 1));
  for (int j = 0; j < P; ++j) M->group_idx[j] = 2 * G + j;
  M->group_idx[P] = -1;
  M->pair = (int*)malloc(sizeof(int) * N); M->grade = (int*)malloc(sizeof(int) * N);
  M->grade_pair = (int*)malloc(sizeof(int) * P);
  M->treat = (float*)malloc(sizeof(float) * N); M->y = (float*)malloc(sizeof(float) * N);
  for (int i = 0; i < N; ++i) { M->pair[i] = pair[i]; M->grade[i] = grade[i]; M->treat[i] = treatment[i]; M->y[i] = y[i]; }
  for (int j = 0; j < P; ++j) M->grade_pair[j] = grade_pair[j];
  M->logp_const = -(double)(M->D + N) * HALF_LOG_2PI - G * log(100.0);
  return M;
}

/* reference models.py:1069-1141.  Parts: sigma_alpha, sigma_mu, alpha0, mu0, ..., alpha{T-1}, mu{T-1}, beta. */
orc_model* orc_time_series_create(int T, const float* x, const float* y) {
  orc_model* M = (orc_model*)calloc(1, sizeof(orc_model));
  M->model = 7; M->N = T; M->D = 3 + 2 * T;
  /* RNG stream layout: sigma_alpha, sigma_mu, beta are replicated; the 2T trend latents are one
   * part sliced in consecutive runs (slot s owns elements s*per_lane .. (s+1)*per_lane-1) */
  M->n_glob = 3; M->n_groups = 2 * T; M->n_local_parts = 1; M->contig = 1; M->lane_unit = 2;
  M->glob_idx[0] = 0; M->glob_idx[1] = 1; M->glob_idx[2] = 2 + 2 * T;
  M->group_idx = (int*)malloc(sizeof(int) * 2 * T);
  for (int j = 0; j < 2 * T; ++j) M->group_idx[j] = 2 + j;
  M->u = (float*)malloc(sizeof(float) * T); M->y = (float*)malloc(sizeof(float) * T);
  for (int t = 0; t < T; ++t) { M->u[t] = x[t]; M->y[t] = y[t]; }
  M->logp_const = -(double)(M->D + T) * HALF_LOG_2PI - T * log(0.12);
  return M;
}

orc_model* orc_german_create(int N, int F, const float* X, const float* y) {
  orc_model* M = (orc_model*)calloc(1, sizeof(orc_model));
  M->model = 2; M->N = N; M->F = F; M->D = 1 + 2 * F;
  /* RNG stream layout: overall_log_scale is the top-level scalar; feature d owns
   * beta_log_scales[d] and beta[d] (two sliced parts, see draw_momentum) */
  M->n_glob = 1; M->n_groups = F; M->n_local_parts = 2; M->contig = 1; M->glob_idx[0] = 0;
  M->group_idx = (int*)malloc(sizeof(int) * 2 * F);
  for (int d = 0; d < F; ++d) { M->group_idx[d] = 1 + d; M->group_idx[F + d] = 1 + F + d; }
  M->X = (float*)malloc(sizeof(float) * (size_t)N * F); memcpy(M->X, X, sizeof(float) * (size_t)N * F);
  M->y = (float*)malloc(sizeof(float) * N); memcpy(M->y, y, sizeof(float) * N);
  M->logp_const = -(1.0 + 2.0 * F) * HALF_LOG_2PI - log(10.0);
  return M;
}

void orc_model_destroy(orc_model* M) {
  if (!M) return;
  free(M->group_idx); free(M->n); free(M->sx); free(M->sy); free(M->u);
  free(M->y); free(M->X); free(M->cell_n); free(M->cell_y);
  free(M->sxx_j); free(M->sxy_j); free(M->syy_j);
  free(M->pair); free(M->grade); free(M->grade_pair); free(M->treat);
  free(M);
}
int orc_model_dim(const orc_model* M) { return M->D; }
double orc_model_logp_const(const orc_model* M) { return M->logp_const; }

/* elements of the sliced axis a slot owns (the RNG stream partition of draw_momentum and the VI draws) */
static int orc_per_lane(const orc_model* M, int lanes) {
  int per_lane = (M->n_groups + lanes - 1) / lanes;
  if (M->lane_unit > 1) per_lane = (per_lane + M->lane_unit - 1) / M->lane_unit * M->lane_unit;
  return per_lane;
}

#define REAL float
#define FN(x) x##_f32
#include "oracle_impl.h"
#undef REAL
#undef FN
#define REAL double
#define FN(x) x##_f64
#include "oracle_impl.h"
#undef REAL
#undef FN
