/*
 * autoreparam.h -- C ABI of the MI355X-native HMC / mean-field-VI engine.
 *
 * This is the drop-in boundary for the hot path of mgorinova/autoreparam.  The
 * reference has NO native interface for this path (it is pure Python on
 * TensorFlow-Probability); each entry point below names the
 * reference call it replaces so a maintainer can bind it with ctypes
 * (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain C types only; every array pointer is a DEVICE pointer (HBM, gfx950)
 *     unless the parameter name ends in `_host`;
 *   - the caller owns every buffer; the library never frees caller memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *     calls only enqueue work, they never synchronise (one exception: arp_vi_run, see there);
 *   - every function returns 0 on success, non-zero on failure, and
 *     arp_last_error() then returns a human readable message (thread local);
 *   - chain states use the reference layout: the model's latent parts,
 *     flattened and concatenated in trace order, one row per chain,
 *     row-major float32 `[C][D]`  (reference: list of `[C,*event]` tensors,
 *     inference.py:207-216).
 *   - threads: a handle is used by one thread at a time; different handles (same device or not) may be
 *     driven from different threads concurrently, each on a stream of its own -- results do not depend on
 *     it.  arp_vi_run's multi-workgroup launches need their groups resident together: the library runs
 *     them one at a time per process; across processes sharing a device, or beside another kernel that
 *     holds the device for seconds, a hand-off that waits 2 s makes the call fail instead of hanging.
 */
#ifndef AUTOREPARAM_H_
#define AUTOREPARAM_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARP_ABI_VERSION 2

/* Models on the hot path (reference models.py:131-166, 809-857, 884-923, 967-1008). */
enum {
  ARP_MODEL_EIGHT_SCHOOLS = 0,
  ARP_MODEL_RADON = 1,
  ARP_MODEL_GERMAN_CREDIT = 2, /* german_credit_lognormalcentered */
  ARP_MODEL_ELECTION = 3,
  ARP_MODEL_NEALS_FUNNEL = 5, /* models.py:671-696; no dataset fields are read */
  ARP_MODEL_RADON_STDDVS = 4, /* radon with per-county observation scales, models.py:763-806 */
  ARP_MODEL_ELECTRIC = 6,     /* electric company, models.py:1011-1066 */
  ARP_MODEL_TIME_SERIES = 7   /* local linear trend, models.py:1069-1141: n_obs = T, x = years, y = series */
};

/* Step-size adaptation wrapped around the HMC transition. */
enum {
  ARP_ADAPT_NONE = 0,   /* fixed step (parity runs) */
  ARP_ADAPT_DUAL = 1,   /* tfp.mcmc.DualAveragingStepSizeAdaptation defaults, inference.py:224-226 */
  ARP_ADAPT_SIMPLE = 2  /* tfp.mcmc.SimpleStepSizeAdaptation(rate, target), inference.py:288-306 */
};

/* Raw model inputs exactly as the reference's `model_args` / `observed_data`
 * hold them (HOST pointers; copied at create time).  Unused fields are NULL/0. */
typedef struct arp_dataset {
  int32_t model;        /* ARP_MODEL_* */
  int32_t n_obs;        /* N */
  int32_t n_groups;     /* radon: J counties; election: n_state; schools: 8; electric: n_pair */
  int32_t n_features;   /* german: 62; electric: n_grade (= n_grade_pair, at most 4) */
  const int32_t* group_host;   /* [N] radon county (0-based) / election state / electric pair (both 1-based, as the reference feeds tf.one_hot) */
  const float* u_host;         /* radon: [J] log uranium; schools: [8] treatment stddevs */
  const float* x_host;         /* radon: [N] floor; election: [N] female; electric: [N] treatment (0/1) */
  const float* x2_host;        /* election: [N] black */
  const float* y_host;         /* [N] observations (Bernoulli outcomes as 0/1 floats); schools: [8] effects */
  const float* X_host;         /* german: [N][n_features] row-major design matrix */
  const int32_t* group2_host;  /* electric: [N] grade (1-based, as fed to tf.one_hot) */
  const int32_t* group3_host;  /* electric: [n_groups] grade_pair (1-based) */
} arp_dataset;

typedef struct arp_model arp_model; /* opaque: frozen data + sufficient statistics, on one device */

/* One segment of `n_steps` HMC transitions over `n_chains` chains (replaces one
 * stretch of tfp.mcmc.sample_chain's while-loop, inference.py:228-236). */
typedef struct arp_hmc_config {
  int32_t n_chains;          /* C on this device */
  int32_t n_leapfrog;        /* L */
  int32_t n_steps;           /* transitions in this call */
  int64_t step_base;         /* transitions already done for these chains (0 on the first call) */
  int64_t chain_offset;      /* global id of chain 0 on this device (RNG keying; multi-GPU sharding) */
  uint64_t seed;
  int32_t adapt_kind;        /* ARP_ADAPT_* */
  int32_t n_adapt;           /* num_adaptation_steps */
  float adapt_target;        /* target accept prob (0.75) */
  float adapt_rate;          /* SIMPLE: adaptation_rate (0.05) */
  /* trace schedule, sample_chain semantics: result s is taken after transition
   * 1 + n_burnin + s*thin (thin = 1 + num_steps_between_results). */
  int32_t n_burnin;
  int32_t thin;
  int32_t n_samples;         /* S: capacity of the trace buffers (rows) */
  int32_t trace_centered;    /* 1: trace rows are mapped to centred coordinates (inference.py:238-239) */
  int32_t lanes_per_chain;   /* 0 = library default; otherwise 1,2,4,8,16 (also partitions the RNG streams) */
  int32_t stats_batch;       /* io.stats: samples per batch of the batch-means accumulators (>= 1) */
  int32_t trace_chains;      /* 0 or >= n_chains: trace rows hold every chain; otherwise only chains [0, trace_chains)
                              * are recorded and a trace row is [trace_chains][D] (a few chains kept next to io.stats) */
  int32_t reserved;
} arp_hmc_config;

/* Per-chain persistent state ("kernel results") + outputs.  All DEVICE pointers. */
typedef struct arp_hmc_io {
  float* q;                  /* [C][D] in/out current state (reparameterised coordinates) */
  float* grad;               /* [C][D] in/out cached gradient at q (ignored when step_base == 0) */
  float* logp;               /* [C]    in/out cached log-density at q (additive constants dropped) */
  float* adapt;              /* [C][4] in/out {step multiplier kappa, error sum, log-averaged multiplier, unused} */
  uint32_t* rng;             /* [C][16][4] in/out stream states, one 16-byte record per RNG slot (MWC64X: x, carry, 0, 0) */
  uint32_t* accept_count;    /* [C] in/out accepted transitions */
  const float* eps0;         /* [D] base step size per element (VI posterior std / (L/4)^2, inference.py:212-216) */
  float* trace;              /* [S][C][D] or NULL */
  uint8_t* trace_accept;     /* [S][C] is_accepted of recorded transitions, or NULL */
  float* stats;              /* [6][C][D] in/out or NULL: streaming statistics of the recorded samples (same schedule and
                              * coordinates as the trace rows), accumulated in the kernel so that a run needs no trace:
                              * {ref = a reference level near the samples, s1 = sum (x - ref), s2 = sum (x - ref)^2, cur = s1 at
                              * the start of the current batch, sb1 = sum of batch means (of x - ref), sb2 = sum of their
                              * squares}; zero it before the first call.  mean = ref + s1/n, var = (s2 - s1^2/n)/(n-1),
                              * batch-means ESS = n var / (stats_batch var(batch means)).  ref is chosen by
                              * the kernel on the first call that records and is only meaningful through these formulas (the
                              * first recorded sample, or -- kernels that keep the running moments of a batch in LDS and touch
                              * the planes once per batch or launch -- the mean of the first stretch they fold); the planes are
                              * sums, so they are exact to rounding however a run is cut into launches */
  uint32_t* rec_accept_count; /* [C] in/out or NULL: accepted transitions among the recorded ones (sum of is_accepted) */
} arp_hmc_io;

int arp_version(void);
const char* arp_last_error(void);

/* Build a model handle on the current HIP device: copies the data, derives the
 * per-group sufficient statistics and uploads them. */
int arp_model_create(const arp_dataset* data, arp_model** out);
int arp_model_destroy(arp_model* m);
int arp_model_dim(const arp_model* m);                 /* D */
/* Per-handle options (no reference counterpart).  "german_math": how german credit's likelihood contraction runs at 4 lanes
 * per chain -- "f32" = f32 matrix cores (v_mfma_f32_16x16x4_f32, exact f32 products), "bf16x3" = bf16 matrix cores with every
 * operand as three bf16 pieces (six leading cross products: f32-equivalent, error ~ 2^-23 per product; needs a design matrix
 * with at most 8 columns that are not exact in one bf16 piece -- the reference's data have 7), "auto" (default) = bf16x3 where
 * the data allow it.  "vi_launch": how arp_vi_run starts a launch whose workgroups wait for each other inside the launch --
 * "cooperative" = hipLaunchCooperativeKernel (the runtime checks the grid against the device's capacity and serialises such
 * launches of the process), "plain" = an ordinary launch sized by an occupancy query, one such launch at a time per process,
 * "auto" (default) = cooperative where the device supports it, plain if the runtime refuses the grid.  (Kernels of other
 * queues or processes can still keep a group from being resident together: arp_vi_run retakes a launch whose hand-offs timed
 * out -- arp_vi_attempts.)  Returns non-zero for an unknown key / value or a model the key does not apply to. */
int arp_model_set_option(arp_model* m, const char* key, const char* value);
/* Additive constant dropped from logp for parameterisation `which` (so callers can
 * report the reference-valued target_log_prob / ELBO): logp_ref = logp + const. */
double arp_model_logp_const(const arp_model* m, int which);

/* Set the parameterisation `which` (0 = primary, 1 = secondary, used by the
 * interleaved kernel): per-element VIP parameters a,b (HOST, [D]).  a=b=1 is CP,
 * a=b=0 is NCP (program_transformations.py:555-600). */
int arp_model_set_param(arp_model* m, int which, const float* a_host, const float* b_host);

/* target_log_prob_fn + its gradient for a batch of chains
 * (vectorize_log_joint_fn, inference.py:172-195 + tf.gradients inside TFP). */
int arp_logp_grad(arp_model* m, int which, const float* x, int n_chains,
                  float* logp, float* grad, int lanes_per_chain, void* stream);

/* State converters (models.py:56-128): dir 0: parameterisation `which` -> centred,
 * dir 1: centred -> parameterisation `which`. */
int arp_transform(arp_model* m, int which, int dir, const float* in, int n_chains,
                  float* out, void* stream);

/* HMC segment (mcmc.HamiltonianMonteCarlo + step-size adaptation + sample_chain): `cfg->n_steps` transitions in ONE
 * launch.  (Internally a launch of 256 steps or more may hand its chains from workgroup to workgroup a few times -- DESIGN.md
 * section 3, relay segments --; chain state, counters and trace rows are bit for bit those of one workgroup per chain block.
 * The streaming statistics in io->stats are the exception: their partial-batch accumulators fold at every segment end, so
 * the sums are grouped differently and agree to float rounding only (~1e-6 relative), and the segment count follows the
 * device's CU count.  The call stays asynchronous; the hand-over waits are bounded, and a launch whose hand-over timed out
 * leaves the chains where they were and is reported by the next call on the handle or by arp_model_check.) */
int arp_hmc_run(arp_model* m, int which, const arp_hmc_config* cfg,
                const arp_hmc_io* io, void* stream);

/* Interleaved CP/NCP segment (interleaved.Interleaved.one_step, interleaved.py:113-155):
 * parameterisation 0 then 1 per step, each with its own leapfrog count, base
 * step sizes and adaptation state; `q` is kept in parameterisation-0 coordinates. */
typedef struct arp_interleaved_io {
  arp_hmc_io k0;             /* q/rng/trace live here; grad/logp are optional: kernels that carry the gradient across the
                                change of coordinates keep it there between calls (NULL: re-bootstrap at every call) */
  float* adapt1;             /* [C][4] adaptation state of kernel 1 */
  uint32_t* accept_count1;   /* [C] */
  const float* eps0_1;       /* [D] */
  uint8_t* trace_accept1;    /* [S][C] or NULL */
  uint32_t* rec_accept_count1; /* [C] in/out or NULL, as k0.rec_accept_count for kernel 1 */
} arp_interleaved_io;
int arp_interleaved_run(arp_model* m, const arp_hmc_config* cfg, int n_leapfrog_1,
                        const arp_interleaved_io* io, void* stream);

/* Mean-field VI (find_best_learning_rate, inference.py:26-154 on top of
 * util.get_mean_field_elbo, util.py:232-268): runs `n_lr` independent Adam
 * optimisations of `n_steps` steps in ONE launch (several when the learning rates do not all fit on the device at
 * once).  Every learning rate's `n_mc` draws are spread over a GROUP of workgroups that exchange their partial
 * gradient sums twice per step inside the launch and add them in a fixed order: a fit is bitwise reproducible, and
 * its draws do not depend on how the group is shaped.  The call is SYNCHRONOUS on `stream` when a group has more than
 * one workgroup (it waits for the launch to check that no in-launch hand-off timed out) and returns non-zero if one
 * did.  It keeps a hand-off workspace on the handle (freed by arp_model_destroy). */
typedef struct arp_vi_config {
  int32_t n_lr;              /* number of learning rates */
  int32_t n_steps;           /* num_optimization_steps */
  int32_t n_mc;              /* num_mc_samples (<= 4096) */
  int32_t learn_a;           /* 1: also optimise the VIP parameter a = sigmoid(w) (cVIP) */
  int32_t tied_b;            /* 1: b := a in the density (tied_pparams as intended); 0: b from set_param, or learned via io.wb */
  int32_t a_prior;           /* 0: none; 1: the reference's --discrete_prior on the learnable parameters (main.py:244-253):
                              * Mixture(logits (0,5,0); Laplace(0,0.1), Uniform(0,1), Laplace(1,0.1)), its log density
                              * added to the objective (inference.py:50-54) */
  uint64_t seed;
} arp_vi_config;
typedef struct arp_vi_io {
  const float* lr;           /* [n_lr] base learning rates */
  float* loc;                /* [n_lr][D] in: initial loc, out: final */
  float* rho;                /* [n_lr][D] in: initial pre-softplus scale, out: final */
  float* w;                  /* [n_lr][D] in/out unconstrained a (only if learn_a) */
  float* wb;                 /* [n_lr][D] in/out unconstrained b, learned separately (untied), or NULL */
  float* elbo;               /* [n_lr][n_steps] ELBO estimate per step (reference-valued, constants included) */
  float* prior;              /* [n_lr][n_steps] or NULL: log density of the a_prior on the learnable parameters at each step's
                              * values (the reference ranks learning rates on elbo + prior and reports elbo, inference.py:50-54,
                              * 120-150) */
  const int32_t* a_group;    /* [D] or NULL.  With untied parameters the reference gives `a` the shape of the variable's loc
                              * (program_transformations.py:486-493, 507-510): a vector variable whose loc is a scalar (german
                              * beta_log_scales, election a) learns ONE shared a.  a_group[d] = index of the first element of
                              * d's group (d itself when its a is its own); groups are contiguous.  DEVICE pointer, like every
                              * array here; arp_vi_run copies it back once and rejects a map that is not of that form */
  const int32_t* b_group;    /* the same for the separately learned b (shape of the variable's scale; only read with wb) */
} arp_vi_io;
int arp_vi_run(arp_model* m, int which, const arp_vi_config* cfg, const arp_vi_io* io, void* stream);
/* Measurement hook: the shape the calling thread's last arp_vi_run gave its launch -- out6 = {threads per workgroup,
 * sample groups G, row parts R (workgroups per learning rate = G x R), learning rates per launch, workgroups resident
 * together (= CUs in use when one fits per CU), workgroups of the kernel one CU holds}.  (bench.py: vi_kernel.) */
int arp_vi_geometry(int32_t* out6);
/* Measurement hook: how many launches the calling thread's last arp_vi_run needed for its slowest chunk of learning rates
 * (1 = every hand-off went through at once; a launch whose in-launch hand-offs ran into their bound -- the device was shared
 * for seconds -- is taken again from the parameters it started with, with 4 x the bound: 2 s, 8 s, 32 s, then an error). */
int arp_vi_attempts(int32_t* out1);

/* Deferred status of the handle's asynchronous launches: 0 if none failed; non-zero (and arp_last_error set) if a relay
 * hand-over inside an arp_hmc_run / arp_interleaved_run launch timed out since the last check -- the chains of that launch
 * were left partly advanced and must be discarded.  Reads a pinned host word: call it after synchronising the stream(s) the
 * launches went to.  Reports once (the word is cleared).  The same word is checked on entry to the next chain launch. */
int arp_model_check(arp_model* m);

/* Measurement hook: what the calling thread's last arp_hmc_run / arp_interleaved_run launch did -- out3 = {relay segments
 * the launch's steps were cut into (1 = one workgroup per chain block), chain blocks, workgroups of
 * the kernel one CU holds (0 where the question did not arise)}. */
int arp_relay_geometry(int32_t* out3);

/* Effective sample size of every series of a recorded trace (replaces tfp.mcmc.effective_sample_size with its
 * defaults, inference.py:240, 327): `trace` holds n_samples rows of `row_stride` floats, series i is column i
 * (i < n_series, e.g. n_series = C*D of a [S][C][D] trace); ess[i] = S / (-1 + 2 sum_k (S-k)/S rho_k) with the
 * auto-correlations cut at the first negative one.  A constant series gives NaN (0/0), as the reference's does. */
int arp_ess(const float* trace, int64_t n_samples, int64_t n_series, int64_t row_stride, float* ess, void* stream);

/* The same statistic with a caller-owned device workspace for LONG traces (the [S = 50 000][k D] trace of a streaming
 * run's kept chains, inference.py:238-240 at main.py:101-110's default schedule): series that are still positively
 * correlated after the coalesced sweeps (48 lags) are copied series-major into the workspace and finished on the matrix
 * cores (csrc/ess_tail.h), instead of per-lane sweeps that re-read the trace 16 lags at a time (arp_ess: 5.5 s on the
 * german-credit trace, this: see profiles/).  arp_ess_workspace_bytes is the size at which every series fits at once
 * (0 when n_samples is short enough for the one-kernel path); a smaller workspace works (the listed series are taken in
 * chunks) as long as the work lists (28 bytes per series) and 64 series rows fit.  256-byte aligned. */
int64_t arp_ess_workspace_bytes(int64_t n_samples, int64_t n_series);
int arp_ess_ws(const float* trace, int64_t n_samples, int64_t n_series, int64_t row_stride, float* ess,
               void* workspace, int64_t workspace_bytes, void* stream);

/* Test hook: the step-size adaptation recurrence of the chain kernels on SCRIPTED log acceptance ratios
 * (tfp.mcmc.DualAveragingStepSizeAdaptation / SimpleStepSizeAdaptation as wired at inference.py:224-226, 288-306;
 * SURVEY.md 8c known answer (7)).  For each of `n` independent rows, applies the update after transitions
 * cfg->step_base + 1 ... cfg->step_base + n_steps with log_accept[s][row]; `adapt` is [n][4] in/out as in arp_hmc_io;
 * kappa_out ([n_steps][n] or NULL) receives the multiplier in force after each update.  Only the adapt_* fields,
 * step_base and n_steps of cfg are read. */
int arp_adapt_probe(const arp_hmc_config* cfg, const float* log_accept, int n, float* adapt, float* kappa_out,
                    void* stream);

/* Measurement hook (no reference counterpart): the shader clock the chip holds under a vector-bound load.  Runs packed FMAs
 * on every SIMD (two resident waves each) for `iters` loop iterations of 64 instructions (~ 0.27 us each) while one wave
 * reads s_memtime (shader cycles) and s_memrealtime (100 MHz) around its loop; cycles_ticks (device, 3 x uint64) receives
 * {shader cycles, 100 MHz ticks, 1}.  bench.py reports cycles / ticks x 0.1 GHz next to every roofline figure. */
int arp_clock_probe(int iters, unsigned long long* cycles_ticks, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AUTOREPARAM_H_ */
