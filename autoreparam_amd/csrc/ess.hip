// Effective sample size of every (chain, element) series of a trace, tfp.mcmc.effective_sample_size
// semantics with its defaults (reference inference.py:240, 327; restated from the published definition):
//   rho_k = c_k / c_0,  c_k = sum_{t < S-k} (x_t - m)(x_{t+k} - m) / (S - k),
//   every lag from the first negative rho on is dropped,  ESS = S / (-1 + 2 sum_k (S - k)/S rho_k).
// The autocovariances are formed directly and only until the first negative one -- no FFT, no work buffers,
// no plan creation.  One thread per series; consecutive threads read consecutive floats of a trace
// row, so every pass streams the trace coalesced.  Sums are accumulated in double.
// The first 16 lags come out of ONE pass over the series, without a mean pass in front of it: with y_t = x_t - r
// (r = the mean of the first 16 samples, so that y is small) and m' = mean(y),
//   sum_{t>=k} (y_t - m')(y_{t-k} - m') = sum_{t>=k} y_t y_{t-k} - m' (2 T - head_k - tail_k) + (S - k) m'^2,
// T = sum_t y_t, head_k / tail_k = the sums of the first / last k values (two 16-sample loops).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "host_common.h"

namespace arp {

#ifndef ARP_ESS_WIN
#define ARP_ESS_WIN 16
#endif
constexpr int kEssWin = ARP_ESS_WIN;      // lags 1..kEssWin come out of the first pass over the series

// One sweep over a series forms kEssWin consecutive auto-covariance sums at once, lags kb+1 .. kb+16 (and c_0 when
// kb == 0): the 16 mean-removed values y_{t-kb-1} .. y_{t-kb-16} sit in a register window that is addressed at compile
// time (the time loop is unrolled by the window length, so nothing is ever shifted), 17 accumulators next to it.  Chains
// that mix stop within a few dozen lags, so with the mean pass almost every series is done after two coalesced sweeps of
// the trace; a series whose first 16 auto-correlations are all positive takes another sweep per 16 lags.
// Loads are issued in batches of 16 independent rows (a thread's consecutive samples are a whole trace row apart: one
// load per iteration with its wait is latency bound -- the first version of this kernel was, at 29 ms for 18.6 GB).
// Accumulation is in float, flushed into doubles every 8 windows = 128 samples (v_fma_f64 issues several times slower
// than v_fma_f32 and made the sweep compute bound); no float sum is longer than 128 products.  Accuracy: a float sum of
// 128 products carries <= 128 x 2^-24 ~ 8e-6 relative error on sum |y y'|; the first sweep centres afterwards from raw
// sums about the reference level r, so with |mean(y)| = m sd the raw sums are (1 + m^2) times the centred ones and rho
// carries up to ~ (1 + m^2) x 1e-5 absolute error.  The sweep is retaken around the mean itself when m^2 > 16 (below;
// a whole WAVE retakes it when one of its 64 series does, so the threshold sits where a stationary chain's first 16
// samples practically never land), which bounds that at 2e-4 -- two orders under the 1 / sqrt(S) sampling noise of rho
// at any S this path sees.
template <bool FIRST>
__device__ __forceinline__ void ess_sweep(const float* __restrict__ x, long long S, long long stride, float mean,
                                          long long kb, double (&dacc)[kEssWin + 1], double& dtot) {
  float acc[kEssWin + 1];
  float tot = 0.0f;
  dtot = 0.0;
  float w[kEssWin];          // w[tt] = y at time t0 + tt - kb of the previous window (0 before the series starts)
#pragma unroll
  for (int j = 0; j <= kEssWin; ++j) { acc[j] = 0.0f; dacc[j] = 0.0; }
#pragma unroll
  for (int j = 0; j < kEssWin; ++j) w[j] = 0.0f;
  // window t0: times t0 .. t0+15 of the leading stream; the lagged stream runs kb behind.  Start where the lagged
  // stream starts (t0 = kb, rounded down to a window): products with times before 0 are zeros.
  for (long long t0 = 0; t0 < S; t0 += kEssWin) {
    float xv[kEssWin], xl[kEssWin];
    if (t0 + kEssWin <= S) {
#pragma unroll
      for (int tt = 0; tt < kEssWin; ++tt) xv[tt] = x[(t0 + tt) * stride];
    } else {   // the last, partial window: the mean past the end, i.e. y = 0, which adds nothing
#pragma unroll
      for (int tt = 0; tt < kEssWin; ++tt) xv[tt] = t0 + tt < S ? x[(t0 + tt) * stride] : mean;
    }
    if (!FIRST) {
      if (t0 - kb >= 0 && t0 - kb + kEssWin <= S) {
#pragma unroll
        for (int tt = 0; tt < kEssWin; ++tt) xl[tt] = x[(t0 + tt - kb) * stride];
      } else {
#pragma unroll
        for (int tt = 0; tt < kEssWin; ++tt) {
          const long long tl = t0 + tt - kb;
          xl[tt] = (tl >= 0 && tl < S) ? x[tl * stride] : mean;
        }
      }
    }
#pragma unroll
    for (int tt = 0; tt < kEssWin; ++tt) {
      const float y = xv[tt] - mean;
      if (FIRST) { acc[0] = fmaf(y, y, acc[0]); tot += y; }
#pragma unroll
      for (int j = 1; j <= kEssWin; ++j) acc[j] = fmaf(y, w[(tt - j + 2 * kEssWin) % kEssWin], acc[j]);
      w[tt] = FIRST ? y : xl[tt] - mean;
    }
    if (((t0 / kEssWin) & 7) == 7 || t0 + kEssWin >= S) {
#pragma unroll
      for (int j = 0; j <= kEssWin; ++j) { dacc[j] += (double)acc[j]; acc[j] = 0.0f; }
      if (FIRST) { dtot += (double)tot; tot = 0.0f; }
    }
  }
}

__global__ __launch_bounds__(256) void ess_kernel(const float* __restrict__ trace, long long S, long long n,
                                                  long long stride, float* __restrict__ ess) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* x = trace + i;
  // reference level r and the sums of the first / last k values (k <= 16)
  const int nh = S < kEssWin ? (int)S : kEssWin;
  float fh[kEssWin], ft[kEssWin];
#pragma unroll
  for (int j = 0; j < kEssWin; ++j) {
    fh[j] = j < nh ? x[(long long)j * stride] : 0.0f;
    ft[j] = j < nh ? x[(S - 1 - j) * stride] : 0.0f;     // ft[j] = x_{S-1-j}
  }
  float r = 0.0f;   // x_0 + mean(x_j - x_0): exactly x_0 for a constant series
#pragma unroll
  for (int j = 1; j < kEssWin; ++j) r += j < nh ? fh[j] - fh[0] : 0.0f;
  r = fh[0] + r / (float)nh;

  double dacc[kEssWin + 1], T, mp, c0;
  float mean;
  for (int attempt = 0;; ++attempt) {
    ess_sweep<true>(x, S, stride, r, 0, dacc, T);
    mp = T / (double)S;                     // mean of y
    mean = r + (float)mp;                   // mean of x, for the sweeps past lag 16
    c0 = (dacc[0] - (double)S * mp * mp) / (double)S;
    // A reference level far from the mean (a series that was still drifting over its first 16 samples) makes the
    // products large against the variance they are meant to resolve: take the pass again around the mean itself.
    if (attempt == 1 || !(mp * mp > 16.0 * c0)) break;
    r = mean;
  }
  // centred sums of lags 1 .. 16
  double head = 0.0, tail = 0.0;            // sums of the first / last k values of y
#pragma unroll
  for (int j = 1; j <= kEssWin; ++j) {
    head += (double)(fh[j - 1] - r); tail += (double)(ft[j - 1] - r);
    dacc[j] = dacc[j] - mp * (2.0 * T - head - tail) + (double)(S - j) * mp * mp;
  }
  if (!(c0 > 0.0)) { ess[i] = __builtin_nanf(""); return; }   // constant series: 0/0 as in the FFT form
  double total = 1.0;   // lag 0: (S - 0)/S * rho_0
  bool done = false;
  for (long long kb = 0; kb < S && !done; kb += kEssWin) {
    if (kb > 0) { double unused; ess_sweep<false>(x, S, stride, mean, kb, dacc, unused); }
#pragma unroll
    for (int j = 1; j <= kEssWin; ++j) {
      const long long k = kb + j;
      if (done || k >= S) { done = true; break; }
      const double rho = dacc[j] / (double)(S - k) / c0;
      if (rho < 0.0) { done = true; break; }
      total += (double)(S - k) / (double)S * rho;
    }
  }
  ess[i] = (float)((double)S / (-1.0 + 2.0 * total));
}

}  // namespace arp

extern "C" int arp_ess(const float* trace, int64_t n_samples, int64_t n_series, int64_t row_stride, float* ess,
                       void* stream) {
  using namespace arp;
  if (!trace || !ess || n_samples <= 0 || n_series <= 0 || row_stride < n_series) {
    set_error("arp_ess: trace/ess, n_samples > 0, n_series > 0 and row_stride >= n_series are required");
    return 1;
  }
  const long long blocks = (n_series + 255) / 256;
  hipLaunchKernelGGL(ess_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, trace, (long long)n_samples,
                     (long long)n_series, (long long)row_stride, ess);
  ARP_HIP_OK(hipGetLastError());
  return 0;
}
