// Effective sample size of every (chain, element) series of a trace, tfp.mcmc.effective_sample_size
// semantics with its defaults (reference inference.py:240, 327; restated from the published definition):
//   rho_k = c_k / c_0,  c_k = sum_{t < S-k} (x_t - m)(x_{t+k} - m) / (S - k),
//   every lag from the first negative rho on is dropped,  ESS = S / (-1 + 2 sum_k (S - k)/S rho_k).
// Chains that mix stop within a few dozen lags, so the autocovariances are formed directly, eight
// lags per pass over the series and only until the first negative one -- no FFT, no work buffers,
// no plan creation.  One thread per series; consecutive threads read consecutive floats of a trace
// row, so every pass streams the trace coalesced.  Sums are accumulated in double.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "host_common.h"

namespace arp {

constexpr int kEssLags = 8;

__global__ __launch_bounds__(256) void ess_kernel(const float* __restrict__ trace, long long S, long long n,
                                                  long long stride, float* __restrict__ ess) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* x = trace + i;
  double m = 0.0;
  for (long long t = 0; t < S; ++t) m += (double)x[t * stride];
  const float mean = (float)(m / (double)S);
  double c0 = 0.0;
  for (long long t = 0; t < S; ++t) { const float d = x[t * stride] - mean; c0 = fma((double)d, (double)d, c0); }
  c0 /= (double)S;
  if (!(c0 > 0.0)) { ess[i] = __builtin_nanf(""); return; }   // constant series: 0/0 as in the FFT form
  double total = 1.0;   // lag 0: (S - 0)/S * rho_0
  bool done = false;
  for (long long k0 = 1; k0 < S && !done; k0 += kEssLags) {
    // lags k0 .. k0+7 in one pass: acc[j] = sum_t y_t * y_{t - k0 - j}
    double acc[kEssLags];
    float w[kEssLags];   // w[j] = y_{t - k0 - j}
#pragma unroll
    for (int j = 0; j < kEssLags; ++j) { acc[j] = 0.0; w[j] = 0.0f; }
    for (long long t = k0; t < S; ++t) {
      const float yt = x[t * stride] - mean;
#pragma unroll
      for (int j = kEssLags - 1; j > 0; --j) w[j] = w[j - 1];
      w[0] = x[(t - k0) * stride] - mean;
#pragma unroll
      for (int j = 0; j < kEssLags; ++j) acc[j] = fma((double)yt, (double)w[j], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < kEssLags; ++j) {
      const long long k = k0 + j;
      if (done || k >= S) break;
      const double rho = acc[j] / (double)(S - k) / c0;
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
