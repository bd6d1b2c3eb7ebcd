// Long series, slowly mixing: the lags past the first sweeps of arp_ess on the MATRIX cores.
//
// A trace of a streaming run's kept chains is [S = 50 000][k D] -- a series is 50 000 floats a whole row apart, and german
// credit at four leapfrogs mixes with an integrated auto-correlation time of ~ 400 samples: tfp's estimator (the sum of
// rho_k up to the first negative one, inference.py:238-240) needs ~ 1 500 lags of 128 000 series, 10^13 multiply-adds.
// Per-lane sweeps that read the leading and the lagged stream 16 lags at a time (ess.hip: ess_sweep_far, the path without
// a workspace) take 5.5 s on that trace.  Here instead:
//   1. ess_kernel hands every series still positive after its coalesced sweeps to a work list (EssDefer);
//   2. ess_gather_kernel copies the listed series series-major, mean removed: ws[p][t] = x[t][idx_p] - mean_p
//      (64 x 64 tiles through LDS: both sides move 256-byte runs);
//   3. ess_tail_kernel, one WAVE per series: the lag sums are blocks of a Toeplitz product,
//        D_j[m][n] = sum_kappa y[16 kappa + m] y[16 kappa + 16 j + n]        (a 16 x 16 block per 16 lags),
//      i.e. v_mfma_f32_16x16x4_f32 with A = 64 consecutive samples (lane l holds y[64 s + l]: ONE coalesced 256-byte
//      load) and B_j = the same series 16 j samples on.  Every product of a block is a distinct term of a lag sum
//      (lag 16 j + n - m: diagonal n - m of block j plus diagonal n - m - 16 of block j + 1), so the matrix pipe's 64
//      FLOP / clock / SIMD are all useful work -- the f32 MFMA is an exact fmaf chain, numerics as the vector form.
//      A round takes 16 blocks (256 lags, 64 accumulator registers); consecutive steps share 12 of their 16 B operands
//      (B_j(s + 1) = B_{j+4}(s)), so a step of 16 MFMAs (512 cycles) costs five 256-byte loads, L1 / L2 hits after the
//      first round.  After a round the diagonals are summed through a 16 x 17 LDS tile (fixed order: deterministic), the
//      first negative lag is found with a ballot, and the wave goes on only while every lag so far is positive.
// Bound: f32 MFMA issue, S x (lags taken, in rounds of 256) multiply-adds per series at 32 per clock and SIMD.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace arp {

typedef float ess_f4 __attribute__((ext_vector_type(4)));

// work list of the series a first-stage wave could not finish (device memory, caller's workspace)
struct EssDefer {
  unsigned* count;      // [1] listed series (zeroed before the first stage)
  unsigned* idx;        // [n_series] series index
  float* mean;          // [n_series] mean of the series
  double* c0;           // [n_series] lag-0 auto-covariance
  double* total;        // [n_series] sum_k (S - k) / S rho_k over the lags already taken
  int* from;            // [n_series] last lag already taken (a multiple of 16)
};

constexpr int kEssTailBlocks = 16;                       // 16-lag blocks per round
constexpr int kEssTailLags = 16 * kEssTailBlocks;        // lags completed per round
constexpr int kEssTailFlush = 1024;                      // 64-sample steps between folds of the float32 accumulators (multiple of 4)

__host__ __device__ inline long long ess_row_floats(long long S) { return (S + 63) & ~63ll; }

// ws[p - p0][t] = trace[t][idx[p]] - mean[p], t < S; zero up to the row's end.  One 256-thread workgroup per tile of 64
// listed series x 64 samples.
__global__ __launch_bounds__(256) void ess_gather_kernel(const float* __restrict__ trace, long long S, long long stride,
                                                         EssDefer W, unsigned p0, unsigned pmax, float* __restrict__ ws) {
  __shared__ float tile[64][65];
  const unsigned count = min(*W.count, pmax);
  const unsigned pb = p0 + blockIdx.x * 64u;
  if (pb >= count) return;
  const long long t0 = (long long)blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const unsigned p = pb + tx;
  const bool have = p < count;
  const unsigned col = have ? W.idx[p] : 0u;
  const float m = have ? W.mean[p] : 0.0f;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const long long t = t0 + ty + 4 * i;
    tile[ty + 4 * i][tx] = (have && t < S) ? trace[t * stride + col] - m : 0.0f;
  }
  __syncthreads();
  const long long SR = ess_row_floats(S);
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int pl = ty + 4 * i;
    if (pb + pl < count && t0 + tx < SR) ws[(long long)(pb - p0 + pl) * SR + t0 + tx] = tile[tx][pl];
  }
}

// one wave per listed series
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 4)))
void ess_tail_kernel(const float* __restrict__ ws, long long S, EssDefer W, unsigned p0, unsigned pmax, float* __restrict__ ess) {
  __shared__ float s_tile[16 * 17];
  __shared__ float s_lag[kEssTailLags + 32];
  const unsigned count = min(*W.count, pmax);
  const unsigned p = p0 + blockIdx.x;
  if (p >= count) return;
  const int lane = threadIdx.x;
  const long long SR = ess_row_floats(S);
  const float* row = ws + (long long)(p - p0) * SR;
  const double c0 = W.c0[p];
  double total = W.total[p];
  const int from = W.from[p];
  const int Si = (int)S;

  // 64 consecutive samples from sample `b` on, lane l the l-th; zero beyond the series (buffer range check)
  const int voff = lane * 4;
  auto ld = [&](int b) -> float {
    const int left = Si - b;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(row + (b < Si ? b : 0)), 0,
                                                                        left > 0 ? left * 4 : 0, 0x00020000);
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
  };

  for (int i = lane; i < kEssTailLags + 32; i += 64) s_lag[i] = 0.0f;
  bool stop = false;
  // round: blocks jb .. jb + 15; s_lag[i] holds the sum of lag 16 jb - 15 + i
  for (int jb = from / 16; !stop; jb += kEssTailBlocks) {
    ess_f4 acc[kEssTailBlocks];
#pragma unroll
    for (int i = 0; i < kEssTailBlocks; ++i) acc[i] = ess_f4{0.0f, 0.0f, 0.0f, 0.0f};
    const int lag0 = 16 * jb;
    const int ns = (Si - lag0 + 63) / 64;            // steps whose B operands are not all zero
    float w[16], a = ld(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = ld(lag0 + 16 * i);
    // The accumulators are float32 chains; they are folded into s_lag every kEssTailFlush steps (4 kEssTailFlush products
    // per element), so no chain is longer than that however long the series is (the workspace path admits 4.19 M samples:
    // one unbroken chain would be 262 144 products, a worst-case relative error of 1.6 % on sum |y y'|).  Series up to
    // 64 kEssTailFlush samples -- the reference's 50 000 among them -- take exactly one fold, as before.
    for (int s0 = 0; s0 < ns; s0 += kEssTailFlush) {
      const int s1 = min(ns, s0 + kEssTailFlush);
      for (int s = s0; s < s1; s += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int sb = 64 * (s + u);
          // the four new B operands of the next step and its A operand go in flight under this step's MFMAs
          float nw[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) nw[k] = ld(sb + lag0 + 16 * (16 + k));
          const float na = ld(sb + 64);
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w[(i + 4 * u) & 15], acc[i], 0, 0, 0);
#pragma unroll
          for (int k = 0; k < 4; ++k) w[(4 * u + k) & 15] = nw[k];
          a = na;
        }
      }
      // diagonals of the 16 blocks -> lag sums.  Block i, diagonal d = n - m: lag 16 (jb + i) + d -> s_lag[16 i + d + 15].
      // D layout of v_mfma_f32_16x16x4_f32: lane l, register r holds D[4 (l / 16) + r][l % 16].
#pragma unroll
      for (int i = 0; i < kEssTailBlocks; ++i) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4; ++r) s_tile[(4 * (lane >> 4) + r) * 17 + (lane & 15)] = acc[i][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < 31) {
          const int d = lane - 15;
          float sum = 0.0f;
#pragma unroll
          for (int m = 0; m < 16; ++m) {
            const int n = m + d;
            sum += (n >= 0 && n < 16) ? s_tile[m * 17 + (n & 15)] : 0.0f;
          }
          s_lag[16 * i + lane] += sum;
        }
        acc[i] = ess_f4{0.0f, 0.0f, 0.0f, 0.0f};
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // the round's complete lags, in order: s_lag[0 .. 255] = lags lag0 - 15 .. lag0 + 240 (lags <= `from` were taken by
    // the first stage); four per lane, first stop found with a ballot
    double part = 0.0;
    int my_stop = 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = lag0 - 15 + 4 * lane + q;
      const bool counted = k > from;
      const bool in_range = k < Si;
      const double rho = (double)s_lag[4 * lane + q] / (double)(in_range ? Si - k : 1) / c0;
      const bool st = counted && (!in_range || rho < 0.0);
      if (st && my_stop == 4) my_stop = q;
      if (counted && my_stop == 4) part += (double)(Si - k) / (double)Si * rho;
    }
    const unsigned long long sm = __ballot(my_stop < 4);
    const int first = sm ? __builtin_ctzll(sm) : 64;
    if (lane > first) part = 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    total += part;
    stop = first < 64;
    // carry the 15 incomplete lags (s_lag[256 .. 270]) to the front, clear the rest
    __builtin_amdgcn_wave_barrier();
    float carry = 0.0f;
    if (lane < 15) carry = s_lag[kEssTailLags + lane];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < kEssTailLags + 32; i += 64) s_lag[i] = i < 15 ? carry : 0.0f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (lane == 0) ess[W.idx[p]] = (float)((double)S / (-1.0 + 2.0 * total));
}

}  // namespace arp
