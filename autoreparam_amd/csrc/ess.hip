// Effective sample size of every (chain, element) series of a trace, tfp.mcmc.effective_sample_size
// semantics with its defaults (reference inference.py:240, 327; restated from the published definition):
//   rho_k = c_k / c_0,  c_k = sum_{t < S-k} (x_t - m)(x_{t+k} - m) / (S - k),
//   every lag from the first negative rho on is dropped,  ESS = S / (-1 + 2 sum_k (S - k)/S rho_k).
// The autocovariances are formed directly and only until the first negative one -- no FFT, no work buffers,
// no plan creation.  One thread per series; consecutive threads read consecutive floats of a trace
// row, so every pass streams the trace coalesced.  Sums are accumulated in double.
//
// Bound: HBM.  A series is S floats a whole trace row apart, so the kernel is a strided stream of the trace; the work
// per sample (W + 1 multiply-adds for W lags) is ~ 1 ms of vector issue for the 18.6 GB headline trace against >= 2.3 ms
// of HBM time.  What decides the time is (i) how many bytes are in flight -- every batch's loads are issued one batch
// AHEAD of their use (two register buffers, ping-pong), four waves per SIMD -- and (ii) how often the trace is read:
//   sweep 1  lags 0 .. W = 16 (kEssWin) and the mean in ONE pass (no mean pass in front of it): with y_t = x_t - r (r = the
//            mean of the first W samples, so that y is small) and m' = mean(y),
//              sum_{t>=k} (y_t - m')(y_{t-k} - m') = sum_{t>=k} y_t y_{t-k} - m' (2 T - head_k - tail_k) + (S - k) m'^2,
//            T = sum_t y_t, head_k / tail_k = the sums of the first / last k values (two W-sample loops).  Window sums of
//            128 products in registers, flushed into the thread's LDS column of W + 2 float running totals.
//   dense    a WAVE with more than kEssDenseAbove (3) series still positive at lag W takes lags W+1 .. W+32 of all its
//            series in one more coalesced pass, about the mean, from a 48-deep register ring (ess_sweep_dense).  Its
//            sums are S-long FLOAT accumulations with no flush: relative error ~ sqrt(S) 2^-24 typical (2e-5 at
//            S = 50 000), S 2^-24 worst case (3e-3) on sum |y y'| -- these lags only decide where the sum is cut and add
//            a few per cent to it, against a sampling noise of 1 / sqrt(S).
//   tail     what is still positive after that: S + 72 <= 2 304 floats -- one series at a time in the wave's LDS block,
//            64 lags per round by all lanes (ess_tail_cooperative); longer series per lane, 16 lags per pass with the
//            leading and the lagged stream both read (ess_sweep_far; sums flushed into doubles every 128 samples).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "host_common.h"
#include "ess_tail.h"

namespace arp {

#ifndef ARP_ESS_WIN
#define ARP_ESS_WIN 16
#endif
#ifndef ARP_ESS_MINB
#define ARP_ESS_MINB 4
#endif
constexpr int kEssWin = ARP_ESS_WIN;      // lags per sweep
constexpr int kEssFar = 16;               // lags per sweep past the first (series too long for the LDS block)
constexpr int kEssDenseLags = 32;         // lags of the dense continuation sweep (ring = kEssWin + 32 = 48 values)
constexpr int kEssDenseAbove = 3;         // ... taken by a wave with more than this many series still positive
#ifndef ARP_ESS_LOAD
#define ARP_ESS_LOAD (ARP_ESS_WIN % 16 == 0 ? 16 : 8)
#endif
// cache policy of the trace loads: 2 = non-temporal (`buffer_load_dword ... nt`): every sample is read once per pass, and
// the pass is longer than every cache -- 1 - 1.5 % faster than the default policy on the 18.6 GB trace
#ifndef ARP_ESS_AUX
#define ARP_ESS_AUX 2
#endif
#ifndef ARP_ESS_DEPTH
#define ARP_ESS_DEPTH 2
#endif
constexpr int kEssLoad = ARP_ESS_LOAD;    // samples per load batch of the first sweep
constexpr int kEssDepth = ARP_ESS_DEPTH;  // load buffers: kEssDepth - 1 batches are in flight while one is consumed

// Accuracy: blocked float summation -- window sums of <= 128 products, added into float running totals (S / 128 of them
// per lag): relative error ~ sqrt(128 + S / 128) x 2^-24 ~ 1e-6 typical, (128 + S / 128) x 2^-24 <= 4e-5 worst case at
// S = 50 000, on sum |y y'|.  Sweep 1 centres afterwards from raw sums about the reference level r; with |mean(y)| = m sd
// the raw sums are (1 + m^2) times the centred ones, so rho carries ~ (1 + m^2) x that.  The sweep is retaken around the
// mean itself when m^2 > 16 (a whole WAVE retakes it when one of its 64 series does, so the threshold sits where a
// stationary chain's first samples practically never land): rho is good to ~ 2e-5 typical, 7e-4 worst case -- against a
// sampling noise of 1 / sqrt(S) >= 4e-3.  ess_sweep_far flushes into doubles (registers are free there); the dense sweep
// keeps float sums (header).

// A series is addressed as (wave-uniform row base)[32-bit lane byte offset] through BUFFER loads: the row base is a
// buffer resource in scalar registers (rebuilt per row with two scalar adds), the lane's offset one VGPR shared by all of
// a window's loads -- `buffer_load_dword v, v_off, s[rsrc], 0 offen`.  With plain pointers the compiler folds the lane
// index into the pointer and keeps W loop-invariant 64-bit lane addresses, which alone pushed the sweep over the 128
// registers that four waves per SIMD allow (and serialised the loads behind their reloads from scratch).
struct EssSeries {
  const float* __restrict__ base;   // trace (uniform)
  unsigned boff;                    // 4 x series index (lane): byte offset within a row
  long long stride;
  unsigned row_bytes;               // bytes of a row that hold series: the buffer's range
  __device__ __forceinline__ float at(long long t) const {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + t * stride), 0,
                                                                        (int)row_bytes, 0x00020000);   // raw dword buffer
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)boff, 0, ARP_ESS_AUX));
  }
};

template <int N, int I = 0, class F>
__device__ __forceinline__ void ess_unrolled_batches(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    ess_unrolled_batches<N, I + 1>(f);
  }
}

template <int W>
__device__ __forceinline__ void ess_load(const EssSeries& x, long long t0, long long S, float fill, float (&v)[W]) {
  if (t0 + W <= S) {
#pragma unroll
    for (int tt = 0; tt < W; ++tt) v[tt] = x.at(t0 + tt);
  } else {   // the last, partial window: `fill` (the centre) past the end, i.e. y = 0, which adds nothing
#pragma unroll
    for (int tt = 0; tt < W; ++tt) v[tt] = t0 + tt < S ? x.at(t0 + tt) : fill;
  }
}

// sweep 1: c_0, lags 1 .. W and the total, about the centre r.  `tot` is the lane's column of W + 2 running totals in its
// wave's LDS block (stride 64 floats): [0] = sum y^2, [j] = sum y_t y_{t-j}, [W + 1] = sum y.
template <int W>
__device__ __forceinline__ void ess_sweep_first(const EssSeries& x, long long S, float r, float* __restrict__ tot) {
  constexpr int LB = kEssLoad;
  static_assert(W % LB == 0 && (128 % LB) == 0, "load batches tile the window and the flush interval");
  float acc[W + 1], w[W];
  float sy = 0.0f;
#pragma unroll
  for (int j = 0; j <= W; ++j) { acc[j] = 0.0f; tot[j * 64] = 0.0f; }
  tot[(W + 1) * 64] = 0.0f;
#pragma unroll
  for (int j = 0; j < W; ++j) w[j] = 0.0f;     // ring of the last W values of y: w[t mod W] = y_t (0 before the start)
  auto flush = [&]() {
#pragma unroll
    for (int j = 0; j <= W; ++j) { tot[j * 64] += acc[j]; acc[j] = 0.0f; }
    tot[(W + 1) * 64] += sy; sy = 0.0f;
  };
  // batch `ph` (compile time: its position within the ring) of LB consecutive samples
  auto block = [&](const float (&v)[LB], auto ph) {
    constexpr int P = decltype(ph)::value;
#pragma unroll
    for (int tl = 0; tl < LB; ++tl) {
      constexpr int dummy = 0; (void)dummy;
      const int tt = (P * LB + tl) % W;
      const float y = v[tl] - r;
      acc[0] = fmaf(y, y, acc[0]); sy += y;
#pragma unroll
      for (int j = 1; j <= W; ++j) acc[j] = fmaf(y, w[(tt - j + 2 * W) % W], acc[j]);
      w[tt] = y;
    }
  };
  // kEssDepth load buffers rotate: batch P is consumed from buffer P mod kEssDepth while the batches up to
  // kEssDepth - 1 ahead of it are in flight.  The time loop is unrolled by a number of batches that is a whole number of
  // buffer rotations AND of ring lengths, so buffers and ring positions are compile-time constants.
  constexpr int DEPTH = kEssDepth, RB = W / LB;
  constexpr int NB = (RB % DEPTH == 0) ? RB : RB * DEPTH;
  float buf[DEPTH][LB];
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d)
    if ((long long)d * LB < S) ess_load<LB>(x, (long long)d * LB, S, r, buf[d]);
  int nb = 0;
  for (long long t0 = 0; t0 < S; t0 += NB * LB) {
    ess_unrolled_batches<NB>([&](auto ph) {
      constexpr int P = decltype(ph)::value;
      const long long tb = t0 + P * LB;            // first sample of this batch
      if (tb >= S) return;
      const long long ta = tb + (DEPTH - 1) * LB;  // the batch that goes in flight now
      if (ta < S) ess_load<LB>(x, ta, S, r, buf[(P + DEPTH - 1) % DEPTH]);
      block(buf[P % DEPTH], ph);
      if (++nb == 128 / LB) { flush(); nb = 0; }
    });
  }
  flush();
}

// sweeps 2+: lags kb+1 .. kb+W (W = kEssFar here): the W mean-removed values y_{t-kb-1} .. y_{t-kb-W} of the LAGGED stream sit in
// the register window, the leading stream is read next to it
template <int W>
__device__ __forceinline__ void ess_sweep_far(const EssSeries& x, long long S, float mean,
                                              long long kb, double (&dacc)[W + 1]) {
  float acc[W + 1], w[W];
#pragma unroll
  for (int j = 0; j <= W; ++j) { acc[j] = 0.0f; dacc[j] = 0.0; }
#pragma unroll
  for (int j = 0; j < W; ++j) w[j] = 0.0f;
  int nb = 0;
  for (long long t0 = 0; t0 < S; t0 += W) {
    float xv[W], xl[W];
    ess_load<W>(x, t0, S, mean, xv);
    if (t0 - kb >= 0 && t0 - kb + W <= S) {
#pragma unroll
      for (int tt = 0; tt < W; ++tt) xl[tt] = x.at(t0 + tt - kb);
    } else {
#pragma unroll
      for (int tt = 0; tt < W; ++tt) {
        const long long tl = t0 + tt - kb;
        xl[tt] = (tl >= 0 && tl < S) ? x.at(tl) : mean;
      }
    }
#pragma unroll
    for (int tt = 0; tt < W; ++tt) {
      const float y = xv[tt] - mean;
#pragma unroll
      for (int j = 1; j <= W; ++j) acc[j] = fmaf(y, w[(tt - j + 2 * W) % W], acc[j]);
      w[tt] = xl[tt] - mean;
    }
    if (++nb == 128 / W) {
#pragma unroll
      for (int j = 1; j <= W; ++j) { dacc[j] += (double)acc[j]; acc[j] = 0.0f; }
      nb = 0;
    }
  }
#pragma unroll
  for (int j = 1; j <= W; ++j) dacc[j] += (double)acc[j];
}

// Dense continuation: lags KB+1 .. KB+W2 of ALL the wave's series in one more coalesced pass, about the mean, from a
// ring of the last KB + W2 values (positions addressed at compile time: the time loop is unrolled by the ring length).
// Taken by a wave when MANY of its series are still positive at lag KB -- long-trajectory HMC decorrelates slowly but
// evenly: on the headline flow's kept candidate (8 + 8 leapfrogs) the median series is cut at lag 10 and 27 % run past 16,
// i.e. ~ 17 of a wave's 64; one series at a time (below) that was 20 of the kernel's 24 ms.
template <int KB, int W2>
__device__ __forceinline__ void ess_sweep_dense(const EssSeries& x, long long S, float mean, float (&acc)[W2 + 1]) {
  constexpr int R = KB + W2, LB = 8;
  static_assert(R % LB == 0 && (R / LB) % 2 == 0, "load batches tile the ring, the two buffers alternate evenly");
  float ring[R];
#pragma unroll
  for (int j = 0; j <= W2; ++j) acc[j] = 0.0f;
#pragma unroll
  for (int j = 0; j < R; ++j) ring[j] = 0.0f;
  // float sums: lags this far out only decide where the sum is cut and add a few per cent to it; W2 x S / 2 products
  // of a centred series (relative error ~ sqrt(S) 2^-24) need no double accumulation
  float buf[2][LB];
  ess_load<LB>(x, 0, S, mean, buf[0]);
  for (long long t0 = 0; t0 < S; t0 += R) {
    ess_unrolled_batches<R / LB>([&](auto ph) {
      constexpr int P = decltype(ph)::value;
      const long long tb = t0 + P * LB;
      if (tb >= S) return;
      if (tb + LB < S) ess_load<LB>(x, tb + LB, S, mean, buf[(P + 1) % 2]);
#pragma unroll
      for (int tl = 0; tl < LB; ++tl) {
        constexpr int dummy = 0; (void)dummy;
        const int tt = P * LB + tl;
        const float y = buf[P % 2][tl] - mean;
#pragma unroll
        for (int j = 1; j <= W2; ++j) acc[j] = fmaf(y, ring[(tt - KB - j + 2 * R) % R], acc[j]);
        ring[tt] = y;
      }
    });
  }
}

// Lags beyond W of ONE series, by the whole wave.  After the first sweep a wave typically holds a few series that are
// still positive at lag W (a chain's slowly mixing elements) next to sixty that are finished; sweeping on per lane would
// issue a full wave's instructions for three live lanes.  Instead the wave copies the series into its LDS block once
// (mean removed, zero padded) and lane l forms lag kb + 1 + l over the whole series -- 64 lags per round, one broadcast
// ds_read_b128 and two ds_read2_b32 per four products -- then the first negative lag is found with a ballot.
// Returns the sum over the lags taken of (S - k) / S rho_k.
__device__ __forceinline__ double ess_tail_cooperative(const float* __restrict__ trace, long long stride, long long S,
                                                       unsigned idx_u, float mean_u, double c0_u, int W,
                                                       float* __restrict__ wbuf, int lane) {
  const int Si = (int)S;
  // the series, mean removed; zeros behind it so that every lane can run to the longest lag sum's end
  // (a lane's samples are 64 rows apart: every load is its own sector, so eight are put in flight at a time)
  for (int tb = 0; tb < Si + 72; tb += 8 * 64) {
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int t = tb + q * 64 + lane;
      v[q] = t < Si ? trace[(long long)t * stride + idx_u] : mean_u;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int t = tb + q * 64 + lane;
      if (t < Si + 72) wbuf[t] = v[q] - mean_u;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  double sum = 0.0;
  for (int kb = W; kb < Si; kb += 64) {
    const int k = kb + 1 + lane;
    // sum_u y_u y_{u+k}, u = 0 .. S-k-1; every lane runs to u < S - kb - 1 (products past its own end meet the zero pad)
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    const float* lead = wbuf + k;
    const int nu = Si - kb - 1;
#pragma unroll 4
    for (int u = 0; u < nu; u += 4) {
      const float4 b = *reinterpret_cast<const float4*>(wbuf + u);          // same address in every lane: a broadcast
      a0 = fmaf(b.x, lead[u], a0); a1 = fmaf(b.y, lead[u + 1], a1);
      a2 = fmaf(b.z, lead[u + 2], a2); a3 = fmaf(b.w, lead[u + 3], a3);
    }
    const bool in_range = k < Si;
    const double rho = ((double)a0 + (double)a1 + ((double)a2 + (double)a3)) / (double)(in_range ? Si - k : 1) / c0_u;
    const unsigned long long stop = __ballot(!in_range || rho < 0.0);
    const int first = stop ? __builtin_ctzll(stop) : 64;
    double part = lane < first ? (double)(Si - k) / (double)Si * rho : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    sum += part;
    if (first < 64) break;
  }
  __builtin_amdgcn_wave_barrier();
  return sum;
}

constexpr int kEssRows = 36;   // LDS floats per lane: the first sweep's totals, then (per wave) a whole series + padding

__global__ __launch_bounds__(256, ARP_ESS_MINB) void ess_kernel(const float* __restrict__ trace, long long S, long long n,
                                                                long long stride, float* __restrict__ ess, EssDefer defer) {
  constexpr int W = kEssWin, WF = kEssFar;
  static_assert(W + 2 <= kEssRows, "the totals of the first sweep fit the lane's LDS column");
  __shared__ __attribute__((aligned(16))) float s_buf[kEssRows * 256];   // one block of kEssRows x 64 floats per wave
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i0 < n;
  const long long i = valid ? i0 : n - 1;          // lanes past the end shadow the last series and never store
  const int lane = threadIdx.x & 63;
  float* wbuf = s_buf + (threadIdx.x >> 6) * (kEssRows * 64);
  float* tot = wbuf + lane;
  const EssSeries x{trace, (unsigned)i * 4u, stride, (unsigned)n * 4u};   // n_series < 2^30 is checked on the host
  // reference level r = x_0 + mean(x_j - x_0) over the first W samples: exactly x_0 for a constant series
  const int nh = S < W ? (int)S : W;
  float r;
  {
    float f0 = x.at(0), d = 0.0f;
#pragma unroll
    for (int j = 1; j < W; ++j) d += j < nh ? x.at(j) - f0 : 0.0f;
    r = f0 + d / (float)nh;
  }
  double T, mp, c0;
  float mean;
  for (int attempt = 0;; ++attempt) {
    ess_sweep_first<W>(x, S, r, tot);
    T = (double)tot[(W + 1) * 64];
    mp = T / (double)S;                     // mean of y
    mean = r + (float)mp;                   // mean of x, for the lags past W
    c0 = ((double)tot[0] - (double)S * mp * mp) / (double)S;
    // A reference level far from the mean (a series that was still drifting over its first samples) makes the
    // products large against the variance they are meant to resolve: take the pass again around the mean itself.
    if (attempt == 1 || !(mp * mp > 16.0 * c0)) break;
    r = mean;
  }
  const bool constant = !(c0 > 0.0);        // constant series: 0/0 as in the FFT form
  // centred sums of lags 1 .. W (the first / last W values are read again here rather than kept across the sweep)
  double total = 1.0;   // lag 0: (S - 0)/S * rho_0
  bool done = constant;
  {
    double head = 0.0, tail = 0.0;            // sums of the first / last k values of y
#pragma unroll
    for (int j = 1; j <= W; ++j) {
      const bool have = j - 1 < nh;
      head += have ? (double)(x.at(j - 1) - r) : 0.0;
      tail += have ? (double)(x.at(S - j) - r) : 0.0;          // x_{S-1-(j-1)}
      const double cj = (double)tot[j * 64] - mp * (2.0 * T - head - tail) + (double)(S - j) * mp * mp;
      const bool in_range = j < S;
      const double rho = cj / (double)(in_range ? S - j : 1) / c0;
      done = done || !in_range || rho < 0.0;
      total += done ? 0.0 : (double)(S - j) / (double)S * rho;
    }
  }
  // lanes past the end shadow series n - 1: they must not vote a wave into further sweeps (nor have the cooperative
  // tail run once more per shadow lane)
  done = done || S <= W || !valid;
  int tail_from = W;                          // first lag not yet taken
  if (__builtin_popcountll(__ballot(!done)) > kEssDenseAbove && S > W + kEssDenseLags) {
    // many of the wave's series go on: the next kEssDenseLags lags of all of them in one more coalesced pass
    float acc2[kEssDenseLags + 1];
    ess_sweep_dense<W, kEssDenseLags>(x, S, mean, acc2);
#pragma unroll
    for (int j = 1; j <= kEssDenseLags; ++j) {
      const long long k = W + j;
      const bool in_range = k < S;
      const double rho = (double)acc2[j] / (double)(in_range ? S - k : 1) / c0;
      done = done || !in_range || rho < 0.0;
      total += done ? 0.0 : (double)(S - k) / (double)S * rho;
    }
    tail_from = W + kEssDenseLags;
  }
  if (S + 72 <= kEssRows * 64) {
    // the wave's unfinished series, one after the other, 64 lags at a time by all lanes
    __builtin_amdgcn_wave_barrier();          // every lane has consumed its totals: the block is free
    unsigned long long pending = __ballot(!done);
    while (pending) {
      const int u = __builtin_amdgcn_readfirstlane(__builtin_ctzll(pending));
      pending &= pending - 1;
      const unsigned idx_u = __builtin_amdgcn_readlane((int)(x.boff >> 2), u);
      const float mean_u = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mean), u));
      const unsigned long long c0b = __builtin_bit_cast(unsigned long long, c0);
      const unsigned long long c0u = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(c0b >> 32), u) << 32) |
                                     (unsigned)__builtin_amdgcn_readlane((int)c0b, u);
      const double add = ess_tail_cooperative(trace, stride, S, idx_u, mean_u, __builtin_bit_cast(double, c0u), tail_from, wbuf, lane);
      if (lane == u) total += add;
    }
  } else if (defer.count) {
    // a series too long for the wave's LDS block, and the caller gave a workspace: what is still positive goes to the
    // work list of the matrix-core tail (ess_tail.h), which writes its ESS
    const unsigned long long m = __ballot(!done);
    if (m) {
      const int leader = __builtin_ctzll(m);
      unsigned base = 0;
      if (lane == leader) base = atomicAdd(defer.count, (unsigned)__builtin_popcountll(m));
      base = (unsigned)__builtin_amdgcn_readlane((int)base, leader);
      if (!done) {
        const unsigned pos = base + (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1ull));
        defer.idx[pos] = (unsigned)i; defer.mean[pos] = mean; defer.c0[pos] = c0; defer.total[pos] = total;
        defer.from[pos] = tail_from;
      }
    }
    if (!done) return;
  } else {
    // a series too long for the wave's LDS block: per lane, lags kb+1 .. kb+16 per sweep, leading and lagged stream read
    for (long long kb = tail_from; kb < S && !done; kb += WF) {
      double dacc[WF + 1];
      ess_sweep_far<WF>(x, S, mean, kb, dacc);
#pragma unroll
      for (int j = 1; j <= WF; ++j) {      // no early exit: the loop unrolls completely and dacc[] stays in registers
        const long long k = kb + j;
        const bool in_range = k < S;
        const double rho = dacc[j] / (double)(in_range ? S - k : 1) / c0;
        done = done || !in_range || rho < 0.0;
        total += done ? 0.0 : (double)(S - k) / (double)S * rho;
      }
    }
  }
  if (valid) ess[i] = constant ? __builtin_nanf("") : (float)((double)S / (-1.0 + 2.0 * total));
}


// ---------------------------------------------------------------------------
// One pass, short series (32 <= S <= kTileMaxS = 1 024: the reference flow's 1 000 recorded samples).  The two-sweep
// kernel above reads every series that is still positive at lag 16 a second time (2.0 - 2.5 x the trace on a sampler's own
// output: profiles/r05_ess_kernel.json).  Here a workgroup owns a TILE of 32 neighbouring series for their whole length --
// 32 x S floats, 128 KB at S = 1 000, one workgroup per CU -- reads it from HBM exactly once (row segments of 128 bytes)
// into LDS and takes every lag a series needs from there, each series only as far as ITS cut:
//   load     the 512 threads hold the next tile in registers (32 pairs each) while the current one is worked on; the loads
//            go out eight at a time between the compute blocks (a CU's memory queue holds far fewer than a tile's 256
//            wave-loads: issued in one go they block the wave until the first ones return) and the workgroup's barriers
//            wait for LDS only, never for them: persistent workgroups, one per CU, tiles round-robin
//   lags <= 16   thread (pair of series, 1/32 of the time axis): products about a reference level r (the mean of eight
//            samples spread over the series), a 16-deep ring of pairs, 17 packed FMAs per row; summed over a wave's four
//            row slots by permlane swaps and over the eight waves in a fixed order (bitwise reproducible)
//   finish   wave w owns series 4 w .. 4 w + 3, lane = lag: the raw sums are centred about the mean
//              sum_{t>=k} (y_t - m')(y_{t-k} - m') = P_k - m' (2 T - head_k - tail_k) + (S - k) m'^2
//            (T = sum y, head_k / tail_k = the sums of the first / last k values), the first negative lag is a ballot,
//            the sum up to it a DPP row sum.  No loop over lags, no single-wave section.
//   further  a series still positive at lag 16 is taken by its wave, 16 lags per round until its cut: lane g forms the
//            products of rows 17 g .. 17 g + 16 (an odd slice and the row pitch of 33 floats spread the 64 lanes over
//            the 64 banks), then a DPP/permlane sum over the wave
// (S - k)/S rho_k = [sum_k / (S - k)] (S - k)/S / [sum_0 / S] = sum_k / sum_0: no divisions per lag, and rho_k < 0 exactly when
// sum_k < 0.  Algorithmic work follows the data (median cut lag ~ 10, ~ 28 % of the series past 16 on the headline flow's
// trace) instead of giving every series of a wave 48 lags; HBM traffic is the trace, once.
// ---------------------------------------------------------------------------
constexpr bool kEssTileDefault = false; // arp_ess takes 32 <= S <= 1 024 through the tile kernel (see profiles/r06_ess_tile.txt)
constexpr int kTileNS = 32;             // series per tile
constexpr int kTilePitch = 34;          // floats per tile row in LDS (even: a pair of series is one aligned 8-byte access)
constexpr int kTileMaxS = 1024;
constexpr int kTileMinS = 32;
constexpr int kTileThreads = 512;
constexpr int kTileRowsPerThread = kTileMaxS / 32;    // of a PAIR of series

#ifdef ARP_ESS_PROF
__device__ unsigned long long g_ess_prof[16];
#define ARP_PROF_MARK(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); g_ess_prof[k] += t_ - prof_t; prof_t = t_; } } while (0)
#else
#define ARP_PROF_MARK(k) do { } while (0)
#endif
// LDS-only workgroup barrier: the tile prefetch (global loads into registers) stays in flight across it
#define ARP_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// The tile's LDS reads are unconditional (clamped index) and masked afterwards; left alone, the compiler sinks each read
// under its mask again -- a branch per read.  Passing a batch of read values through one empty asm keeps the reads where
// they are, at the price of one wait for the whole batch.
template <int N>
__device__ __forceinline__ void ess_pin(float (&x)[N]) {
  static_assert(N == 8 || N == 16 || N == 17, "batches of 8 / 16 / 17");
  if constexpr (N == 8)
    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
  else if constexpr (N == 16)
    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
                      "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]));
  else
    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
                      "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]),
                      "+v"(x[16]));
}

__device__ __forceinline__ float ess_row_sum(float v) {        // over the 16 lanes of a row, every lane gets the total
  v += dpp_mov<0xB1>(v);     // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);     // quad_perm [2,3,0,1]
  v += dpp_mov<0x124>(v);    // row_ror:4
  v += dpp_mov<0x128>(v);    // row_ror:8
  return v;
}
__device__ __forceinline__ float ess_rows_sum(float v) {       // over the four rows of a wave (lanes l, l^16, l^32, l^48)
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float ess_wave_sum(float v) {       // every lane gets the same total, fixed order
  v = ess_rows_sum(ess_row_sum(v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
__device__ __forceinline__ double ess_readlane(double x, int lane) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), lane), lo = (unsigned)__builtin_amdgcn_readlane((int)b, lane);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// physical LDS row of tile row t: one row of padding behind every 32, so that the 32 time slots of a series (32 rows
// apart) start 33 physical rows apart -- with the even pitch of 34 floats that spreads them over 32 different bank pairs
__device__ __forceinline__ int ess_prow(int t) { return (t + (t >> 5)) * kTilePitch; }

__global__ __launch_bounds__(kTileThreads, 1) void ess_tile_kernel(const float* __restrict__ trace, int S, long long n,
                                                                   long long stride, float* __restrict__ ess, int n_tiles) {
  __shared__ __attribute__((aligned(16))) float s_tile[(kTileMaxS + kTileMaxS / 32) * kTilePitch];   // x_t of the tile's series
  // LOADING: a thread takes a pair of neighbouring series (columns 2 cp, 2 cp + 1: one 8-byte load) and rows rs + 32 i.
  // WORKING: wave w owns series 4 w .. 4 w + 3 for their whole length -- lanes 0 - 31 the pair (4 w, 4 w + 1), lanes 32 - 63
  // the pair (4 w + 2, 4 w + 3), lane & 31 = the 1/32 of the time axis it forms products for.  Nothing is combined
  // between waves.
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, cp = tid & 15, rs = tid >> 4;
  v2f v[kTileRowsPerThread];

  // Rows rs + 32 i of a tile into v, eight at a time (a wave's load covers four rows x 128 bytes).  Every load is
  // unconditional and in range: a row past S reads the thread's last row again (the offset stops advancing), a pair past
  // the last series reads the last pair -- what lands there is never used, except that the last series of an odd count
  // arrives in the SECOND half of the clamped pair (`swap`).  `off` is an offset, not a pointer: through the asm below a
  // pointer would lose its address space and load as FLAT, which the LDS barriers would wait for.
  long long off = 0;
  int cnt = 0;
  bool swap = false, swap_next = false;
  auto load_begin = [&](int tile) {
    const long long c = (long long)tile * kTileNS + 2 * cp;
    swap_next = c == n - 1;
    off = (long long)rs * stride + (c + 1 < n ? c : n - 2);
    cnt = (S - rs + 31) / 32;           // rows rs + 32 i < S  <=>  i < cnt: tests against a constant, no row index kept
    asm volatile("" : "+v"(cnt));       // (and recomputed per tile rather than kept as loop-invariant lane masks)
  };
  auto load_chunk = [&](auto chunk) {
    constexpr int K = decltype(chunk)::value;
    const long long step = 32 * stride;
#pragma unroll
    for (int i = 8 * K; i < 8 * K + 8; ++i) {
      const float* q = trace + off;
      v2f x;
      x[0] = __builtin_nontemporal_load(q); x[1] = __builtin_nontemporal_load(q + 1);      // one global_load_dwordx2
      v[i] = x;
      off += (i + 1 < cnt) ? step : 0;
      asm volatile("" : "+v"(off));     // a running offset: without this the row offsets become loop-invariant registers
    }
  };
  using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>;
  using C2 = std::integral_constant<int, 2>; using C3 = std::integral_constant<int, 3>;

#ifdef ARP_ESS_PROF
  unsigned long long prof_t = __builtin_amdgcn_s_memtime();
#endif
  // A workgroup takes a CONTIGUOUS run of tiles: the workgroups that are resident together then read columns ~ 70 KB
  // apart, spread over the HBM channels
  const int per = (n_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
  int tile = blockIdx.x * per;
  const int tile_end = tile + per < n_tiles ? tile + per : n_tiles;
  if (tile < tile_end) { load_begin(tile); load_chunk(C0{}); load_chunk(C1{}); load_chunk(C2{}); load_chunk(C3{}); }
  const int S_arg = S;
  for (; tile < tile_end; ++tile) {
    // everything below derives its row bounds, masks and LDS addresses from this copy: opaque per tile, so that they are
    // recomputed (a few hundred scalar / vector instructions) instead of living in ~ 100 loop-invariant registers next to
    // the 64 of the prefetched tile
    int S = S_arg;
    asm volatile("" : "+s"(S));
    const long long col0 = (long long)tile * kTileNS;
    const int ncol = (int)(n - col0 < kTileNS ? n - col0 : kTileNS);
    const bool more = tile + 1 < tile_end;
    swap = swap_next;
    ARP_LDS_BARRIER();       // every wave has left the previous tile's LDS
    ARP_PROF_MARK(0);
    // ---- the tile, as it is (rows S .. 1 023 exist in LDS: written unconditionally, never read)
#pragma unroll
    for (int i = 0; i < kTileRowsPerThread; ++i) {
      v2f x = v[i];
      if (swap) x[0] = x[1];
      *reinterpret_cast<v2f*>(s_tile + (rs + 33 * i) * kTilePitch + 2 * cp) = x;        // physical row of rs + 32 i (rs < 32)
    }
    ARP_PROF_MARK(1);
    // (unconditional: behind the last tile the same tile is loaded once more and dropped -- under `if (more)` the loaded pairs
    //  meet the old ones in a phi at the end of the branch, and the register allocator copies them there, i.e. waits for them)
    load_begin(more ? tile + 1 : tile); load_chunk(C0{});
    ARP_LDS_BARRIER();
    ARP_PROF_MARK(2);
    // ---- lags 0 .. 16 of the lane's series pair about r: rows [t0, t1), a 16-deep ring of pairs, 17 packed FMAs per row
    const int hp = l >> 5, slot = l & 31, q4 = l >> 4;
    const float* col = s_tile + 4 * w + 2 * hp;
    auto pair_at = [&](int t) { return *reinterpret_cast<const v2f*>(col + ess_prow(t)); };
    v2f r;
    float X[36];             // raw sums: X[2 j + c] = lag j of the pair's series c, X[34 + c] = T
    {
      const int t0 = 32 * slot, t1 = (t0 + 32 < S) ? t0 + 32 : S;
      v2f acc[17], ring[16], sy = v2f{0.0f, 0.0f};
#pragma unroll
      for (int j = 0; j <= 16; ++j) acc[j] = v2f{0.0f, 0.0f};
      {
        float a0[8], a1[8], r0[16], r1[16];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const v2f q = pair_at((S - 1) * k / 7); a0[k] = q[0]; a1[k] = q[1]; }
#pragma unroll
        for (int j = 0; j < 16; ++j) {                        // ring[m & 15] = y_{t0 + m}, m = -16 .. -1
          const int t = t0 - 16 + j;
          const v2f q = pair_at(t < 0 ? 0 : (t < kTileMaxS ? t : kTileMaxS - 1));
          r0[j] = q[0]; r1[j] = q[1];
        }
        ess_pin(a0); ess_pin(a1); ess_pin(r0); ess_pin(r1);
        r = v2f{((a0[0] + a0[1]) + (a0[2] + a0[3])) + ((a0[4] + a0[5]) + (a0[6] + a0[7])),
                ((a1[0] + a1[1]) + (a1[2] + a1[3])) + ((a1[4] + a1[5]) + (a1[6] + a1[7]))} * 0.125f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const bool in = t0 - 16 + j >= 0 && t0 - 16 + j < t1;
          ring[j] = in ? v2f{r0[j], r1[j]} - r : v2f{0.0f, 0.0f};
        }
      }
      auto half = [&](int tb, auto hh) {                          // eight rows: ring positions 8 H .. 8 H + 7
        constexpr int H = decltype(hh)::value;
        float y0[8], y1[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int t = tb + 8 * H + i;                           // (clamped to the tile)
          const v2f q = pair_at(t < kTileMaxS ? t : kTileMaxS - 1);
          y0[i] = q[0]; y1[i] = q[1];
        }
        ess_pin(y0); ess_pin(y1);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          constexpr int dummy = 0; (void)dummy;
          const int p = 8 * H + i;
          const v2f y = tb + p < t1 ? v2f{y0[i], y1[i]} - r : v2f{0.0f, 0.0f};
          acc[0] = vfma(y, y, acc[0]);
          sy += y;
#pragma unroll
          for (int j = 1; j <= 16; ++j) acc[j] = vfma(y, ring[(p - j + 32) & 15], acc[j]);
          ring[p] = y;
        }
      };
      if (t0 < t1) { half(t0, C0{}); half(t0, C1{}); }
      ARP_PROF_MARK(3);
      load_chunk(C1{});
      ARP_PROF_MARK(4);
      if (t0 + 16 < t1) { half(t0 + 16, C0{}); half(t0 + 16, C1{}); }
      ARP_PROF_MARK(5);
      load_chunk(C2{});
      ARP_PROF_MARK(6);
#pragma unroll
      for (int j = 0; j <= 16; ++j) { X[2 * j] = acc[j][0]; X[2 * j + 1] = acc[j][1]; }
      X[34] = sy[0]; X[35] = sy[1];
    }
    // ---- the 32 time slots of a pair: over the 16 lanes of a row by DPP, then the pair's two rows by a permlane swap that
    // leaves series 0's totals in the even row and series 1's in the odd row -- row q4 of the wave now holds series
    // 4 w + q4, every lane of the row the same 18 numbers
    // (the swap first -- it halves what the DPP row sums have to add: 18 values instead of 36)
    float Y[18];
#pragma unroll
    for (int m = 0; m < 18; ++m) {
      const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(X[2 * m]), __float_as_uint(X[2 * m + 1]), false, false);
      Y[m] = ess_row_sum(__uint_as_float(sw[0]) + __uint_as_float(sw[1]));
    }
    ARP_PROF_MARK(7);
    load_chunk(C3{});
    ARP_PROF_MARK(8);
    // ---- centre about the mean and cut at the first negative lag (every lane of the row does the same arithmetic)
    {
      const int u = 4 * w + q4;
      const float ru = (q4 & 1) ? r[1] : r[0];
      const float* colu = s_tile + u;
      float h[16], g[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) { h[i] = colu[ess_prow(i)]; g[i] = colu[ess_prow(S - 1 - i)]; }     // (S >= 32)
      ess_pin(h); ess_pin(g);
      // float throughout: r sits within about a standard deviation of the mean, so the correction terms are of the size of the
      // sums they correct (|m'|^2 <~ c_0) and their rounding is that of the sums themselves (~ 1e-6 relative)
      const float T = Y[17], Sf = (float)S, mp = T / Sf;
      const float sum0 = fmaf(-Sf * mp, mp, Y[0]);
      const bool constant = !(sum0 > 0.0f);
      const float inv_f = 1.0f / sum0;
      const double inv = (double)inv_f;
      float tsum = 0.0f, ht = 0.0f;            // ht = head_j + tail_j: the sums of the first and the last j values of y
      bool done = constant;
      const float twoT = 2.0f * T, mp2 = mp * mp;
#pragma unroll
      for (int j = 1; j <= 16; ++j) {
        ht += (h[j - 1] - ru) + (g[j - 1] - ru);
        const float cj = fmaf(Sf - (float)j, mp2, fmaf(-mp, twoT - ht, Y[j]));
        done = done || cj < 0.0f;
        tsum += done ? 0.0f : cj * inv_f;
      }
      const double total = 1.0 + (double)tsum;
      const bool valid = u < ncol;
      const bool open = !done && S > 17 && valid;
      if ((l & 15) == 0 && valid && !open) ess[col0 + u] = constant ? __builtin_nanf("") : (float)((double)S / (-1.0 + 2.0 * total));
      ARP_PROF_MARK(9);
      // ---- the wave's series that go on, one after the other: 16 lags per round by all 64 lanes
      const unsigned long long open_m = __ballot(open && (l & 15) == 0);
      const float mu = ru + (float)mp;                             // the series' mean
      const int SLd = ((S + 63) / 64) | 1;                         // odd slice: lane g's rows are 17 g .. 17 g + 16 at S = 1 000
      for (int qq = 0; qq < 4; ++qq) {
        if (!((open_m >> (16 * qq)) & 1ull)) continue;
        const int uu = 4 * w + qq;
        const double inv_u = ess_readlane(inv, 16 * qq);
        double tot = ess_readlane(total, 16 * qq);
        const float mu_u = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mu), 16 * qq));
        const float* cu = s_tile + uu;
        const int t0 = l * SLd;
        bool fin = false;
        for (int kb = 16; !fin; kb += 16) {
          float acc[17], ring[16];
#pragma unroll
          for (int jj = 1; jj <= 16; ++jj) acc[jj] = 0.0f;
          auto at = [&](int t) { return cu[ess_prow(t < 0 ? 0 : (t < kTileMaxS ? t : kTileMaxS - 1))]; };
#pragma unroll
          for (int jj = 0; jj < 16; ++jj) ring[jj] = at(t0 - 16 + jj - kb);      // ring[m & 15] = y_{t0 + m - kb}, m = -16 .. -1
          float ya[17], yl[17];
#pragma unroll
          for (int i = 0; i < 17; ++i) { ya[i] = at(t0 + i); yl[i] = at(t0 + i - kb); }
          ess_pin(ring); ess_pin(ya); ess_pin(yl);
#pragma unroll
          for (int jj = 0; jj < 16; ++jj) { const int t = t0 - 16 + jj - kb; ring[jj] = (t >= 0 && t < S) ? ring[jj] - mu_u : 0.0f; }
#pragma unroll
          for (int i = 0; i < 17; ++i) {
            const int t = t0 + i;
            const bool in = i < SLd && t < S;
            const float y = in ? ya[i] - mu_u : 0.0f;
#pragma unroll
            for (int jj = 1; jj <= 16; ++jj) acc[jj] = fmaf(y, ring[(i - jj + 32) & 15], acc[jj]);
            ring[i & 15] = (in && t - kb >= 0) ? yl[i] - mu_u : 0.0f;
          }
#pragma unroll
          for (int jj = 1; jj <= 16; ++jj) {
            const float c = ess_wave_sum(acc[jj]);
            fin = fin || kb + jj >= S || c < 0.0f;
            tot += fin ? 0.0 : (double)c * inv_u;
          }
        }
        if (l == 0) ess[col0 + uu] = (float)((double)S / (-1.0 + 2.0 * tot));
      }
    }
    ARP_PROF_MARK(10);
  }
}

}  // namespace arp

namespace arp {
// workspace layout of arp_ess_ws: [count | idx | mean | c0 | total | from | rows], every array 256-byte aligned
struct EssWsLayout {
  size_t off_idx, off_mean, off_c0, off_total, off_from, off_rows;
  static size_t up(size_t x) { return (x + 255) & ~(size_t)255; }
  explicit EssWsLayout(int64_t n) {
    off_idx = 256;
    off_mean = off_idx + up((size_t)n * 4);
    off_c0 = off_mean + up((size_t)n * 4);
    off_total = off_c0 + up((size_t)n * 8);
    off_from = off_total + up((size_t)n * 8);
    off_rows = off_from + up((size_t)n * 4);
  }
};
}  // namespace arp

#ifdef ARP_ESS_PROF
extern "C" int arp_debug_ess_prof(unsigned long long* out16, int reset) {
  if (out16) ARP_HIP_OK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(arp::g_ess_prof), 16 * sizeof(unsigned long long)));
  if (reset) { unsigned long long z[16] = {0}; ARP_HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(arp::g_ess_prof), z, sizeof(z))); }
  return 0;
}
#endif

extern "C" int64_t arp_ess_workspace_bytes(int64_t n_samples, int64_t n_series) {
  using namespace arp;
  if (n_samples <= 0 || n_series <= 0) return 0;
  if (n_samples + 72 <= kEssRows * 64) return 0;          // short series finish inside the first-stage kernel
  // series rows are gathered 64 at a time, so the row area holds a whole number of 64-row blocks (at least one):
  // a caller that allocates exactly this many bytes gets every listed series in ONE chunk, also for n_series < 64
  const size_t rows = ((size_t)n_series + 63) & ~(size_t)63;
  return (int64_t)(EssWsLayout(n_series).off_rows + rows * (size_t)ess_row_floats(n_samples) * 4);
}

extern "C" int arp_ess_ws(const float* trace, int64_t n_samples, int64_t n_series, int64_t row_stride, float* ess,
                          void* workspace, int64_t workspace_bytes, void* stream) {
  using namespace arp;
  if (!trace || !ess || n_samples <= 0 || n_series <= 0 || row_stride < n_series) {
    set_error("arp_ess: trace/ess, n_samples > 0, n_series > 0 and row_stride >= n_series are required");
    return 1;
  }
  if (n_series >= (1ll << 30)) { set_error("arp_ess: at most 2^30 - 1 series per call (32-bit lane offsets)"); return 1; }
  hipStream_t st = (hipStream_t)stream;
  const long long blocks = (n_series + 255) / 256;
  EssDefer D{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  long long rows_cap = 0;
  const bool long_series = n_samples + 72 > kEssRows * 64;
  const long long SR = ess_row_floats(n_samples);
  if (long_series && workspace) {
    if (SR / 64 > 65535) { set_error("arp_ess_ws: at most 4 194 240 samples per series on the workspace path"); return 1; }
    if (((uintptr_t)workspace & 255) != 0) { set_error("arp_ess_ws: the workspace must be 256-byte aligned"); return 1; }
    const EssWsLayout Lw(n_series);
    rows_cap = workspace_bytes > (int64_t)Lw.off_rows ? ((workspace_bytes - (int64_t)Lw.off_rows) / (SR * 4)) & ~63ll : 0;
    if (rows_cap < 64) {
      set_error("arp_ess_ws: workspace too small (the work lists and at least 64 series rows: see arp_ess_workspace_bytes)");
      return 1;
    }
    if (rows_cap > n_series) rows_cap = (n_series + 63) & ~63ll;
    char* w = (char*)workspace;
    D = EssDefer{(unsigned*)w, (unsigned*)(w + Lw.off_idx), (float*)(w + Lw.off_mean), (double*)(w + Lw.off_c0),
                 (double*)(w + Lw.off_total), (int*)(w + Lw.off_from)};
    ARP_HIP_OK(hipMemsetAsync(w, 0, 256, st));
  }
  // short series: the one-pass tile kernel (the trace is read once; every lag from LDS)
  bool tile_path = n_samples >= kTileMinS && n_samples <= kTileMaxS && n_series >= 2 && row_stride < (1ll << 29);
  {
    // which kernel takes short series: the library default below, or (experiments, ARP_DEBUG=1) ARP_ESS_TILE=0 / 1
    const char* e = getenv("ARP_ESS_TILE");
    const char* d = getenv("ARP_DEBUG");
    const bool dbg = e && d && d[0] == '1' && d[1] == 0;
    if (!kEssTileDefault) tile_path = tile_path && dbg && e[0] == '1';
    else if (dbg && e[0] == '0') tile_path = false;
  }
  if (tile_path) {
    static thread_local int cus_of[64] = {0};
    int dev = 0;
    ARP_HIP_OK(hipGetDevice(&dev));
    int& cus = cus_of[dev & 63];
    if (cus <= 0) ARP_HIP_OK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const long long n_tiles = (n_series + kTileNS - 1) / kTileNS;
    const unsigned grid = (unsigned)std::min<long long>(n_tiles, cus > 0 ? cus : 256);
    hipLaunchKernelGGL(ess_tile_kernel, dim3(grid), dim3(kTileThreads), 0, st, trace, (int)n_samples, (long long)n_series,
                       (long long)row_stride, ess, (int)n_tiles);
    ARP_HIP_OK(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(ess_kernel, dim3((unsigned)blocks), dim3(256), 0, st, trace, (long long)n_samples,
                     (long long)n_series, (long long)row_stride, ess, D);
  ARP_HIP_OK(hipGetLastError());
  if (D.count) {
    // The listed series, `rows_cap` at a time; how many there are stays on the device (no host synchronisation): every
    // chunk that could hold listed series is launched, and workgroups past the count leave at once.
    float* rows = (float*)((char*)workspace + EssWsLayout(n_series).off_rows);
    for (long long p0 = 0; p0 < n_series; p0 += rows_cap) {
      const unsigned pmax = (unsigned)std::min<long long>(p0 + rows_cap, n_series);
      const unsigned nrow = pmax - (unsigned)p0;
      hipLaunchKernelGGL(ess_gather_kernel, dim3((nrow + 63) / 64, (unsigned)(SR / 64)), dim3(256), 0, st, trace,
                         (long long)n_samples, (long long)row_stride, D, (unsigned)p0, pmax, rows);
      hipLaunchKernelGGL(ess_tail_kernel, dim3(nrow), dim3(64), 0, st, (const float*)rows, (long long)n_samples, D,
                         (unsigned)p0, pmax, ess);
    }
    ARP_HIP_OK(hipGetLastError());
  }
  return 0;
}

extern "C" int arp_ess(const float* trace, int64_t n_samples, int64_t n_series, int64_t row_stride, float* ess,
                       void* stream) {
  // no workspace: long, slowly mixing series take the per-lane far sweeps (slow, but no memory is needed)
  return arp_ess_ws(trace, n_samples, n_series, row_stride, ess, nullptr, 0, stream);
}
