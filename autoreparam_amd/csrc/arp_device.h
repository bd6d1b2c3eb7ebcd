// Device-side building blocks for the gfx950 (CDNA4, wave64) sampling kernels:
// sub-wave group reductions on DPP, the per-slot RNG and the normal generator.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ARP_DEV __device__ __forceinline__

namespace arp {

constexpr int kBlock = 256;      // 4 waves per workgroup
constexpr int kViBlock = 512;

// register pair for the packed f32 VALU forms (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32)
typedef float v2f __attribute__((ext_vector_type(2)));
ARP_DEV v2f vfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// 16-byte global store; STREAM = a trace row: written once, read by a later kernel after 18 MB x rows of other
// traffic -- stored non-temporally (`global_store_dwordx4 ... nt`), so the rows do not displace the model tables and the
// chain state in L2 (headline launch: 0.8 - 1.1 % shorter)
typedef float v4f_nt __attribute__((ext_vector_type(4)));
template <bool STREAM>
ARP_DEV void store_f4(float4* dst, const float4& v) {
  if constexpr (STREAM) __builtin_nontemporal_store(v4f_nt{v.x, v.y, v.z, v.w}, reinterpret_cast<v4f_nt*>(dst));
  else *dst = v;
}

template <bool STREAM>
ARP_DEV void store_v4(v4f_nt* dst, v4f_nt v) {
  if constexpr (STREAM) __builtin_nontemporal_store(v, dst);
  else *dst = v;
}

// ---------------------------------------------------------------------------
// Cross-lane helpers.  A chain is spread over K consecutive lanes (K | 16), so a
// chain never straddles a DPP row of 16 lanes; all exchanges are VALU DPP
// modifiers (no LDS traffic).
// ---------------------------------------------------------------------------
template <int CTRL>
ARP_DEV float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// Sum over the K lanes of a chain; every lane receives the bitwise identical
// total (butterfly of commutative adds: quad xor 1, quad xor 2, half-row mirror,
// row mirror).
template <int K>
ARP_DEV float group_sum(float v) {
  // The argument is made opaque first: with -ffp-contract=fast a multiply that feeds the first add would be fused into
  // it, fma(x_own, y, round(x_partner y)) -- own product unrounded, partner's rounded -- and the K lanes would no
  // longer hold the same bits (seen as a run that depended on how it was cut into launches: the replicated top-level
  // scalars are stored from slot 0 only).
  if (K >= 2) asm volatile("" : "+v"(v));
  if (K >= 2) v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  if (K >= 4) v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  if (K >= 8) v += dpp_mov<0x141>(v);   // row_half_mirror
  if (K >= 16) v += dpp_mov<0x140>(v);  // row_mirror
  return v;
}

template <int K>
ARP_DEV float group_max(float v) {
  if (K >= 2) v = fmaxf(v, dpp_mov<0xB1>(v));
  if (K >= 4) v = fmaxf(v, dpp_mov<0x4E>(v));
  if (K >= 8) v = fmaxf(v, dpp_mov<0x141>(v));
  if (K >= 16) v = fmaxf(v, dpp_mov<0x140>(v));
  return v;
}

// Value held by slot 0 of the chain, broadcast to all of its K lanes.
template <int K>
ARP_DEV float group_bcast0(float v, int slot) {
  if (K == 1) return v;
  return group_sum<K>(slot == 0 ? v : 0.0f);
}

// ---------------------------------------------------------------------------
// RNG.  Each (chain, slot) owns one MWC64X stream (a 32-bit multiply-with-carry generator,
// x' = lo(A x + c), c' = hi(A x + c), output x ^ c of the state before the step; D. B. Thomas'
// published parameters A = 4294883355, period ~2^63; passes BigCrush).  One step is a single
// v_mad_u64_u32 plus an xor -- a third of the issue time of a 128-bit xorshift-family step, and
// the momentum draw is the largest per-transition cost of the chain kernels -- and the state is
// two registers.  The 64-bit state is seeded once per run by Philox4x32-10 (Salmon et al., SC'11)
// keyed on the user seed with counter (global chain id, slot, lanes_per_chain), so a stream
// depends on the global chain id only, never on which GPU or workgroup runs the chain.
// ---------------------------------------------------------------------------
struct Rng {
  uint32_t x, c;
};

constexpr uint32_t kMwcA = 4294883355u;

ARP_DEV uint32_t rng_next(Rng& r) {
  const uint32_t res = r.x ^ r.c;
  const uint64_t t = (uint64_t)r.x * kMwcA + r.c;
  r.x = (uint32_t)t;
  r.c = (uint32_t)(t >> 32);
  return res;
}

// Jump-ahead.  With v = c 2^32 + x one step is v' = A x + c = A v mod m for m = A 2^32 - 1 (A 2^32 = 1 mod m, so
// multiplying by A divides by 2^32), hence n steps are v -> A^n v mod m: one modular multiplication instead of n steps.
// mwc_montmul(a, b) = a b 2^-64 mod m by two rounds of P <- (P >> 32) + A (P mod 2^32) on the 128-bit product -- the
// generator's own step applied to a wider value; mwc_pow(n) = A^n 2^64 mod m (Montgomery form), so that
// mwc_montmul(v, mwc_pow(n)) = A^n v mod m.  (tests/test_rng.py restates this in Python integers against n single steps;
// the VI kernel, which skips the words other lanes consume, is held to the oracle's sequential draws.)
constexpr uint64_t kMwcM = ((uint64_t)kMwcA << 32) - 1u;
constexpr uint64_t kMwcMontOne = 0u - kMwcM;                                   // 2^64 mod m (m < 2^64 < 2 m)
constexpr uint64_t kMwcMontA = (uint64_t)(((unsigned __int128)kMwcA * kMwcMontOne) % kMwcM);
ARP_DEV uint64_t mwc_montmul(uint64_t a, uint64_t b) {
  const uint64_t lo = a * b, hi = __umul64hi(a, b);
  uint64_t t = (uint64_t)(uint32_t)lo * kMwcA;
  uint64_t l1 = (hi << 32) | (lo >> 32), h1 = hi >> 32;        // P >> 32 = h1 : l1
  l1 += t; h1 += l1 < t ? 1u : 0u;                             // h1 <= 2^32
  t = (uint64_t)(uint32_t)l1 * kMwcA;
  uint64_t c2 = h1 >> 32, l2 = (h1 << 32) + (l1 >> 32);        // (h1 : l1) >> 32 = c2 : l2
  l2 += t; c2 += l2 < t ? 1u : 0u;                             // the value is below 2 m + 2^33 < 2^65: c2 <= 1
  if (c2) l2 += kMwcMontOne;                                   // 2^64 = kMwcMontOne mod m; no wrap (see the bound)
  if (l2 >= kMwcM) l2 -= kMwcM;
  if (l2 >= kMwcM) l2 -= kMwcM;
  return l2;
}
ARP_DEV uint64_t mwc_pow(unsigned n) {
  uint64_t r = kMwcMontOne, base = kMwcMontA;
  for (; n; n >>= 1) {
    if (n & 1u) r = mwc_montmul(r, base);
    base = mwc_montmul(base, base);
  }
  return r;
}
// r advanced by the n steps whose multiplier is `jump` = mwc_pow(n)
ARP_DEV void rng_jump(Rng& r, uint64_t jump) {
  const uint64_t v = mwc_montmul(((uint64_t)r.c << 32) | r.x, jump);
  r.x = (uint32_t)v;
  r.c = (uint32_t)(v >> 32);
}

ARP_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                           uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

ARP_DEV Rng rng_seed(uint64_t seed, uint64_t chain, uint32_t slot, uint32_t lanes) {
  uint32_t o[4];
  philox4x32_10((uint32_t)chain, (uint32_t)(chain >> 32), slot, lanes,
                (uint32_t)seed, (uint32_t)(seed >> 32), o);
  Rng r{o[0], o[1] >> 1};      // carry < 2^31 < A: a state on the generator's cycle
  if ((r.x | r.c) == 0u) r.x = 1u;  // (0, 0) is a fixed point
  return r;
}

// uniform in (0,1] with 24 random bits
ARP_DEV float u01_open0(uint32_t w) { return (float)((w >> 8) + 1u) * 5.9604644775390625e-08f; }

// Two standard normals from two 32-bit words (Box-Muller on the hardware
// transcendental units: v_log_f32 is log2, v_sin/v_cos take revolutions).
// u = (float)w0 * 2^-32 + 2^-33 lies in (0, 1] after rounding (one conversion and one multiply-add, no shifts); the
// angle is the float in [1, 2) whose mantissa is the low 23 bits of w1, in revolutions (sin/cos are periodic, so the
// leading 1 is free): one v_and_or_b32 instead of a conversion (half rate on gfx950) and a multiply.
ARP_DEV float angle_rev(uint32_t w1) { return __uint_as_float(0x3f800000u | (w1 & 0x007fffffu)); }
ARP_DEV void normal_pair(uint32_t w0, uint32_t w1, float& z0, float& z1) {
  float u = fmaf((float)w0, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
  float rev = angle_rev(w1);
  // r = sqrt(-2 ln u) = sqrt(-2 ln2 * log2 u)
  float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u));
  z0 = r * __builtin_amdgcn_cosf(rev);
  z1 = r * __builtin_amdgcn_sinf(rev);
}

// Numerically careful pieces shared by the Bernoulli-logit models.
ARP_DEV float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
ARP_DEV float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
ARP_DEV float sigmoidf_(float x) {
  // 1/(1+e^-x), no overflow for large |x|
  float e = fast_exp(-fabsf(x));
  float s = __builtin_amdgcn_rcpf(1.0f + e);
  return x >= 0.0f ? s : e * s;
}
ARP_DEV float softplusf_(float x) {
  // log(1+e^x) = max(x,0) + log1p(e^-|x|)
  float e = fast_exp(-fabsf(x));
  return fmaxf(x, 0.0f) + fast_log(1.0f + e);
}


#ifdef ARP_EXP_TIMING
// timing experiments only: per-workgroup segment timers (core-clock cycles), printed by block 0 at kernel end
ARP_DEV unsigned long long* exp_t() { __shared__ unsigned long long t[16]; return t; }
ARP_DEV unsigned long long exp_now() { return __builtin_readcyclecounter(); }
#define ARP_T(k, t0) do { unsigned long long n_ = arp::exp_now(); if (threadIdx.x == 0) arp::exp_t()[k] += n_ - t0; t0 = n_; } while (0)
#define ARP_T0(t0) unsigned long long t0 = arp::exp_now()
#else
#define ARP_T(k, t0) do {} while (0)
#define ARP_T0(t0) do {} while (0)
#endif
}  // namespace arp
