// German credit, log-normal centred scales (reference models.py:884-923) under the
// general VIP parameterisation.  Parts in trace order:
//   overall_log_scale (ols), beta_log_scales[F] (bls), beta[F].
//
//   ols ~ N(0,10)                      top level: ot ~ N(0, 10^b), ols = 10^(1-b) ot
//   bls_d ~ N(ols, 1)                  sigma = 1: blt_d ~ N(a_d ols, 1), bls_d = blt_d + (1 - a_d) ols
//   beta_d ~ N(0, exp(bls_d))          mu = 0:   bt_d ~ N(0, exp(b_d bls_d)), beta_d = exp((1-b_d) bls_d) bt_d
//   y_n ~ Bernoulli(logit = X beta)    [N x F] dense contraction: this model is compute bound
//
// A chain is spread over K lanes that own F/K consecutive features each.  The
// design matrix (padded to 64 columns) streams through LDS in tiles shared by the
// whole workgroup; every lane forms its partial logit, the K partials are summed
// with DPP, and the residual y - sigmoid(eta) is scattered back to the lane's own
// columns.  With v_d = sum_n X_nd (y_n - sigmoid(eta_n)):
//   d/dbt_d  = -bt_d e_d^2 + v_d exp((1-b_d) bls_d),   e_d = exp(-b_d bls_d)
//   hb_d     = b_d (bt_d e_d)^2 - b_d + v_d (1-b_d) beta_d
//   d/dblt_d = hb_d - r_d,  r_d = blt_d - a_d ols;   d/dols = sum a_d r_d + (1-a_d) hb_d
#pragma once
#include "arp_device.h"

namespace arp {

constexpr int kGermanCols = 64;   // padded row length of the device design matrix
constexpr int kGermanTile = 64;   // rows per LDS tile

struct GermanArgs {
  const float* X;   // [N][64] row-major, columns >= F are zero
  const float* y;   // [N]
  int N, F;
};

template <int K_, int NLS_>
struct GermanLane {
  static constexpr int K = K_;
  static constexpr int NG = 1;          // overall_log_scale
  static constexpr int NLS = NLS_;      // features owned by this lane: d = slot*NLS + i
  static constexpr int NL = 2 * NLS;    // local elements: bls slices, then beta slices
  static constexpr int ND = NG + NL;
  static constexpr int NGRP = NLS_;   // groups owned by a lane (what the host matches against ceil(groups / K))
  static constexpr int DCAP = 1 + 2 * kGermanCols;   // upper bound of the flattened state dimension D
  static_assert(K_ * NLS_ == kGermanCols, "lanes x features per lane must cover the padded row");
  static constexpr bool HAS_MODES = false;
  static constexpr bool HAS_CARRY = false;
  static constexpr bool HAS_FUSED = false;
  static constexpr int MINW = 1;   // waves per SIMD the register allocator must leave room for
  using Args = GermanArgs;

  float a[NLS], b[NLS];     // a of bls_d, b of beta_d (the only ones that matter)
  float s0i, c0;            // 1/10^b0, 10^(1-b0)
  const float* X; const float* y;
  int N, F, slot, nown;

  static ARP_DEV int gg(int) { return 0; }
  ARP_DEV int lbase(int i) const { return i < NLS ? 1 + slot * NLS : 1 + F + slot * NLS; }
  static constexpr ARP_DEV int loff(int i) { return i < NLS ? i : i - NLS; }
  ARP_DEV int lidx(int i) const { return i < NLS ? 1 + slot * NLS + i : 1 + F + slot * NLS + (i - NLS); }
  ARP_DEV bool lvalid(int i) const { return (i < NLS ? i : i - NLS) < nown; }

  ARP_DEV void init(const Args& A, const float* av, const float* bv, int slot_) {
    slot = slot_;
    X = A.X; y = A.y; N = A.N; F = A.F;
    nown = F - slot * NLS;
    nown = nown < 0 ? 0 : (nown > NLS ? NLS : nown);
    set_param(av, bv);
  }
  ARP_DEV void set_param(const float* av, const float* bv) {
    s0i = __builtin_amdgcn_exp2f(-bv[0] * 3.321928094887362f);
    c0 = 10.0f * s0i;
#pragma unroll
    for (int i = 0; i < NLS; ++i) {
      bool ok = i < nown;
      a[i] = ok ? av[1 + slot * NLS + i] : 0.0f;
      b[i] = ok ? bv[1 + F + slot * NLS + i] : 0.0f;
    }
  }

  template <bool LOGP>
  ARP_DEV float grad(const float (&q)[ND], float (&g)[ND]) const {
    __shared__ float tile[kGermanTile * kGermanCols];
    __shared__ float ytile[kGermanTile];
    const float ols = c0 * q[0];
    float beta[NLS], v[NLS], bls[NLS], r[NLS];
#pragma unroll
    for (int i = 0; i < NLS; ++i) {
      r[i] = fmaf(-a[i], ols, q[NG + i]);
      bls[i] = r[i] + ols;
      beta[i] = fast_exp((1.0f - b[i]) * bls[i]) * q[NG + NLS + i];   // padding: q = 0 -> beta = 0
      v[i] = 0.0f;
    }
    float lp = 0.0f;
    const int nthreads = blockDim.x;
    for (int n0 = 0; n0 < N; n0 += kGermanTile) {
      __syncthreads();   // previous tile fully consumed
      const int rows = min(kGermanTile, N - n0);
      // cooperative, coalesced float4 copy of `rows` x 64 floats
      const float4* src = reinterpret_cast<const float4*>(X + (size_t)n0 * kGermanCols);
      float4* dst = reinterpret_cast<float4*>(tile);
      for (int t = threadIdx.x; t < rows * (kGermanCols / 4); t += nthreads) dst[t] = src[t];
      if ((int)threadIdx.x < rows) ytile[threadIdx.x] = y[n0 + threadIdx.x];
      __syncthreads();
      for (int n = 0; n < rows; ++n) {
        const float* xr = tile + n * kGermanCols + slot * NLS;
        float x[NLS];
#pragma unroll
        for (int i = 0; i < NLS; ++i) x[i] = xr[i];
        float eta = 0.0f;
#pragma unroll
        for (int i = 0; i < NLS; ++i) eta = fmaf(x[i], beta[i], eta);
        eta = group_sum<K>(eta);
        float ex = fast_exp(-fabsf(eta));
        float rc = __builtin_amdgcn_rcpf(1.0f + ex);
        float sg = eta >= 0.0f ? rc : ex * rc;
        float yn = ytile[n];
        float w = yn - sg;
#pragma unroll
        for (int i = 0; i < NLS; ++i) v[i] = fmaf(x[i], w, v[i]);
        if (LOGP) lp += fmaf(yn, eta, -(fmaxf(eta, 0.0f) + fast_log(1.0f + ex)));
      }
    }
    // every lane of the chain accumulated the same likelihood value
    float lq = 0.0f, g_ols = 0.0f;
#pragma unroll
    for (int i = 0; i < NLS; ++i) {
      float e = fast_exp(-b[i] * bls[i]);
      float zb = q[NG + NLS + i] * e;
      float hb = fmaf(b[i], fmaf(zb, zb, -1.0f), v[i] * (1.0f - b[i]) * beta[i]);
      bool ok = i < nown;
      g[NG + NLS + i] = ok ? fmaf(v[i], fast_exp((1.0f - b[i]) * bls[i]), -zb * e) : 0.0f;
      g[NG + i] = ok ? hb - r[i] : 0.0f;
      g_ols += ok ? fmaf(a[i], r[i], (1.0f - a[i]) * hb) : 0.0f;
      if (LOGP) lq += ok ? fmaf(-0.5f * r[i], r[i], fmaf(-0.5f * zb, zb, -b[i] * bls[i])) : 0.0f;
    }
    g_ols = group_sum<K>(g_ols);
    const float u0 = q[0] * s0i;
    g[0] = fmaf(c0, g_ols, -u0 * s0i);
    if (LOGP) lp += group_sum<K>(lq) - 0.5f * u0 * u0;
    return lp;
  }

  ARP_DEV void dparam(const float (&q)[ND], const float (&g)[ND], float (&da)[ND], float (&db)[ND]) const {
    const float ols = c0 * q[0];
#pragma unroll
    for (int i = 0; i < ND; ++i) { da[i] = 0.0f; db[i] = 0.0f; }
    db[0] = -2.302585092994046f * fmaf(q[0], g[0], 1.0f);
#pragma unroll
    for (int i = 0; i < NLS; ++i) {
      bool ok = i < nown;
      float bls = fmaf(-a[i], ols, q[NG + i]) + ols;
      da[NG + i] = ok ? -ols * g[NG + i] : 0.0f;
      db[NG + NLS + i] = ok ? -bls * fmaf(q[NG + NLS + i], g[NG + NLS + i], 1.0f) : 0.0f;
    }
  }

  ARP_DEV void to_centered(const float (&q)[ND], float (&x)[ND]) const {
    const float ols = c0 * q[0];
    x[0] = ols;
#pragma unroll
    for (int i = 0; i < NLS; ++i) {
      float bls = q[NG + i] + (1.0f - a[i]) * ols;
      x[NG + i] = bls;
      x[NG + NLS + i] = fast_exp((1.0f - b[i]) * bls) * q[NG + NLS + i];
    }
  }
  ARP_DEV void from_centered(const float (&x)[ND], float (&q)[ND]) const {
    const float ols = x[0];
    q[0] = ols / c0;
#pragma unroll
    for (int i = 0; i < NLS; ++i) {
      bool ok = i < nown;
      q[NG + i] = ok ? x[NG + i] - (1.0f - a[i]) * ols : 0.0f;
      q[NG + NLS + i] = ok ? x[NG + NLS + i] * fast_exp(-(1.0f - b[i]) * x[NG + i]) : 0.0f;
    }
  }
};

}  // namespace arp
