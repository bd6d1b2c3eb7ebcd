// Eight schools (reference models.py:131-166) under the general VIP
// parameterisation.  Parts in trace order: mu, log_tau, theta[8].
//
//   mu ~ N(0,5), log_tau ~ N(0,5)      top level: xt ~ N(0, 5^b), x = 5^(1-b) xt
//   theta_k ~ N(mu, tau), tau = exp(log_tau):
//       tt_k ~ N(a_k mu, tau^b_k),  z_k = (tt_k - a_k mu) exp(-b_k lt),  theta_k = mu + tau z_k
//   y_k ~ N(theta_k, s_k)
// With w_k = (y_k - theta_k)/s_k^2:
//   d/dtt_k = e_k (tau w_k - z_k) =: g_k      d/dmu  = sum (w_k - a_k g_k)
//   d/dlt   = sum (b_k z_k^2 - b_k + w_k tau z_k (1 - b_k))
#pragma once
#include "arp_device.h"

namespace arp {

struct SchoolsArgs {
  const float* y;      // [8] treatment effects
  const float* sigma;  // [8] treatment stddevs
};

template <int K_, int NL_>
struct SchoolsLane {
  static constexpr int K = K_;
  static constexpr int NG = 2;   // mu, log_tau
  static constexpr int NL = NL_; // schools owned by this lane: k = slot + K*i
  static constexpr int ND = NG + NL;
  static constexpr int NGRP = NL_;   // groups owned by a lane (what the host matches against ceil(groups / K))
  static constexpr int DCAP = 10;   // upper bound of the flattened state dimension D
  static constexpr int LBASE = 2;
  // sliced element i of this lane: flattened index and validity
  ARP_DEV int lbase(int) const { return LBASE + slot; }
  static constexpr ARP_DEV int loff(int i) { return K * i; }
  ARP_DEV int lidx(int i) const { return LBASE + slot + K * i; }
  // only the last slice can be padding: NL == ceil(groups / K) is enforced by the host
  ARP_DEV bool lvalid(int i) const { return i < NL - 1 ? true : last_ok; }
  bool last_ok;
  static constexpr bool HAS_MODES = false;
  static constexpr bool HAS_CARRY = false;
  static constexpr bool HAS_FUSED = false;
  static constexpr bool HAS_VI = true;
  static constexpr int MINW = 1;   // waves per SIMD the register allocator must leave room for
  using Args = SchoolsArgs;

  float y[NL], is2[NL], a[NL], b[NL];
  float s0i, s1i, c0, c1;  // 1/5^b0, 1/5^b1, 5^(1-b0), 5^(1-b1)
  int slot;

  static ARP_DEV int gg(int i) { return i; }

  ARP_DEV void init(const Args& A, const float* av, const float* bv, int slot_) {
    slot = slot_;
    last_ok = slot + K * (NL - 1) < 8;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      int k = slot + K * i;
      bool ok = k < 8;
      y[i] = ok ? A.y[k] : 0.0f;
      float s = ok ? A.sigma[k] : 1.0f;
      is2[i] = ok ? 1.0f / (s * s) : 0.0f;
    }
    set_param(av, bv);
  }
  ARP_DEV void set_param(const float* av, const float* bv) {
    const float l5 = 2.321928094887362f;  // log2(5)
    s0i = __builtin_amdgcn_exp2f(-bv[0] * l5);
    s1i = __builtin_amdgcn_exp2f(-bv[1] * l5);
    c0 = 5.0f * s0i;
    c1 = 5.0f * s1i;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      bool ok = lvalid(i);
      a[i] = ok ? av[LBASE + slot + K * i] : 0.0f;
      b[i] = ok ? bv[LBASE + slot + K * i] : 0.0f;
    }
  }

  template <bool LOGP>
  ARP_DEV float grad(const float (&q)[ND], float (&g)[ND]) const {
    const float mu = c0 * q[0], lt = c1 * q[1];
    const float tau = fast_exp(lt);
    float g_mu = 0.0f, g_lt = 0.0f, lp = 0.0f;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      float e = fast_exp(-b[i] * lt);
      float z = (q[NG + i] - a[i] * mu) * e;
      float th = fmaf(tau, z, mu);
      float r = y[i] - th;
      float w = r * is2[i];
      float gk = e * fmaf(tau, w, -z);
      g[NG + i] = gk;          // padding: q = 0, a = 0 -> z = 0; is2 = 0 -> w = 0 -> gk = 0
      g_mu += fmaf(-a[i], gk, w);
      g_lt += fmaf(b[i], fmaf(z, z, -1.0f), w * tau * z * (1.0f - b[i]));
      if (LOGP) lp += fmaf(-0.5f * z, z, -b[i] * lt) - 0.5f * r * w;
    }
    g_mu = group_sum<K>(g_mu);
    g_lt = group_sum<K>(g_lt);
    float u0 = q[0] * s0i, u1 = q[1] * s1i;
    g[0] = fmaf(c0, g_mu, -u0 * s0i);
    g[1] = fmaf(c1, g_lt, -u1 * s1i);
    if (LOGP) {
      lp = group_sum<K>(lp);
      lp += -0.5f * (u0 * u0 + u1 * u1);
    }
    return lp;
  }

  // d logp / d a, d logp / d b from the state gradient (see model_radon.h)
  ARP_DEV void dparam(const float (&q)[ND], const float (&g)[ND], float (&da)[ND], float (&db)[ND]) const {
    const float ln5 = 1.6094379124341003f;
    const float mu = c0 * q[0], lt = c1 * q[1];
    da[0] = 0.0f; da[1] = 0.0f;
    db[0] = -ln5 * fmaf(q[0], g[0], 1.0f);
    db[1] = -ln5 * fmaf(q[1], g[1], 1.0f);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      bool ok = lvalid(i);
      da[NG + i] = ok ? -mu * g[NG + i] : 0.0f;
      db[NG + i] = ok ? -lt * fmaf(q[NG + i] - a[i] * mu, g[NG + i], 1.0f) : 0.0f;
    }
  }

  ARP_DEV void to_centered(const float (&q)[ND], float (&x)[ND]) const {
    const float mu = c0 * q[0], lt = c1 * q[1];
    x[0] = mu; x[1] = lt;
#pragma unroll
    for (int i = 0; i < NL; ++i)
      x[NG + i] = fmaf(fast_exp((1.0f - b[i]) * lt), q[NG + i] - a[i] * mu, mu);
  }
  ARP_DEV void from_centered(const float (&x)[ND], float (&q)[ND]) const {
    const float mu = x[0], lt = x[1];
    q[0] = mu / c0; q[1] = lt / c1;
#pragma unroll
    for (int i = 0; i < NL; ++i)
      q[NG + i] = lvalid(i) ? fmaf(x[NG + i] - mu, fast_exp(-(1.0f - b[i]) * lt), a[i] * mu) : 0.0f;
  }
};

}  // namespace arp
