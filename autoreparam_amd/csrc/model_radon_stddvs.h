// Radon with per-county observation scales (reference models.py:763-806,
// `radon_stddvs`) under the general VIP parameterisation.  Parts in trace order:
//   mua, b1, b2, m[J], log_m_stddv[J].
//
//   mua, b1, b2 ~ N(0,1);  m_j ~ N(mua + u_j b1, 1);  s_j = log_m_stddv_j ~ N(0,1)
//   y_i ~ N(m_{c_i} + x_i b2, exp(s_{c_i}))
// Only `m` has a non-trivial VIP map (its scale is 1, so only `a` matters); s_j has
// loc 0 / scale 1.  Per county, with the sufficient statistics n, Sx, Sy, Sxx, Sxy, Syy:
//   t = Sy - b2 Sx,  resid = t - n m,  w = exp(-2 s),
//   Q = Syy - 2 b2 Sxy + b2^2 Sxx - m (resid + t)         (sum of squared residuals)
//   loglik_j = -n s - w Q / 2
//   d/dm = w resid,  d/ds = -n + w Q,  d/db2 += w (Sxy - m Sx - b2 Sxx)
#pragma once
#include "arp_device.h"

namespace arp {

struct RadonSdArgs {
  const float *n, *sx, *sy, *sxx, *sxy, *syy, *u;   // [J] each
  int J;
};

template <int K_, int NLS_>
struct RadonSdLane {
  static constexpr int K = K_;
  static constexpr int NG = 3;
  static constexpr int NLS = NLS_;        // counties owned by this lane: j = slot + K*i
  static constexpr int NL = 2 * NLS;      // local elements: m slices, then log_m_stddv slices
  static constexpr int ND = NG + NL;
  static constexpr int NGRP = NLS_;   // groups owned by a lane (what the host matches against ceil(groups / K))
  static constexpr int DCAP = NG + 2 * K_ * NLS_;
  static constexpr bool HAS_MODES = false;
  static constexpr bool HAS_CARRY = false;
  static constexpr bool HAS_FUSED = false;
  static constexpr bool HAS_VI = true;
  static constexpr int MINW = 1;
  using Args = RadonSdArgs;

  float n[NLS], sx[NLS], sy[NLS], sxx[NLS], sxy[NLS], syy[NLS], u[NLS], a[NLS];
  int J, slot;
  bool last_ok;

  static ARP_DEV int gg(int i) { return i; }
  ARP_DEV int lbase(int i) const { return i < NLS ? NG + slot : NG + J + slot; }
  static constexpr ARP_DEV int loff(int i) { return K * (i < NLS ? i : i - NLS); }
  ARP_DEV int lidx(int i) const { return lbase(i) + loff(i); }
  // only the last slice of each part can be padding (NLS == ceil(J / K), enforced by the host)
  ARP_DEV bool lvalid(int i) const { return (i < NLS ? i : i - NLS) < NLS - 1 ? true : last_ok; }

  ARP_DEV void init(const Args& A, const float* av, const float* bv, int slot_) {
    slot = slot_;
    J = A.J;
    last_ok = slot + K * (NLS - 1) < J;
#pragma unroll
    for (int i = 0; i < NLS; ++i) {
      int j = slot + K * i;
      bool ok = j < J;
      n[i] = ok ? A.n[j] : 0.0f;  sx[i] = ok ? A.sx[j] : 0.0f;  sy[i] = ok ? A.sy[j] : 0.0f;
      sxx[i] = ok ? A.sxx[j] : 0.0f;  sxy[i] = ok ? A.sxy[j] : 0.0f;  syy[i] = ok ? A.syy[j] : 0.0f;
      u[i] = ok ? A.u[j] : 0.0f;
    }
    set_param(av, bv);
  }
  ARP_DEV void set_param(const float* av, const float* /*bv*/) {
#pragma unroll
    for (int i = 0; i < NLS; ++i) a[i] = lvalid(i) ? av[NG + slot + K * i] : 0.0f;
  }

  template <bool LOGP>
  ARP_DEV float grad(const float (&q)[ND], float (&g)[ND]) const {
    const float mua = q[0], b1 = q[1], b2 = q[2];
    float acc_h = 0.0f, acc_uh = 0.0f, acc_b2 = 0.0f, lp = 0.0f;
#pragma unroll
    for (int i = 0; i < NLS; ++i) {
      const float mt = q[NG + i], s = q[NG + NLS + i];
      const float mu = fmaf(u[i], b1, mua);
      const float r = fmaf(-a[i], mu, mt);
      const float m = r + mu;
      const float w = fast_exp(-2.0f * s);
      const float t = fmaf(-b2, sx[i], sy[i]);
      const float resid = fmaf(-n[i], m, t);
      const float c = fmaf(b2, fmaf(b2, sxx[i], -2.0f * sxy[i]), syy[i]);
      const float Q = fmaf(-m, resid + t, c);
      const float l = w * resid;                       // d loglik / d m
      const float gm = l - r;
      g[NG + i] = gm;
      g[NG + NLS + i] = fmaf(w, Q, -n[i]) - s;          // padding: n = Q = s = 0
      const float h = fmaf(-a[i], gm, l);
      acc_h += h;
      acc_uh = fmaf(u[i], h, acc_uh);
      acc_b2 = fmaf(w, fmaf(-b2, sxx[i], fmaf(-m, sx[i], sxy[i])), acc_b2);
      if (LOGP) lp += fmaf(-0.5f * r, r, fmaf(-0.5f * w, Q, fmaf(-n[i], s, -0.5f * s * s)));
    }
    acc_h = group_sum<K>(acc_h);
    acc_uh = group_sum<K>(acc_uh);
    acc_b2 = group_sum<K>(acc_b2);
    g[0] = acc_h - mua;
    g[1] = acc_uh - b1;
    g[2] = acc_b2 - b2;
    if (LOGP) lp = group_sum<K>(lp) - 0.5f * (mua * mua + b1 * b1 + b2 * b2);
    return lp;
  }

  ARP_DEV void dparam(const float (&q)[ND], const float (&g)[ND], float (&da)[ND], float (&db)[ND]) const {
#pragma unroll
    for (int i = 0; i < ND; ++i) { da[i] = 0.0f; db[i] = 0.0f; }
#pragma unroll
    for (int i = 0; i < NLS; ++i) da[NG + i] = -fmaf(u[i], q[1], q[0]) * g[NG + i];
  }

  ARP_DEV void to_centered(const float (&q)[ND], float (&x)[ND]) const {
#pragma unroll
    for (int i = 0; i < ND; ++i) x[i] = q[i];
#pragma unroll
    for (int i = 0; i < NLS; ++i) {
      float mu = fmaf(u[i], q[1], q[0]);
      x[NG + i] = fmaf(-a[i], mu, q[NG + i]) + mu;
    }
  }
  ARP_DEV void from_centered(const float (&x)[ND], float (&q)[ND]) const {
#pragma unroll
    for (int i = 0; i < ND; ++i) q[i] = x[i];
#pragma unroll
    for (int i = 0; i < NLS; ++i) {
      float mu = fmaf(u[i], x[1], x[0]);
      q[NG + i] = lvalid(i) ? x[NG + i] - (1.0f - a[i]) * mu : 0.0f;
      q[NG + NLS + i] = lvalid(i) ? x[NG + NLS + i] : 0.0f;
    }
  }
};

}  // namespace arp
