// Radon hierarchical regression (reference models.py:809-857), collapsed to
// per-county sufficient statistics and evaluated under the general VIP
// parameterisation (program_transformations.py:555-600).
//
//   mua, b1, b2 ~ N(0,1)                      (top level: VIP map is the identity)
//   m_j ~ N(mu_j, 1),  mu_j = mua + u_j b1    (sigma = 1, so only `a` matters)
//   y_i ~ N(m_{c_i} + x_i b2, 1)
//
// State holds mt_j with mt_j ~ N(a_j mu_j, 1), m_j = mt_j + (1 - a_j) mu_j.
// With r_j = mt_j - a_j mu_j,  l_j = Sy_j - b2 Sx_j - n_j m_j:
//   d/dmt_j = l_j - r_j =: g_j         h_j := d/dmu_j = l_j - a_j g_j
//   d/dmua = -mua + sum h_j            d/db1 = -b1 + sum u_j h_j
//   d/db2  = -b2 + Sxy - b2 Sxx - sum m_j Sx_j
// (checked against float64 autograd in tests/).
#pragma once
#include "arp_device.h"

namespace arp {

struct RadonArgs {
  const float* n;    // [J] observations per county
  const float* sx;   // [J] sum of floor indicators
  const float* sy;   // [J] sum of log radon
  const float* u;    // [J] log uranium
  float sxy, sxx;    // totals over all observations
  float sy_tot, suy_tot;   // sum_j Sy_j, sum_j u_j Sy_j: gradient of the non-centred log joint at the origin (radon_fast.h)
  int J;
};

template <int K_, int NL_>
struct RadonLane {
  static constexpr int K = K_;
  static constexpr int NG = 3;   // mua, b1, b2 replicated in every lane of the chain
  static constexpr int NL = NL_; // counties owned by this lane: j = slot + K*i
  static constexpr int ND = NG + NL;
  static constexpr int NGRP = NL_;   // groups owned by a lane (what the host matches against ceil(groups / K))
  static constexpr int DCAP = NG + K_ * NL_;   // upper bound of the flattened state dimension D
  // sliced element i of this lane: flattened index and validity
  ARP_DEV int lbase(int) const { return LBASE + slot; }
  static constexpr ARP_DEV int loff(int i) { return K * i; }
  ARP_DEV int lidx(int i) const { return LBASE + slot + K * i; }
  // only the last slice can be padding: NL == ceil(groups / K) is enforced by the host
  ARP_DEV bool lvalid(int i) const { return i < NL - 1 ? true : last_ok; }
  bool last_ok;
  // VALU issue needs two resident waves per SIMD: cap the allocation at 256 VGPRs where the slice fits
  static constexpr int MINW = (NL_ <= 23) ? 2 : 1;
  static constexpr bool HAS_MODES = false;  // centred / non-centred runs use the packed kernels of radon_fast.h
  static constexpr bool HAS_CARRY = false;
  static constexpr bool HAS_FUSED = true;   // kick_drift below  // grad_m / to_centered_m / from_centered_m below
  static constexpr bool HAS_VI = true;
  static constexpr int MOM_SPEC = 1;   // momentum stream layout 1 (kernels.h: hmc_transition, radon_fast.h)
  static constexpr bool HAS_MODE_STATE = false;   // nothing but (a, b) depends on the parameterisation
  using Args = RadonArgs;

  static constexpr int LBASE = 3; // flattened index of m_0 (parts: mua, b1, b2, m[J])

  // tables are held as register pairs so that the packed (v_pk_*_f32) leapfrog pass can use
  // them without shuffles; scalar code reads element i as T[i >> 1][i & 1] (a sub-register)
  static constexpr int NP = (NL + 1) / 2;
  v2f n2[NP], sx2[NP], sy2[NP], u2[NP], a2[NP];
  float sxy, sxx;
  int slot;

  // flattened index of replicated global i
  static ARP_DEV int gg(int i) { return i; }

  ARP_DEV void init(const Args& A, const float* av, const float* bv, int slot_) {
    slot = slot_;
    const int J = A.J;
    last_ok = slot + K * (NL - 1) < J;
    sxy = A.sxy;
    sxx = A.sxx;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      int j = slot + K * i;
      bool ok = j < J;
      n2[i >> 1][i & 1] = ok ? A.n[j] : 0.0f;
      sx2[i >> 1][i & 1] = ok ? A.sx[j] : 0.0f;
      sy2[i >> 1][i & 1] = ok ? A.sy[j] : 0.0f;
      u2[i >> 1][i & 1] = ok ? A.u[j] : 0.0f;
    }
    set_param(av, bv);
  }
  // (re)load the parameterisation-dependent slice (the interleaved kernel switches it twice per step)
  ARP_DEV void set_param(const float* av, const float* /*bv*/) {
#pragma unroll
    for (int i = 0; i < NL; ++i) a2[i >> 1][i & 1] = lvalid(i) ? av[LBASE + slot + K * i] : 0.0f;
  }

  // Gradient of the log joint at q (and the log joint itself, additive
  // constants dropped, when LOGP).  Padding slots (j >= J) keep q = g = 0.
  template <bool LOGP>
  ARP_DEV float grad(const float (&q)[ND], float (&g)[ND]) const {
    const float mua = q[0], b1 = q[1], b2 = q[2];
    float acc_h = 0.0f, acc_uh = 0.0f, acc_ms = 0.0f, lp = 0.0f;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      float mt = q[NG + i];
      float mu = fmaf(u2[i >> 1][i & 1], b1, mua);
      float r = fmaf(-a2[i >> 1][i & 1], mu, mt);
      float m = r + mu;
      float t = fmaf(-b2, sx2[i >> 1][i & 1], sy2[i >> 1][i & 1]);
      float l = fmaf(-n2[i >> 1][i & 1], m, t);
      float gm = l - r;
      g[NG + i] = gm;
      float h = fmaf(-a2[i >> 1][i & 1], gm, l);
      acc_h += h;
      acc_uh = fmaf(u2[i >> 1][i & 1], h, acc_uh);
      acc_ms = fmaf(m, sx2[i >> 1][i & 1], acc_ms);
      if (LOGP) {
        lp = fmaf(-0.5f * r, r, lp);
        lp = fmaf(-0.5f * m, fmaf(n2[i >> 1][i & 1], m, -2.0f * t), lp);
      }
    }
    acc_h = group_sum<K>(acc_h);
    acc_uh = group_sum<K>(acc_uh);
    acc_ms = group_sum<K>(acc_ms);
    g[0] = acc_h - mua;
    g[1] = acc_uh - b1;
    g[2] = fmaf(-b2, sxx, sxy) - acc_ms - b2;
    if (LOGP) {
      lp = group_sum<K>(lp);
      lp += -0.5f * (mua * mua + b1 * b1 + b2 * b2) + b2 * (sxy - 0.5f * b2 * sxx);
    }
    return lp;
  }

  // ---- compile-time parameterisations (kernels.h: kModeCP = 1, kModeNCP = 2) ----
  // CP (a = 1):  r_j = mt_j - mu_j, m_j = mt_j, h_j = r_j.
  // NCP (a = 0): r_j = mt_j, m_j = mt_j + mu_j, h_j = l_j.
  // 8 VALU ops per county instead of 10, and no `a` registers.  Only the last
  // slice of a lane can be padding (j >= J); CP masks its r there.
  template <bool LOGP, int MODE>
  ARP_DEV float grad_m(const float (&q)[ND], float (&g)[ND]) const {
    const float mua = q[0], b1 = q[1], b2 = q[2];
    const v2f vmua = {mua, mua}, vb1 = {b1, b1}, vnb2 = {-b2, -b2}, mhalf = {-0.5f, -0.5f}, mtwo = {-2.0f, -2.0f};
    v2f acc_h = {0.0f, 0.0f}, acc_uh = {0.0f, 0.0f}, acc_ms = {0.0f, 0.0f}, lpv = {0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < NL / 2; ++k) {   // county pairs on the packed f32 pipe
      const int i = 2 * k;
      const v2f mt = {q[NG + i], q[NG + i + 1]};
      const v2f mu = vfma(u2[k], vb1, vmua);
      const v2f t = vfma(vnb2, sx2[k], sy2[k]);
      v2f r, m;
      if (MODE == 1) {
        r = mt - mu;
        if (i + 1 == NL - 1) r[1] = last_ok ? r[1] : 0.0f;
        m = mt;
      } else {
        r = mt;
        m = mt + mu;
      }
      const v2f l = vfma(-n2[k], m, t);
      const v2f gm = l - r;
      g[NG + i] = gm[0]; g[NG + i + 1] = gm[1];
      const v2f h = (MODE == 1) ? r : l;
      acc_h += h;
      acc_uh = vfma(u2[k], h, acc_uh);
      acc_ms = vfma(m, sx2[k], acc_ms);
      if (LOGP) {
        lpv = vfma(mhalf * r, r, lpv);
        lpv = vfma(mhalf * m, vfma(n2[k], m, mtwo * t), lpv);
      }
    }
    float s_h = acc_h[0] + acc_h[1], s_uh = acc_uh[0] + acc_uh[1], s_ms = acc_ms[0] + acc_ms[1];
    float lp = lpv[0] + lpv[1];
    if (NL & 1) {
      constexpr int i = NL - 1;
      const float mt = q[NG + i];
      const float mu = fmaf(u2[i >> 1][0], b1, mua);
      const float t = fmaf(-b2, sx2[i >> 1][0], sy2[i >> 1][0]);
      float r, m;
      if (MODE == 1) { r = last_ok ? mt - mu : 0.0f; m = mt; } else { r = mt; m = mt + mu; }
      const float l = fmaf(-n2[i >> 1][0], m, t);
      g[NG + i] = l - r;
      const float h = (MODE == 1) ? r : l;
      s_h += h;
      s_uh = fmaf(u2[i >> 1][0], h, s_uh);
      s_ms = fmaf(m, sx2[i >> 1][0], s_ms);
      if (LOGP) {
        lp = fmaf(-0.5f * r, r, lp);
        lp = fmaf(-0.5f * m, fmaf(n2[i >> 1][0], m, -2.0f * t), lp);
      }
    }
    s_h = group_sum<K>(s_h);
    s_uh = group_sum<K>(s_uh);
    s_ms = group_sum<K>(s_ms);
    g[0] = s_h - mua;
    g[1] = s_uh - b1;
    g[2] = fmaf(-b2, sxx, sxy) - s_ms - b2;
    if (LOGP) {
      lp = group_sum<K>(lp);
      lp += -0.5f * (mua * mua + b1 * b1 + b2 * b2) + b2 * (sxy - 0.5f * b2 * sxx);
    }
    return lp;
  }

  // Interior leapfrog step in one pass (kernels.h: lane_kick_drift): for every county the
  // gradient is formed, kicked into p and the position drifted at once; the three
  // top-level scalars follow after the group sums.  MODE 0 uses the `a` table.
  template <int MODE>
  ARP_DEV void kick_drift(float (&q)[ND], float (&p)[ND], const float (&eps)[ND]) const {
    const float mua = q[0], b1 = q[1], b2 = q[2];
    const v2f vmua = {mua, mua}, vb1 = {b1, b1}, vnb2 = {-b2, -b2};
    v2f acc_h = {0.0f, 0.0f}, acc_uh = {0.0f, 0.0f}, acc_ms = {0.0f, 0.0f};
    // counties in pairs: every operation below is one v_pk_fma_f32 / v_pk_add_f32 on a register pair
#pragma unroll
    for (int k = 0; k < NL / 2; ++k) {
      const int i = 2 * k;
      const v2f mt = {q[NG + i], q[NG + i + 1]};
      const v2f ev = {eps[NG + i], eps[NG + i + 1]};
      const v2f pv = {p[NG + i], p[NG + i + 1]};
      const v2f mu = vfma(u2[k], vb1, vmua);
      const v2f t = vfma(vnb2, sx2[k], sy2[k]);
      v2f r, m, h;
      if (MODE == 1) {
        r = mt - mu;
        if (i + 1 == NL - 1) r[1] = last_ok ? r[1] : 0.0f;
        m = mt;
      } else if (MODE == 2) {
        r = mt;
        m = mt + mu;
      } else {
        r = vfma(-a2[k], mu, mt);
        m = r + mu;
      }
      const v2f l = vfma(-n2[k], m, t);
      const v2f gm = l - r;
      h = (MODE == 1) ? r : ((MODE == 2) ? l : vfma(-a2[k], gm, l));
      acc_h += h;
      acc_uh = vfma(u2[k], h, acc_uh);
      acc_ms = vfma(m, sx2[k], acc_ms);
      const v2f pn = vfma(ev, gm, pv);
      const v2f qn = vfma(ev, pn, mt);
      p[NG + i] = pn[0]; p[NG + i + 1] = pn[1];
      q[NG + i] = qn[0]; q[NG + i + 1] = qn[1];
    }
    float s_h = acc_h[0] + acc_h[1], s_uh = acc_uh[0] + acc_uh[1], s_ms = acc_ms[0] + acc_ms[1];
    if (NL & 1) {   // odd slice count: the last county on its own
      constexpr int i = NL - 1;
      const float mt = q[NG + i];
      const float mu = fmaf(u2[i >> 1][0], b1, mua);
      const float t = fmaf(-b2, sx2[i >> 1][0], sy2[i >> 1][0]);
      float r, m, h;
      if (MODE == 1) { r = last_ok ? mt - mu : 0.0f; m = mt; }
      else if (MODE == 2) { r = mt; m = mt + mu; }
      else { r = fmaf(-a2[i >> 1][0], mu, mt); m = r + mu; }
      const float l = fmaf(-n2[i >> 1][0], m, t);
      const float gm = l - r;
      h = (MODE == 1) ? r : ((MODE == 2) ? l : fmaf(-a2[i >> 1][0], gm, l));
      s_h += h;
      s_uh = fmaf(u2[i >> 1][0], h, s_uh);
      s_ms = fmaf(m, sx2[i >> 1][0], s_ms);
      const float pn = fmaf(eps[NG + i], gm, p[NG + i]);
      p[NG + i] = pn;
      q[NG + i] = fmaf(eps[NG + i], pn, mt);
    }
    s_h = group_sum<K>(s_h);
    s_uh = group_sum<K>(s_uh);
    s_ms = group_sum<K>(s_ms);
    const float g0 = s_h - mua, g1 = s_uh - b1, g2 = fmaf(-b2, sxx, sxy) - s_ms - b2;
    p[0] = fmaf(eps[0], g0, p[0]); q[0] = fmaf(eps[0], p[0], mua);
    p[1] = fmaf(eps[1], g1, p[1]); q[1] = fmaf(eps[1], p[1], b1);
    p[2] = fmaf(eps[2], g2, p[2]); q[2] = fmaf(eps[2], p[2], b2);
  }

  // Change of coordinates CP <-> NCP of a state AND its gradient (interleaved kernel).  The map
  // m = mt + mu(mua, b1) is a shear with unit Jacobian: the log density is unchanged and
  //   d/dmt_j = d/dm_j,   d/dmua (NCP) = d/dmua (CP) + sum_j d/dm_j,   d/db1 likewise with u_j.
  // FROM == 1: CP -> NCP, FROM == 2: NCP -> CP.
  template <int FROM>
  ARP_DEV void carry(float (&q)[ND], float (&g)[ND]) const {
    float s = 0.0f, su = 0.0f;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const float mu = fmaf(u2[i >> 1][i & 1], q[1], q[0]);
      const float gj = g[NG + i];               // 0 in padding slots
      s += gj;
      su = fmaf(u2[i >> 1][i & 1], gj, su);
      const float qn = (FROM == 1) ? q[NG + i] - mu : q[NG + i] + mu;
      q[NG + i] = lvalid(i) ? qn : 0.0f;
    }
    s = group_sum<K>(s);
    su = group_sum<K>(su);
    g[0] += (FROM == 1) ? s : -s;
    g[1] += (FROM == 1) ? su : -su;
  }

  template <int MODE>
  ARP_DEV void to_centered_m(const float (&q)[ND], float (&x)[ND]) const {
#pragma unroll
    for (int i = 0; i < ND; ++i) x[i] = q[i];
    if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < NL; ++i) x[NG + i] = q[NG + i] + fmaf(u2[i >> 1][i & 1], q[1], q[0]);
    }
  }
  template <int MODE>
  ARP_DEV void from_centered_m(const float (&x)[ND], float (&q)[ND]) const {
#pragma unroll
    for (int i = 0; i < ND; ++i) q[i] = x[i];
    if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < NL; ++i) q[NG + i] = lvalid(i) ? x[NG + i] - fmaf(u2[i >> 1][i & 1], x[1], x[0]) : 0.0f;
    } else {
#pragma unroll
      for (int i = 0; i < NL; ++i) q[NG + i] = lvalid(i) ? x[NG + i] : 0.0f;
    }
  }

  // d logp / d a_i and d logp / d b_i from the state gradient g (cVIP learns a):
  //   d/da = -mu g,   d/db = -log(sigma) (1 + (xt - a mu) g);   here sigma = 1.
  ARP_DEV void dparam(const float (&q)[ND], const float (&g)[ND], float (&da)[ND], float (&db)[ND]) const {
#pragma unroll
    for (int i = 0; i < ND; ++i) { da[i] = 0.0f; db[i] = 0.0f; }
#pragma unroll
    for (int i = 0; i < NL; ++i) da[NG + i] = -fmaf(u2[i >> 1][i & 1], q[1], q[0]) * g[NG + i];
  }

  // reparameterised -> centred coordinates
  ARP_DEV void to_centered(const float (&q)[ND], float (&x)[ND]) const {
    x[0] = q[0]; x[1] = q[1]; x[2] = q[2];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      float mu = fmaf(u2[i >> 1][i & 1], q[1], q[0]);
      x[NG + i] = fmaf(-a2[i >> 1][i & 1], mu, q[NG + i]) + mu;
    }
  }
  // centred -> reparameterised coordinates
  ARP_DEV void from_centered(const float (&x)[ND], float (&q)[ND]) const {
    q[0] = x[0]; q[1] = x[1]; q[2] = x[2];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      float mu = fmaf(u2[i >> 1][i & 1], x[1], x[0]);
      // mt = m - (1-a) mu ; padding slots (a = 0, x = 0 on input) must stay 0
      q[NG + i] = lvalid(i) ? x[NG + i] - (1.0f - a2[i >> 1][i & 1]) * mu : 0.0f;
    }
  }
};

}  // namespace arp
