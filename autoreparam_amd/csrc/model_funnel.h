// Neal's funnel (reference models.py:671-696) under the general VIP parameterisation.
// Parts in trace order: x1, x2.
//   x1 ~ N(0,3)                 top level: xt1 ~ N(0, 3^b1), x1 = 3^(1-b1) xt1
//   x2 ~ N(0, exp(x1/2))        loc 0: xt2 ~ N(0, sigma^b2), sigma = exp(x1/2), x2 = sigma^(1-b2) xt2
// No observations.  log p = -u1^2/2 - z^2/2 - b2 x1/2 with u1 = xt1/3^b1, z = xt2 exp(-b2 x1/2):
//   d/dxt2 = -z e,  d/dx1 = b2 (z^2 - 1)/2,  d/dxt1 = -u1/3^b1 + 3^(1-b1) d/dx1.
// One lane per chain (the model has two dimensions).
#pragma once
#include "arp_device.h"

namespace arp {

struct FunnelArgs { int unused; };

template <int K_, int NL_>
struct FunnelLane {
  static constexpr int K = K_;
  static constexpr int NG = 1, NL = NL_, ND = NG + NL, NGRP = NL_, DCAP = 2, LBASE = 1;
  static_assert(K_ == 1 && NL_ == 1, "the funnel runs one lane per chain");
  static constexpr bool HAS_MODES = false;
  static constexpr bool HAS_CARRY = false;
  static constexpr bool HAS_FUSED = false;
  static constexpr bool HAS_VI = true;
  static constexpr int MINW = 1;
  using Args = FunnelArgs;

  float s1i, c1, b2;   // 1/3^b1, 3^(1-b1), b of x2
  int slot;

  static ARP_DEV int gg(int) { return 0; }
  ARP_DEV int lbase(int) const { return 1; }
  static constexpr ARP_DEV int loff(int) { return 0; }
  ARP_DEV int lidx(int) const { return 1; }
  ARP_DEV bool lvalid(int) const { return true; }

  ARP_DEV void init(const Args&, const float* av, const float* bv, int slot_) {
    slot = slot_;
    set_param(av, bv);
  }
  ARP_DEV void set_param(const float* /*av*/, const float* bv) {
    s1i = __builtin_amdgcn_exp2f(-bv[0] * 1.584962500721156f);   // 3^-b1
    c1 = 3.0f * s1i;
    b2 = bv[1];
  }

  template <bool LOGP>
  ARP_DEV float grad(const float (&q)[ND], float (&g)[ND]) const {
    const float x1 = c1 * q[0];
    const float e = fast_exp(-0.5f * b2 * x1);
    const float z = q[1] * e;
    const float u1 = q[0] * s1i;
    g[1] = -z * e;
    g[0] = fmaf(c1, 0.5f * b2 * fmaf(z, z, -1.0f), -u1 * s1i);
    return LOGP ? fmaf(-0.5f * u1, u1, fmaf(-0.5f * z, z, -0.5f * b2 * x1)) : 0.0f;
  }
  ARP_DEV void dparam(const float (&q)[ND], const float (&g)[ND], float (&da)[ND], float (&db)[ND]) const {
    da[0] = 0.0f; da[1] = 0.0f;
    db[0] = -1.0986122886681098f * fmaf(q[0], g[0], 1.0f);
    db[1] = -0.5f * c1 * q[0] * fmaf(q[1], g[1], 1.0f);
  }
  ARP_DEV void to_centered(const float (&q)[ND], float (&x)[ND]) const {
    x[0] = c1 * q[0];
    x[1] = fast_exp(0.5f * (1.0f - b2) * x[0]) * q[1];
  }
  ARP_DEV void from_centered(const float (&x)[ND], float (&q)[ND]) const {
    q[0] = x[0] / c1;
    q[1] = x[1] * fast_exp(-0.5f * (1.0f - b2) * x[0]);
  }
};

}  // namespace arp
