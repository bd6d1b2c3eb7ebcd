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
  static constexpr int MINW = 2;   // waves per SIMD the register allocator must leave room for (the tables are in LDS)
  using Args = RadonSdArgs;

  float a[NLS];
  int J, slot;
  bool last_ok;

  // The seven per-county statistics depend on (slot, slice) only, not on the chain: they live in an LDS table shared by
  // the workgroup (two ds_read_b128 per county and gradient; a wave's 64 lanes read K distinct entries, a broadcast)
  // instead of 7 NLS registers per lane -- what lets two waves share a SIMD at 8 lanes per chain.
  //   entry (slice i, slot) = [n sx sy sxx][-2 sxy, syy, u, sxy], 12 floats apart.
  // Bank conflicts (MI355X_MICROARCH.md, LDS): a ds_read_b128 is served in groups of 16 lanes over 64 banks, and the
  // K distinct entries a group touches must fall on distinct 4-bank windows: a stride of 12 dwords does that for 8 and
  // for 16 lanes per chain (12 s mod 64 is a permutation of the multiples of 4), a stride of 8 only for 8.  Both
  // reads must BE ds_read_b128: with the second quad's last float unused the compiler shrank it to ds_read_b96, which
  // is served in groups of 8 lanes over 32 banks -- slots s and s + 4 collided, 8 extra LDS cycles per county and
  // gradient (round 3: SQ_LDS_BANK_CONFLICT / SQ_INSTS_LDS = 3.0).  The fourth float now carries -2 sxy's partner.
  static constexpr int kEntry = 12;
  static ARP_DEV float* county_table() {
    __shared__ __attribute__((aligned(16))) float tab[NLS * K * kEntry];
    return tab;
  }
  struct County { float n, sx, sy, sxx, m2sxy, syy, u, sxy; };
  // the entry's index is laundered: the table is loop invariant, and left to itself the compiler hoists every read out of
  // the leapfrog loop into registers -- the very registers the table is there to save
  ARP_DEV County county(int i) const {
    int e = (i * K + slot) * kEntry;
    asm volatile("" : "+v"(e));
    const float4* t = reinterpret_cast<const float4*>(county_table() + e);
    const float4 p = t[0], r = t[1];
    return County{p.x, p.y, p.z, p.w, r.x, r.y, r.z, r.w};
  }

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
    float* tab = county_table();
    __syncthreads();   // a previous user of the table (none inside one kernel) is done
    if ((int)threadIdx.x < K) {   // the first chain of the workgroup fills the table for everybody
#pragma unroll 1
      for (int i = 0; i < NLS; ++i) {
        const int j = slot + K * i;
        const bool ok = j < J;
        float* e = tab + (i * K + slot) * kEntry;
        e[0] = ok ? A.n[j] : 0.0f;  e[1] = ok ? A.sx[j] : 0.0f;  e[2] = ok ? A.sy[j] : 0.0f;  e[3] = ok ? A.sxx[j] : 0.0f;
        e[4] = ok ? -2.0f * A.sxy[j] : 0.0f;  e[5] = ok ? A.syy[j] : 0.0f;  e[6] = ok ? A.u[j] : 0.0f;  e[7] = ok ? A.sxy[j] : 0.0f;
        e[8] = e[9] = e[10] = e[11] = 0.0f;
      }
    }
    __syncthreads();
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
      const County c_ = county(i);
      const float n_i = c_.n, sx_i = c_.sx, sy_i = c_.sy, sxx_i = c_.sxx, sxy_i = c_.sxy, syy_i = c_.syy, u_i = c_.u;
      const float m2sxy_i = c_.m2sxy;
      const float mu = fmaf(u_i, b1, mua);
      const float r = fmaf(-a[i], mu, mt);
      const float m = r + mu;
      const float w = fast_exp(-2.0f * s);
      const float t = fmaf(-b2, sx_i, sy_i);
      const float resid = fmaf(-n_i, m, t);
      const float c = fmaf(b2, fmaf(b2, sxx_i, m2sxy_i), syy_i);    // -2 sxy is exact: same value as before
      const float Q = fmaf(-m, resid + t, c);
      const float l = w * resid;                       // d loglik / d m
      const float gm = l - r;
      g[NG + i] = gm;
      g[NG + NLS + i] = fmaf(w, Q, -n_i) - s;          // padding: n = Q = s = 0
      const float h = fmaf(-a[i], gm, l);
      acc_h += h;
      acc_uh = fmaf(u_i, h, acc_uh);
      acc_b2 = fmaf(w, fmaf(-b2, sxx_i, fmaf(-m, sx_i, sxy_i)), acc_b2);
      if (LOGP) lp += fmaf(-0.5f * r, r, fmaf(-0.5f * w, Q, fmaf(-n_i, s, -0.5f * s * s)));
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
    for (int i = 0; i < NLS; ++i) da[NG + i] = -fmaf(county(i).u, q[1], q[0]) * g[NG + i];
  }

  ARP_DEV void to_centered(const float (&q)[ND], float (&x)[ND]) const {
#pragma unroll
    for (int i = 0; i < ND; ++i) x[i] = q[i];
#pragma unroll
    for (int i = 0; i < NLS; ++i) {
      float mu = fmaf(county(i).u, q[1], q[0]);
      x[NG + i] = fmaf(-a[i], mu, q[NG + i]) + mu;
    }
  }
  ARP_DEV void from_centered(const float (&x)[ND], float (&q)[ND]) const {
#pragma unroll
    for (int i = 0; i < ND; ++i) q[i] = x[i];
#pragma unroll
    for (int i = 0; i < NLS; ++i) {
      float mu = fmaf(county(i).u, x[1], x[0]);
      q[NG + i] = lvalid(i) ? x[NG + i] - (1.0f - a[i]) * mu : 0.0f;
      q[NG + NLS + i] = lvalid(i) ? x[NG + NLS + i] : 0.0f;
    }
  }
};

}  // namespace arp
