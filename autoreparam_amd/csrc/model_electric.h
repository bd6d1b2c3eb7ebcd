// Electric company (reference models.py:1011-1066) under the general VIP
// parameterisation.  Parts in trace order: mua[4], sigma_y[4], a[P], b[4].
//
//   mua_k ~ N(0,1), sigma_y_k ~ N(0,1)     unit scale: every (a,b) is the identity
//   a_j ~ N(mu_j, 1), mu_j = 100 mua[grade_pair_j]:   at_j ~ N(al_j mu_j, 1), a_j = at_j + (1 - al_j) mu_j
//   b_k ~ N(0,100)                          top level: bt_k ~ N(0, 100^be_k), b_k = 100^(1-be_k) bt_k
//   y_i ~ N(a[pair_i] + b[grade_i] treatment_i, exp(sigma_y[grade_i]))
//
// pair, grade and grade_pair are 1-based in the data and the reference feeds them to
// tf.one_hot unchanged, so index n falls on an all-zero row: pair P has no pair effect
// (group P below: observations only, no latent), grade 4 has b = 0 and scale exp(0) = 1,
// grade_pair 4 gives mu_j = 0, and a[0], mua[0], sigma_y[0], b[0] see the prior only.
//
// The observations of one pair share a grade g (checked by the host) and treatment is a
// 0/1 indicator, so a group collapses to two cells (control, treated) with count n_t,
// mean ybar_t and the pooled within-cell sum of squares SS:
//   sum_i (y_i - a - b t_i)^2 = n_0 (ybar_0 - a)^2 + n_1 (ybar_1 - a - b)^2 + SS =: Q
//   loglik_j = -(n_0 + n_1) s_g - w Q / 2,  w = exp(-2 s_g)
//   d/da = w (n_0 r_0 + n_1 r_1),  d/db_g = w n_1 r_1,  d/ds_g = w Q - (n_0 + n_1)
// The grade look-ups and scatters are one-hot FMAs against the replicated grade scalars.  The
// one-hot weights and the five cell statistics depend on (slot, slice) only, not on the chain, so
// they live in a 7 KB LDS table shared by the workgroup (four ds_read_b128 per pair and gradient)
// instead of 91 registers per lane: that is what lets two waves share a SIMD without scratch.
#pragma once
#include "arp_device.h"

namespace arp {

constexpr int kElG = 4;   // grades (n_grade = n_grade_pair = 4 in the reference's data)

struct ElectricArgs {
  // all tables are [P+1] (group P = observations without a pair effect), wm/og are [4][P+1]
  const float* wm;    // 100 * one_hot(grade_pair_j): location weights of a_j on mua
  const float* og;    // one_hot(grade of the group's observations)
  const float* n0; const float* y0; const float* n1; const float* y1; const float* ss;
  int P;
};

template <int K_, int NL_>
struct ElectricLane {
  static constexpr int K = K_;
  static constexpr int NG = 3 * kElG;  // mua[4], sigma_y[4], b[4]
  static constexpr int NL = NL_;       // groups owned by this lane: j = slot + K*i, j <= P
  static constexpr int ND = NG + NL;
  static constexpr int NGRP = NL_;
  static constexpr int DCAP = NG + K_ * NL_;
  static constexpr int LBASE = 2 * kElG;
  ARP_DEV int lbase(int) const { return LBASE + slot; }
  static constexpr ARP_DEV int loff(int i) { return K * i; }
  ARP_DEV int lidx(int i) const { return LBASE + slot + K * i; }
  // only the last slice can be padding: NL == ceil(groups / K) is enforced by the host
  ARP_DEV bool lvalid(int i) const { return i < NL - 1 ? true : last_ok; }
  bool last_ok;
  static constexpr bool HAS_MODES = true;         // compile-time centred / non-centred forms (grad_m below)
  static constexpr bool HAS_MODE_STATE = false;   // their scales of b are constants
  static constexpr bool HAS_CARRY = false;
  static constexpr bool HAS_FUSED = false;
  static constexpr bool HAS_VI = true;
  static constexpr int MINW = 2;   // waves per SIMD the register allocator must leave room for
  using Args = ElectricArgs;

  float al[NL];
  float lat_last;   // 1 if the lane's last slice is a pair effect, 0 if it is the observation-only group P or padding
  ARP_DEV float lat(int i) const { return i < NL - 1 ? 1.0f : lat_last; }
  float si[kElG], cs[kElG];   // 1/100^b and 100^(1-b) of b_k
  float nk[kElG];             // observations of the lane's groups by grade: sum_j nn_j s_{g_j} = sum_k nk_k s_k
  int slot, P;
  // the parameterisation as the kernel sees it: run-time (a, b) in the general form; centred (a = b = 1: a_j = at_j,
  // b_k = bt_k) and non-centred (a = b = 0: a_j = at_j + mu_j, b_k = 100 bt_k) as constants
  template <int MODE> ARP_DEV float alv(int i) const { return MODE == kModeCP ? 1.0f : MODE == kModeNCP ? 0.0f : al[i]; }
  template <int MODE> ARP_DEV float csv(int k) const { return MODE == kModeCP ? 1.0f : MODE == kModeNCP ? 100.0f : cs[k]; }
  template <int MODE> ARP_DEV float siv(int k) const { return MODE == kModeCP ? 0.01f : MODE == kModeNCP ? 1.0f : si[k]; }

  // table entry of (slice i, slot): [wm0..3][og0..3][n0 y0 n1 y1][ss z nn -], 20 floats apart (z = 1 - sum_k og_k: 1 for a
  // group whose grade falls on one_hot's all-zero row, nn = n0 + n1).
  // Bank conflicts (MI355X_MICROARCH.md, LDS): a ds_read_b128 is served in groups of 16 lanes over 64 banks; the K
  // distinct entries a group touches must fall on distinct 4-bank windows.  At round 3's stride of 16 dwords slots s
  // and s + 4 shared a window (2-way on each of the three b128 reads) and the lone `ss` went out as a ds_read_b32
  // (groups of 32 lanes over 32 banks: 4-way): 18 extra LDS cycles per pair and gradient on top of 14
  // (SQ_LDS_BANK_CONFLICT / SQ_INSTS_LDS = 4.0).  20 s mod 64 is a permutation of the multiples of 4 for 8 and for 16
  // lanes per chain, and `ss` is read as the fourth b128.
  static constexpr int kEntry = 20;
  static ARP_DEV float* onehot_table() {
    __shared__ __attribute__((aligned(16))) float tab[NL * K * kEntry];
    return tab;
  }
  // The lane's offset into the table is laundered ONCE per use of the model (gradient, coordinate change): the table is
  // loop invariant, and left to itself the compiler hoists every read out of the leapfrog loop into registers -- the
  // very registers the table is there to save.  The slices sit at compile-time distances behind it, so a pair's four
  // reads share one address register and differ in their immediate offsets.
  ARP_DEV int table_off() const {
    int e = slot * kEntry;
    asm volatile("" : "+v"(e));
    return e;
  }
  static ARP_DEV const float4* entry(int t0, int i) {
    return reinterpret_cast<const float4*>(onehot_table() + t0 + i * (K * kEntry));
  }
  ARP_DEV void onehot(int t0, int i, float (&wm)[kElG], float (&og)[kElG]) const {
    const float4* t = entry(t0, i);
    const float4 a = t[0], b = t[1];
    wm[0] = a.x; wm[1] = a.y; wm[2] = a.z; wm[3] = a.w;
    og[0] = b.x; og[1] = b.y; og[2] = b.z; og[3] = b.w;
  }
  // the pair's two cells: counts and means of the control / treated scores, pooled within-cell sum of squares; z and nn
  // as above
  ARP_DEV void cells(int t0, int i, float& n0, float& y0, float& n1, float& y1, float& ss, float& z, float& nn) const {
    const float4* e = entry(t0, i);
    const float4 c = e[2];
    float4 d = e[3];
    // the whole quad passes through an (empty) asm statement, so the compiler cannot narrow the read to a ds_read_b32
    // (served over 32 banks: 4-way conflicts at this stride) again
    asm volatile("" : "+v"(d.x), "+v"(d.y), "+v"(d.z), "+v"(d.w));
    n0 = c.x; y0 = c.y; n1 = c.z; y1 = c.w;
    ss = d.x; z = d.y; nn = d.z;
  }

  // flattened index of replicated scalar i: mua, sigma_y in front of a[P], b behind it
  ARP_DEV int gg(int i) const { return i < 2 * kElG ? i : i + P; }

  ARP_DEV void init(const Args& A, const float* av, const float* bv, int slot_) {
    slot = slot_;
    P = A.P;
    last_ok = slot + K * (NL - 1) < P;   // latent validity (j < P); group j == P has no latent
    const int stride = P + 1;
    float* tab = onehot_table();
    __syncthreads();   // a previous user of the table (none inside one kernel) is done
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int j = slot + K * i;
      const bool has = j <= P;
      if ((int)threadIdx.x < K) {   // the first chain of the workgroup fills the table for everybody
#pragma unroll
        for (int k = 0; k < kElG; ++k) {
          tab[(i * K + slot) * kEntry + k] = has ? A.wm[k * stride + j] : 0.0f;
          tab[(i * K + slot) * kEntry + 4 + k] = has ? A.og[k * stride + j] : 0.0f;
        }
        float* e = tab + (i * K + slot) * kEntry + 8;
        e[0] = has ? A.n0[j] : 0.0f; e[1] = has ? A.y0[j] : 0.0f; e[2] = has ? A.n1[j] : 0.0f; e[3] = has ? A.y1[j] : 0.0f;
        float ogs = 0.0f;
#pragma unroll
        for (int k = 0; k < kElG; ++k) ogs += has ? A.og[k * stride + j] : 0.0f;
        e[4] = has ? A.ss[j] : 0.0f; e[5] = has ? 1.0f - ogs : 0.0f; e[6] = has ? A.n0[j] + A.n1[j] : 0.0f; e[7] = 0.0f;
        e[8] = e[9] = e[10] = e[11] = 0.0f;
      }
    }
    lat_last = slot + K * (NL - 1) < P ? 1.0f : 0.0f;
#pragma unroll
    for (int k = 0; k < kElG; ++k) nk[k] = 0.0f;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int j = slot + K * i;
      if (j <= P) {
#pragma unroll
        for (int k = 0; k < kElG; ++k) nk[k] = fmaf(A.og[k * stride + j], A.n0[j] + A.n1[j], nk[k]);
      }
    }
    __syncthreads();
    set_param(av, bv);
  }
  ARP_DEV void set_param(const float* av, const float* bv) {
    const float l100 = 6.643856189774724f;  // log2(100)
#pragma unroll
    for (int k = 0; k < kElG; ++k) {
      si[k] = __builtin_amdgcn_exp2f(-bv[LBASE + P + k] * l100);
      cs[k] = 100.0f * si[k];
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) al[i] = lvalid(i) ? av[LBASE + slot + K * i] : 0.0f;
  }

  // The grade scalars reach a group through one-hot FMAs.  exp(-2 s_g) is formed once per grade, not once per group:
  // sum_k og_k wk_k + z is the very value exp(-2 sum_k og_k s_k) has (one weight is 1, the others 0; z = 1 covers the
  // all-zero row, exp(0)), so a gradient takes 4 exponentials instead of NL; likewise sum_j nn_j s_{g_j} is taken per
  // grade (nk) and the treatment slope's w e1 is scaled by the grade's w after the scatter.
  template <bool LOGP>
  ARP_DEV float grad(const float (&q)[ND], float (&g)[ND]) const { return grad_m<LOGP, kModeVIP>(q, g); }
  template <bool LOGP, int MODE>
  ARP_DEV float grad_m(const float (&q)[ND], float (&g)[ND]) const {
    float bb[kElG], wk[kElG], dM[kElG], dS[kElG], dB[kElG];
#pragma unroll
    for (int k = 0; k < kElG; ++k) {
      bb[k] = csv<MODE>(k) * q[2 * kElG + k]; wk[k] = fast_exp(-2.0f * q[kElG + k]);
      dM[k] = 0.0f; dS[k] = 0.0f; dB[k] = 0.0f;
    }
    float lp = 0.0f;
    const int t0 = table_off();
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      float wm[kElG], og[kElG];
      onehot(t0, i, wm, og);
      float n0, y0, n1, y1, ss, z, nn;
      cells(t0, i, n0, y0, n1, y1, ss, z, nn);
      float mu = wm[0] * q[0], w = fmaf(og[0], wk[0], z), bg = og[0] * bb[0];   // (a product, not an FMA onto zero)
#pragma unroll
      for (int k = 1; k < kElG; ++k) {
        mu = fmaf(wm[k], q[k], mu);
        w = fmaf(og[k], wk[k], w);
        bg = fmaf(og[k], bb[k], bg);
      }
      const float r = fmaf(-alv<MODE>(i), mu, q[NG + i]);   // group P / padding: q = 0, wm = 0 -> mu = 0, r = 0
      const float aj = r + mu;
      const float r0 = y0 - aj, r1 = (y1 - aj) - bg;
      const float e0 = n0 * r0, e1 = n1 * r1;
      const float dA = w * (e0 + e1);
      const float Q = fmaf(e0, r0, fmaf(e1, r1, ss));
      const float dSv = fmaf(w, Q, -nn);
      const float ga = lat(i) * (dA - r);
      g[NG + i] = ga;
      const float hm = fmaf(-alv<MODE>(i), ga, dA);   // d / d mu_j
#pragma unroll
      for (int k = 0; k < kElG; ++k) {
        dM[k] = fmaf(wm[k], hm, dM[k]);
        dS[k] = fmaf(og[k], dSv, dS[k]);
        dB[k] = fmaf(og[k], e1, dB[k]);
      }
      if (LOGP) lp += fmaf(-0.5f * r, r, -0.5f * w * Q);
    }
    float pri = 0.0f;
#pragma unroll
    for (int k = 0; k < kElG; ++k) {
      const float u = q[2 * kElG + k] * siv<MODE>(k);
      g[k] = group_sum<K>(dM[k]) - q[k];
      g[kElG + k] = group_sum<K>(dS[k]) - q[kElG + k];
      g[2 * kElG + k] = fmaf(csv<MODE>(k), group_sum<K>(wk[k] * dB[k]), -u * siv<MODE>(k));
      if (LOGP) { pri += fmaf(q[k], q[k], fmaf(q[kElG + k], q[kElG + k], u * u)); lp = fmaf(-nk[k], q[kElG + k], lp); }
    }
    if (LOGP) lp = group_sum<K>(lp) - 0.5f * pri;
    return lp;
  }

  // d logp / d a, d logp / d b from the state gradient (see model_radon.h): only a_j has a
  // location parent and only b_k a non-unit scale
  ARP_DEV void dparam(const float (&q)[ND], const float (&g)[ND], float (&da)[ND], float (&db)[ND]) const {
#pragma unroll
    for (int i = 0; i < ND; ++i) { da[i] = 0.0f; db[i] = 0.0f; }
#pragma unroll
    for (int k = 0; k < kElG; ++k)
      db[2 * kElG + k] = -4.605170185988092f * fmaf(q[2 * kElG + k], g[2 * kElG + k], 1.0f);
    const int t0 = table_off();
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      float wm[kElG], og[kElG];
      onehot(t0, i, wm, og);
      float mu = 0.0f;
#pragma unroll
      for (int k = 0; k < kElG; ++k) mu = fmaf(wm[k], q[k], mu);
      da[NG + i] = lvalid(i) ? -mu * g[NG + i] : 0.0f;
    }
  }

  ARP_DEV void to_centered(const float (&q)[ND], float (&x)[ND]) const { to_centered_m<kModeVIP>(q, x); }
  ARP_DEV void from_centered(const float (&x)[ND], float (&q)[ND]) const { from_centered_m<kModeVIP>(x, q); }
  template <int MODE>
  ARP_DEV void to_centered_m(const float (&q)[ND], float (&x)[ND]) const {
#pragma unroll
    for (int k = 0; k < kElG; ++k) { x[k] = q[k]; x[kElG + k] = q[kElG + k]; x[2 * kElG + k] = csv<MODE>(k) * q[2 * kElG + k]; }
    const int t0 = table_off();
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      float wm[kElG], og[kElG];
      onehot(t0, i, wm, og);
      float mu = 0.0f;
#pragma unroll
      for (int k = 0; k < kElG; ++k) mu = fmaf(wm[k], q[k], mu);
      x[NG + i] = fmaf(1.0f - alv<MODE>(i), mu, q[NG + i]);
    }
  }
  template <int MODE>
  ARP_DEV void from_centered_m(const float (&x)[ND], float (&q)[ND]) const {
#pragma unroll
    for (int k = 0; k < kElG; ++k) { q[k] = x[k]; q[kElG + k] = x[kElG + k]; q[2 * kElG + k] = x[2 * kElG + k] / csv<MODE>(k); }
    const int t0 = table_off();
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      float wm[kElG], og[kElG];
      onehot(t0, i, wm, og);
      float mu = 0.0f;
#pragma unroll
      for (int k = 0; k < kElG; ++k) mu = fmaf(wm[k], x[k], mu);
      q[NG + i] = lvalid(i) ? fmaf(-(1.0f - alv<MODE>(i)), mu, x[NG + i]) : 0.0f;
    }
  }
};

}  // namespace arp
