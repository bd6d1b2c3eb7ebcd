// Election88 hierarchical logistic regression (reference models.py:967-1008)
// under the general VIP parameterisation, with the 11 566 Bernoulli observations
// collapsed to (state, female, black) cells (n = observations, y = ones).
// Parts in trace order: mua, log_sigma_a, a[S], b1, b2.
//
//   mua ~ N(0,100), lsa ~ N(0,10), b1, b2 ~ N(0,100)   top level: xt ~ N(0, s^b), x = s^(1-b) xt
//   a_t ~ N(mua, sigma), sigma = exp(lsa):
//       at_t ~ N(al_t mua, sigma^be_t), z_t = (at_t - al_t mua) exp(-be_t lsa), a_t = mua + sigma z_t
//   y_i ~ Bernoulli(logit = a[state_i] + female_i b2 + black_i b1)
// The reference feeds the 1-based state index to tf.one_hot(., S): index t uses
// column t for t < S and index S hits an all-zero row, so those observations have
// no state effect (group S below: cells only, no latent).  SURVEY.md 8c-(i).
#pragma once
#include "arp_device.h"

namespace arp {

struct ElectionArgs {
  const float* cell_n;  // [S+1][4] cell index = female + 2*black
  const float* cell_y;  // [S+1][4]
  int S;
};

template <int K_, int NL_>
struct ElectionLane {
  static constexpr int K = K_;
  static constexpr int NG = 4;   // mua, lsa, b1, b2
  static constexpr int NL = NL_; // groups owned by this lane: t = slot + K*i, t <= S
  static constexpr int ND = NG + NL;
  static constexpr int NGRP = NL_;   // groups owned by a lane (what the host matches against ceil(groups / K))
  static constexpr int DCAP = NG + K_ * NL_;   // upper bound of the flattened state dimension D
  static constexpr int LBASE = 2;
  // sliced element i of this lane: flattened index and validity
  ARP_DEV int lbase(int) const { return LBASE + slot; }
  static constexpr ARP_DEV int loff(int i) { return K * i; }
  ARP_DEV int lidx(int i) const { return LBASE + slot + K * i; }
  // only the last slice can be padding: NL == ceil(groups / K) is enforced by the host
  ARP_DEV bool lvalid(int i) const { return i < NL - 1 ? true : last_ok; }
  bool last_ok;
  static constexpr bool HAS_MODES = true;    // CP / NCP: (a, b) are compile-time constants, no al / be registers
  static constexpr bool HAS_CARRY = false;
  static constexpr bool HAS_FUSED = true;    // kick_drift below
  static constexpr bool HAS_VI = true;
  static constexpr bool HAS_MODE_STATE = true;    // si / cs of the top-level scalars follow b (set_mode)
  static constexpr bool HAS_MODE_B1 = true;
  static constexpr int MOM_SPEC = 1;   // momentum stream layout 1 (kernels.h: hmc_transition, pk_chain.h)       // MODE 3: a free, b = 1 (tied cVIP / dVIP as the reference executes them)
  // three waves per SIMD at K = 4 (LDS: 51 KB per workgroup, three fit a CU): the 4 reciprocals + 1 exponential per state
  // are dependent-latency bound, and a third wave buys 7 % even though the 168-register cap spills a few values
  static constexpr int MINW = K_ == 4 ? 3 : 2;
  using Args = ElectionArgs;

  // The cell tables (n, y of the four (female, black) cells of every state) live in LDS, not in registers: they are
  // the same for every chain, 104 registers per lane at K = 4 -- with them in VGPRs the chain kernels spill inside
  // the leapfrog loop.  A state's eight numbers are two 16-byte reads at an address shared by the 64/K chains of the wave.
  static constexpr int SMEM_FLOATS = 8 * K_ * NL_;
  const float4* tab;
  static ARP_DEV void stage_tables(const Args& A, float* smem) {
    for (int idx = threadIdx.x; idx < K * NL; idx += blockDim.x) {
      const int i = idx / K, t = (idx % K) + K * i;
      const bool cell = t <= A.S;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        smem[idx * 8 + c] = cell ? A.cell_n[t * 4 + c] : 0.0f;
        smem[idx * 8 + 4 + c] = cell ? A.cell_y[t * 4 + c] : 0.0f;
      }
    }
  }
  ARP_DEV void bind_tables(const float* smem) { tab = reinterpret_cast<const float4*>(smem); }
  // (n, y) of slice i.  The index is laundered so that the reads stay inside the pass that uses them: hoisted out
  // of the leapfrog loop they would be 8 registers per state again.
  ARP_DEV void cells(int i, float4& n4, float4& y4) const {
    int k = (i * K + slot) * 2;
    asm volatile("" : "+v"(k));
    n4 = tab[k]; y4 = tab[k + 1];
  }
  float al[NL], be[NL];
  float d1, d2;     // sum over the lane's cells of (y - n) that carry b1 / b2: the cell-independent part of sum y eta - n eta
  float lat_last;   // 1 if the lane's last slice is a state effect, 0 if it is the cell-only group S or padding
  // (only the last slice can be anything but a latent, see lvalid)
  ARP_DEV float lat(int i) const { return i < NL - 1 ? 1.0f : lat_last; }
  // (a, b) of slice i under the compile-time parameterisations
  // (a non-latent slice has a = b = 0 in every mode: its prior terms vanish and q stays 0)
  template <int MODE> ARP_DEV float A(int i) const { return MODE == 1 ? lat(i) : (MODE == 2 ? 0.0f : al[i]); }
  template <int MODE> ARP_DEV float B(int i) const { return (MODE == 1 || MODE == 3) ? lat(i) : (MODE == 2 ? 0.0f : be[i]); }
  // exp(-b_i ls): shared by all slices when b is uniform (always so for CP, NCP and the reference's tied cVIP/dVIP)
  template <int MODE> ARP_DEV float E(int i, float ls, float eu) const {
    if (MODE == 2) return 1.0f;
    if (MODE == 1 || MODE == 3) return i < NL - 1 ? eu : (lat_last != 0.0f ? eu : 1.0f);
    return buni ? eu : fast_exp(-be[i] * ls);
  }
  float si[4], cs[4];   // 1/s^b and s^(1-b) for mua, lsa, b1, b2
  float bbar; bool buni; // every state shares one b (always so for CP, NCP and the reference's tied cVIP/dVIP)
  int slot, S;

  // flattened index of replicated global i (b1 -> 2+S, b2 -> 3+S: S is a run-time value)
  int gmap[NG];
  ARP_DEV int gg(int i) const { return gmap[i]; }

  ARP_DEV void init(const Args& A, const float* av, const float* bv, int slot_) {
    slot = slot_;
    S = A.S;
    last_ok = slot + K * (NL - 1) < S;   // latent validity (t < S); the extra cell group t == S has no latent
    gmap[0] = 0; gmap[1] = 1; gmap[2] = 2 + S; gmap[3] = 3 + S;
    d1 = 0.0f; d2 = 0.0f;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      float4 n4, y4;
      cells(i, n4, y4);
      d2 += (y4.y - n4.y) + (y4.w - n4.w);     // cells with female = 1 carry b2
      d1 += (y4.z - n4.z) + (y4.w - n4.w);     // cells with black = 1 carry b1
    }
    lat_last = slot + K * (NL - 1) < S ? 1.0f : 0.0f;
    set_param(av, bv);
  }
  // si, cs under CP (b = 1: xt = x) and NCP (b = 0: xt = x / s)
  template <int MODE>
  ARP_DEV void set_mode() {
    const float sc[4] = {100.0f, 10.0f, 100.0f, 100.0f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      si[i] = (MODE == 1 || MODE == 3) ? 1.0f / sc[i] : 1.0f;
      cs[i] = (MODE == 1 || MODE == 3) ? 1.0f : sc[i];
    }
  }
  ARP_DEV void set_param(const float* av, const float* bv) {
    const float l100 = 6.643856189774724f, l10 = 3.321928094887362f;  // log2
    const float lg[4] = {l100, l10, l100, l100};
    const float sc[4] = {100.0f, 10.0f, 100.0f, 100.0f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // b = 1 and b = 0 get the exact constants set_mode<> uses, so the general kernel, the compile-time forms
      // and a run cut into several launches all see the same numbers
      const float bb = bv[gmap[i]];
      const float e = __builtin_amdgcn_exp2f(-bb * lg[i]);
      si[i] = bb == 1.0f ? 1.0f / sc[i] : (bb == 0.0f ? 1.0f : e);
      cs[i] = bb == 1.0f ? 1.0f : (bb == 0.0f ? sc[i] : sc[i] * e);
    }
    bbar = bv[LBASE + slot];     // slice 0 is a latent in every lane
    float lo = bbar, hi = bbar;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      bool has = lvalid(i);
      al[i] = has ? av[LBASE + slot + K * i] : 0.0f;
      be[i] = has ? bv[LBASE + slot + K * i] : 0.0f;
      lo = has ? fminf(lo, be[i]) : lo;
      hi = has ? fmaxf(hi, be[i]) : hi;
    }
    buni = group_max<K>(hi) == -group_max<K>(-lo);
  }

  // Interior leapfrog step in one pass (kernels.h: lane_kick_drift).  The four cells of a
  // state share exp(-a_t): exp(-eta) = exp(-a_t) {1, e^-b2, e^-b1, e^-b1-b2}, so a state costs
  // one exp and four reciprocals instead of four of each, and with a shared b the prior's
  // exp(-b lsa) is formed once per gradient.  (The closing gradient of a transition, which
  // also needs the log density, uses the overflow-proof form in grad<>.)
  template <int MODE>
  ARP_DEV void kick_drift(float (&q)[ND], float (&p)[ND], const float (&eps)[ND]) const {
    const float mua = cs[0] * q[0], ls = cs[1] * q[1], b1 = cs[2] * q[2], b2 = cs[3] * q[3];
    const float sig = fast_exp(ls);
    const float E1 = fast_exp(-b1), E2 = fast_exp(-b2), E12 = E1 * E2;
    const float eu = MODE == 2 ? 1.0f : fast_exp(-((MODE == 1 || MODE == 3) ? 1.0f : bbar) * ls);
    float g_mua = 0.0f, g_ls = 0.0f, g_b1 = 0.0f, g_b2 = 0.0f;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const float qt = q[NG + i];
      const float ai = A<MODE>(i), bi = B<MODE>(i), li = lat(i);
      const float e = E<MODE>(i, ls, eu);
      const float z = (qt - ai * mua) * e;
      const float as = li * fmaf(sig, z, mua);
      const float t = fast_exp(-as);
      float4 n4, y4;
      cells(i, n4, y4);
      const float w0 = fmaf(-n4.x, __builtin_amdgcn_rcpf(1.0f + t), y4.x);
      const float w1 = fmaf(-n4.y, __builtin_amdgcn_rcpf(fmaf(t, E2, 1.0f)), y4.y);
      const float w2 = fmaf(-n4.z, __builtin_amdgcn_rcpf(fmaf(t, E1, 1.0f)), y4.z);
      const float w3 = fmaf(-n4.w, __builtin_amdgcn_rcpf(fmaf(t, E12, 1.0f)), y4.w);
      const float W = (w0 + w1) + (w2 + w3);
      g_b2 += w1 + w3;
      g_b1 += w2 + w3;
      const float gt = li * e * fmaf(sig, W, -z);
      g_mua += li * fmaf(-ai, gt, W);
      g_ls += fmaf(bi, fmaf(z, z, -1.0f), li * W * sig * z * (1.0f - bi));
      const float pn = fmaf(eps[NG + i], gt, p[NG + i]);
      p[NG + i] = pn;
      q[NG + i] = fmaf(eps[NG + i], pn, qt);
    }
    g_mua = group_sum<K>(g_mua);
    g_ls = group_sum<K>(g_ls);
    g_b1 = group_sum<K>(g_b1);
    g_b2 = group_sum<K>(g_b2);
    const float gs[4] = {g_mua, g_ls, g_b1, g_b2};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float gi = fmaf(cs[i], gs[i], -(q[i] * si[i]) * si[i]);
      p[i] = fmaf(eps[i], gi, p[i]);
      q[i] = fmaf(eps[i], p[i], q[i]);
    }
  }

  template <bool LOGP>
  ARP_DEV float grad(const float (&q)[ND], float (&g)[ND]) const { return grad_m<LOGP, 0>(q, g); }

  // Gradient (and log density when LOGP).  Two forms of the likelihood part:
  //  * SAFE (the general form MODE 0: logp_grad_kernel, the VI kernel, arbitrary caller-supplied states): every cell
  //    through exp(-|eta|), overflow-proof for any state;
  //  * fast (the compile-time parameterisations, i.e. the closing gradient of every CP / NCP / b = 1 transition): the
  //    four cells of a state share exp(-a_t) as in kick_drift, and with rc = 1 / (1 + e^-eta) = sigmoid(eta)
  //        softplus(eta) = eta - log(rc),  so  y eta - n softplus(eta) = (y - n) eta + n log(rc):
  //    per cell one reciprocal and one logarithm, no exponential, no |.|, no select.  e^-eta overflows only for
  //    logits below -88, where the log density comes out -inf and the proposal is rejected (TFP's non-finite rule).
  template <bool LOGP, int MODE>
  ARP_DEV float grad_m(const float (&q)[ND], float (&g)[ND]) const {
    constexpr bool SAFE = MODE == 0;
    const float mua = cs[0] * q[0], ls = cs[1] * q[1], b1 = cs[2] * q[2], b2 = cs[3] * q[3];
    const float sig = fast_exp(ls);
    const float E1 = SAFE ? 0.0f : fast_exp(-b1), E2 = SAFE ? 0.0f : fast_exp(-b2), E12 = E1 * E2;
    const float eu = MODE == 2 ? 1.0f : fast_exp(-((MODE == 1 || MODE == 3) ? 1.0f : bbar) * ls);
    float g_mua = 0.0f, g_ls = 0.0f, g_b1 = 0.0f, g_b2 = 0.0f, lp = 0.0f;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const float ai = A<MODE>(i), bi = B<MODE>(i), li = lat(i);
      const float e = MODE == 0 ? fast_exp(-be[i] * ls) : E<MODE>(i, ls, eu);
      float z = (q[NG + i] - ai * mua) * e;      // group S / padding: q = 0, a = 0 -> z = 0
      float as = li * fmaf(sig, z, mua);
      float4 n4, y4;
      cells(i, n4, y4);
      const float cn_[4] = {n4.x, n4.y, n4.z, n4.w}, cy_[4] = {y4.x, y4.y, y4.z, y4.w};
      float w[4];
      if (SAFE) {
        const float eta[4] = {as, as + b2, as + b1, as + b1 + b2};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          // sigmoid / softplus share exp(-|eta|)
          float ex = fast_exp(-fabsf(eta[c]));
          float rc = __builtin_amdgcn_rcpf(1.0f + ex);
          float sg = eta[c] >= 0.0f ? rc : ex * rc;
          w[c] = fmaf(-cn_[c], sg, cy_[c]);
          if (LOGP) {
            float sp = fmaxf(eta[c], 0.0f) + fast_log(1.0f + ex);
            lp += fmaf(cy_[c], eta[c], -cn_[c] * sp);
          }
        }
      } else {
        const float t = fast_exp(-as);
        const float den[4] = {1.0f + t, fmaf(t, E2, 1.0f), fmaf(t, E1, 1.0f), fmaf(t, E12, 1.0f)};
        float dn = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float rc = __builtin_amdgcn_rcpf(den[c]);
          w[c] = fmaf(-cn_[c], rc, cy_[c]);
          if (LOGP) {
            lp = fmaf(cn_[c] * 0.6931471805599453f, __builtin_amdgcn_logf(rc), lp);    // n log(rc), v_log_f32 is log2
            dn += cy_[c] - cn_[c];
          }
        }
        if (LOGP) lp = fmaf(dn, as, lp);          // (y - n) eta, the part every cell of the state shares
      }
      float W = (w[0] + w[1]) + (w[2] + w[3]);
      g_b2 += w[1] + w[3];
      g_b1 += w[2] + w[3];
      float gt = li * e * fmaf(sig, W, -z);
      g[NG + i] = gt;
      g_mua += li * fmaf(-ai, gt, W);
      g_ls += fmaf(bi, fmaf(z, z, -1.0f), li * W * sig * z * (1.0f - bi));
      if (LOGP) lp += fmaf(-0.5f * z, z, -bi * ls);
    }
    if (LOGP && !SAFE) lp = fmaf(d1, b1, fmaf(d2, b2, lp));   // (y - n) (b1 black + b2 female) over the lane's cells
    g_mua = group_sum<K>(g_mua);
    g_ls = group_sum<K>(g_ls);
    g_b1 = group_sum<K>(g_b1);
    g_b2 = group_sum<K>(g_b2);
    float u[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = q[i] * si[i];
    g[0] = fmaf(cs[0], g_mua, -u[0] * si[0]);
    g[1] = fmaf(cs[1], g_ls, -u[1] * si[1]);
    g[2] = fmaf(cs[2], g_b1, -u[2] * si[2]);
    g[3] = fmaf(cs[3], g_b2, -u[3] * si[3]);
    if (LOGP) {
      lp = group_sum<K>(lp);
      lp += -0.5f * (u[0] * u[0] + u[1] * u[1] + u[2] * u[2] + u[3] * u[3]);
    }
    return lp;
  }

  // d logp / d a, d logp / d b from the state gradient (see model_radon.h)
  ARP_DEV void dparam(const float (&q)[ND], const float (&g)[ND], float (&da)[ND], float (&db)[ND]) const {
    const float lns[4] = {4.605170185988092f, 2.302585092994046f, 4.605170185988092f, 4.605170185988092f};
    const float mua = cs[0] * q[0], ls = cs[1] * q[1];
#pragma unroll
    for (int i = 0; i < 4; ++i) { da[i] = 0.0f; db[i] = -lns[i] * fmaf(q[i], g[i], 1.0f); }
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      bool ok = lvalid(i);
      da[NG + i] = ok ? -mua * g[NG + i] : 0.0f;
      db[NG + i] = ok ? -ls * fmaf(q[NG + i] - al[i] * mua, g[NG + i], 1.0f) : 0.0f;
    }
  }

  ARP_DEV void to_centered(const float (&q)[ND], float (&x)[ND]) const { to_centered_m<0>(q, x); }
  ARP_DEV void from_centered(const float (&x)[ND], float (&q)[ND]) const { from_centered_m<0>(x, q); }
  template <int MODE>
  ARP_DEV void to_centered_m(const float (&q)[ND], float (&x)[ND]) const {
    const float mua = cs[0] * q[0], ls = cs[1] * q[1];
    x[0] = mua; x[1] = ls; x[2] = cs[2] * q[2]; x[3] = cs[3] * q[3];
    const float sig = MODE == 2 ? fast_exp(ls) : 1.0f;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      if (MODE == 1) x[NG + i] = q[NG + i];                             // a = b = 1: identity
      else if (MODE == 2) x[NG + i] = lat(i) * fmaf(sig, q[NG + i], mua);   // a = b = 0: mua + sigma q
      else if (MODE == 3) x[NG + i] = fmaf(lat(i) - al[i], mua, q[NG + i]);   // b = 1: q + (1 - a) mua
      else x[NG + i] = fmaf(fast_exp((1.0f - be[i]) * ls), q[NG + i] - al[i] * mua, mua);
    }
  }
  template <int MODE>
  ARP_DEV void from_centered_m(const float (&x)[ND], float (&q)[ND]) const {
    const float mua = x[0], ls = x[1];
    q[0] = mua / cs[0]; q[1] = ls / cs[1]; q[2] = x[2] / cs[2]; q[3] = x[3] / cs[3];
    const float isig = MODE == 2 ? fast_exp(-ls) : 1.0f;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      if (MODE == 1) q[NG + i] = lvalid(i) ? x[NG + i] : 0.0f;
      else if (MODE == 2) q[NG + i] = lvalid(i) ? (x[NG + i] - mua) * isig : 0.0f;
      else if (MODE == 3) q[NG + i] = lvalid(i) ? fmaf(al[i] - 1.0f, mua, x[NG + i]) : 0.0f;
      else q[NG + i] = lvalid(i) ? fmaf(x[NG + i] - mua, fast_exp(-(1.0f - be[i]) * ls), al[i] * mua) : 0.0f;
    }
  }
};

}  // namespace arp
