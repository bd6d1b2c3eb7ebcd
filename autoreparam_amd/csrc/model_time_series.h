// Local linear trend ("time_series", reference models.py:1069-1141) under the general VIP
// parameterisation.  Every latent is its own scalar random variable; trace order:
//   sigma_alpha, sigma_mu, alpha_0, mu_0, alpha_1, mu_1, ..., alpha_{T-1}, mu_{T-1}, beta.
//
//   sa, sm, beta ~ N(0,1);  Sa = softplus(sa), Sm = softplus(sm)
//   alpha_t ~ N(m_t, Sa), m_t = alpha_{t-1} + mu_{t-1} (m_0 = 0):
//       at_t ~ N(a m_t, Sa^b),  alpha_t = m_t + c (at_t - a m_t) = f m_t + c at_t,  c = Sa^(1-b), f = 1 - a c
//   mu_t ~ N(mu_{t-1}, Sm) (mu_{-1} = 0):  mt_t likewise with (a', b', c', f')
//   y_t ~ N(alpha_t + beta x_t, 0.12)
//
// The centred values form a chain in t, so the K lanes of a chain own consecutive blocks of B time steps
// (K B >= T; steps beyond T are padding that carries no latent, no observation and a = b = 0) and the two
// recurrences are block scans:
//   forward   (alpha, mu)_t = [[f, f], [0, f']] (alpha, mu)_{t-1} + (c at_t, c' mt_t)
//   backward  with e_t = d loglik / d alpha_t, z = (at - a m) / Sa^b and the messages
//             G_t = d logp / d m_t, H_t = d logp / d mu_{t-1} (through mu_t's prior):
//             Abar_t = e_t + G_{t+1},  Mbar_t = G_{t+1} + H_{t+1},
//             G_t = f Abar_t + a z / Sa^b,  H_t = f' Mbar_t + a' z' / Sm^b'
// Both are affine in the incoming pair, so a lane first runs its block with a zero input while
// composing the block's 2x2 (triangular) map, the chain's value is rippled through the block maps
// lane by lane (K - 1 rounds of DPP row shifts, masked at the chain's ends so that a chain never
// reads a neighbour's values; no map is multiplied into another, which would cost accuracy), and
// the lane reruns its block with the right input.
//
// Lanes per chain: 4 (B = 15) is the fastest split wherever it fills the SIMDs once (>= 16 384 chains) although it
// runs one wave per SIMD (256 + up to 256 registers); 8 (B = 8, T = 60 padded to 64) and 16 (B = 4) run two to three
// waves per SIMD and serve the smaller chain counts (profiles/r03_time_series_sweep.txt).
#pragma once
#include "arp_device.h"

namespace arp {

constexpr int kTsSteps = 60;   // the reference's series length: the only T the host accepts (arp_api.hip: build_time_series)

struct TimeSeriesArgs {
  const float* x;   // [T] regressor (years)
  const float* y;   // [T] observations
  int T;
};

template <int K_, int NL_>
struct TimeSeriesLane {
  static_assert(K_ == 4 || K_ == 8 || K_ == 16, "a chain is 4, 8 or 16 lanes of one DPP row");
  static_assert(NL_ % 2 == 0, "a lane owns whole time steps");
  static constexpr int K = K_;
  static constexpr int NG = 3;          // sigma_alpha, sigma_mu, beta
  static constexpr int NL = NL_;        // trend latents owned by this lane: (alpha, mu) of B consecutive steps
  static constexpr int B = NL_ / 2;
  static constexpr int ND = NG + NL;
  static constexpr int NGRP = NL_;
  static constexpr int DCAP = NG + K_ * NL_;
  static constexpr bool HAS_MODES = true;        // compile-time centred / non-centred / "a free, b = 1" forms (grad_m below)
  static constexpr bool HAS_MODE_STATE = false;
  static constexpr bool HAS_MODE_B1 = true;
  static constexpr bool HAS_CARRY = false;
  static constexpr bool HAS_FUSED = false;
  static constexpr bool HAS_VI = true;
  static constexpr int MINW = NL_ <= 16 ? 2 : 1;   // waves per SIMD the register allocator must leave room for
  using Args = TimeSeriesArgs;

  float xt[B], yt[B], aA[B], bA[B], aM[B], bM[B];
  ARP_DEV bool real_step(int tl) const { return !PADDED || slot * B + tl < T; }
  int slot, T;

  // flattened indices: the lane's latents are one consecutive run; beta sits behind all of them
  ARP_DEV int gg(int i) const { return i < 2 ? i : 2 + 2 * T; }
  ARP_DEV int lbase(int) const { return 2 + slot * NL; }
  static constexpr ARP_DEV int loff(int i) { return i; }
  ARP_DEV int lidx(int i) const { return 2 + slot * NL + i; }
  static constexpr bool PADDED = K_ * B > kTsSteps;   // the last lanes own steps beyond T
  ARP_DEV bool lvalid(int i) const { return PADDED ? slot * NL + i < 2 * T : true; }

  ARP_DEV void init(const Args& A, const float* av, const float* bv, int slot_) {
    slot = slot_;
    T = A.T;
#pragma unroll
    for (int tl = 0; tl < B; ++tl) {
      const int t = slot * B + tl;
      xt[tl] = real_step(tl) ? A.x[t] : 0.0f; yt[tl] = real_step(tl) ? A.y[t] : 0.0f;
    }
    set_param(av, bv);
  }
  ARP_DEV void set_param(const float* av, const float* bv) {
#pragma unroll
    for (int tl = 0; tl < B; ++tl) {
      const int i = 2 + slot * NL + 2 * tl;
      const bool ok = real_step(tl);
      aA[tl] = ok ? av[i] : 0.0f; bA[tl] = ok ? bv[i] : 0.0f; aM[tl] = ok ? av[i + 1] : 0.0f; bM[tl] = ok ? bv[i + 1] : 0.0f;
    }
  }

  // Value held by the lane D slots EARLIER in the chain; `old` where the chain has no such lane (a row shift of the DPP
  // row of 16 lanes, then a select on the slot).
  template <int D>
  ARP_DEV float from_earlier(float old, float v) const {
    static_assert(D == 1, "row_shr:1; a wider shift would want a bank mask instead of the select");
    const float r = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x110 + D, 0xF, 0xF, false));
    return slot >= D ? r : old;
  }
  // ... D slots LATER in the chain
  template <int D>
  ARP_DEV float from_later(float old, float v) const {
    static_assert(D == 1, "row_shl:1");
    const float r = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x100 + D, 0xF, 0xF, false));
    return slot + D < K ? r : old;
  }

  struct Scales { float Sa, Sm, lSa, lSm; };
  ARP_DEV Scales scales(float sa, float sm) const {
    Scales s;
    s.Sa = softplusf_(sa); s.Sm = softplusf_(sm);
    s.lSa = fast_log(s.Sa); s.lSm = fast_log(s.Sm);
    return s;
  }

  // Forward block scan: centred (alpha_t, mu_t) of this lane's steps; eA/eM = Sa^-b, Sm^-b' per step.
  ARP_DEV void forward(const float (&q)[ND], const Scales& S, float (&al)[B], float (&mu)[B], float (&eA)[B],
                       float (&eM)[B]) const {
    // pass 1: zero input, compose the block map (alpha, mu)_out = [[p11, p12], [0, p22]] in + (da, dm)
    float p11 = 1.0f, p12 = 0.0f, p22 = 1.0f, da = 0.0f, dm = 0.0f;
#pragma unroll
    for (int tl = 0; tl < B; ++tl) {
      eA[tl] = fast_exp(-bA[tl] * S.lSa);
      eM[tl] = fast_exp(-bM[tl] * S.lSm);
      const float cA = S.Sa * eA[tl], cM = S.Sm * eM[tl];
      const float fA = fmaf(-aA[tl], cA, 1.0f), fM = fmaf(-aM[tl], cM, 1.0f);
      p12 = fA * (p12 + p22); p11 = fA * p11; p22 = fM * p22;
      da = fmaf(fA, da + dm, cA * q[NG + 2 * tl]);
      dm = fmaf(fM, dm, cM * q[NG + 2 * tl + 1]);
    }
    // ripple the chain's value through the block maps: after j rounds the lanes 0 .. j hold their true input (the output
    // of the lane before them; (0, 0) for the first), K - 1 rounds of three FMAs and two row shifts
    float ap = 0.0f, mp = 0.0f;
#pragma unroll
    for (int j = 1; j < K; ++j) {
      const float ao = fmaf(p11, ap, fmaf(p12, mp, da)), mo = fmaf(p22, mp, dm);
      ap = from_earlier<1>(0.0f, ao); mp = from_earlier<1>(0.0f, mo);
    }
    // pass 2: the block with its real input
#pragma unroll
    for (int tl = 0; tl < B; ++tl) {
      const float cA = S.Sa * eA[tl], cM = S.Sm * eM[tl];
      const float mA = ap + mp;
      al[tl] = fmaf(cA, fmaf(-aA[tl], mA, q[NG + 2 * tl]), mA);
      mu[tl] = fmaf(cM, fmaf(-aM[tl], mp, q[NG + 2 * tl + 1]), mp);
      ap = al[tl]; mp = mu[tl];
    }
  }

  // (alpha, mu) at the step before this lane's first (zero for slot 0)
  ARP_DEV void incoming(const float (&al)[B], const float (&mu)[B], float& ap, float& mp) const {
    ap = from_earlier<1>(0.0f, al[B - 1]); mp = from_earlier<1>(0.0f, mu[B - 1]);
  }

  template <bool LOGP>
  ARP_DEV float grad(const float (&q)[ND], float (&g)[ND]) const {
    const float s2i = 69.44444444444444f;   // 1 / 0.12^2
    const float sa = q[0], sm = q[1], beta = q[2];
    const Scales S = scales(sa, sm);
    float al[B], mu[B], eA[B], eM[B];
    forward(q, S, al, mu, eA, eM);
    float ap, mp;
    incoming(al, mu, ap, mp);
    // residuals: e_t, z_t, z'_t (al / mu are overwritten by zA / zM)
    float e[B], lp = 0.0f, g_beta = 0.0f;
#pragma unroll
    for (int tl = 0; tl < B; ++tl) {
      const float mA = ap + mp, alpha = al[tl], mut = mu[tl];
      const float res = (yt[tl] - alpha) - beta * xt[tl];
      e[tl] = real_step(tl) ? res * s2i : 0.0f;   // a padding step has no observation
      g_beta = fmaf(e[tl], xt[tl], g_beta);
      const float zA = fmaf(-aA[tl], mA, q[NG + 2 * tl]) * eA[tl];
      const float zM = fmaf(-aM[tl], mp, q[NG + 2 * tl + 1]) * eM[tl];
      if (LOGP) lp += fmaf(-0.5f * zA, zA, fmaf(-0.5f * zM, zM, fmaf(-0.5f * res, e[tl], -(bA[tl] * S.lSa + bM[tl] * S.lSm))));
      al[tl] = zA; mu[tl] = zM;
      ap = alpha; mp = mut;
    }
    // backward pass 1: zero input, compose (G, H)_out = [[r11, 0], [r21, r22]] in + (oG, oH)
    float r11 = 1.0f, r21 = 0.0f, r22 = 1.0f, oG = 0.0f, oH = 0.0f;
#pragma unroll
    for (int tl = B - 1; tl >= 0; --tl) {
      const float cA = S.Sa * eA[tl], cM = S.Sm * eM[tl];
      const float fA = fmaf(-aA[tl], cA, 1.0f), fM = fmaf(-aM[tl], cM, 1.0f);
      const float kA = aA[tl] * al[tl] * eA[tl], kM = aM[tl] * mu[tl] * eM[tl];
      r21 = fM * (r11 + r21); r11 = fA * r11; r22 = fM * r22;
      const float nH = fmaf(fM, oG + oH, kM);
      oG = fmaf(fA, e[tl] + oG, kA);
      oH = nH;
    }
    // the same ripple from the chain's end: a lane's input is the output of the lane after it, (0, 0) for the last
    float G = 0.0f, H = 0.0f;
#pragma unroll
    for (int j = 1; j < K; ++j) {
      const float Go = fmaf(r11, G, oG), Ho = fmaf(r21, G, fmaf(r22, H, oH));
      G = from_later<1>(0.0f, Go); H = from_later<1>(0.0f, Ho);
    }
    // backward pass 2: gradients
    float g_lSa = 0.0f, g_lSm = 0.0f;
#pragma unroll
    for (int tl = B - 1; tl >= 0; --tl) {
      const float cA = S.Sa * eA[tl], cM = S.Sm * eM[tl];
      const float fA = fmaf(-aA[tl], cA, 1.0f), fM = fmaf(-aM[tl], cM, 1.0f);
      const float zA = al[tl], zM = mu[tl];
      const float Ab = e[tl] + G, Mb = G + H;
      g[NG + 2 * tl] = fmaf(cA, Ab, -zA * eA[tl]);
      g[NG + 2 * tl + 1] = fmaf(cM, Mb, -zM * eM[tl]);
      g_lSa += fmaf(Ab * (1.0f - bA[tl]), zA * S.Sa, bA[tl] * fmaf(zA, zA, -1.0f));
      g_lSm += fmaf(Mb * (1.0f - bM[tl]), zM * S.Sm, bM[tl] * fmaf(zM, zM, -1.0f));
      G = fmaf(fA, Ab, aA[tl] * zA * eA[tl]);
      H = fmaf(fM, Mb, aM[tl] * zM * eM[tl]);
    }
    g_lSa = group_sum<K>(g_lSa);
    g_lSm = group_sum<K>(g_lSm);
    g_beta = group_sum<K>(g_beta);
    g[0] = fmaf(g_lSa, sigmoidf_(sa) * __builtin_amdgcn_rcpf(S.Sa), -sa);
    g[1] = fmaf(g_lSm, sigmoidf_(sm) * __builtin_amdgcn_rcpf(S.Sm), -sm);
    g[2] = g_beta - beta;
    if (LOGP) lp = group_sum<K>(lp) - 0.5f * (sa * sa + sm * sm + beta * beta);
    return lp;
  }

  // -------------------------------------------------------------------------------------------------------------
  // Compile-time parameterisations (kernels.h): MODE 1 centred (a = b = 1), 2 non-centred (a = b = 0), 3 "a free,
  // b = 1" (what the reference's tied cVIP / dVIP runs execute).  The general form above evaluates
  // S^-b = exp(-b log S) per latent and pass; here it is 1 or 1 / S:
  //   centred      e = 1/S, c = 1, f = 0: the state IS (alpha_t, mu_t) -- no recurrence, no scan;
  //   non-centred  e = 1,   c = S, f = 1: alpha and mu are cumulative sums of the scaled state;
  //   b = 1        e = 1/S, c = 1, f = 1 - a.
  // A padding step (steps beyond T in the chain's last lanes) has a zero latent; only the centred form has to mask its
  // z = (0 - m) / S explicitly, and the terms that count steps use the lane's number of real steps.
  // -------------------------------------------------------------------------------------------------------------
  ARP_DEV float n_real() const {
    if (!PADDED) return (float)B;
    const int n = T - slot * B;
    return (float)(n < 0 ? 0 : (n > B ? B : n));
  }
  template <int MODE>
  ARP_DEV float q_minus_a_m(float a, float m, float qq) const { return MODE == 2 ? qq : fmaf(-a, m, qq); }

  // forward block scan of MODE 2 / 3: centred (alpha_t, mu_t) of this lane's steps
  template <int MODE>
  ARP_DEV void forward_m(const float (&q)[ND], const Scales& S, float (&al)[B], float (&mu)[B]) const {
    static_assert(MODE == 2 || MODE == 3, "");
    constexpr bool NCP = MODE == 2;
    const float cA = NCP ? S.Sa : 1.0f, cM = NCP ? S.Sm : 1.0f;
    float p11 = 1.0f, p12 = 0.0f, p22 = 1.0f, da = 0.0f, dm = 0.0f;
#pragma unroll
    for (int tl = 0; tl < B; ++tl) {
      if constexpr (NCP) {
        p12 += 1.0f;
        da = (da + dm) + cA * q[NG + 2 * tl];
        dm = fmaf(cM, q[NG + 2 * tl + 1], dm);
      } else {
        const float fA = 1.0f - aA[tl], fM = 1.0f - aM[tl];
        p12 = fA * (p12 + p22); p11 = fA * p11; p22 = fM * p22;
        da = fmaf(fA, da + dm, q[NG + 2 * tl]);
        dm = fmaf(fM, dm, q[NG + 2 * tl + 1]);
      }
    }
    float ap = 0.0f, mp = 0.0f;
#pragma unroll
    for (int j = 1; j < K; ++j) {
      const float ao = NCP ? (ap + fmaf(p12, mp, da)) : fmaf(p11, ap, fmaf(p12, mp, da));
      const float mo = NCP ? (mp + dm) : fmaf(p22, mp, dm);
      ap = from_earlier<1>(0.0f, ao); mp = from_earlier<1>(0.0f, mo);
    }
#pragma unroll
    for (int tl = 0; tl < B; ++tl) {
      const float mA = ap + mp;
      if constexpr (NCP) {
        al[tl] = fmaf(cA, q[NG + 2 * tl], mA);
        mu[tl] = fmaf(cM, q[NG + 2 * tl + 1], mp);
      } else {
        al[tl] = fmaf(-aA[tl], mA, q[NG + 2 * tl]) + mA;
        mu[tl] = fmaf(-aM[tl], mp, q[NG + 2 * tl + 1]) + mp;
      }
      ap = al[tl]; mp = mu[tl];
    }
  }

  template <bool LOGP>
  ARP_DEV float grad_cp(const float (&q)[ND], float (&g)[ND]) const {
    const float s2i = 69.44444444444444f;   // 1 / 0.12^2
    const float sa = q[0], sm = q[1], beta = q[2];
    const Scales S = scales(sa, sm);
    const float iSa = __builtin_amdgcn_rcpf(S.Sa), iSm = __builtin_amdgcn_rcpf(S.Sm);
    // the step before the lane's first belongs to the lane before it
    float ap = from_earlier<1>(0.0f, q[NG + NL - 2]), mp = from_earlier<1>(0.0f, q[NG + NL - 1]);
    float zA[B], zM[B], e[B], lp = 0.0f, g_beta = 0.0f, ssA = 0.0f, ssM = 0.0f;
#pragma unroll
    for (int tl = 0; tl < B; ++tl) {
      const float mA = ap + mp, alpha = q[NG + 2 * tl], mut = q[NG + 2 * tl + 1];
      const float res = (yt[tl] - alpha) - beta * xt[tl];
      e[tl] = real_step(tl) ? res * s2i : 0.0f;
      g_beta = fmaf(e[tl], xt[tl], g_beta);
      float za = (alpha - mA) * iSa, zm = (mut - mp) * iSm;
      if (PADDED) { za = real_step(tl) ? za : 0.0f; zm = real_step(tl) ? zm : 0.0f; }
      zA[tl] = za; zM[tl] = zm;
      ssA = fmaf(za, za, ssA); ssM = fmaf(zm, zm, ssM);
      if (LOGP) lp = fmaf(-0.5f * res, e[tl], lp);
      ap = alpha; mp = mut;
    }
    // messages of the step after: G_t = z_t / Sa, H_t = z'_t / Sm (f = 0: nothing is passed further down the chain)
    float G = from_later<1>(0.0f, zA[0] * iSa), H = from_later<1>(0.0f, zM[0] * iSm);
#pragma unroll
    for (int tl = B - 1; tl >= 0; --tl) {
      const float Ab = e[tl] + G, Mb = G + H;
      const float gA = zA[tl] * iSa, gM = zM[tl] * iSm;
      g[NG + 2 * tl] = Ab - gA;
      g[NG + 2 * tl + 1] = Mb - gM;
      G = gA; H = gM;
    }
    const float nr = n_real();
    const float g_lSa = group_sum<K>(ssA - nr), g_lSm = group_sum<K>(ssM - nr);
    g_beta = group_sum<K>(g_beta);
    g[0] = fmaf(g_lSa, sigmoidf_(sa) * iSa, -sa);
    g[1] = fmaf(g_lSm, sigmoidf_(sm) * iSm, -sm);
    g[2] = g_beta - beta;
    if (LOGP) lp = group_sum<K>(lp - 0.5f * (ssA + ssM) - nr * (S.lSa + S.lSm)) - 0.5f * (sa * sa + sm * sm + beta * beta);
    return lp;
  }

  template <bool LOGP, int MODE>
  ARP_DEV float grad_s(const float (&q)[ND], float (&g)[ND]) const {
    static_assert(MODE == 2 || MODE == 3, "");
    constexpr bool NCP = MODE == 2;
    const float s2i = 69.44444444444444f;   // 1 / 0.12^2
    const float sa = q[0], sm = q[1], beta = q[2];
    const Scales S = scales(sa, sm);
    const float iSa = __builtin_amdgcn_rcpf(S.Sa), iSm = __builtin_amdgcn_rcpf(S.Sm);
    float al[B], mu[B];
    forward_m<MODE>(q, S, al, mu);
    float ap, mp;
    incoming(al, mu, ap, mp);
    // residuals: e_t, z_t, z'_t (al / mu are overwritten by zA / zM); a padding step has q = 0 (and a = 0): z = 0
    float e[B], lp = 0.0f, g_beta = 0.0f, ssA = 0.0f, ssM = 0.0f;
#pragma unroll
    for (int tl = 0; tl < B; ++tl) {
      const float mA = ap + mp, alpha = al[tl], mut = mu[tl];
      const float res = (yt[tl] - alpha) - beta * xt[tl];
      e[tl] = real_step(tl) ? res * s2i : 0.0f;
      g_beta = fmaf(e[tl], xt[tl], g_beta);
      const float zA = NCP ? q[NG + 2 * tl] : fmaf(-aA[tl], mA, q[NG + 2 * tl]) * iSa;
      const float zM = NCP ? q[NG + 2 * tl + 1] : fmaf(-aM[tl], mp, q[NG + 2 * tl + 1]) * iSm;
      ssA = fmaf(zA, zA, ssA); ssM = fmaf(zM, zM, ssM);
      if (LOGP) lp = fmaf(-0.5f * res, e[tl], lp);
      al[tl] = zA; mu[tl] = zM;
      ap = alpha; mp = mut;
    }
    // backward pass 1: zero input, compose (G, H)_out = [[r11, 0], [r21, r22]] in + (oG, oH)
    float r11 = 1.0f, r21 = 0.0f, r22 = 1.0f, oG = 0.0f, oH = 0.0f;
#pragma unroll
    for (int tl = B - 1; tl >= 0; --tl) {
      if constexpr (NCP) {     // f = 1, a = 0: G_t = e_t + G_{t+1}, H_t = G_{t+1} + H_{t+1}
        r21 += 1.0f;
        const float nH = oG + oH;
        oG = e[tl] + oG;
        oH = nH;
      } else {
        const float fA = 1.0f - aA[tl], fM = 1.0f - aM[tl];
        const float kA = aA[tl] * al[tl] * iSa, kM = aM[tl] * mu[tl] * iSm;
        r21 = fM * (r11 + r21); r11 = fA * r11; r22 = fM * r22;
        const float nH = fmaf(fM, oG + oH, kM);
        oG = fmaf(fA, e[tl] + oG, kA);
        oH = nH;
      }
    }
    float G = 0.0f, H = 0.0f;
#pragma unroll
    for (int j = 1; j < K; ++j) {
      const float Go = NCP ? (G + oG) : fmaf(r11, G, oG);
      const float Ho = NCP ? (H + fmaf(r21, G, oH)) : fmaf(r21, G, fmaf(r22, H, oH));
      G = from_later<1>(0.0f, Go); H = from_later<1>(0.0f, Ho);
    }
    // backward pass 2: gradients
    float g_lSa = 0.0f, g_lSm = 0.0f;
#pragma unroll
    for (int tl = B - 1; tl >= 0; --tl) {
      const float zA = al[tl], zM = mu[tl];
      const float Ab = e[tl] + G, Mb = G + H;
      if constexpr (NCP) {
        g[NG + 2 * tl] = fmaf(S.Sa, Ab, -zA);
        g[NG + 2 * tl + 1] = fmaf(S.Sm, Mb, -zM);
        g_lSa = fmaf(Ab, zA, g_lSa);      // times Sa below
        g_lSm = fmaf(Mb, zM, g_lSm);
        G = Ab; H = Mb;
      } else {
        const float gA = zA * iSa, gM = zM * iSm;
        g[NG + 2 * tl] = Ab - gA;
        g[NG + 2 * tl + 1] = Mb - gM;
        G = fmaf(1.0f - aA[tl], Ab, aA[tl] * gA);
        H = fmaf(1.0f - aM[tl], Mb, aM[tl] * gM);
      }
    }
    const float nr = n_real();
    if constexpr (NCP) { g_lSa *= S.Sa; g_lSm *= S.Sm; }
    else { g_lSa = ssA - nr; g_lSm = ssM - nr; }
    g_lSa = group_sum<K>(g_lSa);
    g_lSm = group_sum<K>(g_lSm);
    g_beta = group_sum<K>(g_beta);
    g[0] = fmaf(g_lSa, sigmoidf_(sa) * iSa, -sa);
    g[1] = fmaf(g_lSm, sigmoidf_(sm) * iSm, -sm);
    g[2] = g_beta - beta;
    if (LOGP) {
      lp -= 0.5f * (ssA + ssM);
      if constexpr (!NCP) lp -= nr * (S.lSa + S.lSm);
      lp = group_sum<K>(lp) - 0.5f * (sa * sa + sm * sm + beta * beta);
    }
    return lp;
  }

  template <bool LOGP, int MODE>
  ARP_DEV float grad_m(const float (&q)[ND], float (&g)[ND]) const {
    if constexpr (MODE == 1) return grad_cp<LOGP>(q, g);
    else return grad_s<LOGP, MODE>(q, g);
  }
  template <int MODE>
  ARP_DEV void to_centered_m(const float (&q)[ND], float (&x)[ND]) const {
    x[0] = q[0]; x[1] = q[1]; x[2] = q[2];
    if constexpr (MODE == 1) {
#pragma unroll
      for (int i = 0; i < NL; ++i) x[NG + i] = q[NG + i];
    } else {
      const Scales S = scales(q[0], q[1]);
      float al[B], mu[B];
      forward_m<MODE>(q, S, al, mu);
#pragma unroll
      for (int tl = 0; tl < B; ++tl) { x[NG + 2 * tl] = al[tl]; x[NG + 2 * tl + 1] = mu[tl]; }
    }
  }
  template <int MODE>
  ARP_DEV void from_centered_m(const float (&x)[ND], float (&q)[ND]) const {
    q[0] = x[0]; q[1] = x[1]; q[2] = x[2];
    if constexpr (MODE == 1) {
#pragma unroll
      for (int i = 0; i < NL; ++i) q[NG + i] = x[NG + i];
    } else {
      const Scales S = scales(x[0], x[1]);
      const float iSa = __builtin_amdgcn_rcpf(S.Sa), iSm = __builtin_amdgcn_rcpf(S.Sm);
      float ap = from_earlier<1>(0.0f, x[NG + NL - 2]), mp = from_earlier<1>(0.0f, x[NG + NL - 1]);
#pragma unroll
      for (int tl = 0; tl < B; ++tl) {
        const float mA = ap + mp;
        const bool ok = real_step(tl);   // padding steps stay 0
        const float qa = MODE == 2 ? (x[NG + 2 * tl] - mA) * iSa : fmaf(aA[tl], mA, x[NG + 2 * tl] - mA);
        const float qm = MODE == 2 ? (x[NG + 2 * tl + 1] - mp) * iSm : fmaf(aM[tl], mp, x[NG + 2 * tl + 1] - mp);
        q[NG + 2 * tl] = ok ? qa : 0.0f;
        q[NG + 2 * tl + 1] = ok ? qm : 0.0f;
        ap = x[NG + 2 * tl]; mp = x[NG + 2 * tl + 1];
      }
    }
  }

  // d logp / d a = -loc g, d logp / d b = -log(scale) (1 + (xt - a loc) g) (see model_radon.h)
  ARP_DEV void dparam(const float (&q)[ND], const float (&g)[ND], float (&da)[ND], float (&db)[ND]) const {
    const Scales S = scales(q[0], q[1]);
    float al[B], mu[B], eA[B], eM[B];
    forward(q, S, al, mu, eA, eM);
    float ap, mp;
    incoming(al, mu, ap, mp);
    da[0] = da[1] = da[2] = 0.0f; db[0] = db[1] = db[2] = 0.0f;
#pragma unroll
    for (int tl = 0; tl < B; ++tl) {
      const float mA = ap + mp;
      da[NG + 2 * tl] = -mA * g[NG + 2 * tl];
      da[NG + 2 * tl + 1] = -mp * g[NG + 2 * tl + 1];
      const bool ok = real_step(tl);
      db[NG + 2 * tl] = ok ? -S.lSa * fmaf(fmaf(-aA[tl], mA, q[NG + 2 * tl]), g[NG + 2 * tl], 1.0f) : 0.0f;
      db[NG + 2 * tl + 1] = ok ? -S.lSm * fmaf(fmaf(-aM[tl], mp, q[NG + 2 * tl + 1]), g[NG + 2 * tl + 1], 1.0f) : 0.0f;
      ap = al[tl]; mp = mu[tl];
    }
  }

  ARP_DEV void to_centered(const float (&q)[ND], float (&x)[ND]) const {
    const Scales S = scales(q[0], q[1]);
    float al[B], mu[B], eA[B], eM[B];
    forward(q, S, al, mu, eA, eM);
    x[0] = q[0]; x[1] = q[1]; x[2] = q[2];
#pragma unroll
    for (int tl = 0; tl < B; ++tl) { x[NG + 2 * tl] = al[tl]; x[NG + 2 * tl + 1] = mu[tl]; }
  }
  ARP_DEV void from_centered(const float (&x)[ND], float (&q)[ND]) const {
    const Scales S = scales(x[0], x[1]);
    q[0] = x[0]; q[1] = x[1]; q[2] = x[2];
    // every location is a centred value: only the step before the lane's first comes from the neighbour
    float ap = from_earlier<1>(0.0f, x[NG + NL - 2]), mp = from_earlier<1>(0.0f, x[NG + NL - 1]);
#pragma unroll
    for (int tl = 0; tl < B; ++tl) {
      const float mA = ap + mp;
      const bool ok = real_step(tl);   // padding steps stay 0
      q[NG + 2 * tl] = ok ? fmaf(x[NG + 2 * tl] - mA, fast_exp(-(1.0f - bA[tl]) * S.lSa), aA[tl] * mA) : 0.0f;
      q[NG + 2 * tl + 1] = ok ? fmaf(x[NG + 2 * tl + 1] - mp, fast_exp(-(1.0f - bM[tl]) * S.lSm), aM[tl] * mp) : 0.0f;
      ap = x[NG + 2 * tl]; mp = x[NG + 2 * tl + 1];
    }
  }
};

}  // namespace arp
