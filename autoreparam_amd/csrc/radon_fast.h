// Packed-f32 chain kernels for the radon model in its two compile-time parameterisations
// (centred, non-centred): the headline path (BASELINE configs[1] and [3]; reference
// inference.py:198-242, 258-329, interleaved.py:113-155, models.py:826-837).
//
// Same algorithm and same random streams as the generic kernels in kernels.h (which keep
// serving the general VIP form), restated so that EVERYTHING a lane owns lives in register
// PAIRS: a lane's counties j = slot + K*i are held two at a time (pair k = counties 2k, 2k+1)
// and every per-county operation of a transition -- Box-Muller scaling, step sizes, kicks,
// drifts, kinetic energies, the gradient, the log density, the change of coordinates -- is one
// v_pk_*_f32 on a pair.  On gfx950 a v_pk_fma_f32 issues in 4.4 cycles against 2 x 2.5-2.9 for
// two v_fma_f32 at two waves per SIMD, and, as important, the pairs never have to be packed or
// unpacked (the generic float-array form spends ~15 % of its instructions on v_mov).
//
// Two algebraic facts of this model are used:
//  * the log joint is a quadratic form with no constant term, so
//        logp(q) = 1/2 q . (grad(q) + grad(0)),   grad(0) = (c0, c1, Sxy, Sy_j ...)
//    -- two packed operations per county pair next to the gradient instead of six;
//  * CP <-> NCP is a shear with unit Jacobian: logp is unchanged and the gradient follows by
//    the chain rule (`carry`), so the interleaved kernel never re-bootstraps.
#pragma once
#include "pk_chain.h"
#include "model_radon.h"

namespace arp {

template <int K_, int NL_>
struct RadonPk {
  static constexpr int K = K_, NL = NL_, NG = 3, ND = NG + NL_;
  static constexpr int NP = (NL_ + 1) / 2;          // county pairs (the last one half padding when NL is odd)
  static constexpr int DCAP = NG + K_ * NL_;
  static constexpr int LBASE = 3;
  static_assert(NL_ >= 4, "at least two full county pairs per lane");
  static_assert(K_ >= 4, "the packed kernels deal the top-level momenta out over the first slots of a chain");
  static constexpr int MINW = 2;
  static constexpr bool SCALAR_TAIL = true;   // pk_chain.h: an odd last county is taken as scalar operations
  using Args = RadonArgs;

  v2f n2[NP], sx2[NP], sy2[NP], u2[NP];
  v2f a2[NP];       // general form (MODE kModeVIP: cVIP / dVIP runs, a free per county; m has unit scale, so b is inert): dead otherwise
  v2f mlast;        // 1/0: which elements of the LAST pair are real counties
  float sxy, sxx;
  float c_sy, c_suy;   // sum_j Sy_j, sum_j u_j Sy_j over all counties (grad(0) of the NCP form; wave-uniform)
  float c_sya, c_suya; // the same sums weighted by (1 - a_j): grad(0) of the general form (equal in all lanes of a chain)
  int slot;
  bool last_ok;

  // row I/O contract of kernels.h (load_row / store_row_wave / stats_update_wave)
  static ARP_DEV int gg(int i) { return i; }
  ARP_DEV int lbase(int) const { return LBASE + slot; }
  static constexpr ARP_DEV int loff(int i) { return K * i; }
  ARP_DEV bool lvalid(int i) const { return i < NL - 1 ? true : last_ok; }

  ARP_DEV void init(const Args& A, const float* av, const float*, int slot_) {
    slot = slot_;
    const int J = A.J;
    last_ok = slot + K * (NL - 1) < J;
    sxy = A.sxy; sxx = A.sxx;
    c_sy = A.sy_tot; c_suy = A.suy_tot;
    float wsy = 0.0f, wsuy = 0.0f;
#pragma unroll
    for (int i = 0; i < 2 * NP; ++i) {
      const int j = slot + K * i;
      const bool ok = i < NL && j < J;
      const float nj = ok ? A.n[j] : 0.0f, sxj = ok ? A.sx[j] : 0.0f, syj = ok ? A.sy[j] : 0.0f, uj = ok ? A.u[j] : 0.0f;
      n2[i >> 1][i & 1] = nj; sx2[i >> 1][i & 1] = sxj; sy2[i >> 1][i & 1] = syj; u2[i >> 1][i & 1] = uj;
      const float aj = (ok && av) ? av[LBASE + j] : 0.0f;
      a2[i >> 1][i & 1] = aj;
      wsy = fmaf(1.0f - aj, syj, wsy);
      wsuy = fmaf((1.0f - aj) * uj, syj, wsuy);
    }
    c_sya = group_sum<K>(wsy); c_suya = group_sum<K>(wsuy);
    if (NL & 1) mlast = v2f{last_ok ? 1.0f : 0.0f, 0.0f};
    else mlast = v2f{1.0f, last_ok ? 1.0f : 0.0f};
  }

  // state <-> flattened float row (register renaming only)
  static ARP_DEV void unpack(const float (&v)[ND], float (&g3)[3], v2f (&c)[NP]) {
    g3[0] = v[0]; g3[1] = v[1]; g3[2] = v[2];
#pragma unroll
    for (int i = 0; i < 2 * NP; ++i) c[i >> 1][i & 1] = i < NL ? v[NG + i] : 0.0f;
  }
  static ARP_DEV void pack(const float (&g3)[3], const v2f (&c)[NP], float (&v)[ND]) {
    v[0] = g3[0]; v[1] = g3[1]; v[2] = g3[2];
#pragma unroll
    for (int i = 0; i < NL; ++i) v[NG + i] = c[i >> 1][i & 1];
  }

  // One pass over the lane's county pairs at position (qg, qc).
  //   PASS 0  interior leapfrog step: gradient, full kick, drift (q, p updated in place)
  //   PASS 1  closing: gradient -> (gg, gc), logp, and 1/2 |p + eps/2 g|^2 (the momentum itself is dead afterwards)
  //   PASS 2  bootstrap: gradient -> (gg, gc) and logp
  // MODE kModeCP:  r = mt - mu, m = mt,      dlogp/dmu = r
  // MODE kModeNCP: r = mt,      m = mt + mu, dlogp/dmu = l          (model_radon.h has the derivation)
  template <int MODE, int PASS>
  ARP_DEV void pass(float (&qg)[3], v2f (&qc)[NP], float (&pg)[3], v2f (&pc)[NP], const float (&eg)[3],
                    const v2f (&ec)[NP], float (&gg_)[3], v2f (&gc)[NP], float& lp, float& ke) const {
    const float mua = qg[0], b1 = qg[1], b2 = qg[2];
    const v2f vb1 = splat(b1), vnb2 = splat(-b2), vmua = splat(mua);
    const v2f vmua_last = vmua * mlast;     // padding: mu = 0 there (its u is 0), so r = m = 0
    // With an odd number of counties per lane (PA at 4 lanes: 17) the last "pair" is one county and a padding slot: it
    // is taken as SCALAR operations on the pair's first element (a v_fma_f32 issues in 2.5 cycles against 4.4 for the
    // v_pk_fma_f32 whose second half would be padding); the pair's second registers are never read.
    constexpr bool TAIL = (NL & 1) != 0;
    constexpr int NPF = NL / 2;               // full pairs
    v2f ah[1], auh[1], ams[1];      // accumulators, seeded by the first pair's terms
    v2f alp = splat(0.0f), ake = splat(0.0f);
    const v2f half = splat(0.5f);
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
      const v2f mt = qc[k];
      const v2f mu = vfma(u2[k], vb1, (k == NP - 1 && MODE == kModeCP) ? vmua_last : vmua);
      const v2f t = vfma(vnb2, sx2[k], sy2[k]);
      v2f r, m;
      if (MODE == kModeCP) { r = mt - mu; m = mt; }
      else if (MODE == kModeNCP) { r = mt; m = mt + mu; }
      else { r = vfma(-a2[k], mu, mt); m = r + mu; }           // general a: xt ~ N(a mu, 1), m = xt + (1 - a) mu
      const v2f l = vfma(-n2[k], m, t);
      const v2f gm = l - r;
      const v2f h = (MODE == kModeCP) ? r : ((MODE == kModeNCP) ? l : vfma(-a2[k], gm, l));   // d logp / d mu
      if (k == 0) {
        ah[0] = h; auh[0] = u2[k] * h; ams[0] = m * sx2[k];
      } else {
        ah[0] += h;
        auh[0] = vfma(u2[k], h, auh[0]);
        ams[0] = vfma(m, sx2[k], ams[0]);
      }
      if (PASS == 0) {
        const v2f pn = vfma(ec[k], gm, pc[k]);
        pc[k] = pn;
        qc[k] = vfma(ec[k], pn, mt);
      } else {
        gc[k] = gm;
        alp = k == 0 ? mt * (gm + sy2[k]) : vfma(mt, gm + sy2[k], alp);
        if (PASS == 1) {
          const v2f pf = vfma(half, ec[k] * gm, pc[k]);
          ake = k == 0 ? pf * pf : vfma(pf, pf, ake);
        }
      }
    }
    float th_t = 0.0f, tuh_t = 0.0f, tms_t = 0.0f, alp_t = 0.0f, ake_t = 0.0f;
    if constexpr (TAIL) {
      constexpr int k = NP - 1;
      const float mt = qc[k][0], uk = u2[k][0], sxk = sx2[k][0], syk = sy2[k][0], ek = ec[k][0];
      const float mu = fmaf(uk, b1, MODE == kModeCP ? vmua_last[0] : mua);
      const float t = fmaf(-b2, sxk, syk);
      float r, m;
      if (MODE == kModeCP) { r = mt - mu; m = mt; }
      else if (MODE == kModeNCP) { r = mt; m = mt + mu; }
      else { r = fmaf(-a2[k][0], mu, mt); m = r + mu; }
      const float l = fmaf(-n2[k][0], m, t);
      const float gm = l - r;
      const float h = (MODE == kModeCP) ? r : ((MODE == kModeNCP) ? l : fmaf(-a2[k][0], gm, l));
      th_t = h; tuh_t = uk * h; tms_t = m * sxk;
      if (PASS == 0) {
        const float pn = fmaf(ek, gm, pc[k][0]);
        pc[k][0] = pn;
        qc[k][0] = fmaf(ek, pn, mt);
      } else {
        gc[k] = v2f{gm, 0.0f};
        alp_t = mt * (gm + syk);
        if (PASS == 1) {
          const float pf = fmaf(0.5f, ek * gm, pc[k][0]);
          ake_t = pf * pf;
        }
      }
    }
    const v2f th = ah[0], tuh = auh[0], tms = ams[0];
    const float s_h = group_sum<K>(TAIL ? (th[0] + th[1]) + th_t : th[0] + th[1]);
    const float s_uh = group_sum<K>(TAIL ? (tuh[0] + tuh[1]) + tuh_t : tuh[0] + tuh[1]);
    const float s_ms = group_sum<K>(TAIL ? (tms[0] + tms[1]) + tms_t : tms[0] + tms[1]);
    const float g0 = s_h - mua, g1 = s_uh - b1, g2 = fmaf(-b2, sxx, sxy) - s_ms - b2;
    if (PASS == 0) {
      pg[0] = fmaf(eg[0], g0, pg[0]); qg[0] = fmaf(eg[0], pg[0], mua);
      pg[1] = fmaf(eg[1], g1, pg[1]); qg[1] = fmaf(eg[1], pg[1], b1);
      pg[2] = fmaf(eg[2], g2, pg[2]); qg[2] = fmaf(eg[2], pg[2], b2);
    } else {
      gg_[0] = g0; gg_[1] = g1; gg_[2] = g2;
      // logp = 1/2 q . (g + g(0)); g(0) = (0, 0, Sxy, Sy_j) centred, (sum Sy, sum u Sy, Sxy, Sy_j) non-centred
      const float c0 = MODE == kModeCP ? 0.0f : (MODE == kModeNCP ? c_sy : c_sya);
      const float c1 = MODE == kModeCP ? 0.0f : (MODE == kModeNCP ? c_suy : c_suya);
      float top = mua * (g0 + c0);
      top = fmaf(b1, g1 + c1, top);
      top = fmaf(b2, g2 + sxy, top);
      lp = 0.5f * (group_sum<K>(TAIL ? (alp[0] + alp[1]) + alp_t : alp[0] + alp[1]) + top);
      if (PASS == 1) {
        float kg = 0.0f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const float pf = fmaf(0.5f * eg[i], gg_[i], pg[i]);
          kg = fmaf(pf, pf, kg);
        }
        ke = 0.5f * (group_sum<K>(TAIL ? (ake[0] + ake[1]) + ake_t : ake[0] + ake[1]) + kg);
      }
    }
  }

  // Change of coordinates of the state AND its cached gradient (the shear m = mt + mu(mua, b1)):
  // FROM == kModeCP: CP -> NCP, FROM == kModeNCP: NCP -> CP.
  template <int FROM>
  ARP_DEV void carry(float (&qg)[3], v2f (&qc)[NP], float (&gg_)[3], const v2f (&gc)[NP]) const {
    constexpr bool TAIL = (NL & 1) != 0;
    constexpr int NPF = NL / 2;
    const v2f vb1 = splat(qg[1]), vmua = splat(qg[0]);
    const v2f vmua_last = vmua * mlast;
    v2f s = gc[0], su = u2[0] * gc[0];             // gradients of padding elements are 0
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
      const v2f mu = vfma(u2[k], vb1, k == NP - 1 ? vmua_last : vmua);
      if (k > 0) { s += gc[k]; su = vfma(u2[k], gc[k], su); }
      qc[k] = (FROM == kModeCP) ? qc[k] - mu : qc[k] + mu;
    }
    float st = 0.0f, sut = 0.0f;
    if constexpr (TAIL) {
      constexpr int k = NP - 1;
      const float mu = fmaf(u2[k][0], qg[1], vmua_last[0]);
      st = gc[k][0]; sut = u2[k][0] * gc[k][0];
      qc[k][0] = (FROM == kModeCP) ? qc[k][0] - mu : qc[k][0] + mu;
    }
    const float ts = group_sum<K>(TAIL ? (s[0] + s[1]) + st : s[0] + s[1]);
    const float tsu = group_sum<K>(TAIL ? (su[0] + su[1]) + sut : su[0] + su[1]);
    gg_[0] += (FROM == kModeCP) ? ts : -ts;
    gg_[1] += (FROM == kModeCP) ? tsu : -tsu;
  }

  // centred coordinates of a state held in parameterisation MODE
  template <int MODE>
  ARP_DEV void to_centered(const float (&qg)[3], const v2f (&qc)[NP], float (&xg)[3], v2f (&xc)[NP]) const {
    xg[0] = qg[0]; xg[1] = qg[1]; xg[2] = qg[2];
    const v2f vb1 = splat(qg[1]), vmua = splat(qg[0]);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const v2f mu = vfma(u2[k], vb1, k == NP - 1 ? vmua * mlast : vmua);
      xc[k] = (MODE == kModeNCP) ? qc[k] + mu : ((MODE == kModeCP) ? qc[k] : vfma(splat(1.0f) - a2[k], mu, qc[k]));
    }
  }
};

// Interleaved CP / NCP sampling (interleaved.Interleaved.one_step, interleaved.py:113-155): a centred transition,
// the change of coordinates, a non-centred transition, the change back; each inner kernel keeps its own
// step-size adaptation state (inference.py:288-306).  The gradient and log density are carried across the
// shear instead of being recomputed (kernels.h: interleaved_kernel, CARRY), 2*num_ls gradient evaluations per step.
template <class T, bool STATS = false>
__global__ __launch_bounds__(kBlock, T::MINW) void radon_interleaved_kernel(RadonArgs A, HmcParams P) {
  constexpr int K = T::K, NP = T::NP, ND = T::ND;
  const RelayId rid = relay_begin(P);      // kernels.h: the launch's steps in segments, a workgroup per (segment, chain block)
  if (rid.seg < 0) return;                 // a hand-over timed out: leave the state as it is (kernels.h: relay_begin)
  const unsigned bid = rid.bid;
  const unsigned t = bid * (unsigned)kBlock + threadIdx.x;
  const int slot = (int)(t % K);
  int c = (int)(t / K);
  const bool live = c < P.C;
  if (!live) c = P.C - 1;
  const int D = P.D;
  T M;
  M.init(A, nullptr, nullptr, slot);

  __shared__ float s_eps[2][PkBlock<T>::kEps];
  __shared__ __attribute__((aligned(16))) float s_save[PkBlock<T>::kSave];
  __shared__ __attribute__((aligned(16))) float s_stats[STATS ? PkStats<T>::kFloats : 4];   // pk_chain.h: PkStats
  int n_acc = 0;
  float* wsave = s_save + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * pk_save_wave_floats<T>();
  float* save = wsave + 2 * (threadIdx.x & 63);
  float* stage = wsave;   // the wave's staging block aliases its own parked state (dead between transitions)
  // first chain of this wave: wave-uniform, kept in SGPRs so that row addresses are scalar arithmetic
  const long long cw0 = (long long)((bid * (unsigned)kBlock + (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x & ~63)) / K);
  const int cl = (threadIdx.x & 63) / K;
  const int nvalid = (int)(P.C - cw0 < 64 / K ? (P.C - cw0 > 0 ? P.C - cw0 : 0) : 64 / K) * D;
  for (int d = threadIdx.x; d < PkBlock<T>::kEps; d += kBlock) {
    s_eps[0][d] = d < D ? P.eps0[d] : 0.0f;
    s_eps[1][d] = d < D ? P.eps0_1[d] : 0.0f;
  }
  __syncthreads();

  float qg[3], gg_[3]; v2f qc[NP], gc[NP];
  float lp;
  {
    float v[ND];
    load_row(M, P.q + (size_t)c * D, v);
    T::unpack(v, qg, qc);
    if (P.step_base == 0 || !P.grad) {
      float pg[3] = {0.f, 0.f, 0.f}, eg[3] = {0.f, 0.f, 0.f}; v2f pc[NP], ec[NP]; float ke;
      M.template pass<kModeCP, 2>(qg, qc, pg, pc, eg, ec, gg_, gc, lp, ke);
    } else {
      load_row(M, P.grad + (size_t)c * D, v);
      T::unpack(v, gg_, gc);
      lp = P.logp[c];
    }
  }
  float kap[2], es[2], la_[2];
  Rng rng;
  uint32_t* rs = P.rng + ((size_t)c * kRngSlots + slot) * 4;
  uint32_t nacc0, nacc1;
  if (P.step_base == 0) {
    kap[0] = kap[1] = 1.0f; es[0] = es[1] = 0.0f; la_[0] = la_[1] = 0.0f;
    nacc0 = nacc1 = 0u;
    rng = rng_seed(P.seed, (unsigned long long)(P.chain_offset + c), (uint32_t)slot, (uint32_t)K);
  } else {
    kap[0] = P.adapt[(size_t)c * 4 + 0]; es[0] = P.adapt[(size_t)c * 4 + 1]; la_[0] = P.adapt[(size_t)c * 4 + 2];
    kap[1] = P.adapt1[(size_t)c * 4 + 0]; es[1] = P.adapt1[(size_t)c * 4 + 1]; la_[1] = P.adapt1[(size_t)c * 4 + 2];
    nacc0 = P.accept_count[c]; nacc1 = P.accept_count1[c];
    rng = Rng{rs[0], rs[1]};
  }
  int next_rec = P.rec_step, rec_row = P.rec_row, bpos = P.stats_bpos;
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  for (int s = 0; s < P.n_steps; ++s) {
    const long long n = P.step_base + s + 1;
    bool acc0, acc1;
    float la = pk_transition<kModeCP>(M, rng, P.L, kap[0], s_eps[0], qg, qc, gg_, gc, lp, acc0, save);
    nacc0 += acc0 ? 1u : 0u;
    adapt_update(P, n, la, kap[0], es[0], la_[0]);
    M.template carry<kModeCP>(qg, qc, gg_, gc);
    la = pk_transition<kModeNCP>(M, rng, P.L1, kap[1], s_eps[1], qg, qc, gg_, gc, lp, acc1, save);
    nacc1 += acc1 ? 1u : 0u;
    adapt_update(P, n, la, kap[1], es[1], la_[1]);
    M.template carry<kModeNCP>(qg, qc, gg_, gc);

    if (s == next_rec && rec_row < P.n_samples) {
      // the state is back in centred coordinates, which are also the ones the reference records
      const bool to_trace = P.trace && cw0 < P.trace_chains;
      if (to_trace) {
        const int nv = min(nvalid, (int)(P.trace_chains - cw0) * D);
        pk_store_rows<true>(M, stage, P.trace + ((size_t)rec_row * P.trace_chains + cw0) * D, cl, D, nv, qg, qc);
      }
      if (STATS) {
        pk_stats_accumulate<T>(s_stats, ++n_acc, qg, qc);
        const bool bend = bpos + 1 == P.stats_batch;
        if (bend) {
          pk_stats_fold(M, stage, s_stats, P, cw0, cl, D, nvalid, n_acc, rec_row + 1 == n_acc, true);
          n_acc = 0;
        }
        bpos = bend ? 0 : bpos + 1;
      } else if (P.stats) {
        float x[ND];
        T::pack(qg, qc, x);
        const bool bend = bpos + 1 == P.stats_batch;
        stats_update_wave(M, stage, P, cw0, cl, D, nvalid, x, rec_row == 0, bend);
        bpos = bend ? 0 : bpos + 1;
      }
      if (live && slot == 0) {
        unsigned ci = (unsigned)c;
        asm volatile("" : "+v"(ci));
        if (P.trace_accept) (P.trace_accept + (size_t)rec_row * P.C)[ci] = acc0 ? 1 : 0;
        if (P.trace_accept1) (P.trace_accept1 + (size_t)rec_row * P.C)[ci] = acc1 ? 1 : 0;
        if (P.rec_accept) P.rec_accept[ci] += acc0 ? 1u : 0u;
        if (P.rec_accept1) P.rec_accept1[ci] += acc1 ? 1u : 0u;
      }
      next_rec += P.thin;
      rec_row += 1;
    }
  }

  if (STATS && n_acc > 0) pk_stats_fold(M, stage, s_stats, P, cw0, cl, D, nvalid, n_acc, rec_row == n_acc, false);
  size_t c2 = (size_t)c;
  asm volatile("" : "+v"(c2));
  const long long cw2 = cw0;
  pk_store_rows(M, stage, P.q + cw2 * D, cl, D, nvalid, qg, qc);
  if (P.grad) pk_store_rows(M, stage, P.grad + cw2 * D, cl, D, nvalid, gg_, gc);
  if (live) {
    uint32_t* rs2 = P.rng + ((size_t)c2 * kRngSlots + slot) * 4;
    rs2[0] = rng.x; rs2[1] = rng.c; rs2[2] = 0u; rs2[3] = 0u;
    if (slot == 0) {
      if (P.grad) P.logp[c2] = lp;
      P.adapt[c2 * 4 + 0] = kap[0]; P.adapt[c2 * 4 + 1] = es[0]; P.adapt[c2 * 4 + 2] = la_[0];
      P.adapt1[c2 * 4 + 0] = kap[1]; P.adapt1[c2 * 4 + 1] = es[1]; P.adapt1[c2 * 4 + 2] = la_[1];
      P.accept_count[c2] = nacc0; P.accept_count1[c2] = nacc1;
    }
  }
  relay_end(P, rid);
}

}  // namespace arp
