// Packed-f32 chain kernels: the transition, row I/O and the HMC kernel shared by the lane models that keep a chain's
// sliced latents in register PAIRS (radon_fast.h, election_fast.h).  Same algorithm and random streams as the generic
// kernels in kernels.h (stream layout 1); every per-slice operation of a transition is one v_pk_*_f32 on a pair.
//
// A packed lane model T provides
//   K, NL, NG, ND, NP, DCAP, LBASE, MINW, Args, init(A, av, bv, slot), slot, gg(i), lbase(i), loff(i), lvalid(i), mlast,
//   unpack / pack (flattened row <-> (top-level floats, pairs)),
//   pass<MODE, PASS>(qg, qc, pg, pc, eg, ec, gg, gc, lp, ke)   PASS 0 interior (gradient, kick, drift), 1 closing
//                    (gradient, logp, kinetic energy after the closing half kick), 2 bootstrap (gradient, logp),
//   to_centered<MODE>(qg, qc, xg, xc).
#pragma once
#include "kernels.h"

namespace arp {

// The value held by slot SRC (< 4) of a chain, in all of its K lanes: a quad broadcast, then -- the chain's first quad
// being right -- a mirror into the other quads that writes only them (DPP bank masks; a bank is a quad of a 16-lane row).
// Two or three v_mov_dpp instead of a select and a log2(K)-step butterfly of adds.
template <int K, int SRC>
ARP_DEV float group_bcast_from(float v, int) {
  static_assert(SRC < 4, "source slot must lie in the first quad");
  static_assert(K == 1 || K >= 4, "a quad holds one chain or part of one");
  if (K == 1) return v;
  int x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), SRC * 0x55, 0xF, 0xF, true);      // quad_perm [SRC, SRC, SRC, SRC]
  if (K >= 8) x = __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, K == 8 ? 0xA : 0x2, false);     // row_half_mirror -> quad 1 (and 3)
  if (K >= 16) x = __builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xC, false);                   // row_mirror -> quads 2, 3
  return __int_as_float(x);
}

ARP_DEV v2f splat(float x) { return v2f{x, x}; }

// Lane models with an odd number of slices per lane may treat the last "pair" (one slice and a padding slot) as scalar
// operations on the pair's first element everywhere (SCALAR_TAIL = true; radon: the padding half of a v_pk_*_f32 costs
// as much as the real one).  The pair's second registers are then never read by the transition.
template <class T, class = void> struct pk_scalar_tail { static constexpr bool value = false; };
template <class T> struct pk_scalar_tail<T, std::void_t<decltype(T::SCALAR_TAIL)>> { static constexpr bool value = T::SCALAR_TAIL && (T::NL & 1); };

// interior leapfrog passes per block of the transition loop: 3 unless the lane model says otherwise (PASS_BLOCK = 1:
// election's passes are long enough that the copies do not matter and a three-pass block costs it registers)
template <class T, class = void> struct pk_pass_block { static constexpr int value = 3; };
template <class T> struct pk_pass_block<T, std::void_t<decltype(T::PASS_BLOCK)>> { static constexpr int value = T::PASS_BLOCK; };

// Box-Muller pair as a register pair: (r cos, r sin) = one packed multiply
ARP_DEV v2f normal_pair2(uint32_t w0, uint32_t w1) {
  const float u = fmaf((float)w0, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
  const float rev = angle_rev(w1);
  const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u));
  const v2f cs = {__builtin_amdgcn_cosf(rev), __builtin_amdgcn_sinf(rev)};
  return cs * splat(r);
}

// Parked start-of-trajectory state, one region per WAVE (so that a wave's row staging block can alias its own region:
// waves of a workgroup are not in step): pairs as 8-byte columns of 64 lanes (conflict free: consecutive lanes touch
// consecutive 8-byte words), the 2 NG top-level floats behind them as 4-byte columns of the wave's 64 / K CHAINS (the K
// lanes of a chain hold identical copies: they write the same word and read it back as a broadcast).
template <class T> constexpr int pk_save_wave_floats() { return 4 * T::NP * 64 + 2 * T::NG * (64 / T::K); }
template <class T> constexpr int pk_save_floats() { return (kBlock / 64) * pk_save_wave_floats<T>(); }

// One HMC transition (mcmc.HamiltonianMonteCarlo.one_step as wired at inference.py:218-222); the contract of
// kernels.h: hmc_transition with the state in pairs.  Random stream layout 1 (DESIGN.md "Randomness"): a slot
// draws one normal per county it owns, then one more, which is the momentum of top-level scalar `slot`
// (slots 0..2), then the Metropolis word (slot 0's is used).
//
// The transition comes in two halves, the state-independent DRAW and the rest.  (Round 3 tried the draw of the next
// transition between the LDS reads that end a transition -- the parked state of the rejecting lanes, the next base step
// sizes -- and their first use, SQ_WAIT_INST_LDS being 12 % of a wave's time: same 3.38 - 3.41 ms per headline launch,
// six spilled registers.  The other wave of the SIMD already covers those waits.)
template <class T>
struct PkDraw {
  v2f pc[T::NP];       // momenta of the county pairs (padding masked)
  float pg[T::NG];     // momenta of the top-level scalars, replicated
  float logu, ke0;     // log of the Metropolis uniform; kinetic energy of the draw
  v2f ec[T::NP];       // BASE step sizes of the coming transition (pk_eps_load), scaled by the chain's multiplier in pk_run
  float eg[T::NG];
};

// base step sizes from LDS (zero beyond D, so padding elements never move)
template <class T>
ARP_DEV void pk_eps_load(const T& M, const float* __restrict__ s_eps, PkDraw<T>& d) {
  constexpr int K = T::K, NP = T::NP, NG = T::NG;
  constexpr bool TAIL = pk_scalar_tail<T>::value;
  constexpr int NPF = TAIL ? NP - 1 : NP;
  const float* e = s_eps + T::LBASE + M.slot;
#pragma unroll
  for (int k = 0; k < NPF; ++k) d.ec[k] = v2f{e[K * 2 * k], e[K * (2 * k + 1)]};
  if constexpr (TAIL) d.ec[NP - 1] = v2f{e[K * 2 * (NP - 1)], 0.0f};
#pragma unroll
  for (int i = 0; i < NG; ++i) d.eg[i] = s_eps[M.gg(i)];
}

template <class T>
ARP_DEV void pk_draw(const T& M, Rng& rng, PkDraw<T>& d) {
  constexpr int K = T::K, NP = T::NP, NL = T::NL, NG = T::NG;
  constexpr bool TAIL = pk_scalar_tail<T>::value;
  constexpr int NPF = TAIL ? NP - 1 : NP;       // pairs handled as pairs
  float extra;
#pragma unroll
  for (int k = 0; k < NPF; ++k) {
    const uint32_t w0 = rng_next(rng), w1 = rng_next(rng);
    d.pc[k] = normal_pair2(w0, w1);
  }
  if constexpr (TAIL) {
    const uint32_t w0 = rng_next(rng), w1 = rng_next(rng);
    const v2f z = normal_pair2(w0, w1);
    extra = z[1];
    d.pc[NP - 1] = v2f{z[0] * M.mlast[0], 0.0f};
  } else {
    if (NL & 1) {
      extra = d.pc[NP - 1][1];
    } else {
      const uint32_t w0 = rng_next(rng), w1 = rng_next(rng);
      extra = normal_pair2(w0, w1)[0];
    }
    d.pc[NP - 1] *= M.mlast;
  }
  float u = u01_open0(rng_next(rng));
  u = group_bcast_from<K, 0>(u, M.slot);
  d.logu = fast_log(u);
  static_assert(NG <= 4 && NG <= K, "one extra normal per slot covers the top-level scalars");
  d.pg[0] = group_bcast_from<K, 0>(extra, M.slot);
  if constexpr (NG > 1) d.pg[1] = group_bcast_from<K, 1>(extra, M.slot);
  if constexpr (NG > 2) d.pg[2] = group_bcast_from<K, 2>(extra, M.slot);
  if constexpr (NG > 3) d.pg[3] = group_bcast_from<K, 3>(extra, M.slot);
  v2f a = d.pc[0] * d.pc[0];
#pragma unroll
  for (int k = 1; k < NPF; ++k) a = vfma(d.pc[k], d.pc[k], a);
  float at = 0.0f;
  if constexpr (TAIL) at = d.pc[NP - 1][0] * d.pc[NP - 1][0];
  float kg = 0.0f;
#pragma unroll
  for (int i = 0; i < NG; ++i) kg = fmaf(d.pg[i], d.pg[i], kg);
  d.ke0 = 0.5f * (group_sum<K>(TAIL ? (a[0] + a[1]) + at : a[0] + a[1]) + kg);
}

// the rest of the transition, with the draw and the base step sizes in `d`
template <int MODE, class T>
ARP_DEV float pk_run(const T& M, PkDraw<T>& d, int L, float kappa, float (&qg)[T::NG], v2f (&qc)[T::NP],
                     float (&gg_)[T::NG], v2f (&gc)[T::NP], float& lp, bool& accepted, float* __restrict__ save) {
  constexpr int K = T::K, NP = T::NP, NG = T::NG;
  constexpr bool TAIL = pk_scalar_tail<T>::value;
  constexpr int NPF = TAIL ? NP - 1 : NP;
  v2f (&pc)[NP] = d.pc; float (&pg)[NG] = d.pg;
  v2f (&ec)[NP] = d.ec; float (&eg)[NG] = d.eg;
  {
    const v2f vk = splat(kappa);
#pragma unroll
    for (int k = 0; k < NPF; ++k) ec[k] *= vk;
    if constexpr (TAIL) ec[NP - 1][0] *= kappa;
#pragma unroll
    for (int i = 0; i < NG; ++i) eg[i] *= kappa;
  }
  // park the start state
  {
    v2f* s2 = reinterpret_cast<v2f*>(save);
#pragma unroll
    for (int k = 0; k < NPF; ++k) { s2[k * 64] = qc[k]; s2[(NP + k) * 64] = gc[k]; }
    if constexpr (TAIL) {
      reinterpret_cast<float*>(s2 + (NP - 1) * 64)[0] = qc[NP - 1][0];
      reinterpret_cast<float*>(s2 + (2 * NP - 1) * 64)[0] = gc[NP - 1][0];
    }
    float* s1 = save + 4 * NP * 64 - 2 * (threadIdx.x & 63) + (threadIdx.x & 63) / K;   // chain columns behind the pair columns
#pragma unroll
    for (int i = 0; i < NG; ++i) { s1[i * (64 / K)] = qg[i]; s1[(NG + i) * (64 / K)] = gg_[i]; }
  }
  {   // first half kick and first drift
    const v2f half = splat(0.5f);
#pragma unroll
    for (int k = 0; k < NPF; ++k) {
      pc[k] = vfma(half, ec[k] * gc[k], pc[k]);
      qc[k] = vfma(ec[k], pc[k], qc[k]);
    }
    if constexpr (TAIL) {
      constexpr int k = NP - 1;
      const float e0 = ec[k][0];
      const float pn = fmaf(0.5f, e0 * gc[k][0], pc[k][0]);
      pc[k][0] = pn;
      qc[k][0] = fmaf(e0, pn, qc[k][0]);
    }
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      pg[i] = fmaf(0.5f * eg[i], gg_[i], pg[i]);
      qg[i] = fmaf(eg[i], pg[i], qg[i]);
    }
  }
  float dlp, dke;
  // interior steps three at a time (the reference's num_leapfrog_steps = 4 is one such block): a loop around a single
  // pass ends every pass with a round of register copies for its loop-carried state (8 - 9 v_mov per pass)
  int l = 1;
  if constexpr (pk_pass_block<T>::value == 3) {
#pragma unroll 1
    for (; l + 3 <= L; l += 3) {
      M.template pass<MODE, 0>(qg, qc, pg, pc, eg, ec, gg_, gc, dlp, dke);
      M.template pass<MODE, 0>(qg, qc, pg, pc, eg, ec, gg_, gc, dlp, dke);
      M.template pass<MODE, 0>(qg, qc, pg, pc, eg, ec, gg_, gc, dlp, dke);
    }
  }
#pragma unroll 1
  for (; l < L; ++l) M.template pass<MODE, 0>(qg, qc, pg, pc, eg, ec, gg_, gc, dlp, dke);
  float lp1, ke1;
  M.template pass<MODE, 1>(qg, qc, pg, pc, eg, ec, gg_, gc, lp1, ke1);

  // log accept ratio; any non-finite energy error rejects (TFP safe_sum semantics)
  float la = (lp1 - lp) + (d.ke0 - ke1);
  if (!(fabsf(la) <= 3.0e38f)) la = -INFINITY;
  accepted = d.logu < la;
  if (!accepted) {
    const v2f* s2 = reinterpret_cast<const v2f*>(save);
#pragma unroll
    for (int k = 0; k < NPF; ++k) { qc[k] = s2[k * 64]; gc[k] = s2[(NP + k) * 64]; }
    if constexpr (TAIL) {
      qc[NP - 1][0] = reinterpret_cast<const float*>(s2 + (NP - 1) * 64)[0];
      gc[NP - 1][0] = reinterpret_cast<const float*>(s2 + (2 * NP - 1) * 64)[0];
    }
    const float* s1 = save + 4 * NP * 64 - 2 * (threadIdx.x & 63) + (threadIdx.x & 63) / K;
#pragma unroll
    for (int i = 0; i < NG; ++i) { qg[i] = s1[i * (64 / K)]; gg_[i] = s1[(NG + i) * (64 / K)]; }
  }
  lp = accepted ? lp1 : lp;
  return la;
}

// draw + run in one piece (kernels that do not interleave the draw with anything)
template <int MODE, class T>
ARP_DEV float pk_transition(const T& M, Rng& rng, int L, float kappa, const float* __restrict__ s_eps,
                               float (&qg)[T::NG], v2f (&qc)[T::NP], float (&gg_)[T::NG], v2f (&gc)[T::NP], float& lp,
                               bool& accepted, float* __restrict__ save) {
  PkDraw<T> d;
  pk_eps_load(M, s_eps, d);
  pk_draw(M, rng, d);
  return pk_run<MODE>(M, d, L, kappa, qg, qc, gg_, gc, lp, accepted, save);
}

// the copy half of kernels.h: store_row_wave as a real call (cold path; keeps its address arithmetic out of the callers)
__device__ __noinline__ void copy_stage_rows(const float* stage, float* gdst, int nvalid) {
  const int lane = threadIdx.x & 63;
  if ((reinterpret_cast<uintptr_t>(gdst) & 15) == 0) {
    for (int k = lane * 4; k < nvalid; k += 256) {
      const float4 t = *reinterpret_cast<const float4*>(stage + k);
      if (k + 3 < nvalid) {
        *reinterpret_cast<float4*>(gdst + k) = t;
      } else {
        gdst[k] = t.x;
        if (k + 1 < nvalid) gdst[k + 1] = t.y;
        if (k + 2 < nvalid) gdst[k + 2] = t.z;
      }
    }
  } else {
    for (int k = lane; k < nvalid; k += 64) gdst[k] = stage[k];
  }
}

// A wave's rows (its 64/K consecutive chains) to a [C][D] array: store_row_wave of kernels.h with the state in pairs and,
// when the wave is full and D is the instantiation's own dimension (the reference's PA: 68 = 4 x 17 counties), every
// offset a compile-time constant -- the staging writes are ds_write2_b32 off one base, the copy is an unrolled run of
// ds_read_b128 / global_store_dwordx4 off a wave-uniform base: no address arithmetic on the vector pipe.
template <bool STREAM = false, class T>
ARP_DEV void pk_store_rows(const T& M, float* stage, float* gdst, int cl, int D, int nvalid,
                              const float (&xg)[T::NG], const v2f (&xc)[T::NP]) {
  constexpr int K = T::K, NL = T::NL, DC = T::DCAP, NV = (64 / K) * DC;
  if (D == DC && nvalid == NV && (NV & 3) == 0 && (reinterpret_cast<uintptr_t>(gdst) & 15) == 0) {
    float* row = stage + cl * DC;
    if (M.slot == 0) {
#pragma unroll
      for (int i = 0; i < T::NG; ++i) row[M.gg(i)] = xg[i];
    }
    float* e = row + T::LBASE + M.slot;
#pragma unroll
    for (int i = 0; i < NL; ++i) e[K * i] = xc[i >> 1][i & 1];      // D == DCAP: every slice is a real county
    __builtin_amdgcn_wave_barrier();
    // the lane's word index is formed afresh here: kept live across the sampling loop it would cost a register the
    // chain kernels do not have
    int lane = threadIdx.x;
    asm volatile("" : "+v"(lane));
    lane &= 63;
    const float4* s4 = reinterpret_cast<const float4*>(stage);
    float4* g4 = reinterpret_cast<float4*>(gdst);
    // ALL the block's LDS reads first, into registers of their own, then the stores: read -> wait -> store -> read into
    // the same four registers (what the loop compiled to through round 4) is five LDS round trips in a row, each also
    // waiting for the store before it to have taken its data -- 600 - 1 000 cycles per row that the other wave of the
    // SIMD, recording the same step, cannot cover.  The registers are free here: a row is stored between transitions.
    // (a lane past the block's end re-reads and re-stores the block's LAST 16 bytes: the same bytes to the same place,
    // so neither the reads nor the stores are predicated and nothing branches)
    constexpr int NIT = (NV / 4 + 63) / 64;
    v4f_nt t[NIT];      // (a native vector: an array of HIP's float4 structs was left in scratch by one instantiation)
    auto word = [&](int it) { const int k = lane + 64 * it; return ((it + 1) * 64 <= NV / 4 || k < NV / 4) ? k : NV / 4 - 1; };
#pragma unroll
    for (int it = 0; it < NIT; ++it) t[it] = reinterpret_cast<const v4f_nt*>(s4)[word(it)];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < NIT; ++it) store_v4<STREAM>(reinterpret_cast<v4f_nt*>(g4) + word(it), t[it]);
    __builtin_amdgcn_wave_barrier();
  } else if (nvalid == (64 / K) * D && (nvalid & 3) == 0 && (reinterpret_cast<uintptr_t>(gdst) & 15) == 0) {
    // a full wave of a county count that leaves padding slices (D < DCAP: PA at 8 lanes per chain, the strong-scaling
    // shard): run-time row stride, but the copy stays inline -- the out-of-line one below returns through
    // s_waitcnt vmcnt(0), i.e. waits for the row it has just stored, and a lone wave per SIMD has nothing to hide that
    // behind (12 % of the 8 192-chain launch).
    float* row = stage + cl * D;
    if (M.slot == 0) {
#pragma unroll
      for (int i = 0; i < T::NG; ++i) row[M.gg(i)] = xg[i];
    }
    float* e = row + T::LBASE + M.slot;
#pragma unroll
    for (int i = 0; i < NL; ++i)
      if (M.lvalid(i)) e[K * i] = xc[i >> 1][i & 1];
    __builtin_amdgcn_wave_barrier();
    int lane = threadIdx.x;
    asm volatile("" : "+v"(lane));
    lane &= 63;
    const float4* s4 = reinterpret_cast<const float4*>(stage);
    float4* g4 = reinterpret_cast<float4*>(gdst);
    const int n4 = nvalid >> 2;
    constexpr int NIT = (NV / 4 + 63) / 64;
    v4f_nt t[NIT];
    auto word = [&](int it) { const int k = lane + 64 * it; return k < n4 ? k : n4 - 1; };      // n4 >= 1 here
#pragma unroll
    for (int it = 0; it < NIT; ++it) t[it] = reinterpret_cast<const v4f_nt*>(s4)[word(it)];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < NIT; ++it) store_v4<STREAM>(reinterpret_cast<v4f_nt*>(g4) + word(it), t[it]);
    __builtin_amdgcn_wave_barrier();
  } else {   // ragged tail of the launch, or an unaligned destination: general offsets, out-of-line copy
    float* row = stage + cl * D;
    if (M.slot == 0) {
#pragma unroll
      for (int i = 0; i < T::NG; ++i) row[M.gg(i)] = xg[i];
    }
    float* e = row + T::LBASE + M.slot;
#pragma unroll
    for (int i = 0; i < NL; ++i)
      if (M.lvalid(i)) e[K * i] = xc[i >> 1][i & 1];
    __builtin_amdgcn_wave_barrier();
    copy_stage_rows(stage, gdst, nvalid);
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------
// In-kernel statistics of the recorded samples (arp_hmc_io.stats) without HBM traffic per sample: the running mean m
// and centred second moment M2 of the samples recorded since the last fold live in LDS -- Welford updates, so no
// reference level is needed and nothing cancels --, one float4 {m.x, m.y, M2.x, M2.y} per county pair in the thread's
// own column (conflict free) and one float2 {m, M2} per top-level scalar in the chain's column (the K lanes of a chain
// write identical values).  When a batch ends, and when the launch does, the wave stages its rows of m and M2 like
// the memory image and folds them into the six planes (kernels.h: stats_fold_staged).
// ---------------------------------------------------------------------------
template <class T>
struct PkStats {
  static constexpr int kPair = 4 * T::NP * kBlock;              // floats: [k][thread] float4
  static constexpr int kTop = 2 * T::NG * (kBlock / T::K);      // floats: [i][chain] float2
  static constexpr int kFloats = kPair + kTop;
};

// sample number n (1-based, wave uniform) of the current accumulation
template <class T>
ARP_DEV void pk_stats_accumulate(float* s_stats, int n, const float (&xg)[T::NG], const v2f (&xc)[T::NP]) {
  constexpr int K = T::K, NP = T::NP, NG = T::NG;
  // the thread's column index is formed afresh here: kept live across the sampling loop it would cost registers the
  // chain kernels do not have
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  float4* a = reinterpret_cast<float4*>(s_stats) + tid;
  float2* t = reinterpret_cast<float2*>(s_stats + PkStats<T>::kPair) + tid / K;
  if (n == 1) {
#pragma unroll
    for (int k = 0; k < NP; ++k) a[k * kBlock] = float4{xc[k][0], xc[k][1], 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < NG; ++i) t[i * (kBlock / K)] = float2{xg[i], 0.0f};
  } else {
    const float rn = __builtin_amdgcn_rcpf((float)n);
    const v2f vrn = splat(rn);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const float4 v = a[k * kBlock];
      v2f m = {v.x, v.y}, M2 = {v.z, v.w};
      const v2f d = xc[k] - m;
      m = vfma(d, vrn, m);
      M2 = vfma(d, xc[k] - m, M2);
      a[k * kBlock] = float4{m[0], m[1], M2[0], M2[1]};
    }
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      float2 v = t[i * (kBlock / K)];
      const float d = xg[i] - v.x;
      v.x = fmaf(d, rn, v.x);
      v.y = fmaf(d, xg[i] - v.x, v.y);
      t[i * (kBlock / K)] = v;
    }
  }
}

// Fold the n samples the accumulators hold into the planes; `stage` is the wave's parked-state region (dead between
// transitions), which holds two row blocks (pk_save_wave_floats >= 2 stage_floats, checked in PkBlock).
template <class T>
ARP_DEV void pk_stats_fold(const T& M, float* stage, const float* s_stats, const HmcParams& P, long long cw0, int cl, int D,
                           int nvalid, int n, bool first, bool batch_end) {
  constexpr int K = T::K, NP = T::NP, NG = T::NG, NL = T::NL;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const float4* a = reinterpret_cast<const float4*>(s_stats) + tid;
  const float2* t = reinterpret_cast<const float2*>(s_stats + PkStats<T>::kPair) + tid / K;
  float* rm = stage + cl * D;
  float* rM = rm + stage_floats<T>();
  if (M.slot == 0) {
#pragma unroll
    for (int i = 0; i < NG; ++i) { const float2 v = t[i * (kBlock / K)]; rm[M.gg(i)] = v.x; rM[M.gg(i)] = v.y; }
  }
  float* em = rm + T::LBASE + M.slot;
  float* eM = rM + T::LBASE + M.slot;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const float4 v = a[k * kBlock];
    if (M.lvalid(2 * k)) { em[K * 2 * k] = v.x; eM[K * 2 * k] = v.z; }
    if (2 * k + 1 < NL && M.lvalid(2 * k + 1)) { em[K * (2 * k + 1)] = v.y; eM[K * (2 * k + 1)] = v.w; }
  }
  __builtin_amdgcn_wave_barrier();
  stats_fold_staged(stage, stage + stage_floats<T>(), P.stats + cw0 * D, (size_t)P.C * D, nvalid, (float)n, first, batch_end,
                    1.0f / (float)P.stats_batch);
  __builtin_amdgcn_wave_barrier();
}

// LDS of one 256-thread workgroup of the packed chain kernels

template <class T>
struct PkBlock {
  // LDS of one 256-thread workgroup: the parked start-of-trajectory state, which also serves as the waves' row staging
  // blocks (a row is staged only between transitions, when the parked state is dead), and the base step sizes
  static constexpr int kSave = pk_save_floats<T>();
  static constexpr int kStage = (kBlock / 64) * stage_floats<T>();
  static_assert(2 * stage_floats<T>() <= pk_save_wave_floats<T>(),
                "a wave's staging block -- two of them when statistics are folded -- aliases its own parked state");
  static constexpr int kEps = (T::LBASE + 2 * T::K * T::NP + 4 + 3) & ~3;   // every index a lane forms, zero beyond D
  // the statistics accumulators fit next to everything else with two workgroups on a CU (160 KB of LDS)
  static constexpr bool kStatsFit = (kSave + 2 * kEps + PkStats<T>::kFloats + lane_smem<T>::value) * 4 + 256 <= 80 * 1024;
};

// STATS: the instantiation that keeps the statistics accumulators in LDS (P.stats is set); its LDS footprint allows two
// workgroups per CU, so it is compiled for at most two waves per SIMD.  Without STATS a run that asks for statistics
// takes the plane-per-sample route of kernels.h (lane models whose accumulators do not fit: PkBlock::kStatsFit).
#ifndef ARP_PK_VIP_MINW2
#define ARP_PK_VIP_MINW2 (-1)      // experiment: kModeVIP (0) compiles the general-(a, b) form for two waves per SIMD
#endif
template <class T, int MODE, bool STATS = false>
__global__ __launch_bounds__(kBlock, (STATS || MODE == ARP_PK_VIP_MINW2) && T::MINW > 2 ? 2 : T::MINW) void pk_hmc_kernel(
    typename T::Args A, const float* __restrict__ av, const float* __restrict__ bv, HmcParams P) {
  constexpr int K = T::K, NP = T::NP, ND = T::ND, NG = T::NG;
  // chain of this lane (a launch holds fewer than 2^31 / K chains: 32-bit lane arithmetic)
  const RelayId rid = relay_begin(P);
  if (rid.seg < 0) return;                 // a hand-over timed out: leave the state as it is (kernels.h: relay_begin)
  const unsigned t = rid.bid * (unsigned)kBlock + threadIdx.x;
  const int slot = (int)(t % K);
  int c = (int)(t / K);
  const bool live = c < P.C;
  if (!live) c = P.C - 1;  // shadow lanes compute on the last chain but never store
  const int D = P.D;
  ARP_LANE_SMEM(T);
  T M;
  lane_tables(M, A, s_lane_tab);
  M.init(A, av, bv, slot);

  __shared__ float s_eps[PkBlock<T>::kEps];
  __shared__ __attribute__((aligned(16))) float s_save[PkBlock<T>::kSave];
  __shared__ __attribute__((aligned(16))) float s_stats[STATS ? PkStats<T>::kFloats : 4];
  int n_acc = 0;   // recorded samples in the LDS accumulators (STATS)
  // this wave's region: pair columns addressed as v2f[lane], the chains' float columns behind them
  float* wsave = s_save + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * pk_save_wave_floats<T>();
  float* save = wsave + 2 * (threadIdx.x & 63);
  float* stage = wsave;   // the wave's staging block aliases its own parked state (dead between transitions)
  // first chain of this wave: wave-uniform, kept in SGPRs so that row addresses are scalar arithmetic
  const long long cw0 = (long long)((rid.bid * (unsigned)kBlock + (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x & ~63)) / K);
  const int cl = (threadIdx.x & 63) / K;
  const int nvalid = (int)(P.C - cw0 < 64 / K ? (P.C - cw0 > 0 ? P.C - cw0 : 0) : 64 / K) * D;
  for (int d = threadIdx.x; d < PkBlock<T>::kEps; d += kBlock) s_eps[d] = d < D ? P.eps0[d] : 0.0f;
  __syncthreads();

  float qg[NG], gg_[NG]; v2f qc[NP], gc[NP];
  float lp;
  {
    float v[ND];
    load_row(M, P.q + (size_t)c * D, v);
    T::unpack(v, qg, qc);
    if (P.step_base == 0) {
      float pg[NG] = {}, eg[NG] = {}; v2f pc[NP], ec[NP]; float ke;
      M.template pass<MODE, 2>(qg, qc, pg, pc, eg, ec, gg_, gc, lp, ke);
    } else {
      load_row(M, P.grad + (size_t)c * D, v);
      T::unpack(v, gg_, gc);
      lp = P.logp[c];
    }
  }
  float kappa, esum, logavg;
  Rng rng;
  uint32_t* rs = P.rng + ((size_t)c * kRngSlots + slot) * 4;
  if (P.step_base == 0) {
    kappa = 1.0f; esum = 0.0f; logavg = 0.0f;
    rng = rng_seed(P.seed, (unsigned long long)(P.chain_offset + c), (uint32_t)slot, (uint32_t)K);
  } else {
    kappa = P.adapt[(size_t)c * 4 + 0]; esum = P.adapt[(size_t)c * 4 + 1]; logavg = P.adapt[(size_t)c * 4 + 2];
    rng = Rng{rs[0], rs[1]};
  }
  uint32_t nacc = (P.step_base == 0) ? 0u : P.accept_count[c];

  int next_rec = P.rec_step, rec_row = P.rec_row, bpos = P.stats_bpos;
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): nothing loaded is awaited inside the loop (kernels.h: hmc_kernel)
  for (int s = 0; s < P.n_steps; ++s) {
    bool acc;
    const float la = pk_transition<MODE>(M, rng, P.L, kappa, s_eps, qg, qc, gg_, gc, lp, acc, save);
    nacc += acc ? 1u : 0u;
    const long long n = P.step_base + s + 1;
    adapt_update(P, n, la, kappa, esum, logavg);

    if (s == next_rec && rec_row < P.n_samples) {
      const bool to_trace = P.trace && cw0 < P.trace_chains;
      if (to_trace || STATS || P.stats) {
        float xg[NG]; v2f xc[NP];
        if (P.trace_centered) {
          M.template to_centered<MODE>(qg, qc, xg, xc);
        } else {   // the state as it is
#pragma unroll
          for (int i = 0; i < NG; ++i) xg[i] = qg[i];
#pragma unroll
          for (int k = 0; k < NP; ++k) xc[k] = qc[k];
        }
        if (to_trace) {
          const int nv = min(nvalid, (int)(P.trace_chains - cw0) * D);
          pk_store_rows<true>(M, stage, P.trace + ((size_t)rec_row * P.trace_chains + cw0) * D, cl, D, nv, xg, xc);
        }
        if (STATS) {
          pk_stats_accumulate<T>(s_stats, ++n_acc, xg, xc);
          const bool bend = bpos + 1 == P.stats_batch;
          if (bend) {
            pk_stats_fold(M, stage, s_stats, P, cw0, cl, D, nvalid, n_acc, rec_row + 1 == n_acc, true);
            n_acc = 0;
          }
          bpos = bend ? 0 : bpos + 1;
        } else if (P.stats) {
          float x[ND];
          T::pack(xg, xc, x);
          stats_update_wave(M, stage, P, cw0, cl, D, nvalid, x, rec_row == 0, bpos + 1 == P.stats_batch);
          bpos = bpos + 1 == P.stats_batch ? 0 : bpos + 1;
        }
      }
      if (live && slot == 0) {
        // 32-bit lane offsets off wave-uniform bases, formed here: no per-lane pointer stays live across the sampling loop
        unsigned ci = (unsigned)c;
        asm volatile("" : "+v"(ci));
        if (P.trace_accept) (P.trace_accept + (size_t)rec_row * P.C)[ci] = acc ? 1 : 0;
        if (P.rec_accept) P.rec_accept[ci] += acc ? 1u : 0u;
      }
      next_rec += P.thin;
      rec_row += 1;
    }
  }

  // the launch ends inside a batch: what the accumulators hold goes into s1 / s2 now, the batch mean when the batch ends
  if (STATS && n_acc > 0) pk_stats_fold(M, stage, s_stats, P, cw0, cl, D, nvalid, n_acc, rec_row == n_acc, false);
  size_t c2 = (size_t)c;
  asm volatile("" : "+v"(c2));
  const long long cw2 = cw0;
  pk_store_rows(M, stage, P.q + cw2 * D, cl, D, nvalid, qg, qc);
  pk_store_rows(M, stage, P.grad + cw2 * D, cl, D, nvalid, gg_, gc);
  if (live) {
    uint32_t* rs2 = P.rng + ((size_t)c2 * kRngSlots + slot) * 4;
    rs2[0] = rng.x; rs2[1] = rng.c; rs2[2] = 0u; rs2[3] = 0u;
    if (slot == 0) {
      P.logp[c2] = lp;
      P.adapt[c2 * 4 + 0] = kappa; P.adapt[c2 * 4 + 1] = esum; P.adapt[c2 * 4 + 2] = logavg;
      P.accept_count[c2] = nacc;
    }
  }
  relay_end(P, rid);
}

// Interleaved sampling on the packed chain layer for lane models whose change of coordinates is NOT a unit-Jacobian
// shear (election: x = mua + sigma z rescales): interleaved.Interleaved.one_step as the reference runs it
// (interleaved.py:113-155) -- re-bootstrap logp / grad under parameterisation M0, one transition, the state through
// centred coordinates into M1, re-bootstrap, one transition, and back; each inner kernel keeps its own adaptation state.
// 2 (L + 1) gradient evaluations per step.  (Radon's shear carries the gradient instead: radon_fast.h.)
// T additionally provides from_centered<MODE>.
template <class T, int M0, int M1, bool STATS = false>
__global__ __launch_bounds__(kBlock, STATS && T::MINW > 2 ? 2 : T::MINW) void pk_interleaved_kernel(
    typename T::Args A, const float* __restrict__ av0, const float* __restrict__ bv0, HmcParams P) {
  constexpr int K = T::K, NP = T::NP, ND = T::ND, NG = T::NG;
  const RelayId rid = relay_begin(P);
  if (rid.seg < 0) return;                 // a hand-over timed out: leave the state as it is (kernels.h: relay_begin)
  const unsigned t = rid.bid * (unsigned)kBlock + threadIdx.x;
  const int slot = (int)(t % K);
  int c = (int)(t / K);
  const bool live = c < P.C;
  if (!live) c = P.C - 1;
  const int D = P.D;
  ARP_LANE_SMEM(T);
  T M;
  lane_tables(M, A, s_lane_tab);
  M.init(A, av0, bv0, slot);

  __shared__ float s_eps[2][PkBlock<T>::kEps];
  __shared__ __attribute__((aligned(16))) float s_save[PkBlock<T>::kSave];
  __shared__ __attribute__((aligned(16))) float s_stats[STATS ? PkStats<T>::kFloats : 4];
  int n_acc = 0;
  float* wsave = s_save + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * pk_save_wave_floats<T>();
  float* save = wsave + 2 * (threadIdx.x & 63);
  float* stage = wsave;
  const long long cw0 = (long long)((rid.bid * (unsigned)kBlock + (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x & ~63)) / K);
  const int cl = (threadIdx.x & 63) / K;
  const int nvalid = (int)(P.C - cw0 < 64 / K ? (P.C - cw0 > 0 ? P.C - cw0 : 0) : 64 / K) * D;
  for (int d = threadIdx.x; d < PkBlock<T>::kEps; d += kBlock) {
    s_eps[0][d] = d < D ? P.eps0[d] : 0.0f;
    s_eps[1][d] = d < D ? P.eps0_1[d] : 0.0f;
  }
  __syncthreads();

  float qg[NG], gg_[NG]; v2f qc[NP], gc[NP];
  {
    float v[ND];
    load_row(M, P.q + (size_t)c * D, v);
    T::unpack(v, qg, qc);
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
    float lp, ke;
    float xg[NG]; v2f xc[NP];
    {
      float pg[NG] = {}, eg[NG] = {}; v2f pc[NP], ec[NP];
      M.template pass<M0, 2>(qg, qc, pg, pc, eg, ec, gg_, gc, lp, ke);
    }
    float la = pk_transition<M0>(M, rng, P.L, kap[0], s_eps[0], qg, qc, gg_, gc, lp, acc0, save);
    nacc0 += acc0 ? 1u : 0u;
    adapt_update(P, n, la, kap[0], es[0], la_[0]);
    M.template to_centered<M0>(qg, qc, xg, xc);
    M.template from_centered<M1>(xg, xc, qg, qc);
    {
      float pg[NG] = {}, eg[NG] = {}; v2f pc[NP], ec[NP];
      M.template pass<M1, 2>(qg, qc, pg, pc, eg, ec, gg_, gc, lp, ke);
    }
    la = pk_transition<M1>(M, rng, P.L1, kap[1], s_eps[1], qg, qc, gg_, gc, lp, acc1, save);
    nacc1 += acc1 ? 1u : 0u;
    adapt_update(P, n, la, kap[1], es[1], la_[1]);
    M.template to_centered<M1>(qg, qc, xg, xc);
    M.template from_centered<M0>(xg, xc, qg, qc);

    if (s == next_rec && rec_row < P.n_samples) {
      // x holds the centred state, q the parameterisation-0 state the reference records
      const bool to_trace = P.trace && cw0 < P.trace_chains;
      const bool use_x = P.trace_centered;
      if (to_trace) {
        const int nv = min(nvalid, (int)(P.trace_chains - cw0) * D);
        float* dst = P.trace + ((size_t)rec_row * P.trace_chains + cw0) * D;
        if (use_x) pk_store_rows<true>(M, stage, dst, cl, D, nv, xg, xc); else pk_store_rows<true>(M, stage, dst, cl, D, nv, qg, qc);
      }
      if (STATS) {
        ++n_acc;
        if (use_x) pk_stats_accumulate<T>(s_stats, n_acc, xg, xc); else pk_stats_accumulate<T>(s_stats, n_acc, qg, qc);
        const bool bend = bpos + 1 == P.stats_batch;
        if (bend) {
          pk_stats_fold(M, stage, s_stats, P, cw0, cl, D, nvalid, n_acc, rec_row + 1 == n_acc, true);
          n_acc = 0;
        }
        bpos = bend ? 0 : bpos + 1;
      } else if (P.stats) {
        float x[ND];
        if (use_x) T::pack(xg, xc, x); else T::pack(qg, qc, x);
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
  if (live) {
    uint32_t* rs2 = P.rng + ((size_t)c2 * kRngSlots + slot) * 4;
    rs2[0] = rng.x; rs2[1] = rng.c; rs2[2] = 0u; rs2[3] = 0u;
    if (slot == 0) {
      P.adapt[c2 * 4 + 0] = kap[0]; P.adapt[c2 * 4 + 1] = es[0]; P.adapt[c2 * 4 + 2] = la_[0];
      P.adapt1[c2 * 4 + 0] = kap[1]; P.adapt1[c2 * 4 + 1] = es[1]; P.adapt1[c2 * 4 + 2] = la_[1];
      P.accept_count[c2] = nacc0; P.accept_count1[c2] = nacc1;
    }
  }
  relay_end(P, rid);
}

}  // namespace arp
