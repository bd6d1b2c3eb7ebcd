// Generic chain kernels, templated on a lane model (model_*.h).  A chain is
// spread over Lane::K consecutive lanes of a wave64; each lane keeps its slice of
// the state, momentum, gradient and sufficient statistics in VGPRs for the whole
// launch, so an HMC segment of n_steps transitions touches HBM only to load and
// store the chain state once and to append trace rows.
#pragma once
#include <type_traits>
#include "arp_device.h"
#include "../../include/autoreparam.h"

namespace arp {

constexpr int kRngSlots = 16;    // rng buffer stride per chain (max lanes per chain)
constexpr int kMaxD = 256;       // largest state dimension the chain kernels stage in LDS

struct HmcParams {
  int C, L, n_steps;
  long long step_base;
  long long chain_offset;
  unsigned long long seed;
  int adapt_kind, n_adapt;
  float adapt_target, adapt_rate;
  float adapt_log_target, adapt_inv_opr;   // log(target), 1/(1+rate): host-computed, live in SGPRs
  int n_burnin, thin, n_samples, trace_centered;
  int rec_step, rec_row;   // first in-launch step (0-based) that records, and its trace row
  int D;
  float* q; float* grad; float* logp; float* adapt;
  uint32_t* rng; uint32_t* accept_count;
  const float* eps0;
  float* trace; uint8_t* trace_accept;
  int trace_chains;                 // chains a trace row holds (== C unless only the first few are recorded)
  float* stats;                     // [6][C][D] streaming statistics of the recorded samples, or nullptr
  int stats_batch, stats_bpos;      // batch length; recorded samples already in the current batch at launch
  uint32_t* rec_accept;             // [C] accepted among the recorded transitions, or nullptr
  // interleaved kernel only: second transition kernel (parameterisation 1)
  int L1;
  float* adapt1; uint32_t* accept_count1; const float* eps0_1; uint8_t* trace_accept1; uint32_t* rec_accept1;
  // Relay (radon_fast.h: radon_interleaved_kernel): the launch's n_steps cut into `segs` segments of seg_len steps, a workgroup
  // per (segment, chain block); segment s of a block starts when seg_flags[block] == seg_epoch + s.  segs <= 1: one workgroup
  // per block takes all the steps.
  int segs, seg_len, seg_blocks;
  unsigned seg_epoch;
  unsigned* seg_flags;
  // seg_ctrl[0]: the launch's ticket counter (relay_begin), seg_ctrl[1]: set when a hand-over timed out (every waiter leaves);
  // seg_err_host: the handle's pinned host word the same event is reported through (arp_model_check), or nullptr
  unsigned* seg_ctrl;
  unsigned* seg_err_host;
  unsigned long long seg_timeout;   // ticks of the 100 MHz clock a segment waits for the one before it
  int seg_fault;                    // test hook (ARP_DEBUG=1 ARP_RELAY_FAULT=1): segments never raise their flag
};


// ---------------------------------------------------------------------------
// Relay.  The workgroups of a launch that are resident together (one or two per CU) are often exactly one or two rounds,
// and then every CU does the same work however fast it runs: the launch ends with the slowest.  With P.segs > 1 the grid is
// segs x seg_blocks workgroups, each taking seg_len of the launch's steps for one block of chains; the
// state travels from a block's segment to the next through the HBM rows a chunked run uses between launches (q, grad,
// logp, adapt, rng, counters, statistics), so the chain state is bit for bit the unsegmented launch's, and a CU that is free
// takes the next (segment, block) in line whoever ran the block before (profiles/r05_relay_segments.txt).
//
// Which (segment, block) a workgroup takes is decided by a TICKET it draws when it starts (atomicAdd on the launch's own
// counter), segment-major: ticket t = segment t / seg_blocks of block t % seg_blocks.  A workgroup behind segment 0 waits for
// the flag its block's previous segment raises after its stores -- that is ticket t - seg_blocks, and every ticket below t
// was drawn by a workgroup that started before this one: it is running or done, whatever order the dispatcher hands
// workgroups out in and whatever else shares the device.  The lowest unfinished ticket never waits on an unfinished one, so
// the launch always advances (no reliance on in-order dispatch: round 5's form indexed with blockIdx).
// relay_begin rewrites the kernel's OWN copy of the parameters to the segment's view (steps, first transition, recording
// schedule -- the arithmetic of arp_api.hip: fill_params); relay_end raises the flag.  A wait is bounded (seg_timeout):
// on expiry the launch is marked failed -- device word for the other waiters, pinned host word for arp_model_check -- and
// every workgroup still waiting leaves without touching the state (seg < 0: the kernels return at once).
// ---------------------------------------------------------------------------
struct RelayId { unsigned bid; int seg; };
ARP_DEV RelayId relay_begin(HmcParams& P) {
  RelayId r{blockIdx.x, 0};
  if (P.segs > 1) {
    __shared__ unsigned s_relay[2];      // ticket, abort
    if (threadIdx.x == 0) {
#ifdef ARP_EXP_RELAY_BLOCKIDX            // timing A/B only (tools/ab_lib.sh): round 5's index-ordered form
      s_relay[0] = blockIdx.x;
#else
      s_relay[0] = __hip_atomic_fetch_add(P.seg_ctrl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
      s_relay[1] = 0u;
    }
    __syncthreads();
    r.bid = (unsigned)__builtin_amdgcn_readfirstlane((int)s_relay[0]);     // uniform: the segment's view stays in SGPRs
    r.seg = (int)(r.bid / (unsigned)P.seg_blocks);
    r.bid -= (unsigned)r.seg * (unsigned)P.seg_blocks;
    const int start = r.seg * P.seg_len;
    int n = min(P.seg_len, P.n_steps - start);
    P.n_steps = n < 0 ? 0 : n;
    P.step_base += start;
    if (r.seg > 0) {
      long long first_n = 1 + (long long)P.n_burnin, r0 = 0;
      if (P.step_base + 1 > first_n) { r0 = (P.step_base + 1 - first_n + P.thin - 1) / P.thin; first_n += r0 * P.thin; }
      const long long s0 = first_n - P.step_base - 1;
      P.rec_step = s0 < P.n_steps ? (int)s0 : -1;
      P.rec_row = (int)(r0 < 0x7fffffff ? r0 : 0x7fffffff);
      P.stats_bpos = (int)(r0 % P.stats_batch);
      if (threadIdx.x == 0) {
        unsigned* const f = P.seg_flags + r.bid;
        const unsigned want = P.seg_epoch + (unsigned)r.seg;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
          __builtin_amdgcn_s_sleep(8);
          // a segment is milliseconds and the one waited for is resident, so this only expires if the device was taken away
          // for the whole time-out (a minute unless the host says otherwise) -- or in the fault-injection test
          bool failed = __hip_atomic_load(P.seg_ctrl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
          if (!failed && __builtin_amdgcn_s_memrealtime() - t0 > P.seg_timeout) {
            __hip_atomic_store(P.seg_ctrl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (P.seg_err_host) __hip_atomic_store(P.seg_err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            failed = true;
          }
          if (failed) { s_relay[1] = 1u; break; }
        }
      }
      __syncthreads();
      // EVERY wave takes the acquire itself, in front of its own loads (this CU's L1, stale L2 lines of other XCDs' data): the
      // memory model orders a wave's loads behind ITS OWN invalidate, not behind another wave's through a barrier
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      if (__builtin_amdgcn_readfirstlane((int)s_relay[1])) r.seg = -1;
    }
  }
  return r;
}
ARP_DEV void relay_end(const HmcParams& P, RelayId r) {
  if (P.segs > 1) {
    // EVERY storing wave releases its own stores (write-back of the XCD's L2 behind them, then vmcnt(0)) before the barrier.
    // Round 5 / early round 6 let the waves only drain (`s_waitcnt vmcnt(0)`) and thread 0 alone write L2 back after the
    // barrier: with the ticket order (consecutive segments of a block on arbitrary XCDs) one run in ten then handed a stale
    // cache line or two to the next segment (tests/diagnostics/relay_race_probe.py: 14 of 120 repetitions at 8 192 chains;
    // 0 of 120 with this form) -- a write-back issued by ANOTHER wave does not order behind this wave's stores.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0 && !P.seg_fault) {
      __hip_atomic_store(P.seg_flags + r.bid, P.seg_epoch + (unsigned)r.seg + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Lane models that keep chain-independent tables in LDS declare SMEM_FLOATS, stage_tables(Args, smem) (all threads of the
// workgroup) and bind_tables(smem); every kernel gives them the block before Lane::init.
template <class L, class = void> struct lane_smem { static constexpr int value = 0; };
template <class L> struct lane_smem<L, std::void_t<decltype(L::SMEM_FLOATS)>> { static constexpr int value = L::SMEM_FLOATS; };
template <class Lane>
ARP_DEV void lane_tables(Lane& M, const typename Lane::Args& A, float* smem) {
  if constexpr (lane_smem<Lane>::value > 0) {
    Lane::stage_tables(A, smem);
    __syncthreads();
    M.bind_tables(smem);
  }
}
#define ARP_LANE_SMEM(Lane) __shared__ __attribute__((aligned(16))) float s_lane_tab[lane_smem<Lane>::value > 0 ? lane_smem<Lane>::value : 4]

// ---------------------------------------------------------------------------
// Row I/O in the reference layout [C][D].  Lane `slot` of a chain owns the
// replicated globals (flattened index M.gg(i)) and NL sliced elements at
// flattened index M.lidx(i) = M.lbase(i) + Lane::loff(i) (valid when M.lvalid(i)):
// lbase is a per-lane base shared by all slices of a part, loff a compile-time
// offset, so one 64-bit base per row and part plus immediate offsets are enough.
// In the slot + K*i models only the LAST slice of a lane can be padding (the host
// picks NL == ceil(groups / K) exactly), so lvalid folds to `true` for the others.
// ---------------------------------------------------------------------------
template <class Lane>
ARP_DEV void load_row(const Lane& M, const float* __restrict__ row, float (&v)[Lane::ND]) {
#pragma unroll
  for (int i = 0; i < Lane::NG; ++i) v[i] = row[M.gg(i)];
#pragma unroll
  for (int i = 0; i < Lane::NL; ++i) v[Lane::NG + i] = M.lvalid(i) ? (row + M.lbase(i))[Lane::loff(i)] : 0.0f;
}
template <class Lane>
ARP_DEV void store_row(const Lane& M, float* __restrict__ row, const float (&v)[Lane::ND], bool live) {
  if (live && M.slot == 0) {
#pragma unroll
    for (int i = 0; i < Lane::NG; ++i) row[M.gg(i)] = v[i];
  }
#pragma unroll
  for (int i = 0; i < Lane::NL; ++i)
    if (live && M.lvalid(i)) (row + M.lbase(i))[Lane::loff(i)] = v[Lane::NG + i];
}

// ---------------------------------------------------------------------------
// Coalesced row stores.  The 64/K chains of a wave are consecutive, so their rows
// form ONE contiguous block of (64/K)*D floats in a [C][D] array.  The lanes drop
// their slices into the wave's LDS staging area (laid out like the memory image)
// and the wave then streams the block out with 16-byte-per-lane stores, instead
// of D/K dword stores per lane that each touch 64/K different cache lines.
// LDS operations of one wave execute in order, so no barrier is needed.
// ---------------------------------------------------------------------------
template <class Lane>
constexpr int stage_floats() { return (((64 / Lane::K) * Lane::DCAP) + 3) & ~3; }
// The chain kernels' per-wave row-staging blocks.  A lane model whose gradient owns a large LDS work area that is idle
// between gradients (German credit's tile buffers) lends it instead: STAGE_ALIAS = true, stage_mem() (at least
// kStageCap floats), and its grad() must open with a workgroup barrier before it writes the area.
template <class L, class = void> struct lane_stage_alias { static constexpr bool value = false; };
template <class L> struct lane_stage_alias<L, std::void_t<decltype(L::STAGE_ALIAS)>> { static constexpr bool value = L::STAGE_ALIAS; };
template <class Lane>
ARP_DEV float* lane_stage(float* own) {
  if constexpr (lane_stage_alias<Lane>::value) {
    static_assert((kBlock / 64) * stage_floats<Lane>() <= Lane::kStageCap, "the lent area holds every wave's staging block");
    return Lane::stage_mem();
  } else {
    return own;
  }
}
#define ARP_STAGE_SMEM(Lane)                                                                                              \
  __shared__ __attribute__((aligned(16))) float s_stage_own[lane_stage_alias<Lane>::value ? 4 : (kBlock / 64) * stage_floats<Lane>()]; \
  float* const s_stage = lane_stage<Lane>(s_stage_own)

template <bool STREAM = false, class Lane>
ARP_DEV void store_row_wave(const Lane& M, float* stage, float* gdst, int cl, int D, int nvalid,
                            const float (&v)[Lane::ND]) {
  float* row = stage + cl * D;
  if (M.slot == 0) {
#pragma unroll
    for (int i = 0; i < Lane::NG; ++i) row[M.gg(i)] = v[i];
  }
#pragma unroll
  for (int i = 0; i < Lane::NL; ++i)
    if (M.lvalid(i)) (row + M.lbase(i))[Lane::loff(i)] = v[Lane::NG + i];
  __builtin_amdgcn_wave_barrier();
  const int lane = threadIdx.x & 63;
  if ((reinterpret_cast<uintptr_t>(gdst) & 15) == 0) {
    // four 1 KiB slices at a time, their LDS reads all in flight before the first store: one read -> wait -> store per
    // slice is an LDS round trip per KiB in a row (pk_chain.h: pk_store_rows has the measurement)
    for (int k0 = lane * 4; k0 < nvalid; k0 += 1024) {
      v4f_nt t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + 256 * u;
        t[u] = *reinterpret_cast<const v4f_nt*>(stage + (k < nvalid ? k : 0));      // (past the block: any word of it, never stored)
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + 256 * u;
        if (k + 3 < nvalid) {
          store_v4<STREAM>(reinterpret_cast<v4f_nt*>(gdst + k), t[u]);
        } else if (k < nvalid) {
          gdst[k] = t[u][0];
          if (k + 1 < nvalid) gdst[k + 1] = t[u][1];
          if (k + 2 < nvalid) gdst[k + 2] = t[u][2];
        }
      }
    }
  } else {
    for (int k = lane; k < nvalid; k += 64) gdst[k] = stage[k];
  }
  __builtin_amdgcn_wave_barrier();
}

// Streaming statistics of a recorded sample (arp_hmc_io.stats, [6][C][D]): the wave stages its chains'
// rows in LDS as store_row_wave does (stage_row_wave) and then walks the six component planes, whose
// blocks for the wave's chains have the same [chains][D] shape, with 16-byte accesses.  x - ref keeps
// the sums well conditioned in float32 (ref = the first recorded sample of that element).
// The walk is a real function call: inlined, its temporaries push the chain kernels (which sit at
// the 256-VGPR edge) into spilling inside the sampling loop even when no statistics are requested.
__device__ __noinline__ void stats_update_staged(const float* stage, float* g, size_t comp, int nvalid, bool first,
                                                 bool batch_end, float invb) {
  const int lane = threadIdx.x & 63;
  // per sample only ref (read), s1 and s2 (read-modify-write) are touched
  auto upd = [&](float x, float& ref, float& s1, float& s2) {
    ref = first ? x : ref;
    const float dx = x - ref;
    s1 += dx; s2 = fmaf(dx, dx, s2);
  };
  const bool aligned = ((reinterpret_cast<uintptr_t>(g) | (comp * sizeof(float))) & 15) == 0;
  // One 16-byte slice of the planes: loads (`ld`), then arithmetic + stores (`st`).  Slices go two at a time with the
  // plane loads of both issued before the first use: the planes come from the Infinity Cache / HBM, and one round trip
  // per slice (five per sample at D = 71, each also waiting out the previous slice's stores: vmcnt is in order) was
  // most of what the statistics cost.
  struct Slice { float4 x, ref, s1, s2; };
  auto ld = [&](int k, Slice& S) {
    S.x = *reinterpret_cast<const float4*>(stage + k);
    S.ref = *reinterpret_cast<float4*>(g + k); S.s1 = *reinterpret_cast<float4*>(g + comp + k);
    S.s2 = *reinterpret_cast<float4*>(g + 2 * comp + k);
  };
  auto st = [&](int k, Slice& S) {
    upd(S.x.x, S.ref.x, S.s1.x, S.s2.x);
    upd(S.x.y, S.ref.y, S.s1.y, S.s2.y);
    upd(S.x.z, S.ref.z, S.s1.z, S.s2.z);
    upd(S.x.w, S.ref.w, S.s1.w, S.s2.w);
    if (first) *reinterpret_cast<float4*>(g + k) = S.ref;
    *reinterpret_cast<float4*>(g + comp + k) = S.s1; *reinterpret_cast<float4*>(g + 2 * comp + k) = S.s2;
  };
  // unaligned planes or the ragged tail of the wave's block: element by element, nothing past it is touched
  auto tail = [&](int k) {
    for (int j = k; j < nvalid && j < k + 4; ++j) {
      float ref = g[j], s1 = g[comp + j], s2 = g[2 * comp + j];
      upd(stage[j], ref, s1, s2);
      if (first) g[j] = ref;
      g[comp + j] = s1; g[2 * comp + j] = s2;
    }
  };
  for (int k = lane * 4; k < nvalid; k += 512) {
    const int k2 = k + 256;
    if (aligned && k2 + 3 < nvalid) {          // both slices whole
      Slice A, B;
      ld(k, A); ld(k2, B);
      st(k, A); st(k2, B);
    } else {
      if (aligned && k + 3 < nvalid) { Slice A; ld(k, A); st(k, A); } else tail(k);
      if (k2 < nvalid) {
        if (aligned && k2 + 3 < nvalid) { Slice B; ld(k2, B); st(k2, B); } else tail(k2);
      }
    }
  }
  // batch ends only (one sample in `stats_batch`): `cur` holds s1 as it was when the batch began, batch sum = s1 - cur.
  // This lane reads back the s1 it has just written (same addresses, same lane).
  if (batch_end) {
    for (int k = lane * 4; k < nvalid; k += 256) {
      for (int j = k; j < nvalid && j < k + 4; ++j) {
        const float s1 = g[comp + j], cur = g[3 * comp + j];
        const float bm = (s1 - cur) * invb;
        g[4 * comp + j] += bm;
        g[5 * comp + j] = fmaf(bm, bm, g[5 * comp + j]);
        g[3 * comp + j] = s1;
      }
    }
  }
}

// The packed chain kernels (pk_chain.h) keep the running mean m and the centred second moment M2 of the samples of the
// current batch in LDS (Welford updates, no HBM traffic per sample) and FOLD them into the six planes when a batch or the
// launch ends: `sm` / `sM2` are the wave's rows of m and M2 staged like the memory image, n the samples they hold.
//   ref (plane 0) = m of the first fold (any level near the samples serves; the planes are zero before it),
//   d = m - ref,  s1 += n d,  s2 += M2 + n d^2;  at a batch end bm = (s1 - cur) / batch, sb1 += bm, sb2 += bm^2, cur = s1
// -- the same six sums as stats_update_staged up to rounding, with 11 plane slices moved per batch (or launch)
// instead of 5 per sample.
__device__ __noinline__ void stats_fold_staged(const float* sm, const float* sM2, float* g, size_t comp, int nvalid,
                                               float n, bool first, bool batch_end, float invb) {
  const int lane = threadIdx.x & 63;
  auto fold = [&](float m, float M2, float& ref, float& s1, float& s2) {
    ref = first ? m : ref;
    const float d = m - ref, nd = n * d;
    s1 += nd; s2 += fmaf(nd, d, M2);
  };
  auto bend = [&](float s1, float& cur, float& b1, float& b2) {
    const float bm = (s1 - cur) * invb;
    b1 += bm; b2 = fmaf(bm, bm, b2); cur = s1;
  };
  const bool aligned = ((reinterpret_cast<uintptr_t>(g) | (comp * sizeof(float))) & 15) == 0;
  for (int k = lane * 4; k < nvalid; k += 256) {
    if (aligned && k + 3 < nvalid) {
      const float4 m = *reinterpret_cast<const float4*>(sm + k), M2 = *reinterpret_cast<const float4*>(sM2 + k);
      float4 ref = first ? m : *reinterpret_cast<float4*>(g + k);
      float4 s1 = *reinterpret_cast<float4*>(g + comp + k), s2 = *reinterpret_cast<float4*>(g + 2 * comp + k);
      float4 cur, b1, b2;
      if (batch_end) {
        cur = *reinterpret_cast<float4*>(g + 3 * comp + k); b1 = *reinterpret_cast<float4*>(g + 4 * comp + k);
        b2 = *reinterpret_cast<float4*>(g + 5 * comp + k);
      }
      fold(m.x, M2.x, ref.x, s1.x, s2.x); fold(m.y, M2.y, ref.y, s1.y, s2.y);
      fold(m.z, M2.z, ref.z, s1.z, s2.z); fold(m.w, M2.w, ref.w, s1.w, s2.w);
      if (first) *reinterpret_cast<float4*>(g + k) = ref;
      *reinterpret_cast<float4*>(g + comp + k) = s1; *reinterpret_cast<float4*>(g + 2 * comp + k) = s2;
      if (batch_end) {
        bend(s1.x, cur.x, b1.x, b2.x); bend(s1.y, cur.y, b1.y, b2.y);
        bend(s1.z, cur.z, b1.z, b2.z); bend(s1.w, cur.w, b1.w, b2.w);
        *reinterpret_cast<float4*>(g + 3 * comp + k) = cur; *reinterpret_cast<float4*>(g + 4 * comp + k) = b1;
        *reinterpret_cast<float4*>(g + 5 * comp + k) = b2;
      }
    } else {   // unaligned planes or the ragged tail of the wave's block: element by element, nothing past it is touched
      for (int j = k; j < nvalid && j < k + 4; ++j) {
        float ref = g[j], s1 = g[comp + j], s2 = g[2 * comp + j];
        fold(sm[j], sM2[j], ref, s1, s2);
        if (first) g[j] = ref;
        g[comp + j] = s1; g[2 * comp + j] = s2;
        if (batch_end) {
          float cur = g[3 * comp + j], b1 = g[4 * comp + j], b2 = g[5 * comp + j];
          bend(s1, cur, b1, b2);
          g[3 * comp + j] = cur; g[4 * comp + j] = b1; g[5 * comp + j] = b2;
        }
      }
    }
  }
}

template <class Lane>
ARP_DEV void stats_update_wave(const Lane& M, float* stage, const HmcParams& P, long long cw0, int cl, int D,
                               int nvalid, const float (&v)[Lane::ND], bool first, bool batch_end) {
  float* row = stage + cl * D;
  if (M.slot == 0) {
#pragma unroll
    for (int i = 0; i < Lane::NG; ++i) row[M.gg(i)] = v[i];
  }
#pragma unroll
  for (int i = 0; i < Lane::NL; ++i)
    if (M.lvalid(i)) (row + M.lbase(i))[Lane::loff(i)] = v[Lane::NG + i];
  __builtin_amdgcn_wave_barrier();
  stats_update_staged(stage, P.stats + cw0 * D, (size_t)P.C * D, nvalid, first, batch_end, 1.0f / (float)P.stats_batch);
  __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------
// Parameterisation modes.  MODE 0 evaluates the general VIP form with per-element
// (a,b) held in registers; lane models that set HAS_MODES also provide
// compile-time specialisations for the two parameterisations every run uses,
// MODE 1 = centred (a=b=1) and MODE 2 = non-centred (a=b=0): fewer VALU ops per
// group and no (a,b) registers.  MODE 3 (lane models with HAS_MODE_B1) is "a free,
// b = 1": what the reference's tied cVIP / dVIP runs execute.
// ---------------------------------------------------------------------------
constexpr int kModeVIP = 0, kModeCP = 1, kModeNCP = 2, kModeB1 = 3;

template <int MODE, bool LOGP, class Lane>
ARP_DEV float lane_grad(const Lane& M, const float (&q)[Lane::ND], float (&g)[Lane::ND]) {
  if constexpr (MODE != kModeVIP && Lane::HAS_MODES) return M.template grad_m<LOGP, MODE>(q, g);
  else return M.template grad<LOGP>(q, g);
}
// One interior leapfrog step: gradient at q, full kick, drift -- q and p updated in place.
// Lane models with HAS_FUSED do it in a single pass over their groups (the gradient of a
// group is consumed as soon as it is formed, so no gradient array stays live in the loop).
template <int MODE, class Lane>
ARP_DEV void lane_kick_drift(const Lane& M, float (&q)[Lane::ND], float (&p)[Lane::ND],
                             const float (&eps)[Lane::ND]) {
  if constexpr (Lane::HAS_FUSED) {
    M.template kick_drift<MODE>(q, p, eps);
  } else {
    float g[Lane::ND];
    lane_grad<MODE, false>(M, q, g);
#pragma unroll
    for (int i = 0; i < Lane::ND; ++i) {
      p[i] = fmaf(eps[i], g[i], p[i]);
      q[i] = fmaf(eps[i], p[i], q[i]);
    }
  }
}
template <int MODE, class Lane>
ARP_DEV void lane_to_centered(const Lane& M, const float (&q)[Lane::ND], float (&x)[Lane::ND]) {
  if constexpr (MODE != kModeVIP && Lane::HAS_MODES) M.template to_centered_m<MODE>(q, x);
  else M.to_centered(q, x);
}
template <int MODE, class Lane>
ARP_DEV void lane_from_centered(const Lane& M, const float (&x)[Lane::ND], float (&q)[Lane::ND]) {
  if constexpr (MODE != kModeVIP && Lane::HAS_MODES) M.template from_centered_m<MODE>(x, q);
  else M.from_centered(x, q);
}

// Make `M` evaluate the parameterisation the interleaved kernel switches to: the general form
// reloads (a, b); a compile-time mode has nothing to reload unless the lane model derives
// run-time state from (a, b) (election's top-level prior scales), which set_mode<> rebuilds.
template <int MODE, class Lane>
ARP_DEV void switch_param(Lane& M, const float* av, const float* bv) {
  if constexpr (MODE == kModeVIP || !Lane::HAS_MODES) M.set_param(av, bv);
  else if constexpr (Lane::HAS_MODE_STATE) M.template set_mode<MODE>();
}

// ---------------------------------------------------------------------------
// logp + grad for a batch of states (test hook and bootstrap of the cached
// gradient; reference: vectorized target + tf.gradients, inference.py:172-195)
// ---------------------------------------------------------------------------
template <class Lane>
__global__ __launch_bounds__(kBlock) void logp_grad_kernel(
    typename Lane::Args A, const float* __restrict__ av, const float* __restrict__ bv,
    const float* __restrict__ x, int C, int D, float* __restrict__ logp, float* __restrict__ grad) {
  constexpr int K = Lane::K, ND = Lane::ND;
  long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
  int slot = (int)(t % K);
  long long c = t / K;
  bool live = c < C;
  long long cc = live ? c : (long long)C - 1;  // dead lanes shadow the last chain (keeps DPP groups uniform)
  ARP_LANE_SMEM(Lane);
  Lane M;
  lane_tables(M, A, s_lane_tab);
  M.init(A, av, bv, slot);
  float q[ND], g[ND];
  load_row(M, x + cc * D, q);
  float lp = M.template grad<true>(q, g);
  store_row(M, grad + cc * D, g, live);
  if (live && slot == 0) logp[c] = lp;
}

// dir 0: reparameterised -> centred; dir 1: centred -> reparameterised
template <class Lane>
__global__ __launch_bounds__(kBlock) void transform_kernel(
    typename Lane::Args A, const float* __restrict__ av, const float* __restrict__ bv,
    int dir, const float* __restrict__ in, int C, int D, float* __restrict__ out) {
  constexpr int K = Lane::K, ND = Lane::ND;
  long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
  int slot = (int)(t % K);
  long long c = t / K;
  bool live = c < C;
  long long cc = live ? c : (long long)C - 1;
  ARP_LANE_SMEM(Lane);
  Lane M;
  lane_tables(M, A, s_lane_tab);
  M.init(A, av, bv, slot);
  float a[ND], b[ND];
  load_row(M, in + cc * D, a);
  if (dir == 0) M.to_centered(a, b); else M.from_centered(a, b);
  store_row(M, out + cc * D, b, live);
}

// ---------------------------------------------------------------------------
// One HMC transition on the lane slice (mcmc.HamiltonianMonteCarlo.one_step as
// wired at inference.py:218-222): momentum draw, L leapfrog steps with the two
// half kicks of consecutive steps merged, Metropolis test.  Returns the log
// acceptance ratio.  The trajectory is integrated in place in q/g; the starting
// state is parked in the lane's LDS column (`save[i * kBlock]`, conflict free) and
// only the rejecting lanes read it back, so no second copy of the state lives in
// VGPRs and acceptance needs no per-element select.
// ---------------------------------------------------------------------------
// random-stream layout of a lane model's momentum draw (0 unless the model declares MOM_SPEC)
template <class L, class = void> struct lane_mom_spec { static constexpr int value = 0; };
template <class L> struct lane_mom_spec<L, std::void_t<decltype(L::MOM_SPEC)>> { static constexpr int value = L::MOM_SPEC; };

template <class Lane, int MODE = kModeVIP>
ARP_DEV float hmc_transition(const Lane& M, Rng& rng, int L, const float (&eps)[Lane::ND],
                             float (&q)[Lane::ND], float (&g)[Lane::ND], float& lp,
                             bool& accepted, float* save) {
  constexpr int K = Lane::K, ND = Lane::ND, NG = Lane::NG;
  float p[ND];
  float u;
  if constexpr (lane_mom_spec<Lane>::value == 1) {
    // stream layout 1 (radon): a slot draws one normal per slice it owns, then ceil(NG / K) more, of which
    // extra x of slot s is the momentum of top-level scalar s + K*x -- no slot draws a normal it discards
    constexpr int NLs = ND - NG, NE = (NG + K - 1) / K, NN = NLs + NE;
    float z[NN + 1];
#pragma unroll
    for (int i = 0; i < NN; i += 2) {
      uint32_t w0 = rng_next(rng), w1 = rng_next(rng);
      normal_pair(w0, w1, z[i], z[i + 1]);
    }
#pragma unroll
    for (int i = 0; i < NLs; ++i) p[NG + i] = z[i];
    u = u01_open0(rng_next(rng));
    u = group_bcast0<K>(u, M.slot);
#pragma unroll
    for (int i = 0; i < NG; ++i) p[i] = group_sum<K>(M.slot == i % K ? z[NLs + i / K] : 0.0f);
  } else {
    // layout 0: every lane draws ND = NG + ceil(groups / K) normals from its own stream (the host picks
    // NL == ceil(groups / K)); the replicated globals take slot 0's draw, padding slots get none
#pragma unroll
    for (int i = 0; i < ND; i += 2) {
      float z0, z1;
      uint32_t w0 = rng_next(rng), w1 = rng_next(rng);
      normal_pair(w0, w1, z0, z1);
      p[i] = z0;
      if (i + 1 < ND) p[i + 1] = z1;
    }
    u = u01_open0(rng_next(rng));
    u = group_bcast0<K>(u, M.slot);
#pragma unroll
    for (int i = 0; i < NG; ++i) p[i] = group_bcast0<K>(p[i], M.slot);
  }
  float ke0 = 0.0f, keg0 = 0.0f;
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    if (i < NG) {
      keg0 = fmaf(p[i], p[i], keg0);
    } else {
      p[i] = M.lvalid(i - NG) ? p[i] : 0.0f;
      ke0 = fmaf(p[i], p[i], ke0);
    }
  }
  ke0 = 0.5f * (group_sum<K>(ke0) + keg0);

#pragma unroll
  for (int i = 0; i < ND; ++i) {
    save[i * kBlock] = q[i];
    save[(ND + i) * kBlock] = g[i];
    p[i] = fmaf(0.5f * eps[i], g[i], p[i]);
  }
  // first drift, L-1 interior steps (gradient, full kick, drift), then the closing half kick
#pragma unroll
  for (int i = 0; i < ND; ++i) q[i] = fmaf(eps[i], p[i], q[i]);
  for (int l = 1; l < L; ++l) lane_kick_drift<MODE>(M, q, p, eps);
  const float lp1 = lane_grad<MODE, true>(M, q, g);
#pragma unroll
  for (int i = 0; i < ND; ++i) p[i] = fmaf(0.5f * eps[i], g[i], p[i]);
  float ke1 = 0.0f, keg1 = 0.0f;
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    if (i < NG) keg1 = fmaf(p[i], p[i], keg1); else ke1 = fmaf(p[i], p[i], ke1);
  }
  ke1 = 0.5f * (group_sum<K>(ke1) + keg1);

  // log accept ratio; any non-finite energy error rejects (TFP safe_sum semantics)
  float la = (lp1 - lp) + (ke0 - ke1);
  if (!(fabsf(la) <= 3.0e38f)) la = -INFINITY;
  accepted = fast_log(u) < la;
  if (!accepted) {
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      q[i] = save[i * kBlock];
      g[i] = save[(ND + i) * kBlock];
    }
  }
  lp = accepted ? lp1 : lp;
  return la;
}

// Step-size multiplier update after a transition whose 1-based index is n.
// kappa scales the per-element base step eps0 (all elements of a chain see the
// same acceptance probability, so the reference's per-element adaptation state
// collapses to one scalar per chain).
ARP_DEV void adapt_update(const HmcParams& P, long long n, float la,
                          float& kappa, float& esum, float& logavg) {
  const int kind = P.adapt_kind, n_adapt = P.n_adapt;
  const float target = P.adapt_target, rate = P.adapt_rate;
  if (kind == ARP_ADAPT_NONE) return;
  if (kind == ARP_ADAPT_SIMPLE) {   // log(target) < 0, so comparing la itself equals comparing min(la, 0)
    if (n <= n_adapt) kappa *= la > P.adapt_log_target ? 1.0f + rate : P.adapt_inv_opr;
    return;
  }
  float lacc = fminf(la, 0.0f);
  if (kind == ARP_ADAPT_DUAL) {
    if (n <= n_adapt) {
      float t = (float)n;
      esum += target - fast_exp(lacc);
      // log(10 eps0) - esum sqrt(t) / ((t + 10) * 0.05), relative to log eps0
      float ls = 2.302585092994046f - esum * __builtin_amdgcn_sqrtf(t) * __builtin_amdgcn_rcpf((t + 10.0f) * 0.05f);
      float eta = __builtin_amdgcn_exp2f(-0.75f * __builtin_amdgcn_logf(t));
      logavg = eta * ls + (1.0f - eta) * logavg;
      kappa = fast_exp(ls);
    } else if (n_adapt > 0) {
      kappa = fast_exp(logavg);
    }
  } else {  // ARP_ADAPT_SIMPLE
    if (n <= n_adapt) {
      float opr = 1.0f + rate;
      kappa *= lacc > P.adapt_log_target ? opr : P.adapt_inv_opr;
    }
  }
}

template <class Lane, int MODE = kModeVIP>
__global__ __launch_bounds__(kBlock, Lane::MINW) void hmc_kernel(
    typename Lane::Args A, const float* __restrict__ av, const float* __restrict__ bv, HmcParams P) {
  constexpr int K = Lane::K, ND = Lane::ND;
  const RelayId rid = relay_begin(P);
  if (rid.seg < 0) return;                 // a hand-over timed out: leave the state as it is (kernels.h: relay_begin)
  long long t = (long long)rid.bid * kBlock + threadIdx.x;
  const int slot = (int)(t % K);
  long long c = t / K;
  const bool live = c < P.C;
  if (!live) c = P.C - 1;  // shadow lanes compute on the last chain but never store
  const int D = P.D;
  ARP_LANE_SMEM(Lane);
  Lane M;
  lane_tables(M, A, s_lane_tab);
  M.init(A, av, bv, slot);

  // base step sizes stay in LDS for the whole launch: re-reading them from global
  // memory every transition would queue behind the trace stores (vmcnt is in order)
  __shared__ float s_eps[kMaxD];
  __shared__ float s_save[2 * ND * kBlock];   // parked start-of-trajectory state, one column per lane
  ARP_STAGE_SMEM(Lane);
  float* save = s_save + threadIdx.x;
  float* stage = s_stage + (threadIdx.x >> 6) * stage_floats<Lane>();
  // first chain of this wave, this lane's chain within the wave, floats of the wave's live chains
  const long long cw0 = ((long long)rid.bid * kBlock + (threadIdx.x & ~63)) / K;
  const int cl = (threadIdx.x & 63) / K;
  const int nvalid = (int)(P.C - cw0 < 64 / K ? (P.C - cw0 > 0 ? P.C - cw0 : 0) : 64 / K) * D;
  for (int d = threadIdx.x; d < D; d += kBlock) s_eps[d] = P.eps0[d];
  __syncthreads();

  float q[ND], g[ND], eps[ND];
  float* qrow = P.q + c * D;
  float* grow = P.grad + c * D;
  load_row(M, qrow, q);
  float lp;
  if (P.step_base == 0) {
    lp = lane_grad<MODE, true>(M, q, g);
  } else {
    load_row(M, grow, g);
    lp = P.logp[c];
  }
  float kappa, esum, logavg;
  Rng rng;
  uint32_t* rs = P.rng + ((size_t)c * kRngSlots + slot) * 4;
  if (P.step_base == 0) {
    kappa = 1.0f; esum = 0.0f; logavg = 0.0f;
    rng = rng_seed(P.seed, (unsigned long long)(P.chain_offset + c), (uint32_t)slot, (uint32_t)K);
  } else {
    kappa = P.adapt[c * 4 + 0]; esum = P.adapt[c * 4 + 1]; logavg = P.adapt[c * 4 + 2];
    rng = Rng{rs[0], rs[1]};
  }
  uint32_t nacc = (P.step_base == 0) ? 0u : P.accept_count[c];

  int next_rec = P.rec_step, rec_row = P.rec_row, bpos = P.stats_bpos;
  // everything loaded so far has landed: no load result is awaited inside the loop, so the
  // in-order vmcnt never makes a wave wait for its own trace stores
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#ifdef ARP_EXP_TIMING
  if (threadIdx.x == 0) for (int k = 0; k < 16; ++k) exp_t()[k] = 0;
  ARP_T0(tk);
#endif
  for (int s = 0; s < P.n_steps; ++s) {
    load_row(M, s_eps, eps);
#pragma unroll
    for (int i = 0; i < ND; ++i) eps[i] *= kappa;
    bool acc;
    float la = hmc_transition<Lane, MODE>(M, rng, P.L, eps, q, g, lp, acc, save);
    nacc += acc ? 1u : 0u;
    const long long n = P.step_base + s + 1;
    adapt_update(P, n, la, kappa, esum, logavg);

    // sample_chain schedule: result r is the state after transition 1 + burnin + r*thin
    if (s == next_rec && rec_row < P.n_samples) {
      const bool to_trace = P.trace && cw0 < P.trace_chains;
      auto record = [&](const float (&v)[ND]) {
        if (to_trace) {
          const int nv = min(nvalid, (int)(P.trace_chains - cw0) * D);
          store_row_wave<true>(M, stage, P.trace + ((size_t)rec_row * P.trace_chains + cw0) * D, cl, D, nv, v);
        }
        if (P.stats) {
          stats_update_wave(M, stage, P, cw0, cl, D, nvalid, v, rec_row == 0, bpos + 1 == P.stats_batch);
          bpos = bpos + 1 == P.stats_batch ? 0 : bpos + 1;
        }
      };
      if (to_trace || P.stats) {
        if (P.trace_centered) {
          float x[ND];
          lane_to_centered<MODE>(M, q, x);
          record(x);
        } else {
          record(q);
        }
      }
      if (live && slot == 0) {
        if (P.trace_accept) P.trace_accept[(size_t)rec_row * P.C + c] = acc ? 1 : 0;
        if (P.rec_accept) P.rec_accept[c] += acc ? 1u : 0u;
      }
      next_rec += P.thin;
      rec_row += 1;
    }
  }

#ifdef ARP_EXP_TIMING
  ARP_T(8, tk);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    unsigned long long* t = exp_t();
    printf("TIMING total %llu | pre %llu bar %llu start %llu tiles %llu vx %llu post %llu | lik %llu | 9:%llu 10:%llu 11:%llu 12:%llu 13:%llu 14:%llu\n", t[8], t[0], t[1], t[2], t[3], t[4], t[5], t[7], t[9], t[10], t[11], t[12], t[13], t[14]);
  }
#endif
  // recompute the per-lane row addresses here instead of keeping 64-bit pointers alive (in VGPR
  // pairs) across the whole sampling loop
  long long c2 = c;
  asm volatile("" : "+v"(c2));
  long long cw2 = cw0;
  asm volatile("" : "+v"(cw2));
  store_row_wave(M, stage, P.q + cw2 * D, cl, D, nvalid, q);
  store_row_wave(M, stage, P.grad + cw2 * D, cl, D, nvalid, g);
  uint32_t* rs2 = P.rng + ((size_t)c2 * kRngSlots + slot) * 4;
  if (live) {
    rs = rs2;
    rs[0] = rng.x; rs[1] = rng.c; rs[2] = 0u; rs[3] = 0u;
    if (slot == 0) {
      P.logp[c2] = lp;
      P.adapt[c2 * 4 + 0] = kappa; P.adapt[c2 * 4 + 1] = esum; P.adapt[c2 * 4 + 2] = logavg;
      P.accept_count[c2] = nacc;
    }
  }
  relay_end(P, rid);
}

// ---------------------------------------------------------------------------
// Interleaved CP/NCP sampling (interleaved.Interleaved.one_step,
// interleaved.py:113-155): per step
//   re-bootstrap logp/grad at the current state under parameterisation 0,
//   one HMC transition there, map the state to parameterisation 1
//   (to_ncp = from_centred_1 o to_centred_0), re-bootstrap, one HMC transition,
//   map back.  Each inner kernel keeps its own step-size adaptation state, which
//   survives the re-bootstrap because it lives outside the HMC results
//   (inference.py:288-306).  2*(L+1) gradient evaluations per step.
// ---------------------------------------------------------------------------
template <class Lane, int M0 = kModeVIP, int M1 = kModeVIP>
__global__ __launch_bounds__(kBlock, Lane::MINW) void interleaved_kernel(
    typename Lane::Args A, const float* __restrict__ av0, const float* __restrict__ bv0,
    const float* __restrict__ av1, const float* __restrict__ bv1, HmcParams P) {
  constexpr int K = Lane::K, ND = Lane::ND;
  const RelayId rid = relay_begin(P);
  if (rid.seg < 0) return;                 // a hand-over timed out: leave the state as it is (kernels.h: relay_begin)
  long long t = (long long)rid.bid * kBlock + threadIdx.x;
  const int slot = (int)(t % K);
  long long c = t / K;
  const bool live = c < P.C;
  if (!live) c = P.C - 1;
  const int D = P.D;
  ARP_LANE_SMEM(Lane);
  Lane M;
  lane_tables(M, A, s_lane_tab);
  M.init(A, av0, bv0, slot);

  __shared__ float s_eps[2][kMaxD];
  __shared__ float s_save[2 * ND * kBlock];
  ARP_STAGE_SMEM(Lane);
  float* save = s_save + threadIdx.x;
  float* stage = s_stage + (threadIdx.x >> 6) * stage_floats<Lane>();
  const long long cw0 = ((long long)rid.bid * kBlock + (threadIdx.x & ~63)) / K;
  const int cl = (threadIdx.x & 63) / K;
  const int nvalid = (int)(P.C - cw0 < 64 / K ? (P.C - cw0 > 0 ? P.C - cw0 : 0) : 64 / K) * D;
  for (int d = threadIdx.x; d < D; d += kBlock) { s_eps[0][d] = P.eps0[d]; s_eps[1][d] = P.eps0_1[d]; }
  __syncthreads();

  float q[ND], g[ND], eps[ND], x[ND];
  float* qrow = P.q + c * D;
  load_row(M, qrow, q);
  float kap[2], es[2], la_[2];
  Rng rng;
  uint32_t* rs = P.rng + ((size_t)c * kRngSlots + slot) * 4;
  uint32_t nacc0, nacc1;
  if (P.step_base == 0) {
    kap[0] = kap[1] = 1.0f; es[0] = es[1] = 0.0f; la_[0] = la_[1] = 0.0f;
    nacc0 = nacc1 = 0u;
    rng = rng_seed(P.seed, (unsigned long long)(P.chain_offset + c), (uint32_t)slot, (uint32_t)K);
  } else {
    kap[0] = P.adapt[c * 4 + 0]; es[0] = P.adapt[c * 4 + 1]; la_[0] = P.adapt[c * 4 + 2];
    kap[1] = P.adapt1[c * 4 + 0]; es[1] = P.adapt1[c * 4 + 1]; la_[1] = P.adapt1[c * 4 + 2];
    nacc0 = P.accept_count[c]; nacc1 = P.accept_count1[c];
    rng = Rng{rs[0], rs[1]};
  }

  // The reference re-bootstraps logp/grad after every change of coordinates
  // (interleaved.py:120-123, 136-139).  Where the CP <-> NCP map is a shear with unit Jacobian
  // (Lane::HAS_CARRY) the log density is unchanged and the gradient follows by the chain rule, so
  // it is carried across the change of coordinates instead of being recomputed: 2*num_ls instead of
  // 2*num_ls + 2 gradient evaluations per step.  The carried pair is kept in grad/logp between launches.
  constexpr bool CARRY = Lane::HAS_CARRY && M0 == kModeCP && M1 == kModeNCP;
  float lp = 0.0f;
  if (CARRY) {
    if (P.step_base == 0 || !P.grad) {
      lp = lane_grad<M0, true>(M, q, g);
    } else {
      load_row(M, P.grad + c * D, g);
      lp = P.logp[c];
    }
  }
  int next_rec = P.rec_step, rec_row = P.rec_row, bpos = P.stats_bpos;
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see hmc_kernel
  for (int s = 0; s < P.n_steps; ++s) {
    const long long n = P.step_base + s + 1;
    bool acc0, acc1;
    // --- parameterisation 0 ---
    if (!CARRY) lp = lane_grad<M0, true>(M, q, g);
    load_row(M, s_eps[0], eps);
#pragma unroll
    for (int i = 0; i < ND; ++i) eps[i] *= kap[0];
    float la = hmc_transition<Lane, M0>(M, rng, P.L, eps, q, g, lp, acc0, save);
    nacc0 += acc0 ? 1u : 0u;
    adapt_update(P, n, la, kap[0], es[0], la_[0]);
    // --- parameterisation 1 ---
    if constexpr (CARRY) {
      M.template carry<kModeCP>(q, g);
    } else {
      lane_to_centered<M0>(M, q, x);
      switch_param<M1>(M, av1, bv1);
      lane_from_centered<M1>(M, x, q);
      lp = lane_grad<M1, true>(M, q, g);
    }
    load_row(M, s_eps[1], eps);
#pragma unroll
    for (int i = 0; i < ND; ++i) eps[i] *= kap[1];
    la = hmc_transition<Lane, M1>(M, rng, P.L1, eps, q, g, lp, acc1, save);
    nacc1 += acc1 ? 1u : 0u;
    adapt_update(P, n, la, kap[1], es[1], la_[1]);
    if constexpr (CARRY) {
      M.template carry<kModeNCP>(q, g);
    } else {
      lane_to_centered<M1>(M, q, x);
      switch_param<M0>(M, av0, bv0);
      lane_from_centered<M0>(M, x, q);
    }

    if (s == next_rec && rec_row < P.n_samples) {
      // x holds the centred state (CARRY: CP coordinates are the centred ones); q the
      // parameterisation-0 state the reference records
      const bool use_x = P.trace_centered && !CARRY;
      if (P.trace && cw0 < P.trace_chains) {
        const int nv = min(nvalid, (int)(P.trace_chains - cw0) * D);
        float* wrow = P.trace + ((size_t)rec_row * P.trace_chains + cw0) * D;
        if (use_x) store_row_wave<true>(M, stage, wrow, cl, D, nv, x);
        else store_row_wave<true>(M, stage, wrow, cl, D, nv, q);
      }
      if (P.stats) {
        const bool bend = bpos + 1 == P.stats_batch;
        if (use_x) stats_update_wave(M, stage, P, cw0, cl, D, nvalid, x, rec_row == 0, bend);
        else stats_update_wave(M, stage, P, cw0, cl, D, nvalid, q, rec_row == 0, bend);
        bpos = bend ? 0 : bpos + 1;
      }
      if (live && slot == 0) {
        if (P.trace_accept) P.trace_accept[(size_t)rec_row * P.C + c] = acc0 ? 1 : 0;
        if (P.trace_accept1) P.trace_accept1[(size_t)rec_row * P.C + c] = acc1 ? 1 : 0;
        if (P.rec_accept) P.rec_accept[c] += acc0 ? 1u : 0u;
        if (P.rec_accept1) P.rec_accept1[c] += acc1 ? 1u : 0u;
      }
      next_rec += P.thin;
      rec_row += 1;
    }
  }

  long long c2 = c;
  asm volatile("" : "+v"(c2));
  long long cw2 = cw0;
  asm volatile("" : "+v"(cw2));
  store_row_wave(M, stage, P.q + cw2 * D, cl, D, nvalid, q);
  if (CARRY && P.grad) store_row_wave(M, stage, P.grad + cw2 * D, cl, D, nvalid, g);
  uint32_t* rs2 = P.rng + ((size_t)c2 * kRngSlots + slot) * 4;
  if (live) {
    rs = rs2;
    rs[0] = rng.x; rs[1] = rng.c; rs[2] = 0u; rs[3] = 0u;
    if (slot == 0) {
      if (CARRY && P.grad) P.logp[c2] = lp;
      P.adapt[c2 * 4 + 0] = kap[0]; P.adapt[c2 * 4 + 1] = es[0]; P.adapt[c2 * 4 + 2] = la_[0];
      P.adapt1[c2 * 4 + 0] = kap[1]; P.adapt1[c2 * 4 + 1] = es[1]; P.adapt1[c2 * 4 + 2] = la_[1];
      P.accept_count[c2] = nacc0; P.accept_count1[c2] = nacc1;
    }
  }
  relay_end(P, rid);
}

// ---------------------------------------------------------------------------
// Mean-field VI (inference.find_best_learning_rate, inference.py:26-154, on
// util.get_mean_field_elbo, util.py:232-268): q(z) = prod N(loc, softplus(rho)),
// ELBO estimated with n_mc reparameterised draws, Adam on -ELBO with the
// reference's three-stage learning-rate decay, NaN gradients zeroed.  All
// learning rates and all optimisation steps run in ONE launch (instead of
// n_lr x n_steps session round trips), and every learning rate is spread over
// the chip:
//
//   * a learning rate's n_mc draws are split over G sample groups and, for a model whose gradient is a sum over
//     observations that a lane model can restrict (German credit: Lane::HAS_PART), over R row parts: G x R
//     workgroups of B threads form the learning rate's GROUP; consecutive blockIdx, so a group is dispatched together;
//   * every workgroup reduces its draws' gradient terms in a fixed order (lanes of a wave by butterfly, waves in wave
//     order) and PUBLISHES the partial sums; workgroup w of the group owns the items t = w, w + GR, ... and adds the
//     GR partials of each IN WORKGROUP ORDER, publishes the totals, and every workgroup reads all totals back and
//     applies the same Adam update redundantly -- so a fit is bitwise reproducible from run to run (float atomics
//     would add in arrival order) and nothing but two hand-offs per step crosses workgroups;
//   * a hand-off is the data-is-the-flag form of the inter-workgroup recipe (cdna_hip_programming.md, Guideline 16,
//     R2): every value travels as ONE aligned 8-byte {epoch = step + 1, float} granule written by an agent-scope
//     relaxed atomic store (global_store_dwordx2 sc1: write-through, the per-XCD L2s are not coherent) and read by
//     agent-scope relaxed atomic loads (sc1) that are repeated until the tag is the step's; no fence, no flag, no
//     grid barrier.  Two buffers alternate by step parity: a workgroup can only be one hand-off ahead of the slowest
//     member of its group.  Waits are bounded (kViSpinTicks of the 100 MHz clock): a group whose members are not all
//     resident gives up, flags ViParams::err, and the host reports it instead of a hang.  The host sizes the grid so
//     that every group fits on the device at once (arp_api.hip: arp_vi_run).
//
// The draws are part of the sampler's specification and do not depend on G, R or B: draw s of a step uses stream
// s mod (kViBlock / K) (one MWC stream per (learning rate, stream, slot)) at its (s / (kViBlock / K))-th turn, the
// layout of the one-workgroup kernel of rounds 1 - 4 (oracle_impl.h: orc_vi_run, block = 512); a lane skips the words
// other lanes consume.  With learn_a the VIP parameter a = sigmoid(w) is optimised too (cVIP,
// program_transformations.py:507-510).
// ---------------------------------------------------------------------------
constexpr int kViDmax = kMaxD;
constexpr unsigned long long kViSpinTicks = 200000000ull;   // 2 s of s_memrealtime per wait: the FIRST attempt's bound (arp_vi_run retries with 4 x)

struct ViParams {
  int n_steps, n_mc, learn_a, tied_b, a_prior, D;
  unsigned long long seed;
  float const_base;            // parameterisation independent part of the dropped constant
  int n_top; int top_idx[4]; float top_logscale[4];   // -b_i log(scale_i) of the top-level latents
  const float* lr; float* loc; float* rho; float* w; float* wb; float* elbo;
  float* prior;                // [n_lr][n_steps] log prior density of the learnable parameters per step, or nullptr
  const int* a_group;          // [D] leader element of every element's shared `a` (its own index when not shared), or nullptr
  const int* b_group;          // the same for the separately learned `b`
  int G, R;                    // sample groups and row parts per learning rate: G x R workgroups form its group
  int lr0;                     // learning rate of this launch's first group
  unsigned long long* xch;     // hand-off granules, zeroed before the launch: per group [2][G R][Tp] partials + [2][Tp] totals
  int xch_tp;                  // Tp: items per workgroup, padded (items: nq x D sums + the ELBO sum)
  int* err;                    // device flag, set when a hand-off wait ran out
  unsigned long long spin_ticks;   // bound of one hand-off wait, in ticks of the 100 MHz clock
};

// workspace of one launch, in 8-byte granules (shared with the host)
__host__ __device__ inline size_t vi_xch_group_granules(int GR, int Tp) { return (size_t)2 * GR * Tp + (size_t)2 * Tp; }

// Sum over the 64 / K chains of a wave -- the lanes with the same slot -- on the vector pipe alone: inside a row of 16
// lanes by DPP (quad permutations below 4 lanes per chain, row rotations by 4 and by 8), across the four rows by gfx950's
// v_permlane16_swap / v_permlane32_swap (the two halves of a swapped pair are each other's partner rows).  No LDS round
// trip: a ds_bpermute butterfly waited out the LDS latency four times per value (a third of a German-credit VI step).
// Every lane of the wave's FIRST chain ends with the total (lanes of other chains may round in another order).
template <int K>
ARP_DEV float chain_sum(float v) {
  static_assert(K == 1 || K == 2 || K == 4 || K == 8 || K == 16, "lanes per chain");
  if (K <= 1) v += dpp_mov<0xB1>(v);     // quad_perm [1,0,3,2]
  if (K <= 2) v += dpp_mov<0x4E>(v);     // quad_perm [2,3,0,1]
  if (K <= 4) v += dpp_mov<0x124>(v);    // row_ror:4
  if (K <= 8) v += dpp_mov<0x128>(v);    // row_ror:8
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// The same sum over the chains of ONE 16-lane row only (K = 16: a row is a chain, nothing to add): every lane of the row
// ends with its slot's total.  The VI kernel leaves the rows' partial sums in LDS and the parameter's owner adds them in
// row order -- per value one or two DPP adds and a store, where the butterfly over the whole wave cost ten instructions.
template <int K>
ARP_DEV float row_sum(float v) {
  static_assert(K == 1 || K == 2 || K == 4 || K == 8 || K == 16, "lanes per chain");
  if (K <= 1) v += dpp_mov<0xB1>(v);     // quad_perm [1,0,3,2]
  if (K <= 2) v += dpp_mov<0x4E>(v);     // quad_perm [2,3,0,1]
  if (K <= 4) v += dpp_mov<0x124>(v);    // row_ror:4
  if (K <= 8) v += dpp_mov<0x128>(v);    // row_ror:8
  return v;
}

// d/dx log of the reference's --discrete_prior density on a learnable parameter x in (0,1) (main.py:244-253):
// Mixture(logits (0,5,0); Laplace(0, 0.1), Uniform(0,1), Laplace(1, 0.1))
ARP_DEV float discrete_prior_dlogp(float x) {
  const float w1 = 148.4131591025766f;                 // e^5 (the Laplace components have weight 1 before normalisation)
  const float l0 = 5.0f * fast_exp(-10.0f * x);          // Laplace(0, 0.1) density at x > 0
  const float l1 = 5.0f * fast_exp(-10.0f * (1.0f - x)); // Laplace(1, 0.1) density at x < 1
  return 10.0f * (l1 - l0) / (l0 + w1 + l1);
}

// log of that density (normalised: the mixture weights are softmax(0, 5, 0))
ARP_DEV float discrete_prior_logp(float x) {
  const float l0 = 5.0f * fast_exp(-10.0f * x), l1 = 5.0f * fast_exp(-10.0f * (1.0f - x));
  return fast_log(l0 + 148.4131591025766f + l1) - 5.013385943110028f;   // log(2 + e^5)
}

// Hand-off granules: {tag = epoch, value} in one aligned 8-byte word, agent-scope relaxed atomics on GLOBAL pointers
// (global_store_dwordx2 / global_load_dwordx2 with sc1).
typedef __attribute__((address_space(1))) unsigned long long vi_gu64;
ARP_DEV vi_gu64* vi_global(unsigned long long* p) { return (vi_gu64*)p; }
ARP_DEV void vi_put(vi_gu64* g, unsigned epoch, float v) {
  __hip_atomic_store(g, ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
ARP_DEV unsigned long long vi_get(vi_gu64* g) { return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// n <= CH granules at g[0], g[stride], ...: all loads of a sweep in flight together, sweeps repeated until every tag is
// the epoch's; the values, in order, are handed to `take`.  false: the wait ran out (a member of the group is not running).
template <int CH, class F>
ARP_DEV bool vi_gather(vi_gu64* g, size_t stride, int n, unsigned epoch, unsigned long long bound, F take) {
  unsigned long long x[CH];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < CH; ++j) x[j] = j < n ? vi_get(g + (size_t)j * stride) : ((unsigned long long)epoch << 32);
#pragma unroll
    for (int j = 0; j < CH; ++j) ok = ok && (unsigned)(x[j] >> 32) == epoch;
    if (ok) break;
    if (__builtin_amdgcn_s_memrealtime() - t0 > bound) return false;
    __builtin_amdgcn_s_sleep(2);
  }
#pragma unroll
  for (int j = 0; j < CH; ++j)
    if (j < n) take(__uint_as_float((unsigned)x[j]));
  return true;
}

// largest state dimension the VI kernel serves for a lane model (the host checks it): kViDmax unless the model says less
template <class L, class = void> struct lane_vi_dmax { static constexpr int value = kViDmax; };
template <class L> struct lane_vi_dmax<L, std::void_t<decltype(L::VI_DMAX)>> { static constexpr int value = L::VI_DMAX; };
template <class L, class = void> struct lane_has_part { static constexpr bool value = false; };
template <class L> struct lane_has_part<L, std::void_t<decltype(L::HAS_PART)>> { static constexpr bool value = L::HAS_PART; };

#ifdef ARP_VI_TIMING
// timing experiments only (tools/build_variants.sh vit=-DARP_VI_TIMING): shader cycles per phase of a step, summed by
// thread 0 of the launch's first workgroup and printed when the kernel ends
#define VI_T0() unsigned long long vt_[12] = {0,0,0,0,0,0,0,0,0,0,0,0}, vt0_ = __builtin_readcyclecounter()
#define VI_T(k) do { const unsigned long long n_ = __builtin_readcyclecounter(); vt_[k] += n_ - vt0_; vt0_ = n_; } while (0)
#define VI_TPRINT(steps) do { if (blockIdx.x == 0 && threadIdx.x == 0) { printf("vi_kernel cycles/step:"); \
    for (int k_ = 0; k_ < 12; ++k_) printf(" [%d] %.0f", k_, (double)vt_[k_] / (steps)); printf("\n"); } } while (0)
#else
#define VI_T0() do {} while (0)
#define VI_T(k) do {} while (0)
#define VI_TPRINT(steps) do {} while (0)
#endif

template <class Lane, int B>
__global__ __launch_bounds__(B) void vi_kernel(
    typename Lane::Args A, const float* __restrict__ av, const float* __restrict__ bv, ViParams P) {
  constexpr int K = Lane::K, ND = Lane::ND, NG = Lane::NG;
  constexpr int W = B / 64;                    // waves per workgroup
  constexpr int PPT = (lane_vi_dmax<Lane>::value + B - 1) / B;   // parameters owned by a thread: d = tid + u B
  constexpr int CPW = B / K;                   // draws a workgroup takes per pass
  constexpr int cpp = kViBlock / K;            // draw layout: streams per slot (see above)
  constexpr int NDW = (ND + 1) / 2 * 2;        // 32-bit words a lane consumes per draw
  static_assert(B % 64 == 0 && B % K == 0 && kViBlock % B == 0, "workgroup shape");
  constexpr int DM = lane_vi_dmax<Lane>::value;
  __shared__ float s_sig[DM], s_a[DM], s_b[DM];
  // Lane layout of a parameter row: element i of the lane with slot s sits at position i K + s (a replicated global K
  // times; positions of padding slices hold zeros for ever).  A lane reads {loc, sigma, log sigma, 1} of its element with
  // one ds_read_b128 at an immediate offset and writes its sums the same way: no per-element index arithmetic, and no
  // per-element validity masks (German credit has 32 of them: held in scalar registers they spilled).
  constexpr int PM = ND * K;
  constexpr int NR = W * 4;                  // 16-lane rows of the workgroup
  __shared__ float4 s_row[PM];
  __shared__ int s_pos[DM];                  // position of parameter d in that layout
  // every row's partial sums (sum g, sum g*eps, sum dlogp/da, sum dlogp/db), added up in row order
  __shared__ float s_part[NR][4][PM];
  __shared__ float s_elbo_w[W], s_pri[DM];
  __shared__ float s_elbo, s_prior;
  __shared__ float s_red[2][DM];   // shared (a, b) groups: per-element gradient contributions, then the leaders' values
  __shared__ int s_fail;
  const int D = P.D, tid = threadIdx.x;
  const int GR = P.G * P.R;
  const int grp = blockIdx.x / GR, wg = blockIdx.x - grp * GR, sg = wg / P.R, rp = wg - sg * P.R;
  const int lr_i = P.lr0 + grp;
  const int slot = tid % K, chain0 = tid / K;
  const int spp = P.G * CPW;                           // draws the group takes per pass
  const int passes = (P.n_mc + spp - 1) / spp;
  const int SP = (P.n_mc + cpp - 1) / cpp;             // turns of a stream per step
  const int s0 = sg * CPW + chain0;                    // this lane's first draw
  const int p0 = s0 / cpp, pstride = spp / cpp;        // its turn, and turns between its consecutive draws (passes > 1:
  //                                                      the host makes spp a multiple of cpp, so the stream stays)
  const float pw = rp == 0 ? 1.0f : 0.0f;              // the entropy terms count once per draw

  // parameters owned by this thread: Adam moments live in registers
  float loc[PPT], rho[PPT], w[PPT], wb[PPT], m1[PPT][4], m2[PPT][4];
  int lead[PPT][2], gsize[PPT][2];
#pragma unroll
  for (int u = 0; u < PPT; ++u) {
    const int d = tid + u * B;
    loc[u] = rho[u] = w[u] = wb[u] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { m1[u][k] = 0.f; m2[u][k] = 0.f; }
    lead[u][0] = lead[u][1] = d; gsize[u][0] = gsize[u][1] = 1;
    if (d < D) {
      loc[u] = P.loc[(size_t)lr_i * D + d];
      rho[u] = P.rho[(size_t)lr_i * D + d];
      if (P.learn_a) w[u] = P.w[(size_t)lr_i * D + d];
      if (P.wb) wb[u] = P.wb[(size_t)lr_i * D + d];
      s_a[d] = av[d]; s_b[d] = bv[d];
      // untied parameterisation variables that the reference creates with the shape of a SCALAR loc / scale while the
      // random variable is a vector (program_transformations.py:486-533): one value shared by the part, owned by its
      // first element (the leader); members are contiguous
      const int* gp[2] = {P.a_group, P.wb ? P.b_group : nullptr};
      for (int k = 0; k < 2; ++k) {
        if (!gp[k]) continue;
        lead[u][k] = gp[k][d];
        int n = 0;
        for (int e = d; e < D && gp[k][e] == d; ++e) ++n;
        gsize[u][k] = lead[u][k] == d ? n : 0;
      }
    }
  }
  if (tid == 0) s_fail = 0;
  const float base_lr = P.lr[lr_i];
  ARP_LANE_SMEM(Lane);
  Lane M;
  __syncthreads();
  lane_tables(M, A, s_lane_tab);
  M.init(A, s_a, s_b, slot);
  if constexpr (lane_has_part<Lane>::value) { M.set_part(rp, P.R); M.make_resident(); }
  for (int e = tid; e < PM; e += B) s_row[e] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < K) {                                       // the first chain's lanes write the map (every chain has the same)
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      if (i < NG) { if (slot == 0) s_pos[M.gg(i)] = -(i * K) - 1; }        // negative: replicated over the K slots
      else if (M.lvalid(i - NG)) s_pos[M.lidx(i - NG)] = i * K + slot;
    }
  }
  __syncthreads();
  int lpos[PPT], lrep[PPT];
#pragma unroll
  for (int u = 0; u < PPT; ++u) {
    const int d = tid + u * B;
    const int e = d < D ? s_pos[d] : 0;
    lpos[u] = e < 0 ? -(e + 1) : e;
    lrep[u] = e < 0 ? K : 1;
  }
  float* const my_part = &s_part[tid >> 4][0][tid & 15];      // lanes of a row's first chain: (tid & 15) < K
  const bool row_writer = (tid & 15) < K;
  // one stream per (learning rate, stream of the layout, slot)
  Rng rng = rng_seed(P.seed ^ 0x5649564956495649ull, ((unsigned long long)lr_i << 32) | (unsigned)(s0 % cpp),
                     (uint32_t)slot, (uint32_t)K);
  // one draw per lane and step (the usual shape): the lane stands at its own turn from the start and each step ends
  // with ONE jump over the SP - 1 turns of the other lanes (arp_device.h: rng_jump) instead of stepping through them
  const bool single = passes == 1;
  const uint64_t jump_step = mwc_pow((unsigned)((SP - 1) * NDW));
  if (single && p0 < SP) rng_jump(rng, mwc_pow((unsigned)(p0 * NDW)));
  float b1t = 1.0f, b2t = 1.0f;
  const int nq = P.learn_a ? 4 : 2;
  const int T = nq * D + 1, Tp = P.xch_tp;
  vi_gu64* const part_base = GR > 1 ? vi_global(P.xch) + (size_t)grp * vi_xch_group_granules(GR, Tp) : nullptr;
  bool failed = false;

  VI_T0();
  for (int step = 0; step < P.n_steps && !failed; ++step) {
    VI_T(11);
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      const int d = tid + u * B;
      if (d < D) {
        float sp = rho[u] > 20.0f ? rho[u] : fast_log(1.0f + fast_exp(rho[u]));   // softplus
        s_sig[d] = sp;
        const float4 r4 = make_float4(loc[u], sp, fast_log(sp), 1.0f);
        for (int j = 0; j < lrep[u]; ++j) s_row[lpos[u] + j] = r4;
        if (P.learn_a) {
          float a = sigmoidf_(w[u]);
          s_a[d] = a;
          if (P.tied_b) s_b[d] = a;
          if (P.wb) s_b[d] = sigmoidf_(wb[u]);
        }
      }
    }
    __syncthreads();
    if (P.prior) {
      // log prior of the learnable parameters at the values this step's ELBO is evaluated with (inference.py:50-54);
      // a shared parameter is one variable and counts once
#pragma unroll
      for (int u = 0; u < PPT; ++u) {
        const int d = tid + u * B;
        if (d < D) {
          float lpr = 0.f;
          if (P.learn_a) {
            lpr = lead[u][0] == d ? discrete_prior_logp(s_a[d]) : 0.f;
            if (P.wb && lead[u][1] == d) lpr += discrete_prior_logp(s_b[d]);
          }
          s_pri[d] = lpr;
        }
      }
    }
    if (P.learn_a) M.set_param(s_a, s_b);
    VI_T(0);

    float acc[4][ND];
    float elbo = 0.f;
    int pos = 0;                                  // turns of the stream consumed so far in this step (several passes only)
    // One pass = one draw per lane.  FIRST (the only pass unless the group is smaller than the draws): the sums START
    // here, so nothing but the draw is live across the gradient; later passes add to them.
    auto one_pass = [&](int pass, auto first_tag) {
      constexpr bool FIRST = decltype(first_tag)::value;
      const int turn = p0 + pass * pstride;
      const bool live = s0 + pass * spp < P.n_mc;
      float eps[ND], z[ND], g[ND];
#pragma unroll
      for (int i = 0; i < ND; ++i) eps[i] = 0.f;
      if (turn < SP) {
        if (!single)
          for (int n = (turn - pos) * NDW; n > 0; --n) rng_next(rng);    // words of the turns other lanes take
#pragma unroll
        for (int i = 0; i < ND; i += 2) {
          float z0, z1;
          uint32_t w0 = rng_next(rng), w1 = rng_next(rng);
          normal_pair(w0, w1, z0, z1);
          eps[i] = z0;
          if (i + 1 < ND) eps[i + 1] = z1;
        }
        pos = turn + 1;
      }
      VI_T(1);
      float ent = 0.f, entg = 0.f;
      {
        const float4* const rowp = s_row + slot;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
          const float4 r4 = rowp[i * K];          // {loc, sigma, log sigma, 1}, all zero in a padding slice
          if (i < NG) {
            eps[i] = group_bcast0<K>(eps[i], slot);
            entg += fmaf(0.5f * eps[i], eps[i], r4.z);
          } else {
            eps[i] *= r4.w;
            ent += fmaf(0.5f * eps[i], eps[i], r4.z);
          }
          z[i] = fmaf(r4.y, eps[i], r4.x);
        }
      }
      VI_T(2);
      float lp = M.template grad<true>(z, g);
      VI_T(3);
      // one ELBO sample: log p(z) - log q(z), the 0.5 log 2pi per latent added by the host-side constant
      // (a row part's lp is its share of log p; the entropy terms count in row part 0)
      float e = fmaf(pw, group_sum<K>(ent) + entg, lp);
      if (FIRST) {
        elbo = live ? e : 0.f;
#pragma unroll
        for (int i = 0; i < ND; ++i) { acc[0][i] = live ? g[i] : 0.f; acc[1][i] = live ? g[i] * eps[i] : 0.f; }
        if (P.learn_a) {
          float da[ND], db[ND];
          M.dparam(z, g, da, db);
#pragma unroll
          for (int i = 0; i < ND; ++i) { acc[2][i] = live ? da[i] : 0.f; acc[3][i] = live ? db[i] : 0.f; }
        }
      } else if (live) {
        elbo += e;
#pragma unroll
        for (int i = 0; i < ND; ++i) { acc[0][i] += g[i]; acc[1][i] = fmaf(g[i], eps[i], acc[1][i]); }
        if (P.learn_a) {
          float da[ND], db[ND];
          M.dparam(z, g, da, db);
#pragma unroll
          for (int i = 0; i < ND; ++i) { acc[2][i] += da[i]; acc[3][i] += db[i]; }
        }
      }
      VI_T(4);
    };
    one_pass(0, std::true_type{});
    // (a lane model with row parts is launched with one draw per lane only -- arp_api.hip: arp_vi_run -- and does not
    // carry the later passes' code: its sums would have to live across the matrix-core gradient)
    if constexpr (!lane_has_part<Lane>::value)
      for (int pass = 1; pass < passes; ++pass) one_pass(pass, std::false_type{});
    // to the start of this lane's turn of the next step: one jump where a lane takes one draw per step
    if (single) { if (SP > 1) rng_jump(rng, jump_step); }
    else for (int n = (SP - pos) * NDW; n > 0; --n) rng_next(rng);
    VI_T(1);
    // sums over the chains of each 16-lane row in registers (all lanes), then -- under ONE branch, not one execution mask
    // per element -- the rows' partial sums go to LDS in the lane layout
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int i = 0; i < ND; ++i) acc[k][i] = row_sum<K>(acc[k][i]);
    if (P.learn_a) {
#pragma unroll
      for (int k = 2; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < ND; ++i) acc[k][i] = row_sum<K>(acc[k][i]);
    }
    if (row_writer) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int i = 0; i < ND; ++i) my_part[k * PM + i * K] = acc[k][i];
      if (P.learn_a) {
#pragma unroll
        for (int k = 2; k < 4; ++k)
#pragma unroll
          for (int i = 0; i < ND; ++i) my_part[k * PM + i * K] = acc[k][i];
      }
    }
    elbo = chain_sum<K>(elbo);
    if ((tid & 63) == 0) s_elbo_w[tid >> 6] = elbo;
    VI_T(5);
    __syncthreads();
    VI_T(6);
    float tot[PPT][4];
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      const int d = tid + u * B;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float t = 0.f;
        if (d < D && k < nq) {
          t = s_part[0][k][lpos[u]];
#pragma unroll
          for (int r = 1; r < NR; ++r) t += s_part[r][k][lpos[u]];
        }
        tot[u][k] = t;
      }
    }
    float elbo_tot = 0.f;
    if (tid == 0) {
      elbo_tot = s_elbo_w[0];
#pragma unroll
      for (int wv = 1; wv < W; ++wv) elbo_tot += s_elbo_w[wv];
    }
    if (GR > 1) {
      const unsigned epoch = (unsigned)step + 1u;
      vi_gu64* const p1 = part_base + (size_t)(step & 1) * GR * Tp;             // [GR][Tp]
      // ---- hand-off 1: this workgroup's partial sums -> the owners of the items
      vi_gu64* const p2 = part_base + (size_t)2 * GR * Tp + (size_t)(step & 1) * Tp;   // [Tp]
#pragma unroll
      for (int u = 0; u < PPT; ++u) {
        const int d = tid + u * B;
        if (d < D)
          for (int k = 0; k < nq; ++k) vi_put(p1 + (size_t)wg * Tp + k * D + d, epoch, tot[u][k]);
      }
      if (tid == 0) vi_put(p1 + (size_t)wg * Tp + nq * D, epoch, elbo_tot);
      VI_T(7);
      // ---- the items this workgroup owns: the GR partials of each, added in workgroup order
      for (int t = wg + GR * tid; t < T; t += GR * B) {
        float sum = 0.f;
        bool ok = true;
        for (int src = 0; src < GR && ok; src += 32)
          ok = vi_gather<32>(p1 + (size_t)src * Tp + t, (size_t)Tp, min(32, GR - src), epoch, P.spin_ticks, [&](float v) { sum += v; });
        if (!ok) { failed = true; sum = __builtin_nanf(""); }
        vi_put(p2 + t, epoch, sum);
      }
      VI_T(8);
      // ---- hand-off 2: every workgroup reads every total
#pragma unroll
      for (int u = 0; u < PPT; ++u) {
        const int d = tid + u * B;
        if (d < D) {
          int k = 0;
          if (!vi_gather<4>(p2 + d, (size_t)D, nq, epoch, P.spin_ticks, [&](float v) { tot[u][k++] = v; })) failed = true;
        }
      }
      if (tid == B - 1 && !vi_gather<1>(p2 + nq * D, 1, 1, epoch, P.spin_ticks, [&](float v) { s_elbo = v; })) failed = true;
      if (failed) { s_fail = 1; if (P.err) *P.err = 1; }
      VI_T(9);
    } else if (tid == 0) {
      s_elbo = elbo_tot;
    }
    if (tid == 0) {
      float pr = 0.f;
      if (P.prior) for (int d = 0; d < D; ++d) pr += s_pri[d];
      s_prior = pr;
    }

    // Adam (tf.train.AdamOptimizer defaults) on -ELBO with NaN gradients zeroed
    // (inference.py:47, 62-66) and the learning-rate schedule of inference.py:69-75
    float lr = base_lr;
    if (3 * step > 2 * P.n_steps) lr = base_lr / 20.0f; else if (3 * step > P.n_steps) lr = base_lr / 5.0f;
    b1t *= 0.9f; b2t *= 0.999f;
    const float lr_t = lr * __builtin_amdgcn_sqrtf(1.0f - b2t) / (1.0f - b1t);
    // gradient of -(ELBO + prior) w.r.t. the unconstrained w (a = sigmoid(w)) and w_b; a shared variable sums the
    // likelihood terms of its members (contiguous, behind the leader) and takes the prior once
    const bool grouped = P.a_group || (P.wb && P.b_group);
    const float inv = 1.0f / (float)P.n_mc;
    float ga[PPT], gb[PPT];
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      const int d = tid + u * B;
      ga[u] = gb[u] = 0.f;
      if (d < D) {
        ga[u] = P.learn_a ? (tot[u][2] + (P.tied_b ? tot[u][3] : 0.f)) * inv : 0.f;
        gb[u] = P.wb ? tot[u][3] * inv : 0.f;
        if (grouped) { s_red[0][d] = ga[u]; s_red[1][d] = gb[u]; }
      }
    }
    if (grouped) __syncthreads();
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      const int d = tid + u * B;
      if (d < D) {
        if (grouped) {
          for (int e = 1; e < gsize[u][0]; ++e) ga[u] += s_red[0][d + e];
          for (int e = 1; e < gsize[u][1]; ++e) gb[u] += s_red[1][d + e];
        }
        const float sgm = s_sig[d];
        const float a = s_a[d], bb = s_b[d];
        const float pa = P.a_prior ? discrete_prior_dlogp(a) : 0.f;
        const float pb = P.a_prior ? discrete_prior_dlogp(bb) : 0.f;
        float gr[4];
        gr[0] = -tot[u][0] * inv;
        gr[1] = -(tot[u][1] * inv + 1.0f / sgm) * sigmoidf_(rho[u]);
        gr[2] = P.learn_a ? -(ga[u] + pa) * a * (1.0f - a) : 0.f;
        gr[3] = P.wb ? -(gb[u] + pb) * bb * (1.0f - bb) : 0.f;
        float* par[4] = {&loc[u], &rho[u], &w[u], &wb[u]};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float gk = gr[k];
          if (!(gk == gk)) gk = 0.f;
          m1[u][k] = 0.9f * m1[u][k] + 0.1f * gk;
          m2[u][k] = 0.999f * m2[u][k] + 0.001f * gk * gk;
          if (k < 2 || (k == 2 && P.learn_a) || (k == 3 && P.wb))
            *par[k] -= lr_t * m1[u][k] / (__builtin_amdgcn_sqrtf(m2[u][k]) + 1e-8f);
        }
      }
    }
    if (grouped) {   // members take their leader's value
      __syncthreads();
#pragma unroll
      for (int u = 0; u < PPT; ++u) { const int d = tid + u * B; if (d < D) { s_red[0][d] = w[u]; s_red[1][d] = wb[u]; } }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < PPT; ++u) { const int d = tid + u * B; if (d < D) { w[u] = s_red[0][lead[u][0]]; wb[u] = s_red[1][lead[u][1]]; } }
    }
    __syncthreads();
    if (tid == 0 && wg == 0) {
      float c = P.const_base + 0.9189385332046727f * (float)D;   // + 0.5 log 2pi per latent from -log q
      for (int k = 0; k < P.n_top; ++k) c -= s_b[P.top_idx[k]] * P.top_logscale[k];
      P.elbo[(size_t)lr_i * P.n_steps + step] = s_elbo / (float)P.n_mc + c;
      if (P.prior) P.prior[(size_t)lr_i * P.n_steps + step] = s_prior;
    }
    failed = s_fail != 0;
    __syncthreads();
    VI_T(10);
  }
  VI_TPRINT(P.n_steps);
  if (wg == 0) {
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      const int d = tid + u * B;
      if (d < D) {
        P.loc[(size_t)lr_i * D + d] = loc[u];
        P.rho[(size_t)lr_i * D + d] = rho[u];
        if (P.learn_a) P.w[(size_t)lr_i * D + d] = w[u];
        if (P.wb) P.wb[(size_t)lr_i * D + d] = wb[u];
      }
    }
  }
}

}  // namespace arp
