// Host-side plumbing shared by the API translation units: the opaque model
// handle, error reporting and the per-model launcher tables.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <type_traits>
#include <utility>
#include <map>
#include <mutex>
#include <vector>
#include "../../include/autoreparam.h"
#include "kernels.h"
#include "model_radon.h"
#include "radon_fast.h"
#include "election_fast.h"
#include "model_schools.h"
#include "model_election.h"
#include "model_german.h"
#include "model_radon_stddvs.h"
#include "model_funnel.h"
#include "model_electric.h"
#include "model_time_series.h"

namespace arp {

void set_error(const std::string& msg);

#define ARP_HIP_OK(expr)                                                          \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess) {                                                       \
      ::arp::set_error(std::string(#expr) + ": " + hipGetErrorString(e_));        \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

// Launchers a model family exports for one (lanes-per-chain, slice-size) pair.
struct LaneOps {
  int K, NL;   // lanes per chain; groups (counties, states, features ...) owned by one lane
  void (*logp_grad)(const void* args, const float* a, const float* b, const float* x, int C, int D,
                    float* logp, float* grad, hipStream_t s);
  void (*transform)(const void* args, const float* a, const float* b, int dir, const float* in,
                    int C, int D, float* out, hipStream_t s);
  void (*hmc)(const void* args, const float* a, const float* b, const HmcParams& P, hipStream_t s);
  void (*interleaved)(const void* args, const float* a0, const float* b0, const float* a1, const float* b1,
                      const HmcParams& P, hipStream_t s);
  // mean-field VI: `n_groups` learning rates x (P.G x P.R) workgroups of vi_block threads (kernels.h: vi_kernel)
  // coop: hipLaunchCooperativeKernel -- the runtime itself guarantees that every workgroup of the grid is resident (or
  // refuses the launch), which is what the in-launch hand-offs of a learning rate's group need
  hipError_t (*vi)(const void* args, const float* a, const float* b, const ViParams& P, int n_groups, bool coop, hipStream_t s);
  // compile-time parameterisations (nullptr when the lane model has none): hmc for CP / NCP,
  // interleaved for the (CP, NCP) pair
  void (*hmc_cp)(const void* args, const float* a, const float* b, const HmcParams& P, hipStream_t s);
  void (*hmc_ncp)(const void* args, const float* a, const float* b, const HmcParams& P, hipStream_t s);
  void (*interleaved_cp_ncp)(const void* args, const float* a0, const float* b0, const float* a1, const float* b1,
                             const HmcParams& P, hipStream_t s);
  // hmc for "a free, b = 1" (nullptr when the lane model has no such form)
  void (*hmc_b1)(const void* args, const float* a, const float* b, const HmcParams& P, hipStream_t s);
  // the general per-element (a, b) on the packed chain layer, where a lane model has it (election); nullptr: the generic kernel
  void (*hmc_vip_pk)(const void* args, const float* a, const float* b, const HmcParams& P, hipStream_t s);
  // shape of the VI kernel's workgroups: threads, whether the lane model splits a gradient's observations into row
  // parts (German credit), and how many workgroups of it one CU holds (occupancy query)
  int vi_block = 0;
  int vi_dmax = 0;
  bool vi_parts = false;
  int (*vi_occ)() = nullptr;
};

// threads of a VI workgroup: 128 (two waves: with 16 - 32 workgroups per learning rate five learning rates cover
// 80 - 160 CUs) unless the lane model names its own
template <class L, class = void> struct lane_vi_block { static constexpr int value = 128; };
template <class L> struct lane_vi_block<L, std::void_t<decltype(L::VI_BLOCK)>> { static constexpr int value = L::VI_BLOCK; };

// Relay segments (kernels.h: relay_begin): the API hands a launcher the flags, the epoch and `segs` = -1 (allowed, not yet
// decided), a forced count (experiments) or 1 (not allowed); the launcher knows its kernel and decides with its occupancy:
// segments pay where the chain blocks are at least one round of resident workgroups (profiles/r05_relay_segments.txt) --
// 8 from 512 steps per launch on, 4 from 256.
// CUs of the device the calling thread's launch goes to: relay_prepare sets it from the handle (its device, cached at
// arp_model_create) right before the launcher runs on the same thread
inline int& relay_device_cus() {
  static thread_local int cus = 0;
  return cus;
}
// what the calling thread's last chain launch did (arp_relay_geometry, a measurement hook): segments, chain blocks, workgroups of
// the kernel one CU holds (0 where the launcher did not have to ask)
struct RelayGeometry { int v[3]; };
inline RelayGeometry& relay_last() {
  static thread_local RelayGeometry g = {{1, 0, 0}};
  return g;
}
template <class F>
inline HmcParams relay_plan(const HmcParams& P, int blocks, F kernel) {
  HmcParams Q = P;
  int segs = P.segs;
  int occ_seen = 0;
  if (segs == -1) {
    static std::mutex mu;
    static std::map<const void*, int> occ_of;
    int occ = 0;
    {
      std::lock_guard<std::mutex> lock(mu);
      auto it = occ_of.find((const void*)kernel);
      if (it == occ_of.end()) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, kBlock, 0) != hipSuccess) occ = 0;
        occ_of[(const void*)kernel] = occ;
      } else {
        occ = it->second;
      }
    }
    occ_seen = occ;
    const long long slots = (long long)occ * relay_device_cus();
    segs = (slots > 0 && blocks >= slots) ? (P.n_steps >= 512 ? 8 : (P.n_steps >= 256 ? 4 : 1)) : 1;
    // one workgroup per CU (German credit and time_series at 4 lanes per chain: 100 - 155 KB of LDS): a workgroup that waits for
    // its block holds the whole CU, and every hand-over invalidates the XCD's L2 under the tile stream -- measured + 0.1 % and
    // + 4.2 % (tools/experiments/relay_ab.py), so these launches stay whole
    if (occ < 2) segs = 1;
  }
  if (segs < 1 || !P.seg_flags || blocks > P.seg_blocks) segs = 1;      // (seg_blocks arrives as the flags' capacity)
  Q.segs = segs;
  Q.seg_len = (P.n_steps + segs - 1) / segs;
  Q.seg_blocks = blocks;
  relay_last() = {{segs, blocks, occ_seen}};
  return Q;
}

// launch KERNEL over NBLOCKS chain blocks (x the segments relay_plan decides), the kernel's own arguments first, P last
#define ARP_RELAY_LAUNCH(KERNEL, NBLOCKS, STREAM, P, ...)                                                   \
  do {                                                                                                      \
    const HmcParams Q_ = relay_plan(P, NBLOCKS, KERNEL);                                                     \
    hipLaunchKernelGGL(KERNEL, dim3((NBLOCKS) * Q_.segs), dim3(kBlock), 0, STREAM, __VA_ARGS__, Q_);         \
  } while (0)

template <class Lane>
struct Launch {
  static int blocks(int C) { return (int)(((long long)C * Lane::K + kBlock - 1) / kBlock); }
  static void logp_grad(const void* args, const float* a, const float* b, const float* x, int C, int D,
                        float* logp, float* grad, hipStream_t s) {
    hipLaunchKernelGGL(logp_grad_kernel<Lane>, dim3(blocks(C)), dim3(kBlock), 0, s,
                       *(const typename Lane::Args*)args, a, b, x, C, D, logp, grad);
  }
  static void transform(const void* args, const float* a, const float* b, int dir, const float* in,
                        int C, int D, float* out, hipStream_t s) {
    hipLaunchKernelGGL(transform_kernel<Lane>, dim3(blocks(C)), dim3(kBlock), 0, s,
                       *(const typename Lane::Args*)args, a, b, dir, in, C, D, out);
  }
  static void hmc(const void* args, const float* a, const float* b, const HmcParams& P, hipStream_t s) {
    const HmcParams Q = relay_plan(P, blocks(P.C), hmc_kernel<Lane>);
    hipLaunchKernelGGL(hmc_kernel<Lane>, dim3(blocks(P.C) * Q.segs), dim3(kBlock), 0, s,
                       *(const typename Lane::Args*)args, a, b, Q);
  }
  static void interleaved(const void* args, const float* a0, const float* b0, const float* a1, const float* b1,
                          const HmcParams& P, hipStream_t s) {
    const HmcParams Q = relay_plan(P, blocks(P.C), interleaved_kernel<Lane>);
    hipLaunchKernelGGL(interleaved_kernel<Lane>, dim3(blocks(P.C) * Q.segs), dim3(kBlock), 0, s,
                       *(const typename Lane::Args*)args, a0, b0, a1, b1, Q);
  }
  static constexpr int kViB = lane_vi_block<Lane>::value;
  static hipError_t vi(const void* args, const float* a, const float* b, const ViParams& P, int n_groups, bool coop, hipStream_t s) {
    if constexpr (Lane::HAS_VI) {
      if (coop) {
        void* kargs[] = {const_cast<void*>(args), (void*)&a, (void*)&b, const_cast<ViParams*>(&P)};
        return hipLaunchCooperativeKernel((const void*)vi_kernel<Lane, kViB>, dim3(n_groups * P.G * P.R), dim3(kViB), kargs, 0, s);
      }
      hipLaunchKernelGGL((vi_kernel<Lane, kViB>), dim3(n_groups * P.G * P.R), dim3(kViB), 0, s,
                         *(const typename Lane::Args*)args, a, b, P);
    }
    return hipGetLastError();
  }
  static int vi_occ() {
    int n = 0;
    if constexpr (Lane::HAS_VI)
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, vi_kernel<Lane, kViB>, kViB, 0) != hipSuccess) n = 0;
    return n;
  }
  static void set_vi(LaneOps& o) {
    if constexpr (Lane::HAS_VI) {
      o.vi = &vi; o.vi_block = kViB; o.vi_parts = lane_has_part<Lane>::value; o.vi_occ = &vi_occ;
      o.vi_dmax = lane_vi_dmax<Lane>::value;
    }
  }
  template <int MODE>
  static void hmc_m(const void* args, const float* a, const float* b, const HmcParams& P, hipStream_t s) {
    const HmcParams Q = relay_plan(P, blocks(P.C), hmc_kernel<Lane, MODE>);
    hipLaunchKernelGGL((hmc_kernel<Lane, MODE>), dim3(blocks(P.C) * Q.segs), dim3(kBlock), 0, s,
                       *(const typename Lane::Args*)args, a, b, Q);
  }
  static void interleaved_m(const void* args, const float* a0, const float* b0, const float* a1, const float* b1,
                            const HmcParams& P, hipStream_t s) {
    const HmcParams Q = relay_plan(P, blocks(P.C), interleaved_kernel<Lane, kModeCP, kModeNCP>);
    hipLaunchKernelGGL((interleaved_kernel<Lane, kModeCP, kModeNCP>), dim3(blocks(P.C) * Q.segs), dim3(kBlock), 0, s,
                       *(const typename Lane::Args*)args, a0, b0, a1, b1, Q);
  }
  // only the VI launcher (a lane sized for the VI kernel's workgroup; nothing else is instantiated)
  static LaneOps vi_only() {
    LaneOps o{Lane::K, Lane::NGRP, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    set_vi(o);
    return o;
  }
  template <class L, class = void> struct has_b1 : std::false_type {};
  template <class L> struct has_b1<L, std::enable_if_t<L::HAS_MODE_B1>> : std::true_type {};
  static LaneOps ops() {
    LaneOps o{Lane::K, Lane::NGRP, &logp_grad, &transform, &hmc, &interleaved, nullptr, nullptr, nullptr, nullptr, nullptr};
    set_vi(o);
    if constexpr (has_b1<Lane>::value) o.hmc_b1 = &hmc_m<kModeB1>;
    if constexpr (Lane::HAS_MODES) {
      o.hmc_cp = &hmc_m<kModeCP>;
      o.hmc_ncp = &hmc_m<kModeNCP>;
      o.interleaved_cp_ncp = &interleaved_m;
    }
    return o;
  }
};

// experiments only: ARP_STATS_LDS=0 sends statistics runs down the plane-per-sample route of kernels.h again.  Honoured
// under ARP_DEBUG=1 only and announced on stderr: a stray variable must not change which kernel a production run takes.
inline bool stats_lds_enabled() {
  static const bool on = [] {
    const char* e = getenv("ARP_STATS_LDS");
    if (!(e && e[0] == '0')) return true;
    const char* d = getenv("ARP_DEBUG");
    if (!(d && d[0] == '1' && d[1] == 0)) {
      fprintf(stderr, "libautoreparam_hip: ARP_STATS_LDS=0 IGNORED (experiment switch; set ARP_DEBUG=1 to enable it)\n");
      return true;
    }
    fprintf(stderr, "libautoreparam_hip: DEBUG SWITCH ARP_STATS_LDS=0 is in effect (statistics take the plane-per-sample route)\n");
    return false;
  }();
  return on;
}

// Radon: the generic lane kernels serve the general VIP form, the packed kernels of radon_fast.h the two
// compile-time parameterisations (centred, non-centred) and their interleaving.
template <int K, int NL>
LaneOps radon_lane_ops() {
  LaneOps o = Launch<RadonLane<K, NL>>::ops();
  // (the packed layer wants at least two county pairs per lane: the 13- and 15-county states at 8 / 16 lanes per chain run
  // on the generic kernels)
  if constexpr (K >= 4 && NL >= 4) {
    using T = RadonPk<K, NL>;
    // a run that accumulates statistics takes the instantiation with the accumulators in LDS when they fit
    constexpr bool SL = PkBlock<T>::kStatsFit;
    o.hmc_cp = [](const void* args, const float*, const float*, const HmcParams& P, hipStream_t s) {
      const int nb = Launch<RadonLane<K, NL>>::blocks(P.C);
      if (SL && P.stats && stats_lds_enabled()) ARP_RELAY_LAUNCH((pk_hmc_kernel<T, kModeCP, SL>), nb, s, P, *(const RadonArgs*)args, nullptr, nullptr);
      else ARP_RELAY_LAUNCH((pk_hmc_kernel<T, kModeCP>), nb, s, P, *(const RadonArgs*)args, nullptr, nullptr);
    };
    o.hmc_ncp = [](const void* args, const float*, const float*, const HmcParams& P, hipStream_t s) {
      const int nb = Launch<RadonLane<K, NL>>::blocks(P.C);
      if (SL && P.stats && stats_lds_enabled()) ARP_RELAY_LAUNCH((pk_hmc_kernel<T, kModeNCP, SL>), nb, s, P, *(const RadonArgs*)args, nullptr, nullptr);
      else ARP_RELAY_LAUNCH((pk_hmc_kernel<T, kModeNCP>), nb, s, P, *(const RadonArgs*)args, nullptr, nullptr);
    };
    // cVIP / dVIP runs: a free per county (m has unit scale, so b is inert: "a free, b = 1" and the untied form are the
    // same kernel)
    o.hmc_vip_pk = [](const void* args, const float* a, const float* b, const HmcParams& P, hipStream_t s) {
      const int nb = Launch<RadonLane<K, NL>>::blocks(P.C);
      if (SL && P.stats && stats_lds_enabled()) ARP_RELAY_LAUNCH((pk_hmc_kernel<T, kModeVIP, SL>), nb, s, P, *(const RadonArgs*)args, a, b);
      else ARP_RELAY_LAUNCH((pk_hmc_kernel<T, kModeVIP>), nb, s, P, *(const RadonArgs*)args, a, b);
    };
    o.hmc_b1 = o.hmc_vip_pk;
    o.interleaved_cp_ncp = [](const void* args, const float*, const float*, const float*, const float*,
                              const HmcParams& P, hipStream_t s) {
      const int nb = Launch<RadonLane<K, NL>>::blocks(P.C);
      if (SL && P.stats && stats_lds_enabled()) ARP_RELAY_LAUNCH((radon_interleaved_kernel<T, SL>), nb, s, P, *(const RadonArgs*)args);
      else ARP_RELAY_LAUNCH((radon_interleaved_kernel<T>), nb, s, P, *(const RadonArgs*)args);
    };
  }
  return o;
}

// Election: the generic lane kernels serve the general VIP form and the interleaved sampler, the packed kernels
// (election_fast.h on pk_chain.h) the three compile-time parameterisations of a plain HMC run.
template <int K, int NL>
LaneOps election_lane_ops() {
  LaneOps o = Launch<ElectionLane<K, NL>>::ops();
  if constexpr (K >= 4) {
    using T = ElectionPk<K, NL>;
    constexpr bool SL = PkBlock<T>::kStatsFit;   // statistics accumulators in LDS (two workgroups per CU instead of three)
#define ARP_EL(MODE)                                                                                              \
    [](const void* args, const float* a, const float* b, const HmcParams& P, hipStream_t s) {                     \
      const int nb = Launch<ElectionLane<K, NL>>::blocks(P.C);                                                      \
      if (SL && P.stats && stats_lds_enabled())                                                                    \
        ARP_RELAY_LAUNCH((pk_hmc_kernel<T, MODE, SL>), nb, s, P, *(const ElectionArgs*)args, a, b); \
      else                                                                                                         \
        ARP_RELAY_LAUNCH((pk_hmc_kernel<T, MODE>), nb, s, P, *(const ElectionArgs*)args, a, b);  \
    }
    o.hmc_cp = ARP_EL(kModeCP);
    o.hmc_ncp = ARP_EL(kModeNCP);
    o.hmc_b1 = ARP_EL(kModeB1);
    o.hmc_vip_pk = ARP_EL(kModeVIP);
#undef ARP_EL
    // --method=i: centred / non-centred interleaving on the packed layer (pk_chain.h: pk_interleaved_kernel)
    o.interleaved_cp_ncp = [](const void* args, const float* a0, const float* b0, const float*, const float*,
                              const HmcParams& P, hipStream_t s) {
      const int nb = Launch<ElectionLane<K, NL>>::blocks(P.C);
      if (SL && P.stats && stats_lds_enabled())
        ARP_RELAY_LAUNCH((pk_interleaved_kernel<T, kModeCP, kModeNCP, SL>), nb, s, P, *(const ElectionArgs*)args, a0, b0);
      else
        ARP_RELAY_LAUNCH((pk_interleaved_kernel<T, kModeCP, kModeNCP>), nb, s, P, *(const ElectionArgs*)args, a0, b0);
    };
  }
  return o;
}

// per-family tables (defined in inst_*.hip)
const std::vector<LaneOps>& radon_ops();
const std::vector<LaneOps>& schools_ops();
const std::vector<LaneOps>& election_ops();
const std::vector<LaneOps>& german_ops();
const LaneOps& german_bf3_ops();   // 4 lanes per chain, likelihood on bf16 matrix cores with three-piece operands
const std::vector<LaneOps>& radon_sd_ops();
const std::vector<LaneOps>& funnel_ops();
const std::vector<LaneOps>& electric_ops();
const std::vector<LaneOps>& time_series_ops();

}  // namespace arp

struct arp_model {
  int model = -1;
  int D = 0;
  int device = 0;
  bool host_only = false;
  int german_math = 0;       // 0 auto (bf16 x 3 where the data allow), 1 f32 matrix cores, 2 bf16 x 3 (arp_model_set_option)    // test hook (arp_api.hip: host_only): no device behind this handle
  int n_groups = 0;          // slice axis length (radon J, election 52, schools 8)
  float* dev_tables = nullptr;   // one allocation holding all frozen tables
  float* dev_ab[2] = {nullptr, nullptr};  // [2][D]: a then b, per parameterisation
  bool has_param[2] = {false, false};
  int param_kind[2] = {0, 0};   // kModeVIP / kModeCP / kModeNCP / kModeB1, detected in arp_model_set_param
  double logp_const[2] = {0.0, 0.0};
  arp::RadonArgs radon{};
  arp::SchoolsArgs schools{};
  arp::ElectionArgs election{};
  arp::GermanArgs german{};
  arp::RadonSdArgs radon_sd{};
  arp::FunnelArgs funnel{};
  arp::ElectricArgs electric{};
  arp::TimeSeriesArgs time_series{};
  std::vector<float> host_tables;
  // hand-off workspace of the VI kernel (granules + the error flag in its first 256 bytes), grown on demand
  void* vi_ws = nullptr;
  size_t vi_ws_bytes = 0;
  // the parameters a VI launch starts from, kept until its hand-offs are known to have gone through (arp_vi_run retries)
  float* vi_snap = nullptr;
  size_t vi_snap_floats = 0;
  // pinned, device-visible word a relay launch sets when a hand-over timed out (kernels.h: relay_begin; arp_model_check)
  unsigned* relay_err = nullptr;
  unsigned* relay_err_dev = nullptr;
  bool coop_ok = false;      // the device supports cooperative launches (hipDeviceAttributeCooperativeLaunch)
  int vi_launch = 0;         // 0 auto (cooperative where supported), 1 plain launch + process-wide mutex, 2 cooperative (arp_model_set_option "vi_launch")
  int cus = 0;               // CUs of `device` (relay decisions are made for the handle's device, not the current one)
  double const_base = 0.0;                       // parameterisation independent part of the dropped constant
  std::vector<std::pair<int, double>> top_scale; // (flattened index, log prior scale) of top-level latents
};
