// German credit, log-normal centred scales (reference models.py:884-923) under the
// general VIP parameterisation.  Parts in trace order:
//   overall_log_scale (ols), beta_log_scales[F] (bls), beta[F].
//
//   ols ~ N(0,10)                      top level: ot ~ N(0, 10^b), ols = 10^(1-b) ot
//   bls_d ~ N(ols, 1)                  sigma = 1: blt_d ~ N(a_d ols, 1), bls_d = blt_d + (1 - a_d) ols
//   beta_d ~ N(0, exp(bls_d))          mu = 0:   bt_d ~ N(0, exp(b_d bls_d)), beta_d = exp((1-b_d) bls_d) bt_d
//   y_n ~ Bernoulli(logit = X beta)    [N x F] dense contraction: this model is compute bound
//
// A chain is spread over K lanes that own F/K consecutive features each.  The
// design matrix (padded to 64 columns) streams through LDS in tiles shared by the
// whole workgroup; every lane forms its partial logit, the K partials are summed
// with DPP, and the residual y - sigmoid(eta) is scattered back to the lane's own
// columns.  With v_d = sum_n X_nd (y_n - sigmoid(eta_n)):
//   d/dbt_d  = -bt_d e_d^2 + v_d exp((1-b_d) bls_d),   e_d = exp(-b_d bls_d)
//   hb_d     = b_d (bt_d e_d)^2 - b_d + v_d (1-b_d) beta_d
//   d/dblt_d = hb_d - r_d,  r_d = blt_d - a_d ols;   d/dols = sum a_d r_d + (1-a_d) hb_d
#pragma once
#include "arp_device.h"

namespace arp {

constexpr int kGermanCols = 64;   // padded row length of the device design matrix
// Tile image of the matrix-core likelihood (built by arp_api.hip: build_german): the design matrix in tiles of 128
// observations, each stored as the exact byte image of its LDS copy so that LDS-DMA (global_load_lds_dwordx4: 64 lanes
// x 16 contiguous bytes per instruction, no registers, no VALU) moves it.  A tile is 32 pieces of 1 KiB = 4 rows of 64
// floats; the 16-byte chunk c of row r is stored at chunk position c ^ (r & 11), followed by one piece with the 128
// outcomes.  With that XOR both operand reads of the matrix-core products are bank-conflict free as ds_read_b128
// (forward: 16 rows x one chunk of a 64-byte feature block; backward: 4 rows x 16 consecutive chunks per lane group),
// checked lane by lane against the gfx950 bank rules (4 groups of 16 lanes, bank = dword address mod 64).
constexpr int kGermanTileRows = 128;
#ifndef ARP_GERMAN_VI_BLOCK
#define ARP_GERMAN_VI_BLOCK 256
#endif
constexpr int kGermanViBlock = ARP_GERMAN_VI_BLOCK;   // threads of a VI workgroup (kernels.h: vi_kernel), 16 draws per wave: four waves,
//   one per SIMD, and with 256 draws 4 sample groups x 8 row parts (one resident 128-row tile each) per learning rate:
//   68 ms per fit against 80 ms with two-wave workgroups of two tiles (profiles/r05_vi_kernel.txt)
constexpr int kGermanImgTile = 33 * 256;   // floats per tile of the image

// Tile image of the bf16 x 3 likelihood (round 5; built by arp_api.hip: build_german_bf3).  Every f32 value is the exact
// sum of three bf16 pieces x = h + m + l (8 + 8 + 8 significant bits, by truncation), and a product of two such values is
// the sum of nine bf16 products, of which the six leading ones carry it to 2^-23: matrix-core work at 16 x the f32 rate.
// The data make it cheaper still: a column of zeros and ones (54 one-hot columns and the intercept of German credit) IS
// its h piece, so only the few SPLIT columns (the standardised numerics: at most 8) have m and l pieces at all, and all
// their cross terms fit ONE extra K = 32 step per product:
//   forward   eta = Xh (bh + bm + bl)  +  [Xm | Xm | Xl | 0] [bh ; bm ; bh ; 0]          (split columns only)
//   backward  v   = Xh' (wh + wm + wl) +  [Xm ; Xl]' wh + [Xm ; Xl]' wm                   (16 extra output rows)
// i.e. 7 v_mfma_f32_16x16x32_bf16 per 16 observations forward and 14 per 32 backward (224 cycles per 32 observations
// against 2 048 on v_mfma_f32_16x16x4_f32), operands read from LDS with 11 ds_read_b128 per 32 observations and wave.
// A tile holds 64 observations in 23 pieces of 1 KiB (the exact byte image of its LDS copy, moved by LDS-DMA):
//   XhF [64 rows][64 features] bf16, 16-byte chunk c of row r at chunk c ^ ((r >> 1) & 7)           8 KiB  forward A operand
//   XaF [64 rows][xm(8) | xm(8) | xl(8) | 0(8)] bf16, chunk g of row r at g ^ ((r >> 2) & 3)        4 KiB
//   XhB [2 k-steps][64 features][4 lane groups g][8] bf16: element j of group g is observation
//       32 s + (j < 4 ? 4 g + j : 16 + 4 g + j - 4) -- the order in which two forward blocks leave their residuals in a
//       lane's registers --, chunk g of feature f at g ^ ((f >> 2) & 3)                             8 KiB  backward A operand
//   XaB [2 k-steps][16 rows: xm of split column o, xl of split column o - 8][4][8] bf16, same order  2 KiB
//   y   [64] f32                                                                                   256 B
// (both operand reads are conflict free: 16 lanes x 16 bytes cover the 64 banks once).
constexpr int kBf3Rows = 64;
constexpr int kBf3Pieces = 23;
constexpr int kBf3ImgTile = kBf3Pieces * 256;   // floats per tile of the image
constexpr int kBf3XhF = 0, kBf3XaF = 8192, kBf3XhB = 12288, kBf3XaB = 20480, kBf3Y = 22528;   // byte offsets in a tile
constexpr int kBf3MaxSplit = 8;

// x = h + m + l exactly, each piece a bf16 (as the f32 bit pattern with a zero low half): truncation twice, the third
// piece is what is left (at most 8 significant bits).  The same on host (the images) and device (beta, the residuals).
__host__ __device__ inline void bf3_split(float x, uint32_t& h, uint32_t& m, uint32_t& l) {
  union { float f; uint32_t u; } a, b_, c_;
  a.f = x; h = a.u & 0xffff0000u;
  b_.u = h; b_.f = x - b_.f; m = b_.u & 0xffff0000u;
  c_.u = m; c_.f = b_.f - c_.f; l = c_.u;
}

struct GermanArgs {
  const float* X;   // [N][64] row-major, columns >= F are zero
  const float* y;   // [N]
  const float* Xt;  // tile image, ceil(N / 128) x kGermanImgTile floats
  int N, F;
  const float* Xb;  // bf16 x 3 tile image, ceil(N / 64) x kBf3ImgTile floats, or nullptr (more than 8 split columns)
  int sidx[kBf3MaxSplit];   // the split columns (-1: unused slot)
};

// W_: waves per workgroup of the kernels the lane is used in (sizes the per-wave LDS areas of the
// matrix-core path): kBlock / 64 for the chain kernels, the VI kernel's own workgroup size / 64 for the VI kernel.
// PART_: the VI kernel's form (kernels.h: vi_kernel).  The observations of one gradient are split over the
// workgroups of a learning rate: a lane evaluates the tiles [tlo, thi) only and the prior terms of the log density and of
// the gradient with weight `pw` (1 in the workgroup that owns row part 0, 0 elsewhere), so that the SUM over the row
// parts of everything grad() and dparam() return is the whole model's -- they are affine in the likelihood's v.
// BF3_: the likelihood on bf16 matrix cores with three-piece operands (above) instead of f32 matrix cores; K = 4 only.
template <int K_, int NLS_, int W_ = kBlock / 64, bool PART_ = false, bool BF3_ = false>
struct GermanLane {
  static_assert(!BF3_ || K_ == 4, "the bf16 x 3 likelihood serves the 4-lane kernels");
  static constexpr int kTileObs = BF3_ ? kBf3Rows : kGermanTileRows;   // observations per tile of this lane's image
  static constexpr int K = K_;
  static constexpr int NG = 1;          // overall_log_scale
  static constexpr int NLS = NLS_;      // features owned by this lane: d = slot*NLS + i
  static constexpr int NL = 2 * NLS;    // local elements: bls slices, then beta slices
  static constexpr int ND = NG + NL;
  static constexpr int NGRP = NLS_;   // groups owned by a lane (what the host matches against ceil(groups / K))
  static constexpr int DCAP = 1 + 2 * kGermanCols;   // upper bound of the flattened state dimension D
  static_assert(K_ * NLS_ == kGermanCols, "lanes x features per lane must cover the padded row");
  static constexpr bool HAS_MODES = false;
  static constexpr bool HAS_CARRY = false;
  static constexpr bool HAS_FUSED = false;
  static constexpr int MINW = K_ >= 8 ? 2 : 1;   // waves per SIMD the register allocator must leave room for
  using Args = GermanArgs;

  float a[NLS], b[NLS];     // a of bls_d, b of beta_d (the only ones that matter)
  float s0i, c0;            // 1/10^b0, 10^(1-b0)
  const float* X; const float* y; const float* Xt; const float* Xb;
  int N, F, slot, nown;
  int sidx[BF3_ ? kBf3MaxSplit : 1];
  int tlo_, thi_; float pw_;   // PART_ only: tile range and prior weight of this workgroup's row part
  bool res_;                   // PART_ only: the part's (at most two) tiles stay in the LDS buffers for the whole launch
  static constexpr bool HAS_PART = PART_;
  static constexpr int VI_BLOCK = W_ * 64;
  static constexpr int VI_DMAX = 128;   // LDS arrays of the VI kernel are sized for this (the real data: D = 125)
  ARP_DEV int tlo() const { if constexpr (PART_) return tlo_; else return 0; }
  ARP_DEV int thi() const { if constexpr (PART_) return thi_; else return (N + kTileObs - 1) / kTileObs; }
  ARP_DEV float pw() const { if constexpr (PART_) return pw_; else return 1.0f; }
  // row part `r` of `R`: whole tiles, as even as they go (a part past the data is empty and contributes nothing)
  ARP_DEV void set_part(int r, int R) {
    const int nt = (N + kTileObs - 1) / kTileObs, per = (nt + R - 1) / R;
    tlo_ = min(nt, r * per); thi_ = min(nt, tlo_ + per); pw_ = r == 0 ? 1.0f : 0.0f;
  }
  ARP_DEV bool resident() const { if constexpr (PART_) return res_; else return false; }
  // A part of one or two tiles is copied into the two LDS buffers ONCE (all threads of the workgroup call this after
  // set_part): every later gradient finds its operands in place -- no LDS-DMA, no wait for it, no workgroup barrier, the
  // waves of the workgroup run their draws independently.
  ARP_DEV void make_resident() {
    static_assert(PART_, "row-part form only");
    const int nt = thi_ - tlo_;
    res_ = nt >= 1 && nt <= 2;
    if (!res_) return;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t tile_off = lds_offset(tile_mem());
    for (int n = 0; n < nt; ++n) {
      if constexpr (BF3_) issue_tile_bf3(tlo_ + n, (n + nt) & 1, tile_off, wv, threadIdx.x & 63);
      else issue_tile(tlo_ + n, (n + nt) & 1, tile_off, wv, threadIdx.x & 63);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  static ARP_DEV int gg(int) { return 0; }
  ARP_DEV int lbase(int i) const { return i < NLS ? 1 + slot * NLS : 1 + F + slot * NLS; }
  static constexpr ARP_DEV int loff(int i) { return i < NLS ? i : i - NLS; }
  ARP_DEV int lidx(int i) const { return i < NLS ? 1 + slot * NLS + i : 1 + F + slot * NLS + (i - NLS); }
  ARP_DEV bool lvalid(int i) const { return (i < NLS ? i : i - NLS) < nown; }

  ARP_DEV void init(const Args& A, const float* av, const float* bv, int slot_) {
    slot = slot_;
    X = A.X; y = A.y; Xt = A.Xt; Xb = A.Xb; N = A.N; F = A.F;
    if constexpr (BF3_) {
#pragma unroll
      for (int q = 0; q < kBf3MaxSplit; ++q) sidx[q] = A.sidx[q];
    }
    nown = F - slot * NLS;
    nown = nown < 0 ? 0 : (nown > NLS ? NLS : nown);
    tlo_ = 0; thi_ = (N + kTileObs - 1) / kTileObs; pw_ = 1.0f; res_ = false;
    set_param(av, bv);
  }
  ARP_DEV void set_param(const float* av, const float* bv) {
    s0i = __builtin_amdgcn_exp2f(-bv[0] * 3.321928094887362f);
    c0 = 10.0f * s0i;
#pragma unroll
    for (int i = 0; i < NLS; ++i) {
      bool ok = i < nown;
      a[i] = ok ? av[1 + slot * NLS + i] : 0.0f;
      b[i] = ok ? bv[1 + F + slot * NLS + i] : 0.0f;
    }
  }

  // K = 8, 16: [rows][64] tile of 64 observations + outcomes, filled through registers (fill_tile).
  // K = 4 (matrix cores): [X buffer 0][outcomes 0][outcomes 1][X buffer 1] of the tile image, then a per-wave exchange area.
  static constexpr int kStride = kGermanCols;
  static constexpr int kRows = K_ == 4 ? kGermanTileRows : 64;
  static constexpr int kBlkB = 16 * kGermanCols * 4;   // bytes of a 16-row block
  static constexpr int kXBufB = kGermanTileRows * kGermanCols * 4;
  static constexpr int kYBufB = 1024;
  static constexpr int kBufStep = kXBufB + 2 * kYBufB;   // X buffer 0 -> X buffer 1
  static constexpr int kYBase = kXBufB;                  // outcomes of buffer 0
  static constexpr int kXchStride = kGermanCols + 4;  // exchange rows padded: conflict-free float4 access both ways
  static constexpr int kXchWaves = W_;                // waves per workgroup the exchange area covers
  static constexpr bool HAS_VI = PART_;   // the VI kernel is built from the row-part form, sized for its own workgroup
  static constexpr int kXch = 16 * kXchStride + 64;   // per wave: [16 chains][row] + 64 log-density partials
  static constexpr int kXchBase = (2 * kXBufB + 2 * kYBufB) / 4;
  static constexpr int kTileFloats = K_ == 4 ? kXchBase + kXchWaves * kXch : kRows * kStride + kRows;
  // The chain kernels' row-staging block (kernels.h: ARP_STAGE_SMEM) aliases tile buffer 0 (X and outcomes): it is used
  // between gradients only, and grad() opens with a workgroup barrier before anything is copied into that buffer.
  static constexpr bool STAGE_ALIAS = K_ == 4;
  static ARP_DEV float* stage_mem() { return tile_mem(); }
  static constexpr int kStageCap = (kXBufB + kYBufB) / 4;
  // The [rows x 64] design-matrix tile and its outcomes, shared by the workgroup (one copy per
  // kernel: both instantiations of grad<> go through this function).
  static ARP_DEV float* tile_mem() {
    __shared__ __attribute__((aligned(16))) float tile[kTileFloats];
    return tile;
  }

  // Cooperative, coalesced float4 copy of tile `n0`; rows past N are zero filled (x = 0 adds
  // nothing to the gradient; the log density masks them).
  ARP_DEV void fill_tile(float* tile, int n0) const {
    const int rows = min(kRows, N - n0);
    const float4* src = reinterpret_cast<const float4*>(X + (size_t)n0 * kGermanCols);
    float4* dst = reinterpret_cast<float4*>(tile);
    const int nthreads = blockDim.x;
    for (int t = threadIdx.x; t < kRows * (kGermanCols / 4); t += nthreads)
      dst[(t >> 4) * (kStride / 4) + (t & 15)] = t < rows * (kGermanCols / 4) ? src[t] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float* ytile = tile + kRows * kStride;
    if ((int)threadIdx.x < kRows) ytile[threadIdx.x] = (int)threadIdx.x < rows ? y[n0 + threadIdx.x] : 0.0f;
  }

  // Likelihood pass, any K: every lane forms its partial logit of one observation, the K partials
  // are summed with a DPP butterfly and all K lanes evaluate the same sigmoid.
  template <bool LOGP>
  ARP_DEV float likelihood_generic(const float (&beta)[NLS], float (&v)[NLS]) const {
    float* tile = tile_mem();
    const float* ytile = tile + kRows * kStride;
    float lp = 0.0f;
    for (int n0 = 0; n0 < N; n0 += kRows) {
      __syncthreads();   // previous tile fully consumed
      fill_tile(tile, n0);
      __syncthreads();
      const int rows = min(kRows, N - n0);
      for (int n = 0; n < rows; ++n) {
        const float* xr = tile + n * kStride + slot * NLS;
        float x[NLS];
#pragma unroll
        for (int i = 0; i < NLS; ++i) x[i] = xr[i];
        float eta = 0.0f;
#pragma unroll
        for (int i = 0; i < NLS; ++i) eta = fmaf(x[i], beta[i], eta);
        eta = group_sum<K>(eta);
        float ex = fast_exp(-fabsf(eta));
        float rc = __builtin_amdgcn_rcpf(1.0f + ex);
        float sg = eta >= 0.0f ? rc : ex * rc;
        float yn = ytile[n];
        float w = yn - sg;
#pragma unroll
        for (int i = 0; i < NLS; ++i) v[i] = fmaf(x[i], w, v[i]);
        if (LOGP) lp += fmaf(yn, eta, -(fmaxf(eta, 0.0f) + fast_log(1.0f + ex)));
      }
    }
    return lp;   // every lane of the chain accumulated the same value
  }

  // Likelihood pass for K = 8 (8 features per lane), four observations at a time.  The two quads
  // of a chain split the four rows: a lane calls the two rows its quad finishes `a` and the other
  // two `b`.  It forms partial logits of all four (packed FMAs), hands the `b` partials to its
  // mirror lane in the other quad (which calls those rows `a`), sums the `a` partials over its quad,
  // evaluates two sigmoids instead of four, and fetches the other two residuals from the mirror
  // lane again: 6 DPP adds + 2 DPP moves + 2 sigmoids per 4 observations instead of 12 + 4.
  // The rows of the next block are fetched from LDS while the current one is processed.
  // (The LDS reads are issued as inline asm so that they stay where they are written -- a whole
  // block ahead of their use; left to the scheduler they sink next to the first use to save
  // registers and every block then waits out the LDS latency.  lgkmcnt returns in order for LDS,
  // so waiting until at most the newest block's 9 reads are outstanding completes the older block.)
  typedef float v4f __attribute__((ext_vector_type(4)));
  struct Rows4 {
    v4f a[2][2], b[2][2];   // rows `a` / `b`, features [0,4) and [4,8) of this lane's slice
    v2f y;                  // outcomes of the two `a` rows
  };
  static ARP_DEV uint32_t lds_offset(const float* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
  }
  ARP_DEV void load_rows(uint32_t tile_off, int nb, int half, Rows4& R) const {
    const uint32_t pa = tile_off + (uint32_t)((nb + 2 * half) * kGermanCols + slot * NLS) * 4u;
    const uint32_t pb = tile_off + (uint32_t)((nb + 2 * (1 - half)) * kGermanCols + slot * NLS) * 4u;
    const uint32_t py = tile_off + (uint32_t)(kRows * kGermanCols + nb + 2 * half) * 4u;
    asm volatile("ds_read_b128 %0, %1" : "=v"(R.a[0][0]) : "v"(pa));
    asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(R.a[0][1]) : "v"(pa));
    asm volatile("ds_read_b128 %0, %1 offset:256" : "=v"(R.a[1][0]) : "v"(pa));
    asm volatile("ds_read_b128 %0, %1 offset:272" : "=v"(R.a[1][1]) : "v"(pa));
    asm volatile("ds_read_b128 %0, %1" : "=v"(R.b[0][0]) : "v"(pb));
    asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(R.b[0][1]) : "v"(pb));
    asm volatile("ds_read_b128 %0, %1 offset:256" : "=v"(R.b[1][0]) : "v"(pb));
    asm volatile("ds_read_b128 %0, %1 offset:272" : "=v"(R.b[1][1]) : "v"(pb));
    asm volatile("ds_read_b64 %0, %1" : "=v"(R.y) : "v"(py));
  }
  // all reads of R have landed once at most `newer` younger LDS reads are outstanding
  template <int NEWER>
  static ARP_DEV void wait_rows(Rows4& R) {
    static_assert(NEWER == 0 || NEWER == 9, "one block = 9 LDS reads");
    if (NEWER == 9)
      asm volatile("s_waitcnt lgkmcnt(9)"
                   : "+v"(R.a[0][0]), "+v"(R.a[0][1]), "+v"(R.a[1][0]), "+v"(R.a[1][1]), "+v"(R.b[0][0]),
                     "+v"(R.b[0][1]), "+v"(R.b[1][0]), "+v"(R.b[1][1]), "+v"(R.y));
    else
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(R.a[0][0]), "+v"(R.a[0][1]), "+v"(R.a[1][0]), "+v"(R.a[1][1]), "+v"(R.b[0][0]),
                     "+v"(R.b[0][1]), "+v"(R.b[1][0]), "+v"(R.b[1][1]), "+v"(R.y));
  }
  template <bool LOGP>
  ARP_DEV void use_rows(const Rows4& R, const v2f (&b2)[4], v2f (&v2)[4], float& lp, int ra, int rows) const {
    float wa[2], wb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const v2f xa[4] = {R.a[j][0].xy, R.a[j][0].zw, R.a[j][1].xy, R.a[j][1].zw};
      const v2f xb[4] = {R.b[j][0].xy, R.b[j][0].zw, R.b[j][1].xy, R.b[j][1].zw};
      v2f sa = xa[0] * b2[0], sb = xb[0] * b2[0];
#pragma unroll
      for (int i = 1; i < 4; ++i) { sa = vfma(xa[i], b2[i], sa); sb = vfma(xb[i], b2[i], sb); }
      float eta = (sa.x + sa.y) + dpp_mov<0x141>(sb.x + sb.y);   // row_half_mirror: lane 7-s, the other quad
      eta += dpp_mov<0xB1>(eta);
      eta += dpp_mov<0x4E>(eta);
      const float ex = fast_exp(-fabsf(eta));
      const float rc = __builtin_amdgcn_rcpf(1.0f + ex);
      const float sg = eta >= 0.0f ? rc : ex * rc;
      const float yj = j ? R.y.y : R.y.x;
      wa[j] = yj - sg;
      if (LOGP) {
        const float t = fmaf(yj, eta, -(fmaxf(eta, 0.0f) + fast_log(1.0f + ex)));
        lp += ra + j < rows ? t : 0.0f;
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) wb[j] = dpp_mov<0x141>(wa[j]);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const v2f xa[4] = {R.a[j][0].xy, R.a[j][0].zw, R.a[j][1].xy, R.a[j][1].zw};
      const v2f xb[4] = {R.b[j][0].xy, R.b[j][0].zw, R.b[j][1].xy, R.b[j][1].zw};
      const v2f wa2 = v2f{wa[j], wa[j]}, wb2 = v2f{wb[j], wb[j]};
#pragma unroll
      for (int i = 0; i < 4; ++i) { v2[i] = vfma(xa[i], wa2, v2[i]); v2[i] = vfma(xb[i], wb2, v2[i]); }
    }
  }
  template <bool LOGP>
  ARP_DEV float likelihood_k8(const float (&beta)[NLS], float (&v)[NLS]) const {
    static_assert(NLS == 8, "K = 8 owns 8 features per lane");
    float* tile = tile_mem();
    const uint32_t tile_off = lds_offset(tile);
    const int half = (slot >> 2) & 1;
    v2f b2[4], v2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { b2[i] = v2f{beta[2 * i], beta[2 * i + 1]}; v2[i] = v2f{0.0f, 0.0f}; }
    float lp = 0.0f;
    for (int n0 = 0; n0 < N; n0 += kRows) {
      __syncthreads();   // previous tile fully consumed
      fill_tile(tile, n0);
      __syncthreads();
      const int rows = min(kRows, N - n0);
      Rows4 A, B;
      load_rows(tile_off, 0, half, A);
      for (int nb = 0; nb < rows; nb += 8) {   // the tile is zero filled up to its 64 rows
        load_rows(tile_off, nb + 4, half, B);
        wait_rows<9>(A);
        use_rows<LOGP>(A, b2, v2, lp, nb + 2 * half, rows);
        load_rows(tile_off, (nb + 8) & (kRows - 1), half, A);
        wait_rows<9>(B);
        use_rows<LOGP>(B, b2, v2, lp, nb + 4 + 2 * half, rows);
      }
      wait_rows<0>(A);   // drain the look-ahead read before the tile is overwritten
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = v2[i].x; v[2 * i + 1] = v2[i].y; }
    // the four lanes of a quad accumulated the same two rows per block: add the two quads
    if (LOGP) lp += dpp_mov<0x141>(lp);
    return lp;
  }

  // Likelihood pass for K = 4 on the matrix cores.  A wave holds 16 chains; logits = X beta is
  // a [obs x 64] x [64 x 16 chains] product and v = X^T (y - sigmoid(logits)) a
  // [64 x obs] x [obs x 16 chains] one, both with f32 inputs and f32 accumulation
  // (v_mfma_f32_16x16x4_f32: exact f32 FMAs at the vector FMA rate -- and on the vector FMA datapath: vector
  // instructions of ANY wave on the SIMD wait while an f32 MFMA runs, tools/mfma_overlap.hip -- but X is read from
  // LDS once per 16 chains instead of once per chain and every (observation, chain) sigmoid is evaluated exactly once).
  //   state layout : lane 4c+t   holds beta[16t .. 16t+15] of chain c
  //   MFMA layout  : lane 16g+j  supplies B[k = g][col = chain j]; a 16 x 16 result has
  //                  col = chain j on the lane and rows 4g .. 4g+3 in its 4 registers
  // beta moves to the MFMA layout (and v back) through a per-wave LDS area, once per gradient; it is scaled by
  // -log2(e) on the way, so the forward product delivers the argument of v_exp_f32 directly.
  // Forward, per 16 rows: 16 steps of k = 4, step s taking column 16g+s from lane group g (the
  // order of a sum is free as long as A and B agree).  The residuals come out with rows 4g+r in
  // register r, which is exactly the B operand of the backward product if its step s takes
  // row 4g+s from lane group g: no movement between the two products.
  //
  // Tiles of 128 observations travel global memory -> LDS by LDS-DMA (see "tile image" above) into two buffers: while
  // tile n is multiplied, tile n+1 lands.  One workgroup barrier per tile, taken in front of the tile's LAST backward
  // block: by then all of the wave's LDS reads of tile n are complete (the operands of that block are in registers),
  // so the same barrier publishes tile n+1 (every wave has waited for its own pieces) and frees tile n's buffer for
  // tile n+2, whose LDS-DMA and the first operand reads of tile n+1 then go out under those last 16 MFMAs.
  // Tile n uses buffer (n + tiles) & 1, so the last tile of a gradient is always in buffer 1 and buffer 0 is free for
  // the kernels' row staging between gradients (STAGE_ALIAS above).
  static ARP_DEV void glds16(const float* sbase, uint32_t voff, uint32_t lds_dst) {
    unsigned keep;   // M0 = LDS destination of the wave's 1 KiB; compiler-reserved, so saved and restored in the statement
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
  }
  // four consecutive pieces: the instruction offset moves the global and the LDS address alike
  static ARP_DEV void glds16x4(const float* sbase, uint32_t voff, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
  }
  // this wave's share of tile n into buffer `buf`: 32 / W pieces of the design matrix, wave 0 also the outcomes
  ARP_DEV void issue_tile(int n, int buf, uint32_t tile_off, int wv, int lane) const {
    constexpr int PW = 32 / W_;
    static_assert(PW % 4 == 0, "pieces go out four at a time");
    const float* src = Xt + (size_t)n * kGermanImgTile + wv * (PW * 256);
    const uint32_t dst = tile_off + (uint32_t)buf * kBufStep + (uint32_t)wv * (PW * 1024u);
    const uint32_t voff = (uint32_t)lane * 16u;
#pragma unroll
    for (int p = 0; p < PW; p += 4) glds16x4(src + p * 256, voff, dst + (uint32_t)p * 1024u);
    if (wv == 0) glds16(Xt + (size_t)n * kGermanImgTile + 32 * 256, voff, tile_off + kYBase + (uint32_t)buf * kYBufB);
  }
  // LDS reads of the matrix-core path, pinned with inline asm a block ahead of their use (see the
  // note at Rows4: left to the scheduler they sink to the first use and the single wave per SIMD
  // waits out every LDS round trip with the matrix pipe idle).
  // A operand of the forward product for block BLK (rows 16 BLK ..): lane (g, j) takes row j, columns 16g .. 16g+15
  // (feature block g), chunk i from a_off[i] (the chunk XOR differs from lane to lane).
  template <int BLK>
  static ARP_DEV void issue_a(const uint32_t (&a_off)[4], v4f (&xa)[4]) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xa[0]) : "v"(a_off[0]), "n"(BLK * kBlkB));
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xa[1]) : "v"(a_off[1]), "n"(BLK * kBlkB));
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xa[2]) : "v"(a_off[2]), "n"(BLK * kBlkB));
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xa[3]) : "v"(a_off[3]), "n"(BLK * kBlkB));
  }
  template <int BLK>
  static ARP_DEV void issue_y(uint32_t y_off, v4f& y4) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(y4) : "v"(y_off), "n"(BLK * 64));
  }
  // A operand of the backward product for block BLK: lane (g, j) takes rows 4g+s (s < 4) from b_off[s], columns
  // 4j .. 4j+3: accumulator k of the product gets feature 4j+k on output row j, so a lane (g', chain) ends up with
  // features 16g' + 4r + k in acc[k][r].
  template <int BLK>
  static ARP_DEV void issue_b(const uint32_t (&b_off)[4], v4f (&xb)[4]) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xb[0]) : "v"(b_off[0]), "n"(BLK * kBlkB));
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xb[1]) : "v"(b_off[1]), "n"(BLK * kBlkB));
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xb[2]) : "v"(b_off[2]), "n"(BLK * kBlkB));
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xb[3]) : "v"(b_off[3]), "n"(BLK * kBlkB));
  }
  template <int IMM>
  static ARP_DEV void rd128(v4f& dst, uint32_t addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM)); }
  // every LDS read older than the NEWER youngest has landed (LDS reads return in order); the operands are tied to the
  // statement so that no use of them moves above it
  template <int NEWER>
  static ARP_DEV void wait_ops(v4f (&xa)[4], v4f (&xb)[4], v4f& y4) {
    asm volatile("s_waitcnt lgkmcnt(%9)"
                 : "+v"(xa[0]), "+v"(xa[1]), "+v"(xa[2]), "+v"(xa[3]), "+v"(xb[0]), "+v"(xb[1]), "+v"(xb[2]), "+v"(xb[3]),
                   "+v"(y4)
                 : "n"(NEWER));
  }
  // y - sigmoid(eta) of a lane's four rows (and their log density terms); z = -eta log2(e) comes out of the forward
  // product.  Written stage by stage over the four rows (a transcendental's result needs a wait state before its first
  // use, four independent ones in a row need none), additions two rows at a time (v_pk_add_f32 / v_pk_fma_f32).
  // LOGP: with ex = 2^-|z| and rc = 1 / (1 + ex) in [1/2, 1],
  //   sigmoid(eta) = 1/2 + copysign(rc - 1/2, eta)           (= rc for eta >= 0, 1 - rc otherwise)
  //   log2 of the Bernoulli term  y eta - softplus(eta)  =  log2(rc) - y z + min(z, 0)
  // accumulated in log2 units in two packed accumulators (lp2; scaled by ln 2 once per gradient).  MASK: the tile ends
  // inside the block range (last tile of the data set), rows >= `rows` are padding and must not count.
  // min(z, 0) is a compiler-visible v_med3_f32 in every path.  (Rounds 3 - 5 had an inline-asm v_min_f32 here; it reads the
  // matrix-core result directly, and the hazard recogniser pads nothing in front of an asm statement: scheduled right
  // behind a v_mfma_f32_16x16x32_bf16 it read the accumulator's OLD contents in the lanes written last -- a log density
  // wrong in lanes 0 - 15 only.  The f32 matrix-core path never showed it, but nothing kept the scheduler from it.)
  template <bool LOGP, bool MASK>
  static ARP_DEV void residuals(const v4f& z, const v4f& yv, int row, int rows, float (&w)[4], v2f (&lp2)[2]) {
    float ex[4], rc[4];
    const v2f one = v2f{1.0f, 1.0f};
    if (LOGP) {
#pragma unroll
      for (int r_ = 0; r_ < 4; ++r_) ex[r_] = __builtin_amdgcn_exp2f(-fabsf(z[r_]));
      const v2f d01 = v2f{ex[0], ex[1]} + one, d23 = v2f{ex[2], ex[3]} + one;
      rc[0] = __builtin_amdgcn_rcpf(d01[0]); rc[1] = __builtin_amdgcn_rcpf(d01[1]);
      rc[2] = __builtin_amdgcn_rcpf(d23[0]); rc[3] = __builtin_amdgcn_rcpf(d23[1]);
      float lg[4], mz[4];
#pragma unroll
      for (int r_ = 0; r_ < 4; ++r_) lg[r_] = __builtin_amdgcn_logf(rc[r_]);   // log2(1 / (1 + ex)) = -log2(1 + ex)
#pragma unroll
      for (int r_ = 0; r_ < 4; ++r_)    // min(z, 0); fminf() would add a v_max to quiet NaNs first
        mz[r_] = __builtin_amdgcn_fmed3f(z[r_], 0.0f, -3.4028234663852886e38f);
      const v2f y01 = v2f{yv[0], yv[1]}, y23 = v2f{yv[2], yv[3]};
      v2f t01 = vfma(-y01, v2f{z[0], z[1]}, v2f{mz[0], mz[1]}) + v2f{lg[0], lg[1]};
      v2f t23 = vfma(-y23, v2f{z[2], z[3]}, v2f{mz[2], mz[3]}) + v2f{lg[2], lg[3]};
      if (MASK) {
        t01 = v2f{row + 0 < rows ? t01[0] : 0.0f, row + 1 < rows ? t01[1] : 0.0f};
        t23 = v2f{row + 2 < rows ? t23[0] : 0.0f, row + 3 < rows ? t23[1] : 0.0f};
      }
      lp2[0] += t01; lp2[1] += t23;
      const v2f half = v2f{0.5f, 0.5f};
      const v2f h01 = v2f{rc[0], rc[1]} - half, h23 = v2f{rc[2], rc[3]} - half;   // |sigmoid - 1/2|
      // sign(eta) = -sign(z): magnitude from h, sign bit from z, subtracted instead of added
      const v2f c01 = v2f{__builtin_copysignf(h01[0], z[0]), __builtin_copysignf(h01[1], z[1])};
      const v2f c23 = v2f{__builtin_copysignf(h23[0], z[2]), __builtin_copysignf(h23[1], z[3])};
      const v2f w01 = (y01 - half) + c01, w23 = (y23 - half) + c23;
      w[0] = w01[0]; w[1] = w01[1]; w[2] = w23[0]; w[3] = w23[1];
    } else {
      // 1 / (1 + e^-eta): e^-eta = inf gives 0, no NaN
#pragma unroll
      for (int r_ = 0; r_ < 4; ++r_) ex[r_] = __builtin_amdgcn_exp2f(z[r_]);
      const v2f d01 = v2f{ex[0], ex[1]} + one, d23 = v2f{ex[2], ex[3]} + one;
      rc[0] = __builtin_amdgcn_rcpf(d01[0]); rc[1] = __builtin_amdgcn_rcpf(d01[1]);
      rc[2] = __builtin_amdgcn_rcpf(d23[0]); rc[3] = __builtin_amdgcn_rcpf(d23[1]);
      const v2f w01 = v2f{yv[0], yv[1]} - v2f{rc[0], rc[1]}, w23 = v2f{yv[2], yv[3]} - v2f{rc[2], rc[3]};
      w[0] = w01[0]; w[1] = w01[1]; w[2] = w23[0]; w[3] = w23[1];
    }
  }
  static constexpr int kNB = kGermanTileRows / 16;   // 16-row blocks per tile

  // The software pipeline over the blocks of a tile.  Operands of block m live in xa[m & 1] (forward), xb[m & 1]
  // (backward), y4[m & 1] (outcomes).  Phase I multiplies FORWARD block I+1 (one accumulation chain) interleaved 1:1
  // with BACKWARD block I (four chains): a dependent MFMA would wait 40 cycles for its predecessor, with an
  // independent one in between both issue every 32.  Vector instructions cannot hide under f32 MFMAs (same datapath,
  // tools/mfma_overlap.hip), so everything else is kept minimal and placed after the MFMAs: the LDS reads of the
  // operands two phases ahead (into the registers the MFMAs just consumed; their issue also covers the MFMA result
  // latency), the four residuals of block I+1, then the outcome read.
  //   reads issued at the end of phase I: A(I+3), B(I+2) | residuals | Y(I+3)      (9 instructions when all exist)
  template <bool LOGP, bool MASK, int I>
  static ARP_DEV void phase(const uint32_t (&a_off)[4], const uint32_t (&b_off)[4], uint32_t y_off, int gk, int rows,
                            const float (&bB)[16], v4f (&xa)[2][4], v4f (&xb)[2][4], v4f (&y4)[2], float (&w)[4],
                            v4f (&acc)[4], v2f (&lp)[2]) {
    constexpr int fa = (I + 1) & 1, bb = I & 1;
    // younger reads at this point: those issued at the end of phase I-1
    constexpr int newer = (I + 2 < kNB ? 5 : 0) + (I + 1 < kNB ? 4 : 0);
    wait_ops<newer>(xa[fa], xb[bb], y4[fa]);
    __builtin_amdgcn_sched_barrier(0);
    v4f e = v4f{0.0f, 0.0f, 0.0f, 0.0f};
    // Quarter q = k-steps 4q .. 4q+3: forward chunk q and backward row q.  Once its MFMAs are issued those two registers
    // are free, and the reads that refill them (a single wave can issue one instruction of any kind every 4 cycles, so
    // they go into the 32-cycle shadow of an MFMA instead of behind the whole block).
#pragma unroll
    for (int q_ = 0; q_ < 4; ++q_) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        e = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[fa][q_][k], bB[4 * q_ + k], e, 0, 0, 0);
        acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[bb][q_][k], w[q_], acc[k], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (I + 3 < kNB) {
        if (q_ == 0) rd128<(I + 3) * kBlkB>(xa[fa][0], a_off[0]);
        if (q_ == 1) rd128<(I + 3) * kBlkB>(xa[fa][1], a_off[1]);
        if (q_ == 2) rd128<(I + 3) * kBlkB>(xa[fa][2], a_off[2]);
        if (q_ == 3) rd128<(I + 3) * kBlkB>(xa[fa][3], a_off[3]);
      }
      if constexpr (I + 2 < kNB) {
        if (q_ == 0) rd128<(I + 2) * kBlkB>(xb[bb][0], b_off[0]);
        if (q_ == 1) rd128<(I + 2) * kBlkB>(xb[bb][1], b_off[1]);
        if (q_ == 2) rd128<(I + 2) * kBlkB>(xb[bb][2], b_off[2]);
        if (q_ == 3) rd128<(I + 2) * kBlkB>(xb[bb][3], b_off[3]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    residuals<LOGP, MASK>(e, y4[fa], 16 * (I + 1) + 4 * gk, rows, w, lp);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (I + 3 < kNB) issue_y<I + 3>(y_off, y4[fa]);
    __builtin_amdgcn_sched_barrier(0);
  }
  template <bool LOGP, bool MASK, int I>
  static ARP_DEV void phases(const uint32_t (&a_off)[4], const uint32_t (&b_off)[4], uint32_t y_off, int gk, int rows,
                             const float (&bB)[16], v4f (&xa)[2][4], v4f (&xb)[2][4], v4f (&y4)[2], float (&w)[4],
                             v4f (&acc)[4], v2f (&lp)[2]) {
    if constexpr (I + 1 < kNB) {
      phase<LOGP, MASK, I>(a_off, b_off, y_off, gk, rows, bB, xa, xb, y4, w, acc, lp);
      phases<LOGP, MASK, I + 1>(a_off, b_off, y_off, gk, rows, bB, xa, xb, y4, w, acc, lp);
    }
  }
  // the first reads of a tile: A(0), Y(0), A(1), Y(1), B(0)
  static ARP_DEV void first_reads(const uint32_t (&a_off)[4], const uint32_t (&b_off)[4], uint32_t y_off,
                                  v4f (&xa)[2][4], v4f (&xb)[2][4], v4f (&y4)[2]) {
    issue_a<0>(a_off, xa[0]);
    issue_y<0>(y_off, y4[0]);
    issue_a<1>(a_off, xa[1]);
    issue_y<1>(y_off, y4[1]);
    issue_b<0>(b_off, xb[0]);
    __builtin_amdgcn_sched_barrier(0);
  }
  // forward block 0 on its own (two chains: nothing to interleave with), then as the end of a phase
  template <bool LOGP, bool MASK>
  static ARP_DEV void head(const uint32_t (&a_off)[4], const uint32_t (&b_off)[4], uint32_t y_off, int gk, int rows,
                           const float (&bB)[16], v4f (&xa)[2][4], v4f (&xb)[2][4], v4f (&y4)[2], float (&w)[4], v2f (&lp)[2]) {
    asm volatile("s_waitcnt lgkmcnt(9)" : "+v"(xa[0][0]), "+v"(xa[0][1]), "+v"(xa[0][2]), "+v"(xa[0][3]), "+v"(y4[0]));
    __builtin_amdgcn_sched_barrier(0);
    v4f e0 = v4f{0.0f, 0.0f, 0.0f, 0.0f}, e1 = v4f{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int q_ = 0; q_ < 4; ++q_) {
      e0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[0][q_][0], bB[4 * q_], e0, 0, 0, 0);
      e1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[0][q_][1], bB[4 * q_ + 1], e1, 0, 0, 0);
      e0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[0][q_][2], bB[4 * q_ + 2], e0, 0, 0, 0);
      e1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[0][q_][3], bB[4 * q_ + 3], e1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (q_ == 0) { rd128<2 * kBlkB>(xa[0][0], a_off[0]); rd128<kBlkB>(xb[1][0], b_off[0]); }
      if (q_ == 1) { rd128<2 * kBlkB>(xa[0][1], a_off[1]); rd128<kBlkB>(xb[1][1], b_off[1]); }
      if (q_ == 2) { rd128<2 * kBlkB>(xa[0][2], a_off[2]); rd128<kBlkB>(xb[1][2], b_off[2]); }
      if (q_ == 3) { rd128<2 * kBlkB>(xa[0][3], a_off[3]); rd128<kBlkB>(xb[1][3], b_off[3]); }
      __builtin_amdgcn_sched_barrier(0);
    }
    residuals<LOGP, MASK>(e0 + e1, y4[0], 4 * gk, rows, w, lp);
    __builtin_amdgcn_sched_barrier(0);
    issue_y<2>(y_off, y4[0]);
    __builtin_amdgcn_sched_barrier(0);
  }
  // backward block kNB-1, the last MFMAs of a tile, one quarter (row q of the block) at a time
  static ARP_DEV void tail_quarter(const v4f& xrow, float wq, v4f (&acc)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(xrow[k], wq, acc[k], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }

  // Start of a gradient on the matrix-core path: the staging block (buffer 0) and the previous gradient's tiles are no
  // longer in use by any wave after the barrier; tile 0 then travels while the prior part of the gradient is computed.
  ARP_DEV void first_tile() const {
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if constexpr (BF3_) {
      const int ntb = thi() - tlo();
      if (resident()) return;
      __syncthreads();
      if (!PART_ || ntb > 0) issue_tile_bf3(tlo(), ntb & 1, lds_offset(tile_mem()), wv, threadIdx.x & 63);
      return;
    }
    const int nt = thi() - tlo();
    if (resident()) return;
    __syncthreads();
    if (!PART_ || nt > 0) issue_tile(tlo(), nt & 1, lds_offset(tile_mem()), wv, threadIdx.x & 63);
  }

  template <bool LOGP>
  ARP_DEV float likelihood_mfma(const float (&beta)[NLS], float (&v)[NLS]) const {
    static_assert(NLS == 16, "K = 4 owns 16 features per lane");
    static_assert(kNB >= 4 && (kNB & 1) == 0, "the pipeline's register parities assume an even number of blocks");
    float* tile = tile_mem();
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* xch = tile + kXchBase + wv * kXch;
    float* lpx = xch + 16 * kXchStride;
    const int c = lane >> 2, t = lane & 3;    // state layout (t == slot)
    const int gk = lane >> 4, j = lane & 15;  // MFMA layout
    const uint32_t tile_off = lds_offset(tile);
    const int t0 = tlo(), nt = thi() - t0;   // this workgroup's tiles (all of them outside the VI kernel)
    int buf = nt & 1;
    ARP_T0(tt);   // tile t0 is on its way (first_tile)

    float4* own = reinterpret_cast<float4*>(xch + c * kXchStride + 16 * t);
    const float4* mine = reinterpret_cast<const float4*>(xch + j * kXchStride + 16 * gk);
    constexpr float kNegLog2e = -1.4426950408889634f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      own[i] = make_float4(kNegLog2e * beta[4 * i], kNegLog2e * beta[4 * i + 1], kNegLog2e * beta[4 * i + 2],
                           kNegLog2e * beta[4 * i + 3]);
    __builtin_amdgcn_wave_barrier();
    float bB[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 f = mine[i];
      bB[4 * i] = f.x; bB[4 * i + 1] = f.y; bB[4 * i + 2] = f.z; bB[4 * i + 3] = f.w;
    }
    __builtin_amdgcn_wave_barrier();
    // operand addresses in the current buffer (see "tile image"): chunk c of row r sits at chunk c ^ (r & 11)
    uint32_t a_off[4], b_off[4], y_off;
    {
      const uint32_t xb0 = tile_off + (uint32_t)buf * kBufStep;
      const uint32_t uj = (uint32_t)j, ug = (uint32_t)gk;
      // forward: row j, chunks 4g + i
      const uint32_t abase = xb0 + uj * 256u + (((ug << 2) ^ (uj & 8u)) << 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) a_off[i] = abase + ((((uint32_t)i) ^ (uj & 3u)) << 4);
      // backward: rows 4g + s, chunk j;  (4g + s) & 11 = s | (g & 2) << 2
      const uint32_t bbase = xb0 + ug * 1024u;
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) b_off[s_] = bbase + (uint32_t)s_ * 256u + ((uj ^ (uint32_t)s_ ^ ((ug & 2u) << 2)) << 4);
      y_off = tile_off + kYBase + (uint32_t)buf * kYBufB + ug * 16u;
    }
    v4f acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
    v2f lp2[2] = {v2f{0.0f, 0.0f}, v2f{0.0f, 0.0f}};   // log density of the lane's rows, log2 units
    v4f xa[2][4], y4[2];
    v4f xb[2][4];
    float w[4];
    // tile 0 has landed everywhere; tile 1 into the other buffer
    if (!resident()) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (nt > 1) issue_tile(t0 + 1, buf ^ 1, tile_off, wv, lane);
    }
    ARP_T(2, tt);
    if (!PART_ || nt > 0) first_reads(a_off, b_off, y_off, xa, xb, y4);
    for (int n = 0; n < nt; ++n) {
      const int rows = min(kRows, N - (t0 + n) * kRows);   // the image is zero filled up to the tile's last row
      if (!LOGP || rows == kRows) {     // only the log density cares about padding rows, and only the last tile has any
        head<LOGP, false>(a_off, b_off, y_off, gk, rows, bB, xa, xb, y4, w, lp2);
        phases<LOGP, false, 0>(a_off, b_off, y_off, gk, rows, bB, xa, xb, y4, w, acc, lp2);
      } else {
        head<LOGP, true>(a_off, b_off, y_off, gk, rows, bB, xa, xb, y4, w, lp2);
        phases<LOGP, true, 0>(a_off, b_off, y_off, gk, rows, bB, xa, xb, y4, w, acc, lp2);
      }
      // every LDS read of this tile has landed (the operands of the last backward block among them)
      constexpr int lb = (kNB - 1) & 1;
      wait_ops<0>(xa[0], xb[lb], y4[0]);
      if (n + 1 < nt) {
        // tile n+1 is in LDS for every wave and nobody reads tile n any more: its buffer takes tile n+2, and the first
        // reads of tile n+1 go out in the shadow of the last MFMAs of tile n
        if (!resident()) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
        }
        tail_quarter(xb[lb][0], w[0], acc);
        if (n + 2 < nt) issue_tile(t0 + n + 2, buf, tile_off, wv, lane);
        const uint32_t dx = buf ? (uint32_t)-kBufStep : (uint32_t)kBufStep, dy = buf ? (uint32_t)-kYBufB : (uint32_t)kYBufB;
        buf ^= 1;
        __builtin_amdgcn_sched_barrier(0);
        tail_quarter(xb[lb][1], w[1], acc);
#pragma unroll
        for (int i = 0; i < 4; ++i) a_off[i] += dx;
        y_off += dy;
        __builtin_amdgcn_sched_barrier(0);
        issue_a<0>(a_off, xa[0]);
        issue_y<0>(y_off, y4[0]);
        __builtin_amdgcn_sched_barrier(0);
        tail_quarter(xb[lb][2], w[2], acc);
#pragma unroll
        for (int i = 0; i < 4; ++i) b_off[i] += dx;
        __builtin_amdgcn_sched_barrier(0);
        issue_a<1>(a_off, xa[1]);
        issue_y<1>(y_off, y4[1]);
        issue_b<0>(b_off, xb[0]);
        __builtin_amdgcn_sched_barrier(0);
        tail_quarter(xb[lb][3], w[3], acc);
      } else {
#pragma unroll
        for (int q_ = 0; q_ < 4; ++q_) tail_quarter(xb[lb][q_], w[q_], acc);
      }
    }
    ARP_T(3, tt);
    // v back to the state layout: lane (g, j) holds v[16g + 4r + k] of chain j in acc[k][r]
#pragma unroll
    for (int r_ = 0; r_ < 4; ++r_)
      *reinterpret_cast<float4*>(xch + j * kXchStride + 16 * gk + 4 * r_) = make_float4(acc[0][r_], acc[1][r_], acc[2][r_], acc[3][r_]);
    float lp = 0.0f;
    if (LOGP) lpx[lane] = 0.6931471805599453f * ((lp2[0][0] + lp2[0][1]) + (lp2[1][0] + lp2[1][1]));
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 f = own[i];
      v[4 * i] = f.x; v[4 * i + 1] = f.y; v[4 * i + 2] = f.z; v[4 * i + 3] = f.w;
    }
    if (LOGP) lp = (lpx[c] + lpx[16 + c]) + (lpx[32 + c] + lpx[48 + c]);
    __builtin_amdgcn_wave_barrier();
    ARP_T(4, tt);
    return lp;
  }

  // ---- the bf16 x 3 likelihood (see the tile image at the top of the file) ----
  typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
  typedef uint32_t u4v __attribute__((ext_vector_type(4)));
  // this wave's share of tile n of the bf16 image into buffer `buf`: a contiguous run of ceil(23 / W) pieces, four per
  // M0 set-up (the instruction offset moves the global and the LDS address alike), the rest one by one
  ARP_DEV void issue_tile_bf3(int n, int buf, uint32_t tile_off, int wv, int lane) const {
    constexpr int PW = (kBf3Pieces + W_ - 1) / W_;
    const int p0 = wv * PW;
    const float* src = Xb + (size_t)n * kBf3ImgTile + p0 * 256;
    const uint32_t dst = tile_off + (uint32_t)buf * kBufStep + (uint32_t)p0 * 1024u;
    const uint32_t voff = (uint32_t)lane * 16u;
#pragma unroll
    for (int p = 0; p + 4 <= PW; p += 4)
      if (p0 + p + 4 <= kBf3Pieces) glds16x4(src + p * 256, voff, dst + (uint32_t)p * 1024u);
      else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (p0 + p + k < kBf3Pieces) glds16(src + (p + k) * 256, voff, dst + (uint32_t)(p + k) * 1024u);
      }
#pragma unroll
    for (int p = PW / 4 * 4; p < PW; ++p)
      if (p0 + p < kBf3Pieces) glds16(src + p * 256, voff, dst + (uint32_t)p * 1024u);
  }
  // two bf16 (the high halves of two f32 bit patterns) in one register: element 2k low, 2k + 1 high
  static ARP_DEV uint32_t bf_pack(uint32_t lo, uint32_t hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }
  // eight f32 values -> their three bf16 fragments
  static ARP_DEV void bf3_frags(const float (&x)[8], bf8 (&f)[3]) {
    uint32_t h[8], m[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bf3_split(x[i], h[i], m[i], l[i]);
    u4v fh, fm, fl;
#pragma unroll
    for (int i = 0; i < 4; ++i) { fh[i] = bf_pack(h[2 * i], h[2 * i + 1]); fm[i] = bf_pack(m[2 * i], m[2 * i + 1]); fl[i] = bf_pack(l[2 * i], l[2 * i + 1]); }
    f[0] = __builtin_bit_cast(bf8, fh); f[1] = __builtin_bit_cast(bf8, fm); f[2] = __builtin_bit_cast(bf8, fl);
  }
  // operand fragments travel as four dwords; the LDS reads are inline asm (they stay where they are written) and every
  // use is tied to the wait that completes them (frag_wait: the compiler sees no other dependency between the two)
  static ARP_DEV u4v lds_frag(uint32_t addr) {
    u4v r;
    asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr));
    return r;
  }
  static ARP_DEV void frag_wait(u4v& a, u4v& b, u4v& c, u4v& d, u4v& e) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e));
  }
  static ARP_DEV void frag_wait(u4v& a, u4v& b, u4v& c, u4v& d) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  }
  static ARP_DEV v4f mm(u4v a, bf8 b, v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), b, c, 0, 0, 0);
  }

  template <bool LOGP>
  ARP_DEV float likelihood_bf3(const float (&beta)[NLS], float (&v)[NLS]) const {
    static_assert(NLS == 16, "K = 4 owns 16 features per lane");
    float* tile = tile_mem();
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float* xch = tile + kXchBase + wv * kXch;
    float* lpx = xch + 16 * kXchStride;
    const int c = lane >> 2, t = lane & 3;    // state layout (t == slot)
    const int g = lane >> 4, j = lane & 15;   // matrix-core layout: lane group g, row / chain j
    const uint32_t tile_off = lds_offset(tile);
    const int t0 = tlo(), nt = thi() - t0;   // this workgroup's tiles (all of them outside the VI kernel)
    const bool res = resident();              // (VI kernel) the part's one or two tiles are in the buffers for good
    int buf = nt & 1;
    ARP_T0(tt);

    // beta, scaled by -log2 e, from the state layout to [chain][feature] in the wave's exchange area ...
    float4* own = reinterpret_cast<float4*>(xch + c * kXchStride + 16 * t);
    constexpr float kNegLog2e = -1.4426950408889634f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      own[i] = make_float4(kNegLog2e * beta[4 * i], kNegLog2e * beta[4 * i + 1], kNegLog2e * beta[4 * i + 2],
                           kNegLog2e * beta[4 * i + 3]);
    __builtin_amdgcn_wave_barrier();
    // ... and from there into the B fragments of the forward product: pieces h, m, l of features 32 kh + 8 g .. + 7 of
    // chain j, and the split columns' [bh ; bm ; bh ; 0] for the extra K = 32 step
    bf8 Bb[2][3], Bba;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      const float4* src = reinterpret_cast<const float4*>(xch + j * kXchStride + 32 * kh + 8 * g);
      const float4 f0 = src[0], f1 = src[1];
      const float x8[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
      bf3_frags(x8, Bb[kh]);
    }
    {
      float x8[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) x8[q] = sidx[q] >= 0 ? xch[j * kXchStride + sidx[q]] : 0.0f;
      bf8 f3[3];
      bf3_frags(x8, f3);
      const u4v z4 = u4v{0u, 0u, 0u, 0u};
      Bba = g == 1 ? f3[1] : (g == 3 ? __builtin_bit_cast(bf8, z4) : f3[0]);
    }
    __builtin_amdgcn_wave_barrier();

    ARP_T(1, tt);
    v4f acc[4], accA;
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
    accA = v4f{0.0f, 0.0f, 0.0f, 0.0f};
    v2f lp2[2] = {v2f{0.0f, 0.0f}, v2f{0.0f, 0.0f}};
    // operand addresses relative to the buffer: row / feature / output row j, chunk g (XOR-permuted per row)
    const uint32_t uj = (uint32_t)j, ug = (uint32_t)g;
    const uint32_t a_h0 = kBf3XhF + uj * 128u + (((ug) ^ ((uj >> 1) & 7u)) << 4);          // kh = 0: chunk g
    const uint32_t a_h1 = kBf3XhF + uj * 128u + (((4u + ug) ^ ((uj >> 1) & 7u)) << 4);     // kh = 1: chunk 4 + g
    const uint32_t a_a = kBf3XaF + uj * 64u + ((ug ^ ((uj >> 2) & 3u)) << 4);
    const uint32_t b_h = kBf3XhB + uj * 64u + ((ug ^ ((uj >> 2) & 3u)) << 4);              // + fb * 1024 + s * 4096
    const uint32_t b_a = kBf3XaB + uj * 64u + ((ug ^ ((uj >> 2) & 3u)) << 4);              // + s * 1024
    const uint32_t y_o = kBf3Y + ug * 16u;                                                 // + block * 64

    // The pipeline over the 64-observation tiles.  Per tile: the 16 forward reads (three fragments and the outcomes of
    // each of the four 16-observation blocks) were issued while the previous tile's last matrix-core instructions ran;
    // the 10 backward reads go out first thing and land under the forward products.  Forward = 4 x 7 matrix-core
    // instructions, each block's residuals (vector pipe: two transcendentals per observation) free to run beside the
    // next block's products; backward = 2 x 14, the second k-step's after the hand-over so that they cover the latency of
    // the next tile's forward reads.  ONE workgroup barrier per tile, at the hand-over: every LDS read of the tile has
    // landed by then (the operands were waited for), so the barrier frees its buffer for tile n + 2 and publishes n + 1.
    u4v F[4][3], Y[4], Bk[2][5];
    // operand reads: one address register per operand kind (buffer base + the lane's part), the block / k-step / feature
    // block as the instruction's immediate offset
    auto rd = [](u4v& dst, uint32_t addr, auto off_tag) {
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(decltype(off_tag)::value));
    };
#define ARP_OFF(x) std::integral_constant<int, (x)>{}
    auto fwd_reads = [&](uint32_t base) {
      const uint32_t p0 = base + a_h0, p1 = base + a_h1, pa = base + a_a, py = base + y_o;
      rd(F[0][0], p0, ARP_OFF(0)); rd(F[0][1], p1, ARP_OFF(0)); rd(F[0][2], pa, ARP_OFF(0)); rd(Y[0], py, ARP_OFF(0));
      rd(F[1][0], p0, ARP_OFF(2048)); rd(F[1][1], p1, ARP_OFF(2048)); rd(F[1][2], pa, ARP_OFF(1024)); rd(Y[1], py, ARP_OFF(64));
      rd(F[2][0], p0, ARP_OFF(4096)); rd(F[2][1], p1, ARP_OFF(4096)); rd(F[2][2], pa, ARP_OFF(2048)); rd(Y[2], py, ARP_OFF(128));
      rd(F[3][0], p0, ARP_OFF(6144)); rd(F[3][1], p1, ARP_OFF(6144)); rd(F[3][2], pa, ARP_OFF(3072)); rd(Y[3], py, ARP_OFF(192));
    };
    auto bwd_reads = [&](uint32_t base) {
      const uint32_t ph = base + b_h, pa = base + b_a;
      rd(Bk[0][0], ph, ARP_OFF(0)); rd(Bk[0][1], ph, ARP_OFF(1024)); rd(Bk[0][2], ph, ARP_OFF(2048)); rd(Bk[0][3], ph, ARP_OFF(3072));
      rd(Bk[0][4], pa, ARP_OFF(0));
      rd(Bk[1][0], ph, ARP_OFF(4096)); rd(Bk[1][1], ph, ARP_OFF(5120)); rd(Bk[1][2], ph, ARP_OFF(6144)); rd(Bk[1][3], ph, ARP_OFF(7168));
      rd(Bk[1][4], pa, ARP_OFF(1024));
    };
#undef ARP_OFF
    // product i (0 .. 6) of forward block b: X_h's two halves against beta's three pieces, then the split columns' step
    auto fmm = [&](int b, int i, v4f e) {
      return i < 6 ? mm(F[b][i & 1], Bb[i & 1][i >> 1], e) : mm(F[b][2], Bba, e);
    };
    // product i (0 .. 13) of backward k-step s_: four feature blocks against the residuals' three pieces, then the split
    // columns' 16 extra rows against the two leading pieces
    auto bmm = [&](int s_, int i, const bf8 (&Bw)[3]) {
      if (i < 12) acc[i / 3] = mm(Bk[s_][i / 3], Bw[i % 3], acc[i / 3]);
      else accA = mm(Bk[s_][4], Bw[i - 12], accA);
    };
    // residuals of a block WITHOUT the log density, in six stages of two vector instructions each (they ride in the
    // shadows of the next block's products): y - 1 / (1 + 2^z) for the lane's four observations
    struct Res { float ex[4], rc[4]; v2f d01, d23; };
    auto res_stage = [&](int k, Res& R, const v4f& z, const v4f& yv, float (&wo)[4]) {
      const v2f one = v2f{1.0f, 1.0f};
      if (k == 0) { R.ex[0] = __builtin_amdgcn_exp2f(z[0]); R.ex[1] = __builtin_amdgcn_exp2f(z[1]); }
      if (k == 1) { R.ex[2] = __builtin_amdgcn_exp2f(z[2]); R.ex[3] = __builtin_amdgcn_exp2f(z[3]); }
      if (k == 2) { R.d01 = v2f{R.ex[0], R.ex[1]} + one; R.d23 = v2f{R.ex[2], R.ex[3]} + one; }
      if (k == 3) { R.rc[0] = __builtin_amdgcn_rcpf(R.d01[0]); R.rc[1] = __builtin_amdgcn_rcpf(R.d01[1]); }
      if (k == 4) { R.rc[2] = __builtin_amdgcn_rcpf(R.d23[0]); R.rc[3] = __builtin_amdgcn_rcpf(R.d23[1]); }
      if (k == 5) {
        const v2f w01 = v2f{yv[0], yv[1]} - v2f{R.rc[0], R.rc[1]}, w23 = v2f{yv[2], yv[3]} - v2f{R.rc[2], R.rc[3]};
        wo[0] = w01[0]; wo[1] = w01[1]; wo[2] = w23[0]; wo[3] = w23[1];
      }
    };
#define ARP_OFF(x) std::integral_constant<int, (x)>{}
    // the backward operand reads of the current tile, one per slot (issued beside the forward products)
    auto bwd_read_slot = [&](int k, uint32_t base) {
      const uint32_t ph = base + b_h, pa = base + b_a;
      if (k == 0) rd(Bk[0][0], ph, ARP_OFF(0));
      if (k == 1) rd(Bk[0][1], ph, ARP_OFF(1024));
      if (k == 2) rd(Bk[0][2], ph, ARP_OFF(2048));
      if (k == 3) rd(Bk[0][3], ph, ARP_OFF(3072));
      if (k == 4) rd(Bk[0][4], pa, ARP_OFF(0));
      if (k == 5) rd(Bk[1][0], ph, ARP_OFF(4096));
      if (k == 6) rd(Bk[1][1], ph, ARP_OFF(5120));
      if (k == 7) rd(Bk[1][2], ph, ARP_OFF(6144));
      if (k == 8) rd(Bk[1][3], ph, ARP_OFF(7168));
      if (k == 9) rd(Bk[1][4], pa, ARP_OFF(1024));
    };
#undef ARP_OFF

    // tile 0 has landed everywhere; tile 1 into the other buffer
    if (!res) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (nt > 1) issue_tile_bf3(t0 + 1, buf ^ 1, tile_off, wv, lane);
    }
    if (!PART_ || nt > 0) fwd_reads(tile_off + (uint32_t)buf * kBufStep);
    ARP_T(2, tt);

    // The tile loop, written slot by slot: every matrix-core instruction is followed by the one or two other
    // instructions that ride in its shadow (it holds vector issue for 8 of its 16 cycles; LDS reads and LDS-DMA issue
    // beside it), and a scheduling barrier keeps the slot together -- left to itself the scheduler issues all 28
    // products of a phase back to back and the vector work, the 26 operand reads and the six LDS-DMA pieces (60 - 100
    // cycles of issue each) behind them: 2 800 cycles per tile against 900 of matrix-core time.
    //   forward : 4 blocks x 7 products; slots carry the 10 backward reads of this tile and the previous block's residuals
    //   hand-over: this tile's reads are complete, the next tile has landed: ONE barrier
    //   backward: 2 k-steps x 14 products; slots carry the LDS-DMA of tile n + 2, the second k-step's split and the 16
    //             forward reads of tile n + 1
    for (int n = 0; n < nt; ++n) {
      const int rows = min(kBf3Rows, N - (t0 + n) * kBf3Rows);
      const uint32_t base = tile_off + (uint32_t)buf * kBufStep;
      ARP_T(3, tt);
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(F[0][0]), "+v"(F[0][1]), "+v"(F[0][2]), "+v"(F[1][0]), "+v"(F[1][1]), "+v"(F[1][2]), "+v"(F[2][0]),
                     "+v"(F[2][1]), "+v"(F[2][2]), "+v"(F[3][0]), "+v"(F[3][1]), "+v"(F[3][2]), "+v"(Y[0]), "+v"(Y[1]),
                     "+v"(Y[2]), "+v"(Y[3]));
      ARP_T(9, tt);
      float w[4][4];
      if constexpr (LOGP) {
        // the closing pass of a trajectory (one gradient in L): the log density's extra vector work is left to the scheduler
        bwd_reads(base);
        auto forward = [&](auto mask_tag) {
          constexpr bool MASK = decltype(mask_tag)::value;
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            v4f e = v4f{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int i = 0; i < 7; ++i) e = fmm(b, i, e);
            residuals<LOGP, MASK>(e, __builtin_bit_cast(v4f, Y[b]), 16 * b + 4 * g, rows, w[b], lp2);
          }
        };
        if (rows == kBf3Rows) forward(std::false_type{});
        else forward(std::true_type{});
      } else {
        v4f e[4];
        Res R;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          e[b] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
          for (int i = 0; i < 7; ++i) {
            e[b] = fmm(b, i, e[b]);
            if (b < 2 && i < 5) bwd_read_slot(5 * b + i, base);
            if (b > 0 && i < 6) res_stage(i, R, e[b - 1], __builtin_bit_cast(v4f, Y[b - 1]), w[b - 1]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) res_stage(i, R, e[3], __builtin_bit_cast(v4f, Y[3]), w[3]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) asm volatile("" : "+v"(w[b][0]), "+v"(w[b][1]), "+v"(w[b][2]), "+v"(w[b][3]));
      ARP_T(10, tt);
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(Bk[0][0]), "+v"(Bk[0][1]), "+v"(Bk[0][2]), "+v"(Bk[0][3]), "+v"(Bk[0][4]), "+v"(Bk[1][0]),
                     "+v"(Bk[1][1]), "+v"(Bk[1][2]), "+v"(Bk[1][3]), "+v"(Bk[1][4]));
      ARP_T(11, tt);
      if (n + 1 < nt) {
        // Hand-over, between the two products: every LDS read of tile n has landed (all its operands are in registers)
        // and this wave's share of tile n + 1 too; after the barrier tile n + 1 is complete for every wave and tile n's
        // buffer is free for tile n + 2.  The next tile's forward reads go out at once and land under the backward product.
        if (!res) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          if (n + 2 < nt) issue_tile_bf3(t0 + n + 2, buf, tile_off, wv, lane);
        }
        buf ^= 1;
        fwd_reads(tile_off + (uint32_t)buf * kBufStep);
      }
      // backward: a lane's eight residuals of blocks 2 s and 2 s + 1 are its B fragment of k-step s; the second k-step's
      // split rides in the shadows of the first one's products (scheduling groups below)
      bf8 Bw0[3], Bw1[3];
      {
        const float x0[8] = {w[0][0], w[0][1], w[0][2], w[0][3], w[1][0], w[1][1], w[1][2], w[1][3]};
        bf3_frags(x0, Bw0);
      }
#pragma unroll
      for (int i = 0; i < 14; ++i) bmm(0, i, Bw0);
      {
        const float x1[8] = {w[2][0], w[2][1], w[2][2], w[2][3], w[3][0], w[3][1], w[3][2], w[3][3]};
        bf3_frags(x1, Bw1);
      }
#pragma unroll
      for (int i = 0; i < 14; ++i) bmm(1, i, Bw1);
      __builtin_amdgcn_sched_group_barrier(0x002, 46, 1);
#pragma unroll
      for (int i = 0; i < 14; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 1);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 14, 1);
    }
    ARP_T(3, tt);
    // v back to the state layout: lane (g, j) holds v[16 fb + 4 g + r] of chain j in acc[fb][r]; the split columns'
    // m and l parts (accA: output row 4 g + r = split column q (rows 0 - 7) / q + 8 (rows 8 - 15)) are added in LDS,
    // the two halves of the wave one after the other (LDS operations of a wave execute in order)
#pragma unroll
    for (int fb = 0; fb < 4; ++fb)
      *reinterpret_cast<float4*>(xch + j * kXchStride + 16 * fb + 4 * g) = make_float4(acc[fb][0], acc[fb][1], acc[fb][2], acc[fb][3]);
    float lp = 0.0f;
    if (LOGP) lpx[lane] = 0.6931471805599453f * ((lp2[0][0] + lp2[0][1]) + (lp2[1][0] + lp2[1][1]));
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if ((g >> 1) == half) {
#pragma unroll
        for (int r_ = 0; r_ < 4; ++r_) {
          // split column 4 (g & 1) + r_: both candidates by CONSTANT index (a run-time index into the lane object would
          // pin the whole object -- a[], b[], the pointers -- in scratch or LDS instead of registers)
          const int sq = (g & 1) ? sidx[4 + r_] : sidx[r_];
          if (sq >= 0) xch[j * kXchStride + sq] += accA[r_];
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 f = own[i];
      v[4 * i] = f.x; v[4 * i + 1] = f.y; v[4 * i + 2] = f.z; v[4 * i + 3] = f.w;
    }
    if (LOGP) lp = (lpx[c] + lpx[16 + c]) + (lpx[32 + c] + lpx[48 + c]);
    __builtin_amdgcn_wave_barrier();
    ARP_T(4, tt);
    return lp;
  }

  template <bool LOGP>
  ARP_DEV float grad(const float (&q)[ND], float (&g)[ND]) const {
    ARP_T0(tg);
    if constexpr (K == 4) first_tile();
    const float ols = c0 * q[0];
    float beta[NLS], v[NLS], bls[NLS], r[NLS];
#pragma unroll
    for (int i = 0; i < NLS; ++i) {
      r[i] = fmaf(-a[i], ols, q[NG + i]);
      bls[i] = r[i] + ols;
      beta[i] = fast_exp((1.0f - b[i]) * bls[i]) * q[NG + NLS + i];   // padding: q = 0 -> beta = 0
      v[i] = 0.0f;
    }
    float lp;
    ARP_T(0, tg);
    if constexpr (K == 8) lp = likelihood_k8<LOGP>(beta, v);
    else if constexpr (K == 4 && BF3_) lp = likelihood_bf3<LOGP>(beta, v);
    else if constexpr (K == 4) lp = likelihood_mfma<LOGP>(beta, v);
    else lp = likelihood_generic<LOGP>(beta, v);
    ARP_T(7, tg);
    float lq = 0.0f, g_ols = 0.0f;
    if constexpr (PART_) {
      // row-part form: the prior terms carry the weight pw (the model's own formulas below with pw = 1)
      const float w_ = pw();
#pragma unroll
      for (int i = 0; i < NLS; ++i) {
        float e = fast_exp(-b[i] * bls[i]);
        float zb = q[NG + NLS + i] * e;
        float hb = fmaf(w_ * b[i], fmaf(zb, zb, -1.0f), v[i] * (1.0f - b[i]) * beta[i]);
        bool ok = i < nown;
        g[NG + NLS + i] = ok ? fmaf(v[i], fast_exp((1.0f - b[i]) * bls[i]), -w_ * zb * e) : 0.0f;
        g[NG + i] = ok ? fmaf(-w_, r[i], hb) : 0.0f;
        g_ols += ok ? fmaf(w_ * a[i], r[i], (1.0f - a[i]) * hb) : 0.0f;
        if (LOGP) lq += ok ? fmaf(-0.5f * r[i], r[i], fmaf(-0.5f * zb, zb, -b[i] * bls[i])) : 0.0f;
      }
      g_ols = group_sum<K>(g_ols);
      const float u0 = q[0] * s0i;
      g[0] = fmaf(c0, g_ols, -w_ * u0 * s0i);
      if (LOGP) lp += w_ * (group_sum<K>(lq) - 0.5f * u0 * u0);
      return lp;
    }
#pragma unroll
    for (int i = 0; i < NLS; ++i) {
      float e = fast_exp(-b[i] * bls[i]);
      float zb = q[NG + NLS + i] * e;
      float hb = fmaf(b[i], fmaf(zb, zb, -1.0f), v[i] * (1.0f - b[i]) * beta[i]);
      bool ok = i < nown;
      g[NG + NLS + i] = ok ? fmaf(v[i], fast_exp((1.0f - b[i]) * bls[i]), -zb * e) : 0.0f;
      g[NG + i] = ok ? hb - r[i] : 0.0f;
      g_ols += ok ? fmaf(a[i], r[i], (1.0f - a[i]) * hb) : 0.0f;
      if (LOGP) lq += ok ? fmaf(-0.5f * r[i], r[i], fmaf(-0.5f * zb, zb, -b[i] * bls[i])) : 0.0f;
    }
    g_ols = group_sum<K>(g_ols);
    const float u0 = q[0] * s0i;
    g[0] = fmaf(c0, g_ols, -u0 * s0i);
    if (LOGP) lp += group_sum<K>(lq) - 0.5f * u0 * u0;
    ARP_T(5, tg);
    return lp;
  }

  ARP_DEV void dparam(const float (&q)[ND], const float (&g)[ND], float (&da)[ND], float (&db)[ND]) const {
    const float ols = c0 * q[0];
#pragma unroll
    for (int i = 0; i < ND; ++i) { da[i] = 0.0f; db[i] = 0.0f; }
    // (PART_: g is this row part's share of the gradient; the constant 1 of the two affine forms counts once)
    db[0] = -2.302585092994046f * fmaf(q[0], g[0], pw());
#pragma unroll
    for (int i = 0; i < NLS; ++i) {
      bool ok = i < nown;
      float bls = fmaf(-a[i], ols, q[NG + i]) + ols;
      da[NG + i] = ok ? -ols * g[NG + i] : 0.0f;
      db[NG + NLS + i] = ok ? -bls * fmaf(q[NG + NLS + i], g[NG + NLS + i], pw()) : 0.0f;
    }
  }

  ARP_DEV void to_centered(const float (&q)[ND], float (&x)[ND]) const {
    const float ols = c0 * q[0];
    x[0] = ols;
#pragma unroll
    for (int i = 0; i < NLS; ++i) {
      float bls = q[NG + i] + (1.0f - a[i]) * ols;
      x[NG + i] = bls;
      x[NG + NLS + i] = fast_exp((1.0f - b[i]) * bls) * q[NG + NLS + i];
    }
  }
  ARP_DEV void from_centered(const float (&x)[ND], float (&q)[ND]) const {
    const float ols = x[0];
    q[0] = ols / c0;
#pragma unroll
    for (int i = 0; i < NLS; ++i) {
      bool ok = i < nown;
      q[NG + i] = ok ? x[NG + i] - (1.0f - a[i]) * ols : 0.0f;
      q[NG + NLS + i] = ok ? x[NG + NLS + i] * fast_exp(-(1.0f - b[i]) * x[NG + i]) : 0.0f;
    }
  }
};

}  // namespace arp
