// Packed-f32 lane model of election88 (reference models.py:967-1008) for its three compile-time parameterisations --
// centred, non-centred, and "a free, b = 1" (what the reference's tied cVIP / dVIP runs execute) -- and,
// round 4, for the GENERAL per-element (a, b) of `--tied_pparams=False` runs (program_transformations.py:513-533, 555-600;
// MODE kModeVIP: one more exponential per state and pass, sigma^{-b_t}), on the packed chain kernels of pk_chain.h.
// model_election.h has the model, its one-hot quirk and the generic float-array form (the checker of this one in the
// density tests, and the interleaved / VI kernels' lane model).
//
// A lane's states t = slot + K*i are held two at a time (pair k = states 2k, 2k+1), so everything per state but the
// transcendentals is one v_pk_*_f32 per pair: per pass and pair 24 packed operations, two exponentials (exp(-a_t), shared
// by the four (female, black) cells of a state) and eight reciprocals; the closing pass adds eight logarithms
// (softplus(eta) = eta - log sigmoid(eta), from the reciprocal it already has).  The cell tables are chain independent and
// live in LDS (5 x 16 bytes per pair: n and y of the four cells of both states, and sum_c (y - n)), so a lane needs
// ~120 registers and three to four waves fit a SIMD -- the reciprocal / exponential chains are latency bound, and the
// generic form (model_election.h, 100+ table registers per lane at K = 4) runs at two with spills.
#pragma once
#include "pk_chain.h"
#include "model_election.h"

namespace arp {

template <int K_, int NL_>
struct ElectionPk {
  static constexpr int K = K_, NL = NL_, NG = 4, ND = NG + NL_;
  static constexpr int NP = (NL_ + 1) / 2;
  static constexpr int DCAP = NG + K_ * NL_;
  static constexpr int LBASE = 2;
  static_assert(K_ >= 4, "the packed kernels deal the top-level momenta out over the first slots of a chain");
  static_assert(NL_ >= 3, "at least two state pairs per lane");
  // three waves per SIMD (measured at K = 4: four waves cap the lane at 128 registers, spill 26 values and run 4 % slower)
  static constexpr int MINW = NL_ <= 13 ? 3 : 2;
  static constexpr int PASS_BLOCK = 1;   // pk_chain.h: pk_transition's interior loop, one pass per iteration
  using Args = ElectionArgs;

  // ---- cell tables in LDS: entry (pair k, slot s) = 5 float4 ----
  //   {n0a n0b n1a n1b} {n2a n2b n3a n3b} {y0a y0b y1a y1b} {y2a y2b y3a y3b} {dna dnb - -},  a / b = states 2k / 2k+1
  static constexpr int SMEM_FLOATS = 20 * K_ * NP;
  const float4* tab;
  static ARP_DEV void stage_tables(const Args& A, float* smem) {
    for (int idx = threadIdx.x; idx < K * NP; idx += blockDim.x) {
      const int k = idx / K, s = idx % K;
      float dn[2] = {0.0f, 0.0f};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int i = 2 * k + h, t = s + K * i;
        const bool cell = i < NL && t <= A.S;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float n = cell ? A.cell_n[t * 4 + c] : 0.0f, y = cell ? A.cell_y[t * 4 + c] : 0.0f;
          smem[idx * 20 + (c >> 1) * 4 + (c & 1) * 2 + h] = n;
          smem[idx * 20 + 8 + (c >> 1) * 4 + (c & 1) * 2 + h] = y;
          dn[h] += y - n;
        }
      }
      smem[idx * 20 + 16] = dn[0]; smem[idx * 20 + 17] = dn[1]; smem[idx * 20 + 18] = 0.0f; smem[idx * 20 + 19] = 0.0f;
    }
  }
  ARP_DEV void bind_tables(const float* smem) { tab = reinterpret_cast<const float4*>(smem); }
  // the entry of pair k; the index is laundered so that the reads stay inside the pass that uses them
  ARP_DEV const float4* entry(int k) const {
    int e = (k * K + slot) * 5;
    asm volatile("" : "+v"(e));
    return tab + e;
  }

  v2f al2[NP];      // a of the state effects (MODE b = 1 and the general form; dead otherwise)
  v2f bl2[NP];      // b of the state effects (general form only)
  float nbl;        // sum of b over the lane's state effects (general form: -b ls per state in the log density)
  float csr[NG], sir[NG];   // general form: s_i^(1 - b_i) and s_i^(-b_i) of the top-level scalars (s = 100, 10, 100, 100)
  v2f mlast;        // 1/0: which elements of the LAST pair are state effects (the cell-only group S and padding are not)
  float nlat;       // state effects owned by the lane
  float d1, d2;     // sum over the lane's cells of (y - n) that carry b1 / b2
  int slot, S;
  bool last_ok;
  int gmap2, gmap3;

  ARP_DEV int gg(int i) const { return i == 0 ? 0 : (i == 1 ? 1 : (i == 2 ? gmap2 : gmap3)); }
  ARP_DEV int lbase(int) const { return LBASE + slot; }
  static constexpr ARP_DEV int loff(int i) { return K * i; }
  ARP_DEV bool lvalid(int i) const { return i < NL - 1 ? true : last_ok; }

  ARP_DEV void init(const Args& A, const float* av, const float* bv, int slot_) {
    slot = slot_;
    S = A.S;
    gmap2 = 2 + S; gmap3 = 3 + S;
    last_ok = slot + K * (NL - 1) < S;
    const float ll = last_ok ? 1.0f : 0.0f;
    if (NL & 1) mlast = v2f{ll, 0.0f}; else mlast = v2f{1.0f, ll};
    nlat = (float)(NL - 1) + ll;
    d1 = 0.0f; d2 = 0.0f;
    nbl = 0.0f;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      // wave uniform: log2 of the prior scale (100, 10, 100, 100) times (1 - b), -b
      const float l2s = i == 1 ? 3.3219280948873623f : 6.643856189774724f;
      const float bi = bv ? bv[gg(i)] : 1.0f;
      csr[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, __builtin_amdgcn_exp2f((1.0f - bi) * l2s))));
      sir[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, __builtin_amdgcn_exp2f(-bi * l2s))));
    }
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const float4* e = entry(k);
      const float4 n01 = e[0], n23 = e[1], y01 = e[2], y23 = e[3];
      // cells: 0 = (f0,b0), 1 = female, 2 = black, 3 = both; b2 multiplies female, b1 black
      d2 += ((y01.z - n01.z) + (y01.w - n01.w)) + ((y23.z - n23.z) + (y23.w - n23.w));
      d1 += ((y23.x - n23.x) + (y23.y - n23.y)) + ((y23.z - n23.z) + (y23.w - n23.w));
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int i = 2 * k + h;
        const bool lat = i < NL && slot + K * i < S;
        al2[k][h] = (lat && av) ? av[LBASE + slot + K * i] : 0.0f;
        bl2[k][h] = (lat && bv) ? bv[LBASE + slot + K * i] : 0.0f;
        nbl += bl2[k][h];
      }
    }
  }

  static ARP_DEV void unpack(const float (&v)[ND], float (&g4)[NG], v2f (&c)[NP]) {
#pragma unroll
    for (int i = 0; i < NG; ++i) g4[i] = v[i];
#pragma unroll
    for (int i = 0; i < 2 * NP; ++i) c[i >> 1][i & 1] = i < NL ? v[NG + i] : 0.0f;
  }
  static ARP_DEV void pack(const float (&g4)[NG], const v2f (&c)[NP], float (&v)[ND]) {
#pragma unroll
    for (int i = 0; i < NG; ++i) v[i] = g4[i];
#pragma unroll
    for (int i = 0; i < NL; ++i) v[NG + i] = c[i >> 1][i & 1];
  }

  // top-level scalars: xt ~ N(0, s^b), x = s^(1-b) xt.  b = 1 (centred, and "a free, b = 1"): x = xt, 1/s in the prior;
  // b = 0 (non-centred): x = s xt, unit prior.
  template <int MODE> static constexpr ARP_DEV float cs(int i) {
    return MODE == kModeNCP ? (i == 1 ? 10.0f : 100.0f) : 1.0f;
  }
  template <int MODE> static constexpr ARP_DEV float si(int i) {
    return MODE == kModeNCP ? 1.0f : (i == 1 ? 0.1f : 0.01f);
  }

  template <int MODE> ARP_DEV float csv(int i) const { if constexpr (MODE == kModeVIP) return csr[i]; else return cs<MODE>(i); }
  template <int MODE> ARP_DEV float siv(int i) const { if constexpr (MODE == kModeVIP) return sir[i]; else return si<MODE>(i); }

  // PASS 0 interior (gradient, kick, drift), 1 closing (gradient, logp, kinetic energy), 2 bootstrap (gradient, logp)
  template <int MODE, int PASS>
  ARP_DEV void pass(float (&qg)[NG], v2f (&qc)[NP], float (&pg)[NG], v2f (&pc)[NP], const float (&eg)[NG],
                    const v2f (&ec)[NP], float (&gg_)[NG], v2f (&gc)[NP], float& lp, float& ke) const {
    constexpr bool GEN = MODE == kModeVIP;          // per-element (a, b): z = (q - a mua) sigma^{-b}
    constexpr bool B1ISH = MODE == kModeCP || MODE == kModeB1;   // b = 1 on the state effects (centred, or a free with b = 1)
    const float mua = csv<MODE>(0) * qg[0], ls = csv<MODE>(1) * qg[1], b1 = csv<MODE>(2) * qg[2], b2 = csv<MODE>(3) * qg[3];
    const float sig = fast_exp(ls);
    const float eu = B1ISH ? fast_exp(-ls) : 1.0f;
    const float E1 = fast_exp(-b1), E2 = fast_exp(-b2), E12 = E1 * E2;
    const v2f vmua = splat(mua), vmua_last = vmua * mlast, vsig = splat(sig), veu = splat(eu);
    const v2f vE1 = splat(E1), vE2 = splat(E2), vE12 = splat(E12), one = splat(1.0f), half = splat(0.5f);
    const v2f nl2e = splat(-1.4426950408889634f);
    const v2f vnls2 = splat(-1.4426950408889634f * ls);      // general form: sigma^{-b} = exp2(b (-ls log2 e))
    v2f a_mua = splat(0.0f), a_ls = splat(0.0f), a_b1 = splat(0.0f), a_b2 = splat(0.0f);
    // log density: |logp| ~ 7 600 while Metropolis needs its differences to ~1e-3, so the per-pair terms (a few hundred
    // each) go into a compensated (Kahan) sum; the pairs' own four-cell sums start from zero
    v2f lps = splat(0.0f), lpc = splat(0.0f), ake = splat(0.0f);
    const v2f ln2 = splat(0.6931471805599453f);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const bool last = k == NP - 1;
      const v2f qt = qc[k];
      v2f z;
      if (MODE == kModeCP) z = (qt - (last ? vmua_last : vmua)) * veu;
      else if (MODE == kModeNCP) z = qt;
      else if (MODE == kModeB1) z = vfma(-al2[k], vmua, qt) * veu;
      v2f euk = one;
      if (GEN) {
        const v2f e = bl2[k] * vnls2;
        euk = v2f{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
        z = vfma(-al2[k], vmua, qt) * euk;
      }
      v2f as = vfma(vsig, z, vmua);
      if (last) as = as * mlast;            // the cell-only group and padding see no state effect
      const v2f e2 = as * nl2e;
      const v2f t = {__builtin_amdgcn_exp2f(e2[0]), __builtin_amdgcn_exp2f(e2[1])};
      const float4* en = entry(k);
      const float4 n01 = en[0], n23 = en[1], y01 = en[2], y23 = en[3];
      const v2f den[4] = {one + t, vfma(t, vE2, one), vfma(t, vE1, one), vfma(t, vE12, one)};
      v2f rc[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) rc[c] = v2f{__builtin_amdgcn_rcpf(den[c][0]), __builtin_amdgcn_rcpf(den[c][1])};
      const v2f nn[4] = {v2f{n01.x, n01.y}, v2f{n01.z, n01.w}, v2f{n23.x, n23.y}, v2f{n23.z, n23.w}};
      const v2f yy[4] = {v2f{y01.x, y01.y}, v2f{y01.z, y01.w}, v2f{y23.x, y23.y}, v2f{y23.z, y23.w}};
      v2f w[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) w[c] = vfma(-nn[c], rc[c], yy[c]);
      const v2f s13 = w[1] + w[3], s23 = w[2] + w[3];
      const v2f W = (w[0] + w[2]) + s13;
      a_b2 += s13;
      a_b1 += s23;
      v2f gt, sW;
      if (GEN) { sW = vsig * W; gt = (sW - z) * euk; }
      else gt = vfma(vsig, W, -z);                  // (the three compile-time forms keep round 3's arithmetic bit for bit)
      if (B1ISH) gt = gt * veu;
      if (last) gt = gt * mlast;
      // d/dmua: the likelihood through a_t, minus what the (a mua) location of the prior takes back
      v2f hm;
      if (MODE == kModeCP) hm = W - gt;
      else if (MODE == kModeNCP) hm = W;
      else hm = vfma(-al2[k], gt, W);                 // b = 1 and the general form alike
      a_mua += last ? hm * mlast : hm;
      // d/dls: b (z^2 - 1) + (1 - b) W sigma z; the constants (-1 per state effect, the factor sigma) follow the reduction
      if (GEN) {
        const v2f swz = sW * z;
        a_ls += vfma(bl2[k], vfma(z, z, -swz), swz);          // b z^2 + (1 - b) sigma W z
      } else {
        a_ls = B1ISH ? vfma(z, z, a_ls) : vfma(W, z, a_ls);
      }
      if (PASS == 0) {
        const v2f pn = vfma(ec[k], gt, pc[k]);
        pc[k] = pn;
        qc[k] = vfma(ec[k], pn, qt);
      } else {
        gc[k] = gt;
        // sum_c y eta - n softplus(eta) = sum_c n log(rc) + (sum_c y - n) a_t + [(y - n) b-terms: d1, d2 below]
        v2f pl = nn[0] * v2f{__builtin_amdgcn_logf(rc[0][0]), __builtin_amdgcn_logf(rc[0][1])};   // v_log_f32 is log2
#pragma unroll
        for (int c = 1; c < 4; ++c)
          pl = vfma(nn[c], v2f{__builtin_amdgcn_logf(rc[c][0]), __builtin_amdgcn_logf(rc[c][1])}, pl);
        const float4 dn4 = en[4];
        v2f term = vfma(pl, ln2, v2f{dn4.x, dn4.y} * as);
        term = vfma(-half * z, z, term);                  // the prior of z: -z^2 / 2 (0 for the cell-only group)
        const v2f yk = term - lpc, tk = lps + yk;
        lpc = (tk - lps) - yk;
        lps = tk;
        if (PASS == 1) {
          const v2f pf = vfma(half, ec[k] * gt, pc[k]);
          ake = vfma(pf, pf, ake);
        }
      }
    }
    const float s_mua = group_sum<K>(a_mua[0] + a_mua[1]);
    const float zz = a_ls[0] + a_ls[1];                       // b = 1: sum z^2; b = 0: sum W z
    const float s_ls = group_sum<K>(GEN ? zz - nbl : (B1ISH ? zz - nlat : zz * sig));
    const float s_b1 = group_sum<K>(a_b1[0] + a_b1[1]);
    const float s_b2 = group_sum<K>(a_b2[0] + a_b2[1]);
    const float gs[4] = {s_mua, s_ls, s_b1, s_b2};
    float gi[4], u[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u[i] = qg[i] * siv<MODE>(i);
      gi[i] = fmaf(csv<MODE>(i), gs[i], -u[i] * siv<MODE>(i));
    }
    if (PASS == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        pg[i] = fmaf(eg[i], gi[i], pg[i]);
        qg[i] = fmaf(eg[i], pg[i], qg[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) gg_[i] = gi[i];
      float part = (lps[0] + lps[1]) - (lpc[0] + lpc[1]);
      part += fmaf(d1, b1, d2 * b2) - (GEN ? nbl * ls : (B1ISH ? nlat * ls : 0.0f));   // (y - n) b-terms; - b ls per state effect
      lp = group_sum<K>(part) - 0.5f * ((u[0] * u[0] + u[1] * u[1]) + (u[2] * u[2] + u[3] * u[3]));
      if (PASS == 1) {
        float kg = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float pf = fmaf(0.5f * eg[i], gi[i], pg[i]);
          kg = fmaf(pf, pf, kg);
        }
        ke = 0.5f * (group_sum<K>(ake[0] + ake[1]) + kg);
      }
    }
  }

  // centred coordinates of a state held in parameterisation MODE
  template <int MODE>
  ARP_DEV void to_centered(const float (&qg)[NG], const v2f (&qc)[NP], float (&xg)[NG], v2f (&xc)[NP]) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) xg[i] = csv<MODE>(i) * qg[i];
    const v2f vmua = splat(xg[0]);
    const v2f vsig = splat(MODE == kModeNCP ? fast_exp(xg[1]) : 1.0f);
    const v2f vls2 = splat(1.4426950408889634f * xg[1]);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const bool last = k == NP - 1;
      v2f x;
      if (MODE == kModeCP) x = qc[k];                                           // a = b = 1: identity
      else if (MODE == kModeNCP) x = vfma(vsig, qc[k], vmua);                   // a = b = 0: mua + sigma q
      else if (MODE == kModeB1) x = vfma((last ? mlast : splat(1.0f)) - al2[k], vmua, qc[k]);   // b = 1: q + (1 - a) mua
      else {                                                                    // mua + sigma^(1 - b) (q - a mua)
        const v2f e = (splat(1.0f) - bl2[k]) * vls2;
        const v2f sc = v2f{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
        x = vfma(sc, vfma(-al2[k], vmua, qc[k]), vmua);
      }
      xc[k] = (last && (MODE == kModeNCP || MODE == kModeVIP)) ? x * mlast : x;
    }
  }

  // the inverse: the state in parameterisation MODE of centred coordinates (xg, xc) (models.py:84-102 with election's scales)
  template <int MODE>
  ARP_DEV void from_centered(const float (&xg)[NG], const v2f (&xc)[NP], float (&qg)[NG], v2f (&qc)[NP]) const {
#pragma unroll
    for (int i = 0; i < 4; ++i) qg[i] = xg[i] * (1.0f / csv<MODE>(i));
    const v2f vmua = splat(xg[0]);
    const v2f vis = splat(MODE == kModeNCP ? fast_exp(-xg[1]) : 1.0f);
    const v2f vls2 = splat(1.4426950408889634f * xg[1]);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const bool last = k == NP - 1;
      v2f q;
      if (MODE == kModeCP) q = xc[k];
      else if (MODE == kModeNCP) q = (xc[k] - vmua) * vis;                      // (x - mua) / sigma
      else if (MODE == kModeB1) q = vfma(al2[k] - (last ? mlast : splat(1.0f)), vmua, xc[k]);   // b = 1: x - (1 - a) mua
      else {                                                                    // a mua + (x - mua) sigma^(b - 1)
        const v2f e = (bl2[k] - splat(1.0f)) * vls2;
        const v2f sc = v2f{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
        q = vfma(sc, xc[k] - vmua, al2[k] * vmua);
      }
      qc[k] = (last && (MODE == kModeNCP || MODE == kModeVIP)) ? q * mlast : q;
    }
  }
};

}  // namespace arp
