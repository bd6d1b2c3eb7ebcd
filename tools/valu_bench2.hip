// Per-instruction issue cost on gfx950 at 2 and 4 waves/SIMD (cycles at the measured clock are printed
// relative to v_fma_f32).  Build: hipcc -O3 --offload-arch=gfx950 tools/valu_bench2.hip -o tools/valu_bench2
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP4(x) x x x x
#define BODY(INS) \
  for (int i = 0; i < iters; ++i) { \
    REP4(REP4(asm volatile(INS("%0") INS("%1") INS("%2") INS("%3") INS("%4") INS("%5") INS("%6") INS("%7") \
      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc", "s20", "s21");)) }
#define K(NAME, INS) __global__ __launch_bounds__(256) void NAME(float* out, int iters, float a, float b) { \
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
  BODY(INS) \
  float r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7; if (r == 12345.678f) out[0] = r; }
#define I_FMA(r) "v_fma_f32 " r ", %8, %9, " r "\n"
#define I_FMAC(r) "v_fmac_f32 " r ", %8, %9\n"
#define I_ADD(r) "v_add_f32 " r ", %8, " r "\n"
#define I_MUL(r) "v_mul_f32 " r ", %8, " r "\n"
#define I_FMA3(r) "v_fma_f32 " r ", " r ", %8, %9\n"
#define I_CND(r) "v_cndmask_b32 " r ", %8, " r ", vcc\n"
#define I_XOR(r) "v_xor_b32 " r ", %8, " r "\n"
#define I_SHL(r) "v_lshlrev_b32 " r ", 9, " r "\n"
#define I_ALIGN(r) "v_alignbit_b32 " r ", " r ", " r ", 21\n"
#define I_ADDU(r) "v_add_u32 " r ", %8, " r "\n"
#define I_CVT(r) "v_cvt_f32_u32 " r ", " r "\n"
#define I_MOV(r) "v_mov_b32 " r ", %8\n"
#define I_XOR3(r) "v_xor3_b32 " r ", %8, %9, " r "\n"
#define I_LSHLADD(r) "v_lshl_add_u32 " r ", " r ", 9, %8\n"
#define I_MAX(r) "v_max_f32 " r ", %8, " r "\n"
#define I_SUBREV(r) "v_subrev_f32 " r ", %8, " r "\n"
#define I_CNDS(r) "v_cndmask_b32 " r ", %8, " r ", s[20:21]\n"
#define I_BFI(r) "v_bfi_b32 " r ", %9, %8, " r "\n"
#define I_AND(r) "v_and_b32 " r ", %8, " r "\n"
#define I_CMP(r) "v_cmp_lt_f32 vcc, %8, " r "\n"
#define I_CMPS(r) "v_cmp_lt_f32 s[20:21], %8, " r "\n"
#define I_MIN(r) "v_min_f32 " r ", %8, " r "\n"
#define I_MED(r) "v_med3_f32 " r ", %8, " r ", %9\n"
#define I_DPP(r) "v_add_f32_dpp " r ", " r ", " r " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I_MOVDPP(r) "v_mov_b32_dpp " r ", " r " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
K(k_fma, I_FMA) K(k_fmac, I_FMAC) K(k_add, I_ADD) K(k_mul, I_MUL) K(k_fma3, I_FMA3) K(k_cnd, I_CND) K(k_xor, I_XOR)
K(k_shl, I_SHL) K(k_align, I_ALIGN) K(k_addu, I_ADDU) K(k_cvt, I_CVT) K(k_mov, I_MOV)
K(k_cnds, I_CNDS) K(k_bfi, I_BFI) K(k_and, I_AND) K(k_cmp, I_CMP) K(k_cmps, I_CMPS) K(k_min, I_MIN) K(k_med, I_MED)
K(k_lshladd, I_LSHLADD) K(k_max, I_MAX) K(k_subrev, I_SUBREV) K(k_dpp, I_DPP) K(k_movdpp, I_MOVDPP)
typedef void (*kern)(float*, int, float, float);
int main() {
  float* d; hipMalloc(&d, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct { const char* n; kern k; } ks[] = {{"v_fma_f32", k_fma}, {"v_fmac_f32", k_fmac}, {"v_add_f32", k_add}, {"v_mul_f32", k_mul},
    {"v_fma(acc as src0)", k_fma3}, {"v_cndmask_b32", k_cnd}, {"v_xor_b32", k_xor}, {"v_lshlrev_b32", k_shl}, {"v_alignbit_b32", k_align},
    {"v_add_u32", k_addu}, {"v_cvt_f32_u32", k_cvt}, {"v_mov_b32", k_mov}, {"v_cndmask sgpr", k_cnds}, {"v_bfi_b32", k_bfi}, {"v_and_b32", k_and}, {"v_cmp_lt vcc", k_cmp}, {"v_cmp_lt sgpr", k_cmps}, {"v_min_f32", k_min}, {"v_med3_f32", k_med}, {"v_lshl_add_u32", k_lshladd},
    {"v_max_f32", k_max}, {"v_subrev_f32", k_subrev}, {"v_add_f32_dpp", k_dpp}, {"v_mov_b32_dpp", k_movdpp}};
  const int iters = 4000;
  for (auto& t : ks) {
    printf("%-20s", t.n);
    for (int wps = 1; wps <= 4; wps *= 2) {
      int blocks = 256 * wps;
      t.k<<<blocks, 256>>>(d, 50, 1.0001f, 0.5f); hipDeviceSynchronize();
      hipEventRecord(e0); t.k<<<blocks, 256>>>(d, iters, 1.0001f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double instr = (double)blocks * 4 * iters * 128;
      printf("  w/SIMD=%d: %.2f cyc@2.4GHz", wps, (ms * 1e-3 * 2.4e9) / (instr / 1024.0));
    }
    printf("\n");
  }
  return 0;
}
