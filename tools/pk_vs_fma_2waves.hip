// v_pk_fma_f32 against v_fma_f32 at EXACTLY two waves per SIMD: the kernels clobber v250, so a wave owns 256 registers and
// a SIMD can hold two -- the headline kernel's regime (a small kernel lets the dispatcher pack 4 - 8 waves on some SIMDs and
// none on others, which is what tools/valu_bench.hip's "waves/SIMD=2" rows actually measured).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define R4(x) x x x x
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v250"
#define INIT "v_mov_b32 v1, 1.0001\n v_mov_b32 v2, 0.5\n v_mov_b32 v3, 0.5\n v_mov_b32 v4, 1.0001\n v_mov_b32 v5, 1.0001\n v_mov_b32 v6, 0.25\n v_mov_b32 v7, 0.25\n" \
  "v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n" \
  "v_mov_b32 v48, 0\n v_mov_b32 v49, 0\n v_mov_b32 v50, 0\n v_mov_b32 v51, 0\n v_mov_b32 v52, 0\n v_mov_b32 v53, 0\n v_mov_b32 v54, 0\n v_mov_b32 v55, 0\n v_mov_b32 v250, 0\n"
template <int MODE> __global__ __launch_bounds__(256) void k(float* out, int iters) {
  asm volatile(INIT ::: CLOB);
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) asm volatile(R4("v_fma_f32 v40, v1, v2, v40\n v_fma_f32 v41, v1, v2, v41\n v_fma_f32 v42, v1, v2, v42\n v_fma_f32 v43, v1, v2, v43\n"
                                   "v_fma_f32 v44, v1, v2, v44\n v_fma_f32 v45, v1, v2, v45\n v_fma_f32 v46, v1, v2, v46\n v_fma_f32 v47, v1, v2, v47\n") ::: CLOB);
    if (MODE == 1) asm volatile(R4("v_pk_fma_f32 v[40:41], v[2:3], v[4:5], v[40:41]\n v_pk_fma_f32 v[42:43], v[2:3], v[4:5], v[42:43]\n"
                                   "v_pk_fma_f32 v[44:45], v[2:3], v[4:5], v[44:45]\n v_pk_fma_f32 v[46:47], v[2:3], v[4:5], v[46:47]\n"
                                   "v_pk_fma_f32 v[48:49], v[2:3], v[4:5], v[48:49]\n v_pk_fma_f32 v[50:51], v[2:3], v[4:5], v[50:51]\n"
                                   "v_pk_fma_f32 v[52:53], v[2:3], v[4:5], v[52:53]\n v_pk_fma_f32 v[54:55], v[2:3], v[4:5], v[54:55]\n") ::: CLOB);
    if (MODE == 2) asm volatile(R4("v_pk_fma_f32 v[40:41], v[2:3], v[4:5], v[40:41]\n v_fma_f32 v48, v1, v6, v48\n"
                                   "v_pk_fma_f32 v[42:43], v[2:3], v[4:5], v[42:43]\n v_fma_f32 v49, v1, v6, v49\n"
                                   "v_pk_fma_f32 v[44:45], v[2:3], v[4:5], v[44:45]\n v_fma_f32 v50, v1, v6, v50\n"
                                   "v_pk_fma_f32 v[46:47], v[2:3], v[4:5], v[46:47]\n v_fma_f32 v51, v1, v6, v51\n") ::: CLOB);
    if (MODE == 3) asm volatile(R4("v_pk_mul_f32 v[40:41], v[2:3], v[4:5]\n v_pk_mul_f32 v[42:43], v[2:3], v[4:5]\n v_pk_add_f32 v[44:45], v[2:3], v[4:5]\n v_pk_add_f32 v[46:47], v[2:3], v[4:5]\n"
                                   "v_pk_mul_f32 v[48:49], v[2:3], v[4:5]\n v_pk_mul_f32 v[50:51], v[2:3], v[4:5]\n v_pk_add_f32 v[52:53], v[2:3], v[4:5]\n v_pk_add_f32 v[54:55], v[2:3], v[4:5]\n") ::: CLOB);
    if (MODE == 4) asm volatile(R4("v_exp_f32 v40, v1\n v_log_f32 v41, v1\n v_sqrt_f32 v42, v1\n v_sin_f32 v43, v2\n v_cos_f32 v44, v2\n v_rcp_f32 v45, v1\n v_exp_f32 v46, v2\n v_log_f32 v47, v4\n") ::: CLOB);
    if (MODE == 5) asm volatile(R4("v_add_f32_dpp v40, v1, v40 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp v41, v1, v41 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                   "v_mov_b32_dpp v42, v1 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp v43, v2, v43 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                   "v_add_f32_dpp v44, v1, v44 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp v45, v1, v45 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                   "v_mov_b32_dpp v46, v2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp v47, v2, v47 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n") ::: CLOB, "vcc");
    if (MODE == 6) asm volatile(R4("v_mad_u64_u32 v[40:41], vcc, v1, v2, v[40:41]\n v_mad_u64_u32 v[42:43], vcc, v1, v2, v[42:43]\n v_mad_u64_u32 v[44:45], vcc, v1, v2, v[44:45]\n v_mad_u64_u32 v[46:47], vcc, v1, v2, v[46:47]\n"
                                   "v_mad_u64_u32 v[48:49], vcc, v1, v2, v[48:49]\n v_mad_u64_u32 v[50:51], vcc, v1, v2, v[50:51]\n v_mad_u64_u32 v[52:53], vcc, v1, v2, v[52:53]\n v_mad_u64_u32 v[54:55], vcc, v1, v2, v[54:55]\n") ::: CLOB, "vcc");
    if (MODE == 7) asm volatile(R4("v_cvt_f32_u32 v40, v1\n v_cndmask_b32 v41, v1, v2, vcc\n v_lshrrev_b32 v42, 3, v1\n v_cmp_gt_f32 vcc, v1, v2\n v_cvt_f32_u32 v44, v2\n v_cndmask_b32 v45, v2, v1, vcc\n v_max_f32 v46, v1, v2\n v_lshlrev_b32 v47, 5, v2\n") ::: CLOB, "vcc");
    if (MODE == 8) asm volatile(R4("v_mov_b32 v40, v1\n v_xor_b32 v41, v1, v2\n v_and_or_b32 v42, v1, v2, v4\n v_add_f32 v43, v1, v2\n v_mul_f32 v44, v1, v2\n v_sub_f32 v45, v2, v1\n v_mov_b32 v46, v2\n v_fmac_f32 v47, v1, v2\n") ::: CLOB);
  }
  if (iters < 0) out[0] = 1.0f;
}
template <int MODE> void run(const char* name) {
  float* d; (void)hipMalloc(&d, 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int blocks = 512; blocks <= 2048; blocks *= 4) {      // 1, 2, 4 (two rounds), 8 (four rounds) waves per SIMD in total
    const int iters = 40000;
    k<MODE><<<blocks, 256>>>(d, iters); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %4d workgroups (%d waves per SIMD in total, at most 2 resident): %.3f ms, %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n",
           name, blocks, blocks / 256, ms, (ms * 1e-3 * 2.4e9) / ((double)blocks * 4 * iters * 32 / 1024.0));
  }
  (void)hipFree(d);
}
int main() {
  run<0>("v_fma_f32"); run<1>("v_pk_fma_f32"); run<2>("v_pk_fma + v_fma alternating"); run<3>("v_pk_mul / v_pk_add");
  run<4>("transcendentals"); run<5>("DPP add / mov"); run<6>("v_mad_u64_u32"); run<7>("cvt / cndmask / shift / cmp / max"); run<8>("mov / xor / and_or / add / mul / fmac");
  return 0;
}
