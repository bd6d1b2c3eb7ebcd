// Micro-benchmark: do the source operands' VGPR banks change the issue rate of v_fma_f32 / v_pk_fma_f32 on gfx950?
// (explicit physical registers; bank hypothesis: register index mod 4)
// Build: hipcc -O3 --offload-arch=gfx950 tools/vgpr_bank_bench.hip -o gpurun_out/vgpr_bank_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#define R4(x) x x x x
#define INIT "v_mov_b32 v0, 1.0\n v_mov_b32 v1, 1.0\n v_mov_b32 v2, 1.0\n v_mov_b32 v3, 1.0\n v_mov_b32 v4, 1.0\n v_mov_b32 v5, 1.0\n" \
             "v_mov_b32 v6, 1.0\n v_mov_b32 v7, 1.0\n v_mov_b32 v8, 1.0\n v_mov_b32 v9, 1.0\n v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0\n v_mov_b32 v12, 1.0\n v_mov_b32 v13, 1.0\n"
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35"
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  asm volatile(INIT ::: CLOB);
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0)        // three sources in three different banks (1, 2, 3), eight independent destinations
      asm volatile(R4("v_fma_f32 v20, v1, v2, v3\n v_fma_f32 v21, v1, v2, v3\n v_fma_f32 v22, v1, v2, v3\n v_fma_f32 v23, v1, v2, v3\n"
                      "v_fma_f32 v24, v1, v2, v3\n v_fma_f32 v25, v1, v2, v3\n v_fma_f32 v26, v1, v2, v3\n v_fma_f32 v27, v1, v2, v3\n") ::: CLOB);
    else if (MODE == 1)   // three sources in ONE bank (0, 4, 8)
      asm volatile(R4("v_fma_f32 v20, v0, v4, v8\n v_fma_f32 v21, v0, v4, v8\n v_fma_f32 v22, v0, v4, v8\n v_fma_f32 v23, v0, v4, v8\n"
                      "v_fma_f32 v24, v0, v4, v8\n v_fma_f32 v25, v0, v4, v8\n v_fma_f32 v26, v0, v4, v8\n v_fma_f32 v27, v0, v4, v8\n") ::: CLOB);
    else if (MODE == 2)   // two sources in one bank (0, 4), the third elsewhere
      asm volatile(R4("v_fma_f32 v20, v0, v4, v1\n v_fma_f32 v21, v0, v4, v1\n v_fma_f32 v22, v0, v4, v1\n v_fma_f32 v23, v0, v4, v1\n"
                      "v_fma_f32 v24, v0, v4, v1\n v_fma_f32 v25, v0, v4, v1\n v_fma_f32 v26, v0, v4, v1\n v_fma_f32 v27, v0, v4, v1\n") ::: CLOB);
    else if (MODE == 3)   // accumulate form (dst = src2), sources spread over banks: what a real FMA chain looks like
      asm volatile(R4("v_fma_f32 v20, v1, v2, v20\n v_fma_f32 v21, v1, v2, v21\n v_fma_f32 v22, v1, v2, v22\n v_fma_f32 v23, v1, v2, v23\n"
                      "v_fma_f32 v24, v1, v2, v24\n v_fma_f32 v25, v1, v2, v25\n v_fma_f32 v26, v1, v2, v26\n v_fma_f32 v27, v1, v2, v27\n") ::: CLOB);
    else if (MODE == 4)   // packed, pairs in distinct bank pairs: (2:3), (4:5)=(0:1 banks), (6:7)
      asm volatile(R4("v_pk_fma_f32 v[20:21], v[2:3], v[4:5], v[10:11]\n v_pk_fma_f32 v[22:23], v[2:3], v[4:5], v[10:11]\n"
                      "v_pk_fma_f32 v[24:25], v[2:3], v[4:5], v[10:11]\n v_pk_fma_f32 v[26:27], v[2:3], v[4:5], v[10:11]\n"
                      "v_pk_fma_f32 v[28:29], v[2:3], v[4:5], v[10:11]\n v_pk_fma_f32 v[30:31], v[2:3], v[4:5], v[10:11]\n"
                      "v_pk_fma_f32 v[32:33], v[2:3], v[4:5], v[10:11]\n v_pk_fma_f32 v[34:35], v[2:3], v[4:5], v[10:11]\n") ::: CLOB);
    else if (MODE == 5)   // packed, all three pairs start in bank 0: (0:1), (4:5), (8:9)
      asm volatile(R4("v_pk_fma_f32 v[20:21], v[0:1], v[4:5], v[8:9]\n v_pk_fma_f32 v[22:23], v[0:1], v[4:5], v[8:9]\n"
                      "v_pk_fma_f32 v[24:25], v[0:1], v[4:5], v[8:9]\n v_pk_fma_f32 v[26:27], v[0:1], v[4:5], v[8:9]\n"
                      "v_pk_fma_f32 v[28:29], v[0:1], v[4:5], v[8:9]\n v_pk_fma_f32 v[30:31], v[0:1], v[4:5], v[8:9]\n"
                      "v_pk_fma_f32 v[32:33], v[0:1], v[4:5], v[8:9]\n v_pk_fma_f32 v[34:35], v[0:1], v[4:5], v[8:9]\n") ::: CLOB);
    else if (MODE == 6)   // packed, pairs (0:1), (2:3), (4:5): two start in bank 0
      asm volatile(R4("v_pk_fma_f32 v[20:21], v[0:1], v[2:3], v[4:5]\n v_pk_fma_f32 v[22:23], v[0:1], v[2:3], v[4:5]\n"
                      "v_pk_fma_f32 v[24:25], v[0:1], v[2:3], v[4:5]\n v_pk_fma_f32 v[26:27], v[0:1], v[2:3], v[4:5]\n"
                      "v_pk_fma_f32 v[28:29], v[0:1], v[2:3], v[4:5]\n v_pk_fma_f32 v[30:31], v[0:1], v[2:3], v[4:5]\n"
                      "v_pk_fma_f32 v[32:33], v[0:1], v[2:3], v[4:5]\n v_pk_fma_f32 v[34:35], v[0:1], v[2:3], v[4:5]\n") ::: CLOB);
    else if (MODE == 9)   // packed accumulate form (dst = src2)
      asm volatile(R4("v_pk_fma_f32 v[20:21], v[2:3], v[4:5], v[20:21]\n v_pk_fma_f32 v[22:23], v[2:3], v[4:5], v[22:23]\n"
                      "v_pk_fma_f32 v[24:25], v[2:3], v[4:5], v[24:25]\n v_pk_fma_f32 v[26:27], v[2:3], v[4:5], v[26:27]\n"
                      "v_pk_fma_f32 v[28:29], v[2:3], v[4:5], v[28:29]\n v_pk_fma_f32 v[30:31], v[2:3], v[4:5], v[30:31]\n"
                      "v_pk_fma_f32 v[32:33], v[2:3], v[4:5], v[32:33]\n v_pk_fma_f32 v[34:35], v[2:3], v[4:5], v[34:35]\n") ::: CLOB);
    else if (MODE == 10)  // packed, src0 == src1
      asm volatile(R4("v_pk_fma_f32 v[20:21], v[2:3], v[2:3], v[4:5]\n v_pk_fma_f32 v[22:23], v[2:3], v[2:3], v[4:5]\n"
                      "v_pk_fma_f32 v[24:25], v[2:3], v[2:3], v[4:5]\n v_pk_fma_f32 v[26:27], v[2:3], v[2:3], v[4:5]\n"
                      "v_pk_fma_f32 v[28:29], v[2:3], v[2:3], v[4:5]\n v_pk_fma_f32 v[30:31], v[2:3], v[2:3], v[4:5]\n"
                      "v_pk_fma_f32 v[32:33], v[2:3], v[2:3], v[4:5]\n v_pk_fma_f32 v[34:35], v[2:3], v[2:3], v[4:5]\n") ::: CLOB);
    else if (MODE == 11)  // packed, one source an SGPR pair
      asm volatile(R4("v_pk_fma_f32 v[20:21], s[4:5], v[2:3], v[4:5]\n v_pk_fma_f32 v[22:23], s[4:5], v[2:3], v[4:5]\n"
                      "v_pk_fma_f32 v[24:25], s[4:5], v[2:3], v[4:5]\n v_pk_fma_f32 v[26:27], s[4:5], v[2:3], v[4:5]\n"
                      "v_pk_fma_f32 v[28:29], s[4:5], v[2:3], v[4:5]\n v_pk_fma_f32 v[30:31], s[4:5], v[2:3], v[4:5]\n"
                      "v_pk_fma_f32 v[32:33], s[4:5], v[2:3], v[4:5]\n v_pk_fma_f32 v[34:35], s[4:5], v[2:3], v[4:5]\n") ::: CLOB);
    else if (MODE == 12)  // packed multiply (two sources)
      asm volatile(R4("v_pk_mul_f32 v[20:21], v[2:3], v[4:5]\n v_pk_mul_f32 v[22:23], v[2:3], v[4:5]\n"
                      "v_pk_mul_f32 v[24:25], v[2:3], v[4:5]\n v_pk_mul_f32 v[26:27], v[2:3], v[4:5]\n"
                      "v_pk_mul_f32 v[28:29], v[2:3], v[4:5]\n v_pk_mul_f32 v[30:31], v[2:3], v[4:5]\n"
                      "v_pk_mul_f32 v[32:33], v[2:3], v[4:5]\n v_pk_mul_f32 v[34:35], v[2:3], v[4:5]\n") ::: CLOB);
    else if (MODE == 13)  // packed multiply, both pairs start in bank 0
      asm volatile(R4("v_pk_mul_f32 v[20:21], v[0:1], v[4:5]\n v_pk_mul_f32 v[22:23], v[0:1], v[4:5]\n"
                      "v_pk_mul_f32 v[24:25], v[0:1], v[4:5]\n v_pk_mul_f32 v[26:27], v[0:1], v[4:5]\n"
                      "v_pk_mul_f32 v[28:29], v[0:1], v[4:5]\n v_pk_mul_f32 v[30:31], v[0:1], v[4:5]\n"
                      "v_pk_mul_f32 v[32:33], v[0:1], v[4:5]\n v_pk_mul_f32 v[34:35], v[0:1], v[4:5]\n") ::: CLOB);
    else if (MODE == 14)  // packed add
      asm volatile(R4("v_pk_add_f32 v[20:21], v[2:3], v[4:5]\n v_pk_add_f32 v[22:23], v[2:3], v[4:5]\n"
                      "v_pk_add_f32 v[24:25], v[2:3], v[4:5]\n v_pk_add_f32 v[26:27], v[2:3], v[4:5]\n"
                      "v_pk_add_f32 v[28:29], v[2:3], v[4:5]\n v_pk_add_f32 v[30:31], v[2:3], v[4:5]\n"
                      "v_pk_add_f32 v[32:33], v[2:3], v[4:5]\n v_pk_add_f32 v[34:35], v[2:3], v[4:5]\n") ::: CLOB);
    else if (MODE == 15)  // packed fma, src2 repeated = src0 pair (two distinct pairs only)
      asm volatile(R4("v_pk_fma_f32 v[20:21], v[2:3], v[4:5], v[2:3]\n v_pk_fma_f32 v[22:23], v[2:3], v[4:5], v[2:3]\n"
                      "v_pk_fma_f32 v[24:25], v[2:3], v[4:5], v[2:3]\n v_pk_fma_f32 v[26:27], v[2:3], v[4:5], v[2:3]\n"
                      "v_pk_fma_f32 v[28:29], v[2:3], v[4:5], v[2:3]\n v_pk_fma_f32 v[30:31], v[2:3], v[4:5], v[2:3]\n"
                      "v_pk_fma_f32 v[32:33], v[2:3], v[4:5], v[2:3]\n v_pk_fma_f32 v[34:35], v[2:3], v[4:5], v[2:3]\n") ::: CLOB);
    else if (MODE == 16)  // interleaved: pk_fma with v_fma (does a plain op fill the packed op's extra cycles?)
      asm volatile(R4("v_pk_fma_f32 v[20:21], v[2:3], v[4:5], v[10:11]\n v_fma_f32 v28, v1, v6, v7\n"
                      "v_pk_fma_f32 v[22:23], v[2:3], v[4:5], v[10:11]\n v_fma_f32 v29, v1, v6, v7\n"
                      "v_pk_fma_f32 v[24:25], v[2:3], v[4:5], v[10:11]\n v_fma_f32 v30, v1, v6, v7\n"
                      "v_pk_fma_f32 v[26:27], v[2:3], v[4:5], v[10:11]\n v_fma_f32 v31, v1, v6, v7\n") ::: CLOB);
    else if (MODE == 7)   // two-operand form with an SGPR-like constant: v_mul (2 VGPR sources, different banks)
      asm volatile(R4("v_mul_f32 v20, v1, v2\n v_mul_f32 v21, v1, v2\n v_mul_f32 v22, v1, v2\n v_mul_f32 v23, v1, v2\n"
                      "v_mul_f32 v24, v1, v2\n v_mul_f32 v25, v1, v2\n v_mul_f32 v26, v1, v2\n v_mul_f32 v27, v1, v2\n") ::: CLOB);
    else if (MODE == 8)   // v_mul, both sources in one bank
      asm volatile(R4("v_mul_f32 v20, v0, v4\n v_mul_f32 v21, v0, v4\n v_mul_f32 v22, v0, v4\n v_mul_f32 v23, v0, v4\n"
                      "v_mul_f32 v24, v0, v4\n v_mul_f32 v25, v0, v4\n v_mul_f32 v26, v0, v4\n v_mul_f32 v27, v0, v4\n") ::: CLOB);
  }
  if (iters < 0) out[0] = 1.0f;
}
template <int MODE>
void run(const char* name) {
  float* d; hipMalloc(&d, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int wps = 2; wps <= 4; wps *= 2) {
    int blocks = 256 * wps;
    k<MODE><<<blocks, 256>>>(d, 100); hipDeviceSynchronize();
    k<MODE><<<blocks, 256>>>(d, iters); hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr = (double)blocks * 4 * iters * 32;     // wave-instructions
    printf("%-44s waves/SIMD=%d  %.3f ms  %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, wps, ms,
           (ms * 1e-3 * 2.4e9) / (instr / (256.0 * 4)));
  }
  hipFree(d);
}
int main() {
  run<0>("v_fma  srcs in banks 1,2,3");
  run<1>("v_fma  srcs in banks 0,0,0");
  run<2>("v_fma  srcs in banks 0,0,1");
  run<3>("v_fma  accumulate (dst = src2)");
  run<4>("v_pk_fma pairs (2:3),(4:5),(10:11)");
  run<5>("v_pk_fma pairs (0:1),(4:5),(8:9)");
  run<6>("v_pk_fma pairs (0:1),(2:3),(4:5)");
  run<9>("v_pk_fma accumulate (dst = src2)");
  run<10>("v_pk_fma src0 == src1");
  run<11>("v_pk_fma one SGPR-pair source");
  run<12>("v_pk_mul pairs (2:3),(4:5)");
  run<13>("v_pk_mul pairs (0:1),(4:5)");
  run<14>("v_pk_add pairs (2:3),(4:5)");
  run<15>("v_pk_fma src2 == src0 (two distinct pairs)");
  run<16>("v_pk_fma + v_fma alternating (per instruction)");
  run<7>("v_mul  srcs in banks 1,2");
  run<8>("v_mul  srcs in banks 0,0");
  return 0;
}
