// Micro-benchmark: issue rate of f32 VALU forms on gfx950 as a function of waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_bench.hip -o gpurun_out/valu_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  v2f y0 = {x0, x1}, y1 = {x2, x3}, y2 = {x4, x5}, y3 = {x6, x7}, y4 = {x1, x0}, y5 = {x3, x2}, y6 = {x5, x4}, y7 = {x7, x6};
  v2f aa = {a, a}, bb = {b, b};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
      REP8(asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
                        "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n"
                        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));)
    } else if (MODE == 1) {
      REP8(asm volatile("v_pk_fma_f32 %0, %8, %9, %0\n v_pk_fma_f32 %1, %8, %9, %1\n v_pk_fma_f32 %2, %8, %9, %2\n v_pk_fma_f32 %3, %8, %9, %3\n"
                        "v_pk_fma_f32 %4, %8, %9, %4\n v_pk_fma_f32 %5, %8, %9, %5\n v_pk_fma_f32 %6, %8, %9, %6\n v_pk_fma_f32 %7, %8, %9, %7\n"
                        : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7) : "v"(aa), "v"(bb));)
    } else if (MODE == 2) {  // DPP add
      REP8(asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_add_f32_dpp %2, %2, %2 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_add_f32_dpp %6, %6, %6 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_f32_dpp %7, %7, %7 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));)
    } else if (MODE == 3) {  // transcendental
      REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_sin_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
                        : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));)
    } else if (MODE == 4) {  // 32x32->64 multiply-add (Philox)
      unsigned long long z0 = x0, z1 = x1, z2 = x2, z3 = x3;
      unsigned c = (unsigned)a;
      REP8(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                        "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n"
                        : "+v"(z0), "+v"(z1), "+v"(z2), "+v"(z3) : "v"(c), "v"(c) : "vcc");)
      x0 += (float)(z0 + z1 + z2 + z3);
    } else if (MODE == 5) {  // int xor/shift mix (xoshiro-like)
      unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1), u2 = __float_as_uint(x2), u3 = __float_as_uint(x3);
      REP8(asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_alignbit_b32 %2, %2, %2, 21\n v_add_u32 %3, %3, %0\n"
                        "v_xor_b32 %0, %0, %3\n v_lshlrev_b32 %1, 9, %1\n v_xor_b32 %2, %2, %1\n v_add_u32 %3, %3, %2\n"
                        : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));)
      x0 += __uint_as_float(u0 ^ u1 ^ u2 ^ u3);
    }
  }
  float r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + y0.x + y0.y + y1.x + y1.y + y2.x + y2.y + y3.x + y3.y + y4.x + y5.x + y6.x + y7.x;
  if (r == 12345.678f) out[0] = r;
}
template <int MODE>
void run(const char* name, int flops_per_instr) {
  float* d; hipMalloc(&d, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int wps = 1; wps <= 8; wps *= 2) {
    int blocks = 256 * wps;  // 4 waves per block -> wps waves per SIMD if evenly spread
    k<MODE><<<blocks, 256>>>(d, 100, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr = (double)blocks * 4 * iters * 64;  // wave-instructions
    double cyc_per_instr_per_simd = (ms * 1e-3 * 2.4e9) / (instr / (256.0 * 4));
    printf("%-10s waves/SIMD=%d  %.3f ms  %.2f cyc/wave-instr/SIMD (at 2.4GHz)  %.1f Tlane-op/s  %.1f TFLOP/s\n", name, wps, ms,
           cyc_per_instr_per_simd, instr * 64 / (ms * 1e-3) / 1e12, instr * 64 * flops_per_instr / (ms * 1e-3) / 1e12);
  }
  hipFree(d);
}
int main() {
  run<0>("v_fma", 2); run<1>("v_pk_fma", 4); run<2>("add_dpp", 1); run<3>("transc", 1); run<4>("mad_u64", 1); run<5>("int_mix", 1);
  return 0;
}
