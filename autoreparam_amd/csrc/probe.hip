// Measurement hook: the shader clock the chip holds under a vector-bound load.
//
// Every figure bench.py prices against a peak assumes a clock; the 157.3 TFLOP/s FP32 peak is 2.4 GHz, the boxes hold 2.08 -
// 2.22 GHz under the headline kernel (profiles/r04_box_to_box.txt) and the difference is most of the box-to-box spread.
// arp_clock_probe runs packed FMAs on every SIMD (two resident waves each: the kernel claims 256 registers) for a given number
// of loop iterations while one wave reads s_memtime (shader cycles) and s_memrealtime (a constant 100 MHz counter) around its
// loop: cycles / realtime ticks x 100 MHz = the clock held, measured live on the box the numbers come from.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "host_common.h"

namespace arp {

typedef float probe_v2f __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void clock_probe_kernel(unsigned long long* __restrict__ out, int iters) {
  probe_v2f a0 = {1.0f + threadIdx.x, 2.0f}, a1 = {3.0f, 4.0f}, a2 = {5.0f, 6.0f}, a3 = {7.0f, 8.0f};
  probe_v2f a4 = {1.5f, 2.5f}, a5 = {3.5f, 4.5f}, a6 = {5.5f, 6.5f}, a7 = {7.5f, 8.5f};
  const probe_v2f x = {0.999f, 1.001f}, y = {1.0e-3f, -1.0e-3f};
  const unsigned long long c0 = __builtin_readcyclecounter();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      asm volatile("v_pk_fma_f32 %0, %8, %0, %9\n v_pk_fma_f32 %1, %8, %1, %9\n v_pk_fma_f32 %2, %8, %2, %9\n v_pk_fma_f32 %3, %8, %3, %9\n"
                   "v_pk_fma_f32 %4, %8, %4, %9\n v_pk_fma_f32 %5, %8, %5, %9\n v_pk_fma_f32 %6, %8, %6, %9\n v_pk_fma_f32 %7, %8, %7, %9\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y) : "v250");
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  const probe_v2f s = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) {
    out[0] = c1 - c0; out[1] = r1 - r0;
    out[2] = (unsigned long long)(s[0] + s[1] != 0.0f);       // keeps the arithmetic alive
  }
}

}  // namespace arp

extern "C" int arp_clock_probe(int iters, unsigned long long* cycles_ticks, void* stream) {
  using namespace arp;
  if (!cycles_ticks || iters <= 0) { set_error("arp_clock_probe: cycles_ticks (device, 3 x uint64) and iters > 0 are required"); return 1; }
  // 512 workgroups of 256 threads at 256 registers per lane: exactly two resident waves on every SIMD of the chip
  hipLaunchKernelGGL(clock_probe_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, cycles_ticks, iters);
  ARP_HIP_OK(hipGetLastError());
  return 0;
}
