// What an LDS read costs a wave that is streaming f32 MFMAs (one wave per SIMD, the German-credit likelihood's situation):
// 16 independent-chain v_mfma_f32_16x16x4_f32 per iteration plus NR reads of one kind issued in front of them, waited for
// at the end of the iteration (a full iteration of latency hiding).  Prints the extra time per read.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_lds.hip -o /tmp/mfma_lds && /tmp/mfma_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
enum { RD_B128, RD_B64, RD_2B32, RD_B32 };

template <int KIND, int NR>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = seed + i;
  __syncthreads();
  v4f acc[4];
  for (int c = 0; c < 4; ++c) acc[c] = v4f{0, 0, 0, 0};
  float a[16], b[16];
  for (int i = 0; i < 16; ++i) { a[i] = seed + threadIdx.x * 1e-3f + i; b[i] = seed * 0.5f + i; }
  const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 8192;
  v4f r4[8]; v2f r2[8]; float r1[8];
  for (int i = 0; i < 8; ++i) { r4[i] = v4f{0, 0, 0, 0}; r2[i] = v2f{0, 0}; r1[i] = 0; }
  float sink = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int n = 0; n < NR; ++n) {
      if (KIND == RD_B128) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r4[n]) : "v"(la), "n"(n * 1024));
      if (KIND == RD_B64) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r2[n]) : "v"(la), "n"(n * 1024));
      if (KIND == RD_2B32) asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(r2[n]) : "v"(la), "n"(n * 32), "n"(n * 32 + 16));
      if (KIND == RD_B32) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r1[n]) : "v"(la), "n"(n * 1024));
    }
#pragma unroll
    for (int s = 0; s < 16; ++s)
      acc[s % 4] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc[s % 4], 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int n = 0; n < NR; ++n) {
      if (KIND == RD_B128) asm volatile("" :: "v"(r4[n]));
      if (KIND == RD_B64 || KIND == RD_2B32) asm volatile("" :: "v"(r2[n]));
      if (KIND == RD_B32) asm volatile("" :: "v"(r1[n]));
    }
  }
  float r = sink;
  for (int c = 0; c < 4; ++c) r += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

static float run1(void (*kern)(float*, int, float), int iters) {
  const int blocks = 256;
  static float* out = nullptr;
  if (!out) hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
template <int KIND> static void kind(const char* name, float base, int iters) {
  float t4 = run1(k<KIND, 4>, iters), t8 = run1(k<KIND, 8>, iters);
  printf("%-14s  +4 reads/16 MFMA: %.3f ms (%.1f ns per read)   +8 reads: %.3f ms (%.1f ns per read)\n", name, t4,
         (t4 - base) * 1e6 / (iters * 4.0), t8, (t8 - base) * 1e6 / (iters * 8.0));
}
int main() {
  const int iters = 4000;
  float base = run1(k<RD_B128, 0>, iters);
  printf("16 MFMA per iteration alone: %.3f ms (%.2f ns per MFMA)\n", base, base * 1e6 / (iters * 16.0));
  kind<RD_B128>("ds_read_b128", base, iters);
  kind<RD_B64>("ds_read_b64", base, iters);
  kind<RD_2B32>("ds_read2_b32", base, iters);
  kind<RD_B32>("ds_read_b32", base, iters);
  return 0;
}
