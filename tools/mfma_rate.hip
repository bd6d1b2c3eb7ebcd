// Issue rate of v_mfma_f32_16x16x4_f32 / 32x32x2 for a single wave per SIMD (the German-credit
// likelihood's situation): cycles per MFMA with 2 or 4 accumulation chains, operands in VGPRs.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ void __launch_bounds__(256) k16(float* out, int iters, float seed) {
  v4f acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = v4f{0, 0, 0, 0};
  float a[16], b[16];
  for (int i = 0; i < 16; ++i) { a[i] = seed + threadIdx.x * 1e-3f + i; b[i] = seed * 0.5f + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s)
      acc[s % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc[s % CHAINS], 0, 0, 0);
  }
  float r = 0;
  for (int c = 0; c < CHAINS; ++c) r += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
// 16x16x4 MFMAs with NV independent VALU FMAs after each: do they hide under the MFMA?
template <int NV>
__global__ void __launch_bounds__(256) k16v(float* out, int iters, float seed) {
  v4f acc[4];
  for (int c = 0; c < 4; ++c) acc[c] = v4f{0, 0, 0, 0};
  float a[16], b[16], f[8];
  for (int i = 0; i < 16; ++i) { a[i] = seed + threadIdx.x * 1e-3f + i; b[i] = seed * 0.5f + i; }
  for (int i = 0; i < 8; ++i) f[i] = seed + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      acc[s % 4] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc[s % 4], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; ++v) f[v] = __builtin_fmaf(f[v], 1.0001f, 0.5f);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float r = 0;
  for (int c = 0; c < 4; ++c) r += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  for (int i = 0; i < 8; ++i) r += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
__global__ void __launch_bounds__(256) k32(float* out, int iters, float seed) {
  v16f acc[2];
  for (int c = 0; c < 2; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0;
  float a[16], b[16];
  for (int i = 0; i < 16; ++i) { a[i] = seed + threadIdx.x * 1e-3f + i; b[i] = seed * 0.5f + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s)
      acc[s % 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc[s % 2], 0, 0, 0);
  }
  float r = 0;
  for (int c = 0; c < 2; ++c) for (int i = 0; i < 16; ++i) r += acc[c][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <class F>
static void run(const char* name, F launch, int blocks, int iters, double mfma_per_iter, double flop_per_mfma) {
  float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0); launch(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double waves_per_simd = blocks * 4.0 / (256 * 4);
  double n = (double)iters * mfma_per_iter;                     // MFMAs per wave
  printf("%-28s blocks %4d (%.0f waves/SIMD): %.3f ms, %.1f ns per MFMA per SIMD-slot, %.1f TFLOP/s\n", name, blocks,
         waves_per_simd, ms, ms * 1e6 / (n * waves_per_simd), blocks * 4.0 * n * flop_per_mfma / (ms * 1e-3) / 1e12);
  hipFree(out);
}
int main() {
  for (int blocks : {256, 512}) {
    run("16x16x4 f32, 1 chain", [&](float* o, int it) { hipLaunchKernelGGL(k16<1>, dim3(blocks), dim3(256), 0, 0, o, it, 1.0f); }, blocks, 4000, 16, 2048);
    run("16x16x4 f32, 2 chains", [&](float* o, int it) { hipLaunchKernelGGL(k16<2>, dim3(blocks), dim3(256), 0, 0, o, it, 1.0f); }, blocks, 4000, 16, 2048);
    run("16x16x4 f32, 4 chains", [&](float* o, int it) { hipLaunchKernelGGL(k16<4>, dim3(blocks), dim3(256), 0, 0, o, it, 1.0f); }, blocks, 4000, 16, 2048);
    run("16x16x4 f32 + 2 v_fma each", [&](float* o, int it) { hipLaunchKernelGGL(k16v<2>, dim3(blocks), dim3(256), 0, 0, o, it, 1.0f); }, blocks, 4000, 16, 2048);
    run("16x16x4 f32 + 4 v_fma each", [&](float* o, int it) { hipLaunchKernelGGL(k16v<4>, dim3(blocks), dim3(256), 0, 0, o, it, 1.0f); }, blocks, 4000, 16, 2048);
    run("16x16x4 f32 + 8 v_fma each", [&](float* o, int it) { hipLaunchKernelGGL(k16v<8>, dim3(blocks), dim3(256), 0, 0, o, it, 1.0f); }, blocks, 4000, 16, 2048);
    run("32x32x2 f32, 2 chains", [&](float* o, int it) { hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, o, it, 1.0f); }, blocks, 2000, 16, 4096);
  }
  return 0;
}
