// Does work of a SECOND wave on the same SIMD hide under the f32 MFMAs of the first?  (tools/mfma_rate.hip shows
// that VALU instructions of the SAME wave do not.)  512-thread workgroups, one per CU: waves 0-3 run a chain of
// v_mfma_f32_16x16x4_f32, waves 4-7 a stream of one instruction class.  Times: MFMA waves alone, the other waves alone,
// both together.  together ~ max(alone) = the classes overlap; together ~ sum = they share the datapath.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_overlap.hip -o /tmp/mfma_overlap && /tmp/mfma_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

enum { OP_FMA, OP_PKFMA, OP_EXP, OP_RCP, OP_MOV, OP_IADD, OP_LDS, OP_MFMA, OP_ACCRD };

template <int OP>
__device__ __forceinline__ void op_stream(float* out, int iters, float seed, float* lds) {
  float f[8];
  v2f g[8];
  v4f acc[4];
  for (int i = 0; i < 8; ++i) { f[i] = seed + i + threadIdx.x * 1e-3f; g[i] = v2f{f[i], f[i] + 1.0f}; }
  for (int c = 0; c < 4; ++c) acc[c] = v4f{0, 0, 0, 0};
  unsigned u[8];
  for (int i = 0; i < 8; ++i) u[i] = threadIdx.x + i;
  const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)lds + (threadIdx.x & 63) * 16;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        if (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[v]) : "v"(seed));
        if (OP == OP_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(g[v]) : "v"(g[(v + 1) & 7]));
        if (OP == OP_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(f[v]));
        if (OP == OP_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[v]));
        if (OP == OP_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(f[v]) : "v"(f[(v + 1) & 7]));
        if (OP == OP_IADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[v]) : "v"(u[(v + 1) & 7]));
        if (OP == OP_LDS) { v4f t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(la)); if (v == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        if (OP == OP_MFMA && v < 2) acc[(2 * s + v) & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[v], f[v + 2], acc[(2 * s + v) & 3], 0, 0, 0);
      }
    }
  }
  float r = 0;
  for (int i = 0; i < 8; ++i) r += f[i] + g[i].x + g[i].y + (float)u[i];
  for (int c = 0; c < 4; ++c) r += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

__device__ __forceinline__ void mfma_stream(float* out, int iters, float seed) {
  v4f acc[4];
  for (int c = 0; c < 4; ++c) acc[c] = v4f{0, 0, 0, 0};
  float a[16], b[16];
  for (int i = 0; i < 16; ++i) { a[i] = seed + threadIdx.x * 1e-3f + i; b[i] = seed * 0.5f + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s)
      acc[s % 4] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc[s % 4], 0, 0, 0);
  }
  float r = 0;
  for (int c = 0; c < 4; ++c) r += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// mode bit 0: MFMA waves work, bit 1: the other waves work
template <int OP>
__global__ void __launch_bounds__(512) both(float* out, int iters, float seed, int mode) {
  __shared__ float lds[64 * 4 + 64];
  if (threadIdx.x < 320) lds[threadIdx.x] = seed;
  __syncthreads();
  if (threadIdx.x < 256) {
    if (mode & 1) mfma_stream(out, iters, seed);
  } else {
    if (mode & 2) op_stream<OP>(out, iters, seed, lds);
  }
}

template <int OP>
static void run(const char* name, int ops_per_iter) {
  const int blocks = 256, iters = 4000;
  float* out; hipMalloc(&out, (size_t)blocks * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms[4] = {0, 0, 0, 0};
  for (int mode = 1; mode <= 3; ++mode) {
    hipLaunchKernelGGL(both<OP>, dim3(blocks), dim3(512), 0, 0, out, 10, 1.0f, mode);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(both<OP>, dim3(blocks), dim3(512), 0, 0, out, iters, 1.0f, mode);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms[mode], e0, e1);
  }
  printf("%-14s mfma alone %.3f ms (%.1f ns/MFMA) | %s alone %.3f ms (%.2f ns/op) | together %.3f ms  (sum %.3f, max %.3f)\n",
         name, ms[1], ms[1] * 1e6 / (iters * 16.0), name, ms[2], ms[2] * 1e6 / ((double)iters * ops_per_iter), ms[3],
         ms[1] + ms[2], ms[1] > ms[2] ? ms[1] : ms[2]);
  hipFree(out);
}

int main() {
  run<OP_FMA>("v_fma_f32", 128);
  run<OP_PKFMA>("v_pk_fma_f32", 128);
  run<OP_EXP>("v_exp_f32", 128);
  run<OP_RCP>("v_rcp_f32", 128);
  run<OP_MOV>("v_mov_b32", 128);
  run<OP_IADD>("v_add_u32", 128);
  run<OP_LDS>("ds_read_b128", 128);
  run<OP_MFMA>("mfma (2nd wave)", 32);
  return 0;
}
