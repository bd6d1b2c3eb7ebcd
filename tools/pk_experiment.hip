// Experiment: radon CP leapfrog inner pass, scalar f32 vs packed v2f formulation (K=4 lanes, 17/18 counties per lane).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int CTRL> __device__ __forceinline__ float dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true)); }
__device__ __forceinline__ float gsum4(float v) { v += dpp<0xB1>(v); v += dpp<0x4E>(v); return v; }
__device__ __forceinline__ v2f vfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

constexpr int NL = 18;
__global__ __launch_bounds__(256, 2) void k_scalar(const float* tab, float* out, int iters) {
  float u[NL], sx[NL], sy[NL], n[NL], q[NL], p[NL], e[NL];
  int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < NL; ++i) { u[i] = tab[i * 4 + (t & 3)]; sx[i] = tab[100 + i * 4 + (t & 3)]; sy[i] = tab[200 + i * 4 + (t & 3)]; n[i] = tab[300 + i * 4 + (t & 3)];
    q[i] = 0.01f * (t + i); p[i] = 0.02f * i; e[i] = 0.01f; }
  float mua = 0.1f, b1 = 0.2f, b2 = 0.3f, p0 = 0, p1 = 0, p2 = 0;
  for (int it = 0; it < iters; ++it) {
    float ah = 0, auh = 0, ams = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      float mt = q[i];
      float mu = fmaf(u[i], b1, mua);
      float tt = fmaf(-b2, sx[i], sy[i]);
      float r = mt - mu;
      float l = fmaf(-n[i], mt, tt);
      float gm = l - r;
      ah += r; auh = fmaf(u[i], r, auh); ams = fmaf(mt, sx[i], ams);
      float pn = fmaf(e[i], gm, p[i]); p[i] = pn; q[i] = fmaf(e[i], pn, mt);
    }
    ah = gsum4(ah); auh = gsum4(auh); ams = gsum4(ams);
    p0 = fmaf(0.01f, ah - mua, p0); mua = fmaf(0.01f, p0, mua);
    p1 = fmaf(0.01f, auh - b1, p1); b1 = fmaf(0.01f, p1, b1);
    p2 = fmaf(0.01f, -ams - b2, p2); b2 = fmaf(0.01f, p2, b2);
  }
  float s = mua + b1 + b2;
#pragma unroll
  for (int i = 0; i < NL; ++i) s += q[i] + p[i];
  out[blockIdx.x * 256 + t] = s;
}
__global__ __launch_bounds__(256, 2) void k_packed(const float* tab, float* out, int iters) {
  constexpr int NP = NL / 2;
  v2f u[NP], sx[NP], sy[NP], n[NP], q[NP], p[NP], e[NP];
  int t = threadIdx.x;
#pragma unroll
  for (int k = 0; k < NP; ++k) for (int h = 0; h < 2; ++h) { int i = 2 * k + h;
    u[k][h] = tab[i * 4 + (t & 3)]; sx[k][h] = tab[100 + i * 4 + (t & 3)]; sy[k][h] = tab[200 + i * 4 + (t & 3)]; n[k][h] = tab[300 + i * 4 + (t & 3)];
    q[k][h] = 0.01f * (t + i); p[k][h] = 0.02f * i; e[k][h] = 0.01f; }
  float mua = 0.1f, b1 = 0.2f, b2 = 0.3f, p0 = 0, p1 = 0, p2 = 0;
  for (int it = 0; it < iters; ++it) {
    v2f ah = {0, 0}, auh = {0, 0}, ams = {0, 0};
    const v2f vb1 = {b1, b1}, vmua = {mua, mua}, vnb2 = {-b2, -b2};
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      v2f mt = q[k];
      v2f mu = vfma(u[k], vb1, vmua);
      v2f tt = vfma(vnb2, sx[k], sy[k]);
      v2f r = mt - mu;
      v2f l = vfma(-n[k], mt, tt);
      v2f gm = l - r;
      ah += r; auh = vfma(u[k], r, auh); ams = vfma(mt, sx[k], ams);
      v2f pn = vfma(e[k], gm, p[k]); p[k] = pn; q[k] = vfma(e[k], pn, mt);
    }
    float sah = gsum4(ah.x + ah.y), sauh = gsum4(auh.x + auh.y), sams = gsum4(ams.x + ams.y);
    p0 = fmaf(0.01f, sah - mua, p0); mua = fmaf(0.01f, p0, mua);
    p1 = fmaf(0.01f, sauh - b1, p1); b1 = fmaf(0.01f, p1, b1);
    p2 = fmaf(0.01f, -sams - b2, p2); b2 = fmaf(0.01f, p2, b2);
  }
  float s = mua + b1 + b2;
#pragma unroll
  for (int k = 0; k < NP; ++k) s += q[k].x + q[k].y + p[k].x + p[k].y;
  out[blockIdx.x * 256 + t] = s;
}
int main() {
  float *tab, *out; hipMalloc(&tab, 4096); hipMemset(tab, 0, 4096); hipMalloc(&out, 1024 * 256 * 4 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wps = 1; wps <= 2; ++wps) {
    int blocks = 256 * wps * 2;   // two rounds
    for (int v = 0; v < 2; ++v) {
      if (v == 0) k_scalar<<<blocks, 256>>>(tab, out, 10); else k_packed<<<blocks, 256>>>(tab, out, 10);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      if (v == 0) k_scalar<<<blocks, 256>>>(tab, out, 2000); else k_packed<<<blocks, 256>>>(tab, out, 2000);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%s blocks=%d: %.3f ms  -> %.3e lane-leapfrogs/s\n", v ? "packed" : "scalar", blocks, ms, (double)blocks * 256 * 2000 / (ms * 1e-3));
    }
  }
  return 0;
}
