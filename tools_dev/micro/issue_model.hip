// How do MFMA and plain VALU instructions share a SIMD's issue?  (gfx950)
//  mode 0: every wave = MFMA loop with NV independent VALU ops after each MFMA
//  mode 1: even workgroups = pure MFMA loop, odd workgroups = pure VALU loop (2 workgroups per CU -> one of each per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NV>
__global__ __launch_bounds__(256) void same_wave(float* out, int iters, unsigned long long* cyc) {
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
  int v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
  unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < NV; ++k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[k % 8]) : "v"(it));
    }
  }
  unsigned long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i][0];
  int sv = 0;
  for (int i = 0; i < 8; ++i) sv += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + sv;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ __launch_bounds__(256) void mixed(float* out, int iters, unsigned long long* cyc, int valu_per_iter) {
  unsigned long long t0 = clock64();
  float s = 0.f;
  if ((blockIdx.x >> 8) & 1) {        // second wave of 256 workgroups = the co-resident one
    int v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[k % 8]) : "v"(it));
    }
    for (int i = 0; i < 8; ++i) s += v[i];
  } else {
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) s += acc[i][0];
  }
  unsigned long long t1 = clock64();
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NV>
static void run_same(float* out, unsigned long long* cyc, int wgs, int iters) {
  hipLaunchKernelGGL(same_wave<NV>, dim3(wgs), dim3(256), 0, 0, out, iters, cyc);
  hipDeviceSynchronize();
  unsigned long long h[1024];
  hipMemcpy(h, cyc, sizeof(unsigned long long) * wgs, hipMemcpyDeviceToHost);
  double m = 0;
  for (int i = 0; i < wgs; ++i) m += h[i];
  m /= wgs;
  printf("same wave, %d wg, %d VALU per MFMA: %.1f cycles per MFMA (32 = pipe bound)\n", wgs, NV, m / (iters * 4.0));
}
int main() {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, 1024 * 256 * 4);
  hipMalloc(&cyc, 1024 * 8);
  int iters = 20000;
  for (int wgs : {256, 512}) {
    run_same<0>(out, cyc, wgs, iters);
    run_same<1>(out, cyc, wgs, iters);
    run_same<2>(out, cyc, wgs, iters);
    run_same<4>(out, cyc, wgs, iters);
    run_same<6>(out, cyc, wgs, iters);
    run_same<8>(out, cyc, wgs, iters);
  }
  hipLaunchKernelGGL(mixed, dim3(512), dim3(256), 0, 0, out, iters, cyc, 16);
  hipDeviceSynchronize();
  unsigned long long h[512];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double mm = 0, mv = 0;
  for (int i = 0; i < 256; ++i) { mm += h[i]; mv += h[256 + i]; }
  printf("mixed: MFMA wave %.1f cycles per MFMA; VALU wave %.1f cycles per VALU (both loops %d iterations x 16)\n",
         mm / 256 / (iters * 16.0), mv / 256 / (iters * 16.0), iters);
  // VALU alone
  hipLaunchKernelGGL(mixed, dim3(512), dim3(256), 0, 0, out, 0, cyc, 16);
  hipDeviceSynchronize();
  return 0;
}
