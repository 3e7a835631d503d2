// issue rate of v_pk_fma_f32 by operand form (the FMA stems / heads: a broadcast VGPR half x an SGPR weight pair)
//   hipcc --offload-arch=gfx950 -O3 -o pk_fma_rate pk_fma_rate.hip && ./pk_fma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define REP8(X) X X X X X X X X
template <int MODE>
__global__ void k(const float* in, float* out, long long* cyc, int reps) {
  f32x2 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (f32x2){in[i], in[i + 1]};
  f32x2 a = {in[threadIdx.x & 63], in[(threadIdx.x & 63) + 1]};
  f32x2 bv = {in[70], in[71]};
  f32x2 bs = {in[blockDim.x], in[blockDim.x + 1]};          // uniform: an SGPR pair
  __syncthreads();
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(bv));
      if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "s"(bs));
      if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(a), "v"(bv));
      if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(a), "s"(bs));
      if (MODE == 4) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(a.x), "v"(bv.x));
      if (MODE == 5) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(a.x), "s"(bs.x));
      if (MODE == 6) asm volatile("v_pk_mul_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(bv));
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* name, const float* in, float* out, long long* cyc, int waves_per_simd) {
  const int reps = 16384, block = 256 * waves_per_simd;
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(block), 0, 0, in, out, cyc, reps);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(block), 0, 0, in, out, cyc, reps);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double c = 0; for (int i = 0; i < 256; ++i) c += h[i];
  c /= 256;
  printf("%-20s waves/SIMD %d: %6.2f clock64 ticks, %6.3f ns per instruction per SIMD  (%.3f ms)\n", name, waves_per_simd,
         c / (double)(reps * 16 * waves_per_simd), ms * 1e6 / (double)(reps * 16 * waves_per_simd), ms);
}
int main() {
  float* in; float* out; long long* cyc;
  hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
  float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = 1.0f + i * 1e-4f;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  for (int w = 1; w <= 4; w *= 2) {
    run<0>("pk_fma v,v", in, out, cyc, w);
    run<1>("pk_fma v,s", in, out, cyc, w);
    run<2>("pk_fma v(bcast),v", in, out, cyc, w);
    run<3>("pk_fma v(bcast),s", in, out, cyc, w);
    run<4>("fma v,v", in, out, cyc, w);
    run<5>("fma v,s", in, out, cyc, w);
    run<6>("pk_mul v,v", in, out, cyc, w);
  }
  return 0;
}
