// cycles per v_mfma_f32_16x16x32_bf16 for one wavefront per SIMD: 8 independent accumulators, jw-outer order as conv_xb.hip,
// (a) pure, (b) with 8 ds_read_b128 per 16 MFMAs, (c) + 3 global loads per 48
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const bf16x8* g, float* out, long long* cyc, int reps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 36864 / 16; i += 256) reinterpret_cast<bf16x8*>(smem)[i] = g[i & 1023];
  __syncthreads();
  f32x4 acc[8];
  for (int t = 0; t < 8; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 A[3], An[3], B[2][8];
  for (int j = 0; j < 3; ++j) An[j] = g[lane + 64 * j];
  for (int j = 0; j < 3; ++j) A[j] = g[lane + 64 * j];
  for (int t = 0; t < 8; ++t) B[0][t] = B[1][t] = g[lane + 64 * (t + 3)];
  long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int ph = 0; ph < 6; ++ph) {
      if (MODE == 3) {
        // one memory instruction in the shadow of each MFMA
        int m = 0;
#pragma unroll
        for (int jw = 0; jw < 3; ++jw) {
          if (jw > 2 - ph % 3) continue;
#pragma unroll
          for (int t = 0; t < 8; ++t, ++m) {
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[jw], B[ph & 1][t], acc[t], 0, 0, 0);
            if (m < 8) B[(ph + 1) & 1][m] = *reinterpret_cast<const bf16x8*>(smem + lane * 16 + m * 768 + (ph % 3) * 9216 + (r & 3) * 1024);
            else if (ph % 3 == 1 && m < 11) An[m - 8] = g[lane + 64 * ((m - 8) + 3 * ((r + ph) & 63))];
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (ph % 3 == 2) { A[0] = An[0]; A[1] = An[1]; A[2] = An[2]; }
        continue;
      }
      if (MODE >= 1) {
#pragma unroll
        for (int t = 0; t < 8; ++t) B[(ph + 1) & 1][t] = *reinterpret_cast<const bf16x8*>(smem + lane * 16 + t * 768 + (ph % 3) * 9216 + (r & 3) * 1024);
      }
      if (MODE >= 2 && ph % 3 == 0) {
#pragma unroll
        for (int j = 0; j < 3; ++j) A[j] = g[lane + 64 * (j + 3 * ((r + ph) & 63))];
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int jw = 0; jw < 3; ++jw) {
        if (jw > 2 - ph % 3) continue;
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[jw], B[ph & 1][t], acc[t], 0, 0, 0);
      }
    }
  }
  long long t1 = clock64();
  float s = 0.f;
  for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
int main() {
  bf16x8* g; float* out; long long* cyc;
  hipMalloc(&g, 1 << 20); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
  hipMemset(g, 0, 1 << 20);
  const int reps = 200;
  for (int mode = 0; mode < 4; ++mode) {
    for (int it = 0; it < 2; ++it) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 40960, 0, g, out, cyc, reps);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 40960, 0, g, out, cyc, reps);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 40960, 0, g, out, cyc, reps);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 40960, 0, g, out, cyc, reps);
    }
    hipDeviceSynchronize();
    static long long h[1024];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 1024; ++i) s += h[i];
    printf("mode %d: %.2f clock64 ticks per MFMA (96 per repetition)\n", mode, s / 1024 / reps / 96);
  }
  return 0;
}
