// cycles of the fp32 -> three bf16 pieces split: hardware RNE (v_cvt_pk_bf16_f32) vs truncation (v_and / v_perm)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_rne(const float4& v, bf16x4* p0, bf16x4* p1, bf16x4* p2) {
  const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const __bf16 a = (__bf16)x[i];
    const float r1 = x[i] - (float)a;
    const __bf16 b = (__bf16)r1;
    const float r2 = r1 - (float)b;
    (*p0)[i] = a; (*p1)[i] = b; (*p2)[i] = (__bf16)r2;
  }
}
__device__ __forceinline__ void split_trunc(const float4& v, u32x2* p0, u32x2* p1, u32x2* p2) {
  const float x[4] = {v.x, v.y, v.z, v.w};
  unsigned a[4], b[4], c[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = __float_as_uint(x[i]) & 0xffff0000u;
    const float r1 = x[i] - __uint_as_float(a[i]);
    b[i] = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(b[i]);
    c[i] = __float_as_uint(r2);
  }
  (*p0)[0] = __builtin_amdgcn_perm(a[1], a[0], 0x07060302u); (*p0)[1] = __builtin_amdgcn_perm(a[3], a[2], 0x07060302u);
  (*p1)[0] = __builtin_amdgcn_perm(b[1], b[0], 0x07060302u); (*p1)[1] = __builtin_amdgcn_perm(b[3], b[2], 0x07060302u);
  (*p2)[0] = __builtin_amdgcn_perm(c[1], c[0], 0x07060302u); (*p2)[1] = __builtin_amdgcn_perm(c[3], c[2], 0x07060302u);
}
template <int MODE>
__global__ void k(const float4* in, u32x2* out, long long* cyc) {
  float4 v[16];
  for (int i = 0; i < 16; ++i) v[i] = in[threadIdx.x + i * 256];
  __syncthreads();
  long long t0 = clock64();
  u32x2 acc = {0u, 0u};
  for (int rep = 0; rep < 64; ++rep) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      u32x2 q0, q1, q2;
      if (MODE == 0) {
        bf16x4 p0, p1, p2;
        split_rne(v[i], &p0, &p1, &p2);
        q0 = __builtin_bit_cast(u32x2, p0); q1 = __builtin_bit_cast(u32x2, p1); q2 = __builtin_bit_cast(u32x2, p2);
      } else {
        split_trunc(v[i], &q0, &q1, &q2);
      }
      acc ^= q0 ^ q1 ^ q2;
      v[i].x += __uint_as_float(acc[0] & 1u);      // dependence between repetitions
    }
  }
  long long t1 = clock64();
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  float4* in; u32x2* out; long long* cyc;
  hipMalloc(&in, 4096 * 16); hipMalloc(&out, 256 * 256 * 8); hipMalloc(&cyc, 256 * 8);
  hipMemset(in, 0x3f, 4096 * 16);
  for (int mode = 0; mode < 2; ++mode) {
    for (int it = 0; it < 2; ++it) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, in, out, cyc);
      else hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, in, out, cyc);
    }
    hipDeviceSynchronize();
    long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
    printf("%s: %.0f clock64 ticks per 16-slot split (64 floats per lane)\n", mode ? "trunc" : "rne  ", s / 256 / 64);
  }
  // exactness of the truncated split on the host is checked in tools_dev/bf16_split_emulation.py
  return 0;
}
