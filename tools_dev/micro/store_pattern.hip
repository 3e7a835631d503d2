// How fast does gfx950 take 1 GB of stores in the full-resolution decoder's tile pattern (deconv_up_b<8>: a workgroup writes
// 8 planes x 8 rows x 1 KB per tile, rows 10 KB apart, planes 2.6 MB apart) against a linear fill?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void linear_kernel(f32x4* out, long n4, int nt) {
  const f32x4 v = {1.f, 2.f, 3.f, 4.f};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    if (nt) __builtin_nontemporal_store(v, out + i); else out[i] = v;
  }
}
// each workgroup owns a contiguous span, wave stores 1 KB at a time
__global__ __launch_bounds__(256) void span_kernel(f32x4* out, long n4, int nt) {
  const f32x4 v = {1.f, 2.f, 3.f, 4.f};
  const long per = n4 / gridDim.x;
  f32x4* o = out + (long)blockIdx.x * per;
  for (long i = threadIdx.x; i < per; i += 256) {
    if (nt) __builtin_nontemporal_store(v, o + i); else o[i] = v;
  }
}

// TXO: output voxels per row segment (32 B each); TYO rows x TZO planes per tile; wave w: planes [w*TZO/4, ..)
template <int TXO, int TYO, int TZO>
__global__ __launch_bounds__(256) void tile_kernel(float* out, int Do, int Ho, int Wo, int wg, int nt, int order) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = blockIdx.x / wg, lbk = blockIdx.x - grp * wg;
  float* yg = out + (size_t)grp * Do * Ho * Wo * 8;
  const int tiles_x = Wo / TXO, tiles_y = Ho / TYO, tiles_z = Do / TZO;
  const int ntiles = tiles_x * tiles_y * tiles_z;
  const int xcd = lbk & 7, tslot = lbk >> 3, per_xcd = (ntiles + 7) >> 3, spx = wg >> 3;
  const unsigned ybytes = (unsigned)((size_t)Do * Ho * Wo * 32);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(yg, 0, ybytes, 0x00020000);
  const u32x4 bits = {1u, 2u, 3u, 4u};
  constexpr int SEG = TXO * 32 / 1024;        // 1 KB store instructions per row segment
  for (int k = 0;; ++k) {
    int tl;
    if (order == 0) { tl = xcd * per_xcd + tslot + k * spx; if (tslot + k * spx >= per_xcd || tl >= ntiles) break; }
    else { tl = lbk + k * wg; if (tl >= ntiles) break; }       // round robin over all workgroups of the sample
    const int bx = tl % tiles_x, rest = tl / tiles_x, by = rest % tiles_y, bz = rest / tiles_y;
#pragma unroll
    for (int pz = 0; pz < TZO / 4; ++pz)
#pragma unroll
      for (int y = 0; y < TYO; ++y)
#pragma unroll
        for (int s = 0; s < SEG; ++s) {
          const unsigned z = bz * TZO + wave * (TZO / 4) + pz;
          const unsigned off = (((z * Ho + by * TYO + y) * Wo + bx * TXO) * 32u) + s * 1024u + lane * 16u;
          if (nt) __builtin_amdgcn_raw_buffer_store_b128(bits, rs, off, 0, 2);
          else __builtin_amdgcn_raw_buffer_store_b128(bits, rs, off, 0, 0);
        }
  }
}

template <class F>
float timeit(F f, int reps = 10) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  const int G = 8, Do = 192, Ho = 128, Wo = 160;
  const size_t bytes = (size_t)G * Do * Ho * Wo * 32;
  float* out;
  CK(hipMalloc(&out, bytes));
  const double gb = bytes / 1e9;
  auto rep = [&](const char* name, float ms) { printf("%-44s %.3f ms  %.2f TB/s\n", name, ms, gb / ms); };
  for (int nt = 0; nt < 2; ++nt) {
    printf("--- nontemporal = %d\n", nt);
    rep("linear grid-stride, 2048 wg", timeit([&] { hipLaunchKernelGGL(linear_kernel, dim3(2048), dim3(256), 0, 0, (f32x4*)out, (long)(bytes / 16), nt); }));
    rep("linear grid-stride, 256 wg", timeit([&] { hipLaunchKernelGGL(linear_kernel, dim3(256), dim3(256), 0, 0, (f32x4*)out, (long)(bytes / 16), nt); }));
    rep("contiguous span per wg, 256 wg", timeit([&] { hipLaunchKernelGGL(span_kernel, dim3(256), dim3(256), 0, 0, (f32x4*)out, (long)(bytes / 16), nt); }));
    rep("contiguous span per wg, 1024 wg", timeit([&] { hipLaunchKernelGGL(span_kernel, dim3(1024), dim3(256), 0, 0, (f32x4*)out, (long)(bytes / 16), nt); }));
    for (int order = 0; order < 2; ++order) {
      printf("  tile order %s\n", order ? "round robin" : "xcd-dealt (the kernel's)");
      for (int wgs = 1; wgs <= 2; ++wgs) {
        const int wg = 256 * wgs / G;
        printf("  %d workgroup(s) per CU\n", wgs);
        rep("tile 32x x 8y x 8z (the decoder's)", timeit([&] { hipLaunchKernelGGL((tile_kernel<32, 8, 8>), dim3(wg * G), dim3(256), 0, 0, out, Do, Ho, Wo, wg, nt, order); }));
        rep("tile 32x x 16y x 4z", timeit([&] { hipLaunchKernelGGL((tile_kernel<32, 16, 4>), dim3(wg * G), dim3(256), 0, 0, out, Do, Ho, Wo, wg, nt, order); }));
        rep("tile 160x x 8y x 8z (whole rows)", timeit([&] { hipLaunchKernelGGL((tile_kernel<160, 8, 8>), dim3(wg * G), dim3(256), 0, 0, out, Do, Ho, Wo, wg, nt, order); }));
        rep("tile 160x x 16y x 4z (whole rows)", timeit([&] { hipLaunchKernelGGL((tile_kernel<160, 16, 4>), dim3(wg * G), dim3(256), 0, 0, out, Do, Ho, Wo, wg, nt, order); }));
        rep("tile 32x x 32y x 4z", timeit([&] { hipLaunchKernelGGL((tile_kernel<32, 32, 4>), dim3(wg * G), dim3(256), 0, 0, out, Do, Ho, Wo, wg, nt, order); }));
      }
    }
  }
  return 0;
}
