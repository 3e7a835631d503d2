// LDS-DMA probe: buffer_load_dwordx4 ... lds with per-lane offsets, out-of-range lanes (zero fill?), M0 base handling.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* src, float* out, int nbytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 4096 / 4; i += 64) reinterpret_cast<float*>(smem)[i] = -1.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nbytes, 0x00020000);
  // lane l fetches record (63 - l) (a gather on the global side), lanes 10..12 out of range
  unsigned voff = (unsigned)(63 - lane) * 16u;
  if (lane >= 10 && lane <= 12) voff = 0xffffffffu;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (void __attribute__((address_space(3)))*)(smem + 1024), 16, (int)voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 4096 / 4; i += 64) out[i] = reinterpret_cast<float*>(smem)[i];
}
int main() {
  std::vector<float> h(256 * 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
  float *d, *o;
  hipMalloc(&d, h.size() * 4); hipMalloc(&o, 4096);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 4096, 0, d, o, 64 * 16);
  std::vector<float> r(1024);
  hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
  printf("before dest: %g %g; dest lanes 0,1,9,10,12,13,63 first float: %g %g %g %g %g %g %g; after dest: %g\n", r[0], r[255], r[256], r[260], r[256 + 36],
         r[256 + 40], r[256 + 48], r[256 + 52], r[256 + 252], r[512]);
  printf("expected: -1 -1; 252 248 216 0 0 200 0; -1\n");
  return 0;
}
