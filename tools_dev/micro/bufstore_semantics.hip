// Which buffer stores does gfx950 drop?  raw buffer (stride 0), num_records = nbytes, voffset / soffset combinations.
// build: hipcc --offload-arch=gfx950 -O2 bufstore_semantics.hip -o bufstore_semantics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* base, unsigned nbytes, unsigned soff, unsigned bad_voff, int mode) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, nbytes, 0x00020000);
  const unsigned lane = threadIdx.x;
  unsigned voff = lane * 16;
  if (mode == 1 && lane >= 32) voff = bad_voff;      // lanes 32..63 invalid
  u32x4 v = {lane + 1, lane + 1, lane + 1, lane + 1};
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 0);
}
int main() {
  const unsigned nbytes = 4096, guard = 8192;
  float* d;
  hipMalloc(&d, nbytes + guard);
  std::vector<unsigned> h((nbytes + guard) / 4);
  struct { unsigned soff, bad; int mode; const char* what; } cases[] = {
      {0, nbytes, 1, "soffset 0, invalid voffset = num_records"},
      {1024, nbytes, 1, "soffset 1024, invalid voffset = num_records"},
      {1024, nbytes - 1024, 1, "soffset 1024, invalid voffset = num_records - soffset"},
      {3584, 0, 0, "soffset 3584, all lanes valid voffset lane*16 (lanes >= 32 run past num_records)"},
      {1024, 0xffffffffu, 1, "soffset 1024, invalid voffset = 0xffffffff"},
      {1024, 0x80000000u, 1, "soffset 1024, invalid voffset = 0x80000000"},
  };
  for (auto& c : cases) {
    hipMemset(d, 0, nbytes + guard);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, nbytes, c.soff, c.bad, c.mode);
    hipMemcpy(h.data(), d, nbytes + guard, hipMemcpyDeviceToHost);
    int in_lo = 0, in_hi = 0, beyond = 0; long first_beyond = -1;
    for (size_t i = 0; i < h.size(); ++i) {
      if (!h[i]) continue;
      if (i * 4 >= nbytes) { ++beyond; if (first_beyond < 0) first_beyond = (long)i * 4; }
      else if (h[i] <= 32) ++in_lo; else ++in_hi;
    }
    printf("%-90s : words written by lanes<32 inside %d, by lanes>=32 inside %d, beyond num_records %d (first at byte %ld)\n",
           c.what, in_lo, in_hi, beyond, first_beyond);
  }
  return 0;
}
