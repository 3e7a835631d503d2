/* A plain-C caller of the C-ABI (no Python, no C++): loads the library, checks the ABI version against the header it was
 * compiled with, runs a host-side size query and the argument check of a launch entry point (which returns before any HIP
 * call).  Built and run by tests/test_c_abi.py:  cc -std=c99 -Iinclude tests/c/abi_smoke.c -ldl && ./a.out <library> */
#include <dlfcn.h>
#include <stdio.h>
#include "atvsnet_hip.h"

typedef int (*version_fn)(void);
typedef int (*pack_size_fn)(int, int, long*);
typedef int (*stems_fn)(const float*, const float*, const float*, const float*, const float*, const float*, float*, double*, int,
                        int, int, int, atvs_stream_t);

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    fprintf(stderr, "dlopen: %s\n", dlerror());
    return 3;
  }
  version_fn version = (version_fn)dlsym(h, "atvs_abi_version");
  pack_size_fn pack_size = (pack_size_fn)dlsym(h, "atvs_conv3d_b_pack_size");
  stems_fn stems = (stems_fn)dlsym(h, "atvs_refine_stems_f32");
  if (!version || !pack_size || !stems) return 4;
  if (version() != ATVS_ABI_VERSION) {
    fprintf(stderr, "library ABI %d, header ABI %d\n", version(), ATVS_ABI_VERSION);
    return 5;
  }
  long bytes = 0;
  if (pack_size(32, 32, &bytes) != ATVS_OK || bytes != 2L * 14 * 2 * 2 * 1024 + 16) return 6;   /* 2 chunks x 14 steps x 2 tiles x 2 fp16 pieces */
  if (pack_size(8, 32, &bytes) != ATVS_ERR_SHAPE) return 7;
  if (stems(0, 0, 0, 0, 0, 0, 0, 0, 1, 8, 8, 32, 0) != ATVS_ERR_NULL) return 8;
  printf("abi %d ok\n", version());
  dlclose(h);
  return 0;
}
