// Shared by conv.hip (gather form) and conv_tiled.hip (LDS-tiled form).
#pragma once
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Contribution of input channels that are CONSTANT along depth (tf.tile'd reference features,
// model.py:186,311,316,329-330): it equals a 2-D convolution with the kd-summed kernel, computed once
// per (y, x) for the three sets of kd taps that fall inside the volume, and is added like a bias.
//   variant 0: kd = 0 falls before the first plane;  1: all three planes exist;  2: kd = 2 falls past the end.
__device__ __forceinline__ int plane_variant(int z_in_first, int Di) {
  return (z_in_first < 0) ? 0 : ((z_in_first + 2 >= Di) ? 2 : 1);
}

static inline int pow2_tiles(int cout) {
  int nt = (cout + 15) / 16;
  int p = 1;
  while (p < nt) p <<= 1;
  return p;
}

__device__ __forceinline__ float f4get(const float4& a, int s) {
  return s == 0 ? a.x : (s == 1 ? a.y : (s == 2 ? a.z : a.w));
}
