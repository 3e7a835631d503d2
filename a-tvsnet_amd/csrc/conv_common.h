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

// Tap order of the 3x3x3 K steps over 8 input channels (four taps x 8 channels per K = 32 step; aanet_b.hip and the Cin = 8 form of
// conv_c16b.hip, which must accumulate in the same order to agree bit for bit): step S, lane group q -> tap kd * 9 + kh * 3 + kw, or 27
// (zero weights).  Steps 0..2 carry the (kd, kw) combinations {(0,0), (0,1), (0,2), (1,0)} at kh = S, steps 3..5 the combinations
// {(1,1), (1,2), (2,0), (2,1)} at kh = S - 3, step 6 the last combination (2,2) at kh = q (q < 3).  The four lane groups of steps 3 G ..
// 3 G + 2 then differ ONLY in kh: row t of step kh reads halo row t + kh of the same four columns, so aanet_b fetches each of the
// group's ten halo rows ONCE for its three steps (20 + 20 + 16 fragment reads per stage instead of 7 x 16: the LDS array was the
// resource its two roles fought over).
__host__ __device__ constexpr int atvs_tap8(int S, int q) {
  if (S < 6) {
    const int c = (S / 3) * 4 + q;
    return ((c / 3) * 3 + S % 3) * 3 + c % 3;
  }
  return q < 3 ? (2 * 3 + q) * 3 + 2 : 27;
}
