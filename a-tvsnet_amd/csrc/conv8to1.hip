// 3x3x3, 8 -> 1 channel, stride 1, SAME, no bias / activation: the probability heads conv_b2_6_2,
// attention_prob_vol[_refine], global_refined_cost_vol (/root/reference/cnn_wrapper/atvsnet.py:192,213,220,226,242,336).
// One output channel cannot feed a 16-row MFMA tile (1/16 useful; as a banded matrix along x 1/6), so: packed fp32 FMAs from
// an LDS halo tile, weights in scalar registers.  432 FLOP per 36 bytes of traffic: the FMA pipe and the LDS are as close
// to their limits as HBM is, hence
//   * four outputs per thread along z: a (kh, kw) column of six staged voxels serves 4 x 3 tap uses (27 LDS reads per output
//     instead of 54);
//   * v_pk_fma_f32: two partial sums per output (even / odd channel pairs), added at the end;
//   * halo rows 592 bytes apart: the 16 lanes of a ds_read_b128 group (8 voxels of one row, 8 of the next, 32 bytes apart)
//     then fall on 64 different banks (at 576 bytes the rows met two by two);
//   * persistent workgroups, the next tile's halo prefetched into registers: 8 volumes 192x128x160 in 0.43 ms (0.52 with one
//     output per thread and a tile per workgroup); staging alone takes 0.27 ms, the arithmetic alone 0.27 ms.
#include "conv_common.h"

namespace {

constexpr int C81_TZ = 4, C81_TY = 16, C81_TX = 16;
constexpr int C81_HZ = C81_TZ + 2, C81_HY = C81_TY + 2, C81_HX = C81_TX + 2;
constexpr int C81_ROWB = C81_HX * 32 + 16;                     // 592
constexpr int C81_LDS = C81_HZ * C81_HY * C81_ROWB;            // 63,936 bytes: two workgroups per CU
constexpr int C81_SLOTS = C81_HZ * C81_HY * C81_HX * 2;        // float4 halves of the halo voxels
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int C81_MAXS = (C81_SLOTS + 255) / 256;           // 16 per thread
__device__ float c81_zeros[4];

// Persistent: a workgroup walks tiles blockIdx.x, + gridDim.x, ... of all samples; the next tile's halo waits in registers
// while this one is computed (two workgroups per CU: one computes while the other stages).
__global__ __launch_bounds__(256, 2) void conv3d_8to1_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             float* __restrict__ y, int D, int H, int W, int tiles_y,
                                                             int tiles_x, int tiles_per_sample, long ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  int goff[C81_MAXS], laddr[C81_MAXS];
  unsigned pg[C81_MAXS];
#pragma unroll
  for (int i = 0; i < C81_MAXS; ++i) {
    int s = tid + i * 256;
    const bool live = s < C81_SLOTS;
    s = min(s, C81_SLOTS - 1);
    const int c4 = s & 1, v = s >> 1;
    const int xx = v % C81_HX, v2 = v / C81_HX;
    const int yy = v2 % C81_HY, zz = v2 / C81_HY;
    goff[i] = ((zz * H + yy) * W + xx) * 8 + c4 * 4;
    laddr[i] = (zz * C81_HY + yy) * C81_ROWB + xx * 32 + c4 * 16;
    pg[i] = 0x808080u | (unsigned)(live ? zz : 0x7f) | ((unsigned)yy << 8) | ((unsigned)xx << 16);
  }
  const size_t vol = (size_t)D * H * W;
  struct Tile {
    const float* xb;
    int z0, y0, x0, smp;
    unsigned lo, hi1;
  };
  auto tile_of = [&](long t) __attribute__((always_inline)) {
    Tile T;
    T.smp = (int)(t / tiles_per_sample);
    const int tl = (int)(t - (long)T.smp * tiles_per_sample);
    const int bx = tl % tiles_x, rest = tl / tiles_x;
    T.x0 = bx * C81_TX; T.y0 = (rest % tiles_y) * C81_TY; T.z0 = (rest / tiles_y) * C81_TZ;
    const int gz0 = T.z0 - 1, gy0 = T.y0 - 1, gx0 = T.x0 - 1;
    T.xb = x + (size_t)T.smp * vol * 8 + (((long)gz0 * H + gy0) * W + gx0) * 8;
    T.lo = (unsigned)(gz0 < 0) | ((unsigned)(gy0 < 0) << 8) | ((unsigned)(gx0 < 0) << 16);
    T.hi1 = (unsigned)(min(D - 1 - gz0, 0x7e) + 1) | ((unsigned)(min(H - 1 - gy0, 0x7e) + 1) << 8) |
            ((unsigned)(min(W - 1 - gx0, 0x7e) + 1) << 16);
    return T;
  };
  float4 pf[C81_MAXS];
  auto prefetch = [&](const Tile& T) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < C81_MAXS; ++i) {
      const unsigned t1 = pg[i] - T.lo;
      const unsigned t2 = T.hi1 + ~pg[i];
      const bool ok = ((t1 & t2) & 0x808080u) == 0x808080u;
      pf[i] = ld4(ok ? (T.xb + goff[i]) : c81_zeros);      // halo slots outside the volume read 16 bytes of zeros
    }
  };
  const int lx = tid % C81_TX, ly = tid / C81_TX;
  const unsigned char* base = smem + ly * C81_ROWB + lx * 32;

  long t = blockIdx.x;
  if (t >= ntiles) return;
  Tile cur = tile_of(t);
  prefetch(cur);
  for (; t < ntiles; t += gridDim.x) {
    __syncthreads();                         // every wavefront is done reading the previous tile
#pragma unroll
    for (int i = 0; i < C81_MAXS; ++i)
      if (i < C81_MAXS - 1 || tid + i * 256 < C81_SLOTS) *reinterpret_cast<float4*>(smem + laddr[i]) = pf[i];
    __syncthreads();
    const Tile me = cur;
    if (t + gridDim.x < ntiles) {
      cur = tile_of(t + gridDim.x);
      prefetch(cur);
    }
    f32x2 acc[C81_TZ];
#pragma unroll
    for (int o = 0; o < C81_TZ; ++o) acc[o] = (f32x2){0.f, 0.f};
#pragma unroll 1
    for (int kh = 0; kh < 3; ++kh) {         // rolled: unrolled, all 108 reads were hoisted (188 spilled registers)
      // the 72 weights of this kernel row into scalar registers FIRST: scalar loads and LDS reads share lgkmcnt and
      // return out of order, so a scalar load among the fragment reads makes every wait a wait for all of them
      float wr[3][3][8];
#pragma unroll
      for (int kd = 0; kd < 3; ++kd)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int c = 0; c < 8; ++c) wr[kd][kw][c] = w[((kd * 3 + kh) * 3 + kw) * 8 + c];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        // the column's six voxels first (twelve reads in flight), then its FMAs
        float4 va[C81_HZ], vb[C81_HZ];
#pragma unroll
        for (int zz = 0; zz < C81_HZ; ++zz) {
          const unsigned char* tp = base + (zz * C81_HY + kh) * C81_ROWB + kw * 32;
          va[zz] = *reinterpret_cast<const float4*>(tp);
          vb[zz] = *reinterpret_cast<const float4*>(tp + 16);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int zz = 0; zz < C81_HZ; ++zz) {
          const f32x2 v0 = {va[zz].x, va[zz].y}, v1 = {va[zz].z, va[zz].w}, v2 = {vb[zz].x, vb[zz].y}, v3 = {vb[zz].z, vb[zz].w};
#pragma unroll
          for (int kd = 0; kd < 3; ++kd) {
            const int o = zz - kd;                                 // output plane z0 + o reads halo plane zz at tap kd
            if (o < 0 || o >= C81_TZ) continue;
            const float* wk = wr[kd][kw];
            f32x2 s = acc[o];
            s = __builtin_elementwise_fma(v0, (f32x2){wk[0], wk[1]}, s);
            s = __builtin_elementwise_fma(v1, (f32x2){wk[2], wk[3]}, s);
            s = __builtin_elementwise_fma(v2, (f32x2){wk[4], wk[5]}, s);
            s = __builtin_elementwise_fma(v3, (f32x2){wk[6], wk[7]}, s);
            acc[o] = s;
          }
        }
      }
    }
    const int yo = me.y0 + ly, xo = me.x0 + lx;
    if (yo < H && xo < W) {
      float* yg = y + (size_t)me.smp * vol;
#pragma unroll
      for (int o = 0; o < C81_TZ; ++o)
        if (me.z0 + o < D) yg[((size_t)(me.z0 + o) * H + yo) * W + xo] = acc[o].x + acc[o].y;
    }
  }
}

}  // namespace

// x (groups,D,H,W,8), w: the TF kernel [3,3,3,8,1] as 216 floats (device), y (groups,D,H,W).
extern "C" int atvs_conv3d_8to1(const float* x, const float* w, float* y, int groups, int D, int H, int W,
                                atvs_stream_t stream) {
  if (!x || !w || !y) return ATVS_ERR_NULL;
  if (groups <= 0 || D <= 0 || H <= 0 || W <= 0) return ATVS_ERR_SHAPE;
  const int tz = (D + C81_TZ - 1) / C81_TZ, ty = (H + C81_TY - 1) / C81_TY, tx = (W + C81_TX - 1) / C81_TX;
  const long per = (long)tz * ty * tx, ntiles = per * groups;
  if (per > 0x7fffffffL || (double)D * H * W * 8.0 >= 2147483648.0) return ATVS_ERR_SHAPE;      // 31-bit halo-relative offsets
  const long grid = ntiles < 512 ? ntiles : 512;                 // two workgroups per CU
  hipLaunchKernelGGL(conv3d_8to1_kernel, dim3((unsigned)grid), dim3(256), C81_LDS, as_stream(stream), x, w, y, D, H, W, ty, tx,
                     (int)per, ntiles);
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}
