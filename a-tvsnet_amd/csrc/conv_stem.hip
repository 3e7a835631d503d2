// 3x3x3 SAME stride-1 convolution of a volume with ONE or TWO channels to eight channels (gfx950): the probability,
// visual-hull and geometric stems of the refinement network (global_refine_{prob,vishull,geo}_3dconv,
// /root/reference/cnn_wrapper/atvsnet.py:300-311, layer code cnn_wrapper/network.py:172-215).
//
// 2 * 27 * Cin * 8 FLOP per voxel against 4 * Cin B read + 32 B written: HBM-bound by a wide margin (1 channel:
// 432 FLOP per 36 B).  On the matrix cores 1-2 input channels fill 1/4-1/2 of one K group and the layer ran at 9 TF/s;
// here it is plain FMAs: a workgroup stages the (8+2) x (8+2) x (32+2) halo of its tile in LDS, a thread owns one
// (y, x) column of 8 voxels with all their accumulators in registers and walks the 27 taps in the OUTER loop, so that a
// tap's weights -- wave-uniform scalar registers -- are loaded once and serve the whole column.  Epilogue as in the MFMA kernels: depth-plane bias (the D-constant channels
// of the geometric stem), ReLU, channel-last stores into a slice of the concat buffer, per-workgroup batch-norm moments.
#include "conv_common.h"

namespace {

constexpr int ST_TZ = 4, ST_TY = 8, ST_TX = 32;
constexpr int ST_HZ = ST_TZ + 2, ST_HY = ST_TY + 2, ST_HX = ST_TX + 2;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ float rs_zeros[8];       // what a load outside the volume reads
constexpr int RS_PF = (ST_HZ * ST_HY * ST_HX + 255) / 256;     // halo voxels per thread
#ifndef RS_WAVES
#define RS_WAVES 2
#endif
constexpr int RS_GRID = RS_WAVES * 256;   // persistent: RS_WAVES workgroups per CU (registers)
constexpr int RS_STG = 64 * 9;      // refine_stems: float4s of one wavefront's staging rows (64 voxels, 9 float4 pitch)

template <int CIN>
__global__ __launch_bounds__(256) void conv_stem_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ pbias, float* __restrict__ y,
                                                        double* __restrict__ stats, int D, int H, int W, int ldy, int ycoff,
                                                        int relu, int tiles_y, int tiles_x, int tiles) {
  __shared__ float tile[ST_HZ * ST_HY * ST_HX * CIN];
  __shared__ double s_red[4][2][8];
  const int tid = threadIdx.x;
  const int grp = blockIdx.x / tiles, t = blockIdx.x - grp * tiles;
  const int x0 = (t % tiles_x) * ST_TX, y0 = ((t / tiles_x) % tiles_y) * ST_TY, z0 = (t / (tiles_x * tiles_y)) * ST_TZ;
  const size_t vol = (size_t)D * H * W;
  const float* xg = x + (size_t)grp * vol * CIN;
  for (int s = tid; s < ST_HZ * ST_HY * ST_HX * CIN; s += 256) {
    const int c = s % CIN, v = s / CIN;
    const int xx = v % ST_HX, yy = (v / ST_HX) % ST_HY, zz = v / (ST_HX * ST_HY);
    const int gz = z0 + zz - 1, gy = y0 + yy - 1, gx = x0 + xx - 1;
    float val = 0.f;
    if ((unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
      val = xg[(((size_t)gz * H + gy) * W + gx) * CIN + c];
    tile[s] = val;
  }
  __syncthreads();
  const int lx = tid & 31, ly = tid >> 5;
  const int xo = x0 + lx, yo = y0 + ly;
  const bool col_ok = xo < W && yo < H;
  // tap-outer, z-inner: the 8 * CIN weights of a tap are wave-uniform (scalar registers) and serve the thread's
  // whole column of ST_TZ voxels
  float acc[ST_TZ][8];
#pragma unroll
  for (int z = 0; z < ST_TZ; ++z)
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[z][k] = 0.f;
#pragma unroll 1
  for (int kd = 0; kd < 3; ++kd)
#pragma unroll 1
    for (int j = 0; j < 9; ++j) {
      const float* wk = w + ((kd * 9 + j) * CIN) * 8;           // uniform address: scalar loads
      float wr[CIN][8];
#pragma unroll
      for (int c = 0; c < CIN; ++c)
#pragma unroll
        for (int k = 0; k < 8; ++k) wr[c][k] = wk[c * 8 + k];
#pragma unroll
      for (int z = 0; z < ST_TZ; ++z) {
        const float* src = tile + (((z + kd) * ST_HY + ly + j / 3) * ST_HX + lx + j % 3) * CIN;
#pragma unroll
        for (int c = 0; c < CIN; ++c) {
          const float v = src[c];
#pragma unroll
          for (int k = 0; k < 8; ++k) acc[z][k] = fmaf(v, wr[c][k], acc[z][k]);
        }
      }
    }
  float ssum[8], ssq[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) ssum[k] = ssq[k] = 0.f;
  float* yg = y + (size_t)grp * vol * ldy;
  const float* pbg = pbias ? pbias + (size_t)grp * H * W * 24 : nullptr;
#pragma unroll
  for (int z = 0; z < ST_TZ; ++z) {
    const int zo = z0 + z;
    if (!col_ok || zo >= D) continue;
    float* a8 = acc[z];
    if (pbg) {
      const float* pb = pbg + ((size_t)yo * W + xo) * 24 + plane_variant(zo - 1, D) * 8;
      const float4 b0 = ld4(pb), b1 = ld4(pb + 4);
      a8[0] += b0.x; a8[1] += b0.y; a8[2] += b0.z; a8[3] += b0.w;
      a8[4] += b1.x; a8[5] += b1.y; a8[6] += b1.z; a8[7] += b1.w;
    }
    if (relu) {
#pragma unroll
      for (int k = 0; k < 8; ++k) a8[k] = fmaxf(a8[k], 0.f);
    }
    float* dst = yg + (((size_t)zo * H + yo) * W + xo) * ldy + ycoff;
    st4(dst, make_float4(a8[0], a8[1], a8[2], a8[3]));
    st4(dst + 4, make_float4(a8[4], a8[5], a8[6], a8[7]));
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      ssum[k] += a8[k];
      ssq[k] += a8[k] * a8[k];
    }
  }
  if (stats) {
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      double a = (double)ssum[k], b = (double)ssq[k];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        a += __shfl_xor(a, o);
        b += __shfl_xor(b, o);
      }
      if (lane == 0) {
        s_red[wave][0][k] = a;
        s_red[wave][1][k] = b;
      }
    }
    __syncthreads();
    if (tid < 32) {                      // row [2][16]: columns 0..7 = channels, 8..15 = 0 (the x-pair kernels' layout)
      const int which = tid >> 4, col = tid & 15;
      double v = 0.0;
      if (col < 8) v = (s_red[0][which][col] + s_red[1][which][col]) + (s_red[2][which][col] + s_red[3][which][col]);
      stats[((size_t)blockIdx.x * 2 + which) * 16 + col] = v;
    }
  }
}

// one halving step of the statistics butterfly: HALF values kept of 2 * HALF
template <int HALF>
__device__ __forceinline__ void rs_halve(double* v, int o, int lane, int& c0) {
  const bool up = lane & o;
#pragma unroll
  for (int i = 0; i < HALF; ++i) {
    const double send = up ? v[i] : v[i + HALF];
    const double keep = up ? v[i + HALF] : v[i];
    v[i] = keep + __shfl_xor(send, o);
  }
  c0 += up ? HALF : 0;
}

// ---- the four stems of the refinement network as ONE pass over the 32-channel concat buffer -------------------------
// CostVolRefineNet (cnn_wrapper/atvsnet.py:300-313) concatenates photo | geo | prob | vishull stems (8 channels each)
// into the 32-channel input of its U-Net.  Written stem by stem, every launch touches 32 of the 128 bytes of each row
// of that buffer (partial-line HBM writes: 0.7 ms per stem for 4 volumes, whatever the arithmetic).  Here the three
// FMA stems (geo: 2 channels + depth-plane bias, prob: 1, vishull: 1) are computed together and stored, with the RAW
// output of the photo stem (MFMA, dense 8-channel tensor) passed through, as whole 128-byte rows.
//   w: [27 taps][4 input channels: geo0, geo1, prob, hull][8] floats (device);
//   stats rows: [2][24] doubles (geo 0..7 | prob 8..15 | hull 16..23), one row per TILE (the workgroups are persistent).
// Where its time goes (4 volumes of 192 x 128 x 160, by leaving phases out): taps 0.31 ms (packed FMAs at the rate the
// SIMDs issue them), loads 0.28, stores 0.30, statistics 0.19 -> 0.06 with the halving butterfly; the phases add up
// (two wavefronts per SIMD: 217 registers) whatever the start of the second workgroup of a CU is delayed by.
// PLANAR: y is chunk-planar per sample, four planes of [D][H][W][8] `pstride` floats apart (the layout the x-pair kernel
// stages as dense 32-byte voxels: conv_xb.hip, x_planar).  The photo stem has written plane 0 itself, so nothing is passed
// through: the kernel reads 16 and writes 96 bytes per voxel instead of 48 and 128, and every store instruction of a
// wavefront covers ONE whole kilobyte of one plane (a line of 32 voxels).
template <bool PLANAR>
__global__ __launch_bounds__(256, RS_WAVES) void refine_stems_kernel(const float* __restrict__ photo, const float* __restrict__ geo,
                                                           const float* __restrict__ geo_pb, const float* __restrict__ prob,
                                                           const float* __restrict__ hull, const float* __restrict__ w,
                                                           float* __restrict__ y, double* __restrict__ stats, int D, int H,
                                                           int W, int tiles_y, int tiles_x, int tiles, int groups, long pstride) {
  // (geo0, geo1, prob, hull) per halo voxel; after the taps, each wavefront's staging rows for the stores
  __shared__ float4 tile[ST_HZ * ST_HY * ST_HX > 4 * RS_STG ? ST_HZ * ST_HY * ST_HX : 4 * RS_STG];
  __shared__ double s_red[4][2][24];
  const int tid = threadIdx.x;
  const size_t vol = (size_t)D * H * W;
  // PERSISTENT: workgroup b takes tiles b, b + gridDim.x, ...; the next tile's halo is requested (into registers) after the
  // taps and lands in LDS after the stores and the statistics, so that its latency hides behind them and the resident
  // workgroups of a CU drift apart (loads of one beside the FMAs of another) instead of marching phase by phase.
  float4 pf[RS_PF];
  auto request = [&](int id) {
    const int grp = id / tiles, t = id - grp * tiles;
    const int x0 = (t % tiles_x) * ST_TX, y0 = ((t / tiles_x) % tiles_y) * ST_TY, z0 = (t / (tiles_x * tiles_y)) * ST_TZ;
    const float* gg = geo + (size_t)grp * vol * 2;
    const float* pg = prob + (size_t)grp * vol;
    const float* hg = hull + (size_t)grp * vol;
#pragma unroll
    for (int i = 0; i < RS_PF; ++i) {
      const int s = tid + i * 256;
      const int xx = s % ST_HX, yy = (s / ST_HX) % ST_HY, zz = s / (ST_HX * ST_HY);
      const int gz = z0 + zz - 1, gy = y0 + yy - 1, gx = x0 + xx - 1;
      // no branch around a load or a store anywhere in the tile loop (a join makes every later wait a wait for all):
      // slots outside the volume read zeros from rs_zeros
      const bool ok = s < ST_HZ * ST_HY * ST_HX && (unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H &&
                      (unsigned)gx < (unsigned)W;
      const size_t v = ((size_t)gz * H + gy) * W + gx;
      const float2 g2 = *reinterpret_cast<const float2*>(ok ? gg + v * 2 : rs_zeros);
      const float4 val = make_float4(g2.x, g2.y, *(ok ? pg + v : rs_zeros), *(ok ? hg + v : rs_zeros));
      pf[i] = val;
    }
  };
  auto land = [&]() {
#pragma unroll
    for (int i = 0; i < RS_PF; ++i)
      if (tid + i * 256 < ST_HZ * ST_HY * ST_HX) tile[tid + i * 256] = pf[i];
  };
  const int total = tiles * groups;
  request(blockIdx.x);                                     // gridDim.x <= total
  for (int id = blockIdx.x; id < total; id += gridDim.x) {
  const int grp = id / tiles, t = id - grp * tiles;
  const int x0 = (t % tiles_x) * ST_TX, y0 = ((t / tiles_x) % tiles_y) * ST_TY, z0 = (t / (tiles_x * tiles_y)) * ST_TZ;
  __syncthreads();                                         // the previous tile's staging rows and s_red are done with
  land();
  __syncthreads();
  const int lx = tid & 31, ly = tid >> 5;
  const int xo = x0 + lx, yo = y0 + ly;
  const bool col_ok = xo < W && yo < H;
  // tap-outer, z-inner: the 32 weights of a tap are wave-uniform (scalar registers) and serve the thread's whole
  // column of ST_TZ voxels (one ds_read_b128 of (geo0, geo1, prob, hull) per 32 FMAs)
  // packed fp32 fused multiply-adds: two output channels per instruction (v_pk_fma_f32; each half is the same IEEE
  // fused operation as fmaf) -- the kernel is VALU-bound (864 FMAs per voxel against 200 bytes of traffic)
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 acc2[ST_TZ][12];
#pragma unroll
  for (int z = 0; z < ST_TZ; ++z)
#pragma unroll
    for (int k = 0; k < 12; ++k) acc2[z][k] = (f32x2){0.f, 0.f};
#pragma unroll 1
  for (int kd = 0; kd < 3; ++kd)
#pragma unroll 1
    for (int j = 0; j < 9; ++j) {
      const float* wk = w + (kd * 9 + j) * 32;                  // uniform address: scalar loads
      f32x2 wr[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) wr[k] = (f32x2){wk[2 * k], wk[2 * k + 1]};
#pragma unroll
      for (int z = 0; z < ST_TZ; ++z) {
        const float4 v = tile[((z + kd) * ST_HY + ly + j / 3) * ST_HX + lx + j % 3];
        const f32x2 vx = {v.x, v.x}, vy = {v.y, v.y}, vz = {v.z, v.z}, vw = {v.w, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          // geo: channel 0, then channel 1 (the kernel's ci order)
          acc2[z][k] = __builtin_elementwise_fma(vx, wr[k], acc2[z][k]);
          acc2[z][k] = __builtin_elementwise_fma(vy, wr[4 + k], acc2[z][k]);
          acc2[z][4 + k] = __builtin_elementwise_fma(vz, wr[8 + k], acc2[z][4 + k]);
          acc2[z][8 + k] = __builtin_elementwise_fma(vw, wr[12 + k], acc2[z][8 + k]);
        }
      }
    }
  float acc[ST_TZ][24];
#pragma unroll
  for (int z = 0; z < ST_TZ; ++z)
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      acc[z][2 * k] = acc2[z][k].x;
      acc[z][2 * k + 1] = acc2[z][k].y;
    }
  float ssum[24], ssq[24];
#pragma unroll
  for (int k = 0; k < 24; ++k) ssum[k] = ssq[k] = 0.f;
  float* yg = PLANAR ? y + (size_t)grp * 4 * pstride + pstride : y + (size_t)grp * vol * 32;
  // the tile's ST_TZ planes of y as a buffer (H * W * 512 bytes < 4 GB, checked by the caller); PLANAR: one per channel plane
  const unsigned ybytes = (unsigned)((size_t)H * W * (PLANAR ? 32 : 128) * ST_TZ);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(yg + (size_t)z0 * H * W * (PLANAR ? 8 : 32), 0, ybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrsrc1 = __builtin_amdgcn_make_buffer_rsrc(yg + (PLANAR ? pstride : 0) + (size_t)z0 * H * W * 8, 0, ybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yrsrc2 = __builtin_amdgcn_make_buffer_rsrc(yg + (PLANAR ? 2 * pstride : 0) + (size_t)z0 * H * W * 8, 0, ybytes, 0x00020000);
  const float* phg = PLANAR ? nullptr : photo + (size_t)grp * vol * 8;
  const float* pbg = geo_pb ? geo_pb + (size_t)grp * H * W * 24 : nullptr;
  // Stores go through LDS: a thread owns one voxel's 128-byte row, and written by its owner each store instruction would
  // touch 64 rows with 16 bytes each (eight partial writes per line at the L2).  Each wavefront stages its 64 rows (two
  // lines of 32 voxels; 144-byte pitch: conflict-free both ways) and writes them back as whole kilobytes per instruction.
  __syncthreads();                                         // every wavefront is done with the halo tile
  // The loads of plane z + 1 (depth-plane bias, photo row) are issued BEFORE the stores of plane z, and the next tile's halo
  // before all of them: memory operations retire in order, so that no wait of the plane loop covers a store.
  float4 ph[2][2], bb[2][2];
  auto fetch = [&](int z, float4* p, float4* q) {
    const int zo = z0 + z;
    const bool ok = col_ok && zo < D;
    const float* pb = ok && pbg ? pbg + ((size_t)yo * W + xo) * 24 + plane_variant(zo - 1, D) * 8 : rs_zeros;
    q[0] = ld4(pb);
    q[1] = ld4(pb + 4);
    if (!PLANAR) {
      const float* pp = ok ? phg + (((size_t)zo * H + yo) * W + xo) * 8 : rs_zeros;
      p[0] = ld4(pp);
      p[1] = ld4(pp + 4);
    }
  };
  request(id + (int)gridDim.x < total ? id + gridDim.x : id);      // the last tile asks for itself again (never landed)
  fetch(0, ph[0], bb[0]);
  const int lane = tid & 63, wave = tid >> 6;
  float4* stg = tile + wave * RS_STG;
#pragma unroll
  for (int z = 0; z < ST_TZ; ++z) {
    const int zo = z0 + z;
    if (z + 1 < ST_TZ) fetch(z + 1, ph[(z + 1) & 1], bb[(z + 1) & 1]);
    float* a24 = acc[z];
    const float4 p0 = ph[z & 1][0], p1 = ph[z & 1][1];
    {
      const float4 b0 = bb[z & 1][0], b1 = bb[z & 1][1];   // zeros without a depth-plane bias: x + 0.f == x for the statistics
      if (pbg) {                                           // and the stores alike (a -0.f would become +0.f: pbg decides)
        a24[0] += b0.x; a24[1] += b0.y; a24[2] += b0.z; a24[3] += b0.w;
        a24[4] += b1.x; a24[5] += b1.y; a24[6] += b1.z; a24[7] += b1.w;
      }
      const bool ok = col_ok && zo < D;
#pragma unroll
      for (int k = 0; k < 24; ++k) {
        const float a = ok ? a24[k] : 0.f;                 // x + 0.f leaves the sums as they were
        ssum[k] += a;
        ssq[k] += a * a;
      }
    }
    if (!PLANAR) {
      stg[lane * 9] = p0;
      stg[lane * 9 + 1] = p1;
    }
#pragma unroll
    for (int k = 0; k < 24; k += 4) stg[lane * 9 + 2 + k / 4] = make_float4(a24[k], a24[k + 1], a24[k + 2], a24[k + 3]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (PLANAR) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {                        // instruction i: stem i / 2 (its plane), line i % 2: 32 voxels x 32 bytes
        const int c = i >> 1, r = i & 1, xi = lane >> 1, hf = lane & 1;
        const float4 v = stg[(r * 32 + xi) * 9 + 2 + c * 2 + hf];
        const int yy = y0 + 2 * wave + r, xx = x0 + xi;
        const bool ok = yy < H && xx < W && zo < D;
        const u32x4 bits = {__builtin_bit_cast(unsigned, v.x), __builtin_bit_cast(unsigned, v.y),
                            __builtin_bit_cast(unsigned, v.z), __builtin_bit_cast(unsigned, v.w)};
        const unsigned off = ok ? (unsigned)(((z * H + yy) * W + xx) * 8 + hf * 4) * 4u : ybytes;
        __builtin_amdgcn_raw_buffer_store_b128(bits, c == 0 ? yrsrc : (c == 1 ? yrsrc1 : yrsrc2), off, 0, ATVS_BUF_NT);
      }
    } else
#pragma unroll
    for (int i = 0; i < 8; ++i) {                          // instruction i: line i / 4, voxels 8 (i % 4) .. + 7, 16 bytes per lane
      const int r = i >> 2, xi = (i & 3) * 8 + (lane >> 3), c = lane & 7;
      const float4 v = stg[(r * 32 + xi) * 9 + c];
      const int yy = y0 + 2 * wave + r, xx = x0 + xi;
      const bool ok = yy < H && xx < W && zo < D;
      const u32x4 bits = {__builtin_bit_cast(unsigned, v.x), __builtin_bit_cast(unsigned, v.y),
                          __builtin_bit_cast(unsigned, v.z), __builtin_bit_cast(unsigned, v.w)};
      // a buffer store: lanes outside the volume get an offset past the descriptor's range and are dropped
      __builtin_amdgcn_raw_buffer_store_b128(bits, yrsrc, ok ? (unsigned)(((z * H + yy) * W + xx) * 32 + c * 4) * 4u : ybytes, 0, ATVS_BUF_NT);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (stats) {
    // The wavefront's 48 sums as a butterfly that HALVES what a lane carries at each of its first four steps (lane bit
    // set: keep the upper half and send the lower, else the reverse), then two plain steps on the three values left:
    // the same additions in the same order as 48 full butterflies (a + b == b + a), with 51 exchanges instead of 288.
    double v[48];
#pragma unroll
    for (int k = 0; k < 24; ++k) {
      v[k] = (double)ssum[k];
      v[24 + k] = (double)ssq[k];
    }
    int c0 = 0;                                            // first of the channels this lane ends up holding
    rs_halve<24>(v, 1, lane, c0);
    rs_halve<12>(v, 2, lane, c0);
    rs_halve<6>(v, 4, lane, c0);
    rs_halve<3>(v, 8, lane, c0);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      v[i] += __shfl_xor(v[i], 16);
      v[i] += __shfl_xor(v[i], 32);
    }
    if (lane < 16) {
      double* row = &s_red[wave][0][0];                    // [2][24] = 48 in the order of v
#pragma unroll
      for (int i = 0; i < 3; ++i) row[c0 + i] = v[i];
    }
    __syncthreads();
    if (tid < 48) {
      const int which = tid / 24, col = tid % 24;
      stats[((size_t)id * 2 + which) * 24 + col] =
          (s_red[0][which][col] + s_red[1][which][col]) + (s_red[2][which][col] + s_red[3][which][col]);
    }
  }
  }
}

}  // namespace

// workgroups per sample = rows per sample of stats_partial ([2][16] doubles each)
extern "C" long atvs_conv_stem_rows(int D, int H, int W) {
  return (long)((D + ST_TZ - 1) / ST_TZ) * ((H + ST_TY - 1) / ST_TY) * ((W + ST_TX - 1) / ST_TX);
}

// y (G,D,H,W,ldy)[..., y_coff + co] = conv3x3x3(x (G,D,H,W,Cin), w) (+ plane_bias (G,H,W,24), ReLU), co < 8, Cin in {1, 2};
// w: the TF kernel [3,3,3,Cin,8] on the device.  stats_partial: groups * atvs_conv_stem_rows rows or NULL.
extern "C" int atvs_conv_stem_f32(const float* x, const float* w, const float* plane_bias, float* y, double* stats_partial,
                                  int groups, int D, int H, int W, int Cin, int ldy, int y_coff, int relu,
                                  atvs_stream_t stream) {
  if (!x || !w || !y) return ATVS_ERR_NULL;
  if (groups <= 0 || D <= 0 || H <= 0 || W <= 0 || (Cin != 1 && Cin != 2)) return ATVS_ERR_SHAPE;
  if (y_coff < 0 || y_coff + 8 > ldy || (ldy % 4) || (y_coff % 4)) return ATVS_ERR_SHAPE;
  if (plane_bias && D < 2) return ATVS_ERR_ARG;
  const int ty = (H + ST_TY - 1) / ST_TY, tx = (W + ST_TX - 1) / ST_TX;
  const long tiles = atvs_conv_stem_rows(D, H, W);
  if (tiles * groups > 0x7fffffffL) return ATVS_ERR_SHAPE;
  dim3 grid((unsigned)(tiles * groups)), block(256);
  hipStream_t s = as_stream(stream);
  if (Cin == 1)
    hipLaunchKernelGGL((conv_stem_kernel<1>), grid, block, 0, s, x, w, plane_bias, y, stats_partial, D, H, W, ldy, y_coff, relu,
                       ty, tx, (int)tiles);
  else
    hipLaunchKernelGGL((conv_stem_kernel<2>), grid, block, 0, s, x, w, plane_bias, y, stats_partial, D, H, W, ldy, y_coff, relu,
                       ty, tx, (int)tiles);
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}

// The geo | prob | vishull stems of CostVolRefineNet in one pass, written with the raw photo-stem output as whole rows of
// the 32-channel concat buffer: y (G,D,H,W,32) = [photo_raw (G,D,H,W,8) | conv(geo (G,D,H,W,2)) + geo_plane_bias
// (G,H,W,24) | conv(prob (G,D,H,W,1)) | conv(hull (G,D,H,W,1))] (no activation: the batch norm + ReLU of the concat
// follows).  w: [27][geo0, geo1, prob, hull][8] floats on the device (atvs_refine_stems_pack arranges the three TF
// kernels).  stats_partial: groups * atvs_conv_stem_rows rows of [2][24] doubles (the 24 computed channels) or NULL.
// y_planar != 0: y is chunk-planar, (G, 4, y_planar) with planes of [D][H][W][8]; planes 1..3 are written (geo | prob | hull),
// plane 0 belongs to the photo stem (atvs_conv_xb_f32 with y_group_stride) and photo_raw is not read (may be NULL).
extern "C" int atvs_refine_stems_f32(const float* photo_raw, const float* geo, const float* geo_plane_bias, const float* prob,
                                     const float* hull, const float* w, float* y, double* stats_partial, int groups, int D,
                                     int H, int W, long y_planar, atvs_stream_t stream) {
  if ((!photo_raw && !y_planar) || !geo || !prob || !hull || !w || !y) return ATVS_ERR_NULL;
  if (groups <= 0 || D <= 0 || H <= 0 || W <= 0) return ATVS_ERR_SHAPE;
  if (y_planar && y_planar < (long)D * H * W * 8) return ATVS_ERR_ARG;
  if (geo_plane_bias && D < 2) return ATVS_ERR_ARG;
  if ((double)H * W * 128.0 * ST_TZ >= 4294967296.0) return ATVS_ERR_SHAPE;       // a tile's planes are one buffer descriptor
  const int ty = (H + ST_TY - 1) / ST_TY, tx = (W + ST_TX - 1) / ST_TX;
  const long tiles = atvs_conv_stem_rows(D, H, W);
  if (tiles * groups > 0x7fffffffL) return ATVS_ERR_SHAPE;
  const long total = tiles * groups;
  const dim3 grid((unsigned)(total < RS_GRID ? total : RS_GRID));
  if (y_planar)
    hipLaunchKernelGGL(refine_stems_kernel<true>, grid, dim3(256), 0, as_stream(stream), photo_raw, geo, geo_plane_bias, prob,
                       hull, w, y, stats_partial, D, H, W, ty, tx, (int)tiles, groups, y_planar);
  else
    hipLaunchKernelGGL(refine_stems_kernel<false>, grid, dim3(256), 0, as_stream(stream), photo_raw, geo, geo_plane_bias, prob,
                       hull, w, y, stats_partial, D, H, W, ty, tx, (int)tiles, groups, 0L);
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}

// HOST function: w_geo [3,3,3,2,8], w_prob [3,3,3,1,8], w_hull [3,3,3,1,8] (TF layouts) -> packed [27][4][8].
extern "C" int atvs_refine_stems_pack(const float* w_geo, const float* w_prob, const float* w_hull, float* packed) {
  if (!w_geo || !w_prob || !w_hull || !packed) return ATVS_ERR_NULL;
  for (int t = 0; t < 27; ++t)
    for (int k = 0; k < 8; ++k) {
      packed[(t * 4 + 0) * 8 + k] = w_geo[(t * 2 + 0) * 8 + k];
      packed[(t * 4 + 1) * 8 + k] = w_geo[(t * 2 + 1) * 8 + k];
      packed[(t * 4 + 2) * 8 + k] = w_prob[t * 8 + k];
      packed[(t * 4 + 3) * 8 + k] = w_hull[t * 8 + k];
    }
  return ATVS_OK;
}
