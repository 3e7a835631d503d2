// 3x3x3 stride-2 SAME transposed convolution to EIGHT output channels, all 8 output parity classes from one staged
// input tile, one wavefront per SIMD (gfx950).
//
// These are the last decoder layers of the stacked U-Nets / the refinement net (conv_b*_6_0, global_refine_3dconv6_0:
// cnn_wrapper/atvsnet.py StackedUNet / CostVolRefineNet; layer code /root/reference/cnn_wrapper/network.py:510-550,
// tf.layers.conv3d_transpose(3, strides 2, 'same')): half-resolution input, FULL-resolution 8-channel output -- per
// launch they write as many bytes as the widest convolutions and are, after those, the largest block of MFMA work.
//
// Form.  Per axis an even output 2j is  w[0] x[j] + w[2] x[j-1]  and an odd output 2j+1 is  w[1] x[j]:  output voxel
// (2z+pz, 2y+py, 2x+px) (parity class (pz,py,px)) reads the input voxels (z+oz, y+oy, x+ox), o in {0,-1}, a class with
// an odd parity only offset 0 on that axis.  As a GEMM per input voxel: M = 8 classes x 8 channels = 64 rows (4 MFMA
// tiles of 16 rows: tile m = (pz,py), rows = (px, channel)), K = 8 offsets x Cin, of whose 32 (tile, offset) blocks 18
// are non-zero; only those are issued (the x pair shares a tile, so 3/4 of the issued work is useful).
//
// Kernel.  Built like conv_xp.hip (the measurements behind that file apply): ONE workgroup of 4 wavefronts per CU with
// the whole register file, a fully unrolled K loop whose LDS reads are base register + immediate, the next tile's halo
// requested between the MFMAs.  Tile = 4(z) x 8(y) x 16(x) INPUT voxels (8 x 16 x 32 output voxels); wavefront w owns
// input plane z0 + w with 8 rows x 4 tiles = 32 accumulator tiles.  The halo is one-sided (offsets 0 / -1 only):
// 5 x 9 x 17 voxels.  The packed weights of every chunk stay in LDS for the whole launch.
#include <type_traits>

#include "conv_common.h"

namespace {

constexpr int UP_TZ = 4, UP_TY = 8, UP_TX = 16;
constexpr int UP_HZ = UP_TZ + 1, UP_HY = UP_TY + 1, UP_HX = UP_TX + 1;
constexpr int UP_VB = 64;                              // bytes per voxel of a 16-channel chunk in LDS
constexpr int UP_ROWB = UP_HX * UP_VB;                 // bytes per image row
constexpr int UP_IMG = UP_HZ * UP_HY * UP_ROWB;        // 48,960
constexpr int UP_SLOTS = UP_HZ * UP_HY * UP_HX * 4;    // 16-byte halo slots
constexpr int UP_MAXS = (UP_SLOTS + 255) / 256;        // 12 per thread
constexpr int UP_PAIRS = 18;                           // non-zero (offset, tile) blocks per chunk
constexpr int UP_WCH = UP_PAIRS * 1024;                // bytes of packed weights per chunk

struct UpArgs {
  const float* x;
  const float* wp;       // packed weights (atvs_deconv_up_pack) + 4 trailing zeros
  const float* zeros;    // those 16 zero bytes: source of the zero padding
  float* y;
  double* stats;
  int Di, Hi, Wi, Cin;
  int ldy, ycoff;
  int nchunk;
  int tiles_y, tiles_x, ntiles;
  int relu;
  int wg;                // workgroups per sample (gridDim.x = groups * wg)
  long gx, gy;           // elements per sample of x / y
};

// tiles of offset (oz, oy): those whose class can read it (an odd parity reads offset 0 only)
__host__ __device__ constexpr bool up_active(int m, int oz, int oy) {
  return ((m >> 1) == 0 || oz == 0) && ((m & 1) == 0 || oy == 0);
}
// index of (offset o = oz*4 + oy*2 + ox [0 = offset 0, 1 = offset -1], tile m) among the 18 non-zero blocks, K-loop order
__host__ __device__ constexpr int up_pair_index(int o, int m) {
  int n = 0;
  for (int oo = 0; oo < 8; ++oo)
    for (int mm = 0; mm < 4; ++mm) {
      if (!up_active(mm, oo >> 2, (oo >> 1) & 1)) continue;
      if (oo == o && mm == m) return n;
      ++n;
    }
  return -1;
}
static_assert(up_pair_index(7, 0) == UP_PAIRS - 1, "18 non-zero (offset, tile) blocks");

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// SINGLE: Cin == 16 (one chunk per tile, the network's case): the tiles' accumulators are stored as soon as their last
// MFMA is issued, interleaved with the MFMAs of the remaining offsets (tile (1,1) is complete after 2 of the 8 offsets,
// (1,0) after 4, (0,1) after 6), so only a quarter of the epilogue is left behind the K loop.  Those stores are
// branch-free -- a branch would split the unrolled loop's scheduling regions: a buffer store whose offset is out of
// range for rows outside the volume (the hardware drops it).
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <bool SINGLE, bool RELU>
__global__ __launch_bounds__(256, 1) void deconv_up_kernel(UpArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;

  // packed weights of every chunk -> LDS, once
  {
    const float4* src = reinterpret_cast<const float4*>(p.wp);
    float4* dst = reinterpret_cast<float4*>(smem + UP_IMG);
    for (int i = tid; i < p.nchunk * (UP_WCH / 16); i += 256) dst[i] = src[i];
  }

  // LDS read base: this lane's fragment at halo voxel (wave, 0, r) = offset (-1,-1,-1) of row 0 of the wavefront's plane;
  // every (offset, row) is a non-negative immediate from here
  const int fbase = ((wave * UP_HY) * UP_HX + r) * UP_VB + q * 16;
  const int wbase = UP_IMG + lane * 16;

  // per-slot constants of this thread: global element offset from the halo origin, LDS byte address and the packed halo
  // coordinate (zz | yy<<8 | xx<<16, each byte with its top bit set) for the bounds test
  int goff[UP_MAXS], laddr[UP_MAXS];
  unsigned pg[UP_MAXS];
#pragma unroll
  for (int i = 0; i < UP_MAXS; ++i) {
    int s = tid + i * 256;
    const bool live = s < UP_SLOTS;
    s = min(s, UP_SLOTS - 1);
    const int c4 = s & 3, v = s >> 2;
    const int xx = v % UP_HX, v2 = v / UP_HX;
    const int yy = v2 % UP_HY, zz = v2 / UP_HY;
    goff[i] = ((zz * p.Hi + yy) * p.Wi + xx) * p.Cin + c4 * 4;
    laddr[i] = s * 16;
    pg[i] = 0x808080u | (unsigned)(live ? zz : 0x7f) | ((unsigned)yy << 8) | ((unsigned)xx << 16);
  }

  // persistent tile list, dealt so that the workgroups of one XCD (blockIdx % 8) sweep one contiguous eighth of the
  // tile range (halo re-use in that XCD's L2)
  const int G = p.wg;
  const int grp = blockIdx.x / p.wg, lbk = blockIdx.x - grp * p.wg;
  const int xcd = lbk & 7, tslot = lbk >> 3;
  const float* __restrict__ xg = p.x + (size_t)grp * p.gx;
  float* __restrict__ yg = p.y + (size_t)grp * p.gy;
  const int per_xcd = (p.ntiles + 7) >> 3;
  const int slots_per_xcd = G >> 3;
  int my_tiles = 0;
  {
    int last = min(per_xcd, p.ntiles - xcd * per_xcd);
    if (tslot < last) my_tiles = (last - tslot + slots_per_xcd - 1) / slots_per_xcd;
  }
  const int nstage = my_tiles * p.nchunk;

  auto tile_origin = [&](int k, int* z0, int* y0, int* x0) __attribute__((always_inline)) {
    int tl = xcd * per_xcd + tslot + k * slots_per_xcd;
    int bx = tl % p.tiles_x;
    int rest = tl / p.tiles_x;
    *x0 = bx * UP_TX;
    *y0 = (rest % p.tiles_y) * UP_TY;
    *z0 = (rest / p.tiles_y) * UP_TZ;
  };

  struct PfTile {
    const float* xb;      // sample + first channel of the chunk
    int org;              // element offset of the halo origin (may be negative: the first layer is outside)
    unsigned lo, hi1;     // packed bounds: valid iff lo_f <= f <= hi_f in every field (hi1 = hi + 1 per byte)
  };
  auto pf_tile = [&](int stage) __attribute__((always_inline)) {
    PfTile T;
    int k = stage / p.nchunk, ch = stage - k * p.nchunk;
    int z0, y0, x0;
    tile_origin(k, &z0, &y0, &x0);
    const int gz0 = z0 - 1, gy0 = y0 - 1, gx0 = x0 - 1;
    T.xb = xg + ch * 16;
    T.org = ((gz0 * p.Hi + gy0) * p.Wi + gx0) * p.Cin;
    T.lo = (unsigned)(gz0 < 0) | ((unsigned)(gy0 < 0) << 8) | ((unsigned)(gx0 < 0) << 16);
    T.hi1 = (unsigned)(min(p.Di - 1 - gz0, 0x7e) + 1) | ((unsigned)(min(p.Hi - 1 - gy0, 0x7e) + 1) << 8) |
            ((unsigned)(min(p.Wi - 1 - gx0, 0x7e) + 1) << 16);
    return T;
  };
  float4 pf[UP_MAXS];
  auto pf_slot = [&](const PfTile& T, int i) __attribute__((always_inline)) {
    const unsigned t1 = pg[i] - T.lo;          // byte f keeps its top bit iff f >= lo_f
    const unsigned t2 = T.hi1 + ~pg[i];        // byte f has its top bit iff f <= hi_f
    const bool ok = ((t1 & t2) & 0x808080u) == 0x808080u;
    pf[i] = ld4(ok ? (T.xb + (T.org + goff[i])) : p.zeros);
  };

  f32x2 ssum2[2] = {{0.f, 0.f}, {0.f, 0.f}}, ssq2[2] = {{0.f, 0.f}, {0.f, 0.f}};   // channels (q&1)*4 + {0,1 | 2,3}
  f32x4 acc[UP_TY][4];
  // this sample's output as a buffer: stores beyond ybytes are dropped by the range check
  const unsigned ybytes = (unsigned)(p.gy * 4);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(yg, 0, ybytes, 0x00020000);

  if (nstage > 0) {
    const PfTile T0 = pf_tile(0);
#pragma unroll
    for (int i = 0; i < UP_MAXS; ++i) pf_slot(T0, i);
  }

  for (int stage = 0; stage < nstage; ++stage) {
    const int k = stage / p.nchunk, ch = stage - k * p.nchunk;
    if (ch == 0) {
#pragma unroll
      for (int t = 0; t < UP_TY; ++t)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[t][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();                       // every wavefront is done reading the previous stage's image (and, the first
                                           // time, the weights are in LDS after the second barrier below)
#pragma unroll
    for (int i = 0; i < UP_MAXS; ++i)
      if (i < UP_MAXS - 1 || tid + i * 256 < UP_SLOTS) *reinterpret_cast<float4*>(smem + laddr[i]) = pf[i];
    __syncthreads();

    const PfTile T = pf_tile(min(stage + 1, nstage - 1));      // last stage: harmless re-read of its own halo
    const int wb = wbase + ch * UP_WCH;

    // ---- output addressing of this tile: this lane holds channels (q&1)*4..+3 of output voxel (2z+pz, 2(y0+t)+py,
    // 2(x0+r) + (q>>1)), (pz, py) = tile m; 32-bit element offsets inside the sample
    int tz0, ty0, tx0;
    tile_origin(k, &tz0, &ty0, &tx0);
    const int zo = tz0 + wave, xo = tx0 + r;
    const bool evox_ok = zo < p.Di && xo < p.Wi;
    const unsigned Hy = 2u * p.Hi, Wy = 2u * p.Wi;
    const unsigned erow = Wy * p.ldy;                                            // elements per output row
    const unsigned eo = (((unsigned)(2 * zo) * Hy + 2 * ty0) * Wy + 2 * xo + (q >> 1)) * p.ldy + p.ycoff + (q & 1) * 4;
    const unsigned eplane = Hy * erow;
    const unsigned vo_ok = evox_ok ? eo * 4u : ybytes;       // byte offset of this lane, or out of range (store dropped)
    // row t of tile m, branch-free: voffset = the lane's constant byte offset (out of range for rows outside the volume:
    // the buffer's range check drops the store), the uniform (tile, row) displacement travels in the scalar offset; the
    // moments use packed adds / fused multiply-adds (two channels per instruction).  NaN passes ReLU, as in tf.nn.relu
    auto flush = [&](auto MASKED, auto MT, auto TT) __attribute__((always_inline)) {
      constexpr bool masked = decltype(MASKED)::value;
      constexpr int m = decltype(MT)::value, t = decltype(TT)::value;
      float a0 = acc[t][m][0], a1 = acc[t][m][1], a2 = acc[t][m][2], a3 = acc[t][m][3];
      if (RELU) {
        a0 = (a0 < 0.f) ? 0.f : a0; a1 = (a1 < 0.f) ? 0.f : a1;
        a2 = (a2 < 0.f) ? 0.f : a2; a3 = (a3 < 0.f) ? 0.f : a3;
      }
      const unsigned soff = ((m >> 1) * eplane + (2 * t + (m & 1)) * erow) * 4u;      // uniform: scalar offset
      // (bit casts of the scalars: __builtin_bit_cast of an ext-vector ELEMENT picked element 0 twice with this compiler)
      const u32x4 bits = {__builtin_bit_cast(unsigned, a0), __builtin_bit_cast(unsigned, a1),
                          __builtin_bit_cast(unsigned, a2), __builtin_bit_cast(unsigned, a3)};
      f32x2 lo = {a0, a1}, hi = {a2, a3};
      if (masked) {
        const bool row_ok = ty0 + t < p.Hi;                  // uniform
        __builtin_amdgcn_raw_buffer_store_b128(bits, yrsrc, row_ok ? vo_ok : ybytes, soff, 0);
        const bool ok = evox_ok && row_ok;
        lo.x = ok ? lo.x : 0.f; lo.y = ok ? lo.y : 0.f; hi.x = ok ? hi.x : 0.f; hi.y = ok ? hi.y : 0.f;
      } else {
        __builtin_amdgcn_raw_buffer_store_b128(bits, yrsrc, vo_ok, soff, 0);
      }
      ssum2[0] += lo;
      ssum2[1] += hi;
      ssq2[0] = __builtin_elementwise_fma(lo, lo, ssq2[0]);
      ssq2[1] = __builtin_elementwise_fma(hi, hi, ssq2[1]);
    };

    auto kloop = [&](auto MASKED) __attribute__((always_inline)) {
      // ---- K loop: 8 offsets, fully unrolled; fragments and weights of offset o + 1 are requested in front of the MFMAs
      // of offset o, the next halo's slots two per offset behind them
      float4 B[2][UP_TY], Wt[2][4];
      auto request = [&](auto OT) __attribute__((always_inline)) {
        constexpr int o = decltype(OT)::value, oz = o >> 2, oy = (o >> 1) & 1, ox = o & 1;
        constexpr int disp = (((1 - oz) * UP_HY + (1 - oy)) * UP_HX + (1 - ox)) * UP_VB;
  #pragma unroll
        for (int t = 0; t < UP_TY; ++t)
          B[o & 1][t] = *reinterpret_cast<const float4*>(smem + fbase + (disp + t * UP_ROWB));
        if constexpr (up_active(0, oz, oy)) Wt[o & 1][0] = *reinterpret_cast<const float4*>(smem + wb + up_pair_index(o, 0) * 1024);
        if constexpr (up_active(1, oz, oy)) Wt[o & 1][1] = *reinterpret_cast<const float4*>(smem + wb + up_pair_index(o, 1) * 1024);
        if constexpr (up_active(2, oz, oy)) Wt[o & 1][2] = *reinterpret_cast<const float4*>(smem + wb + up_pair_index(o, 2) * 1024);
        if constexpr (up_active(3, oz, oy)) Wt[o & 1][3] = *reinterpret_cast<const float4*>(smem + wb + up_pair_index(o, 3) * 1024);
      };
      // one group = the 8 MFMAs (rows t) of (tile m, k-substep s); behind group number gi of offset o goes, in the SINGLE
      // form, one row of a tile that is already complete: offsets 2,3 -> tile 3, offsets 4,5 -> tile 2 (a row behind every
      // other of their 8 groups), offsets 6,7 -> tile 1 (a row behind each of their 4 groups)
      auto group = [&](auto OT, auto MT, auto ST, auto GI) __attribute__((always_inline)) {
        constexpr int o = decltype(OT)::value, m = decltype(MT)::value, sub = decltype(ST)::value, gi = decltype(GI)::value;
  #pragma unroll
        for (int t = 0; t < UP_TY; ++t)
          acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(Wt[o & 1][m], sub), f4get(B[o & 1][t], sub), acc[t][m], 0, 0, 0);
        if constexpr (SINGLE && o >= 2) {
          if constexpr (o < 6) {
            if constexpr ((gi & 1) == 0)
              flush(MASKED, std::integral_constant<int, (o < 4) ? 3 : 2>{}, std::integral_constant<int, (o & 1) * 4 + gi / 2>{});
          } else {
            flush(MASKED, std::integral_constant<int, 1>{}, std::integral_constant<int, (o & 1) * 4 + gi>{});
          }
        }
      };
      auto tile_groups = [&](auto OT, auto MT, auto G0) __attribute__((always_inline)) {
        constexpr int g0 = decltype(G0)::value;
        group(OT, MT, std::integral_constant<int, 0>{}, std::integral_constant<int, g0>{});
        group(OT, MT, std::integral_constant<int, 1>{}, std::integral_constant<int, g0 + 1>{});
        group(OT, MT, std::integral_constant<int, 2>{}, std::integral_constant<int, g0 + 2>{});
        group(OT, MT, std::integral_constant<int, 3>{}, std::integral_constant<int, g0 + 3>{});
      };
      auto step = [&](auto OT) __attribute__((always_inline)) {
        constexpr int o = decltype(OT)::value, oz = o >> 2, oy = (o >> 1) & 1;
        if constexpr (o + 1 < 8) request(std::integral_constant<int, o + 1>{});
        if constexpr (2 * o < UP_MAXS) pf_slot(T, 2 * o);
        if constexpr (2 * o + 1 < UP_MAXS) pf_slot(T, 2 * o + 1);
        // compiler barrier (keeps the requests from sinking to their uses) + scheduling barrier (keeps them in front of
        // the MFMAs that cover their latency)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        constexpr int n1 = up_active(1, oz, oy) ? 4 : 0, n2 = up_active(2, oz, oy) ? 4 : 0;
        tile_groups(OT, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        if constexpr (up_active(1, oz, oy)) tile_groups(OT, std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{});
        if constexpr (up_active(2, oz, oy)) tile_groups(OT, std::integral_constant<int, 2>{}, std::integral_constant<int, 4 + n1>{});
        if constexpr (up_active(3, oz, oy)) tile_groups(OT, std::integral_constant<int, 3>{}, std::integral_constant<int, 4 + n1 + n2>{});
      };
      request(std::integral_constant<int, 0>{});
      asm volatile("" ::: "memory");
      step(std::integral_constant<int, 0>{});
      step(std::integral_constant<int, 1>{});
      step(std::integral_constant<int, 2>{});
      step(std::integral_constant<int, 3>{});
      step(std::integral_constant<int, 4>{});
      step(std::integral_constant<int, 5>{});
      step(std::integral_constant<int, 6>{});
      step(std::integral_constant<int, 7>{});
      static_assert(2 * 6 >= UP_MAXS, "every halo slot is requested inside the K loop");
      if (!SINGLE && ch != p.nchunk - 1) return;

      // ---- the rest of the epilogue: SINGLE -> tile 0; otherwise every tile
      auto flush_tile = [&](auto MT) __attribute__((always_inline)) {
        flush(MASKED, MT, std::integral_constant<int, 0>{}); flush(MASKED, MT, std::integral_constant<int, 1>{});
        flush(MASKED, MT, std::integral_constant<int, 2>{}); flush(MASKED, MT, std::integral_constant<int, 3>{});
        flush(MASKED, MT, std::integral_constant<int, 4>{}); flush(MASKED, MT, std::integral_constant<int, 5>{});
        flush(MASKED, MT, std::integral_constant<int, 6>{}); flush(MASKED, MT, std::integral_constant<int, 7>{});
      };
      flush_tile(std::integral_constant<int, 0>{});
      if (!SINGLE) {
        flush_tile(std::integral_constant<int, 1>{});
        flush_tile(std::integral_constant<int, 2>{});
        flush_tile(std::integral_constant<int, 3>{});
      }
    };
    // (an unmasked second copy of the loop for tiles inside the volume would save 4 selects per row, but the two copies
    // together no longer fit the register file: 87+ spills)
    kloop(std::true_type{});
  }

  // ---- per-workgroup partial moments (sum, sum of squares) per output channel -> row blockIdx of stats:
  // [2][16] doubles, columns 0..7 = channels, 8..15 = 0 (the layout of conv_xp.hip)
  if (p.stats) {
    __syncthreads();
    double* s_red = reinterpret_cast<double*>(smem);   // [4 waves][2][8]
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      double a = (double)ssum2[kk >> 1][kk & 1], bq = (double)ssq2[kk >> 1][kk & 1];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        a += __shfl_xor(a, o);
        bq += __shfl_xor(bq, o);
      }
      a += __shfl_xor(a, 32);      // lanes q and q^2 hold the same channels (the two x parities)
      bq += __shfl_xor(bq, 32);
      if (r == 0 && q < 2) {
        s_red[(wave * 2 + 0) * 8 + q * 4 + kk] = a;
        s_red[(wave * 2 + 1) * 8 + q * 4 + kk] = bq;
      }
    }
    __syncthreads();
    if (tid < 32) {
      const int which = tid >> 4, col = tid & 15;
      double v = 0.0;
      if (col < 8)
        v = (s_red[(0 * 2 + which) * 8 + col] + s_red[(1 * 2 + which) * 8 + col]) +
            (s_red[(2 * 2 + which) * 8 + col] + s_red[(3 * 2 + which) * 8 + col]);
      p.stats[((size_t)blockIdx.x * 2 + which) * 16 + col] = v;
    }
  }
}

long up_ntiles(int D, int H, int W) {
  return (long)((D + UP_TZ - 1) / UP_TZ) * ((H + UP_TY - 1) / UP_TY) * ((W + UP_TX - 1) / UP_TX);
}

bool up_shape_ok(int Cin, int Cout) { return Cout == 8 && Cin > 0 && Cin % 16 == 0 && Cin <= 64; }

template <bool SINGLE, bool RELU>
int launch_up(const UpArgs& a, long grid, size_t lds, atvs_stream_t stream) {
  // the attribute is per device: one flag per device ordinal of this process (and per instantiation)
  static bool attr_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return ATVS_ERR_LAUNCH;
  if (!attr_set[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(deconv_up_kernel<SINGLE, RELU>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return ATVS_ERR_LAUNCH;
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL((deconv_up_kernel<SINGLE, RELU>), dim3((unsigned)grid), dim3(256), lds, as_stream(stream), a);
  return ATVS_OK;
}

}  // namespace

// Floats of the packed form of a transposed-convolution kernel [3,3,3,Cout,Cin] (Cout == 8, Cin % 16 == 0, Cin <= 64),
// including 4 trailing zeros.
extern "C" int atvs_deconv_up_pack_size(int Cin, int Cout, long* packed_floats) {
  if (!packed_floats) return ATVS_ERR_NULL;
  if (!up_shape_ok(Cin, Cout)) return ATVS_ERR_SHAPE;
  *packed_floats = (long)(Cin / 16) * UP_PAIRS * 256 + 4;
  return ATVS_OK;
}

// HOST function.  w: TF kernel of tf.layers.conv3d_transpose, [3,3,3,Cout,Cin].  packed[chunk][block][lane][s]: block =
// the non-zero (offset, tile) pairs in K-loop order; lane = q*16 + row, row = px*8 + co of tile (pz,py); value =
// w[kd][kh][kw][co][chunk*16 + 4q + s] with k per axis from (parity, offset): (0,0) -> 0, (0,-1) -> 2, (1,0) -> 1,
// (1,-1) -> structural zero.
extern "C" int atvs_deconv_up_pack(const float* w, int Cin, int Cout, float* packed) {
  if (!w || !packed) return ATVS_ERR_NULL;
  long pf;
  int rc = atvs_deconv_up_pack_size(Cin, Cout, &pf);
  if (rc) return rc;
  for (long i = 0; i < pf; ++i) packed[i] = 0.f;
  auto kof = [](int par, int off) { return par ? (off ? -1 : 1) : (off ? 2 : 0); };
  for (int ch = 0; ch < Cin / 16; ++ch)
    for (int o = 0; o < 8; ++o)
      for (int m = 0; m < 4; ++m) {
        const int oz = o >> 2, oy = (o >> 1) & 1, ox = o & 1;
        if (!up_active(m, oz, oy)) continue;
        const int pi = up_pair_index(o, m);
        const int kd = kof(m >> 1, oz), kh = kof(m & 1, oy);
        for (int q = 0; q < 4; ++q)
          for (int row = 0; row < 16; ++row) {
            const int px = row >> 3, co = row & 7;
            const int kw = kof(px, ox);
            if (kw < 0) continue;
            for (int s = 0; s < 4; ++s) {
              const int ci = ch * 16 + 4 * q + s;
              packed[((((size_t)ch * UP_PAIRS + pi) * 64) + q * 16 + row) * 4 + s] =
                  w[((((size_t)kd * 3 + kh) * 3 + kw) * Cout + co) * Cin + ci];
            }
          }
      }
  return ATVS_OK;
}

// workgroups PER SAMPLE of a launch over `groups` independent samples (rows of the statistics buffer = groups * this):
// one workgroup per CU in all, shared out among the samples, a multiple of 8 each
extern "C" long atvs_deconv_up_grid(int D, int H, int W, int groups) {
  if (groups < 1) groups = 1;
  long nt = up_ntiles(D, H, W);
  long share = 256 / groups / 8 * 8;
  if (share < 8) share = 8;
  long g = nt < share ? nt : share;
  return (g + 7) / 8 * 8;
}

// y (2D, 2H, 2W, ldy)[..., y_coff : y_coff + 8] = conv3d_transpose(x (D,H,W,Cin), w, stride 2, SAME) (+ ReLU), `groups`
// independent samples on the leading axis of x / y.  stats_partial: groups * atvs_deconv_up_grid rows of [2][16] doubles
// (partial sums / sums of squares per channel of the stored values), or NULL.
extern "C" int atvs_deconv_up_f32(const float* x, const float* packed_w, float* y, double* stats_partial, int groups, int D,
                                  int H, int W, int Cin, int Cout, int ldy, int y_coff, int relu, atvs_stream_t stream) {
  if (!x || !packed_w || !y) return ATVS_ERR_NULL;
  if (!up_shape_ok(Cin, Cout) || groups <= 0 || D <= 0 || H <= 0 || W <= 0) return ATVS_ERR_SHAPE;
  if (y_coff < 0 || y_coff + 8 > ldy || (ldy % 4) || (y_coff % 4)) return ATVS_ERR_SHAPE;
  if ((double)D * H * W * Cin >= 2147483648.0) return ATVS_ERR_SHAPE;            // 31-bit halo-relative element offsets
  if (8.0 * D * H * W * ldy * 4.0 >= 4294967296.0) return ATVS_ERR_SHAPE;        // 32-bit output BYTE offsets (buffer stores)
  UpArgs a;
  a.x = x; a.wp = packed_w; a.zeros = packed_w + (size_t)(Cin / 16) * UP_PAIRS * 256;
  a.y = y; a.stats = stats_partial;
  a.Di = D; a.Hi = H; a.Wi = W; a.Cin = Cin; a.ldy = ldy; a.ycoff = y_coff; a.nchunk = Cin / 16;
  a.tiles_y = (H + UP_TY - 1) / UP_TY; a.tiles_x = (W + UP_TX - 1) / UP_TX;
  a.ntiles = (int)up_ntiles(D, H, W);
  a.relu = relu;
  const long blocks = atvs_deconv_up_grid(D, H, W, groups);
  a.wg = (int)blocks;
  a.gx = (long)D * H * W * Cin; a.gy = 8L * D * H * W * ldy;
  if (blocks * groups > 0x7fffffffL) return ATVS_ERR_SHAPE;
  const size_t lds = (size_t)UP_IMG + (size_t)a.nchunk * UP_WCH;
  int rc;
  if (a.nchunk == 1) rc = relu ? launch_up<true, true>(a, blocks * groups, lds, stream) : launch_up<true, false>(a, blocks * groups, lds, stream);
  else rc = relu ? launch_up<false, true>(a, blocks * groups, lds, stream) : launch_up<false, false>(a, blocks * groups, lds, stream);
  if (rc) return rc;
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}
