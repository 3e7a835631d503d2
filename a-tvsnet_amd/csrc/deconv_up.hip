// 3x3x3 stride-2 SAME transposed convolution to 8 or 16 output channels, all 8 output parity classes from one staged
// input tile, one wavefront per SIMD (gfx950).
//
// These are the decoder layers of the stacked U-Nets / the refinement net that write the two finest resolutions
// (conv_b*_6_0, global_refine_3dconv6_0: 16 -> 8 channels, full-resolution output; conv_b*_5_0, global_refine_3dconv5_0:
// 32 -> 16; cnn_wrapper/atvsnet.py StackedUNet / CostVolRefineNet; layer code /root/reference/cnn_wrapper/network.py:
// 510-550, tf.layers.conv3d_transpose(3, strides 2, 'same')).  Per launch the 8-channel ones write as many bytes as the
// widest convolutions and are, after those, the largest block of MFMA work.
//
// Form.  Per axis an even output 2j is  w[0] x[j] + w[2] x[j-1]  and an odd output 2j+1 is  w[1] x[j]:  output voxel
// (2z+pz, 2y+py, 2x+px) (parity class (pz,py,px)) reads the input voxels (z+oz, y+oy, x+ox), o in {0,-1}, a class with
// an odd parity only offset 0 on that axis.  As a GEMM per input voxel: M = 8 classes x Cout rows, K = 8 offsets x Cin;
// only the non-zero (16-row tile, offset) blocks are issued:
//   Cout =  8: 4 tiles, tile m = (pz,py), rows = (px, channel); 18 of 32 blocks (the x pair shares a tile: 3/4 useful);
//   Cout = 16: 8 tiles, tile m = (pz,py,px), rows = channel;    27 of 64 blocks (all useful).
//
// Kernel.  Built like conv_xw.hip (the measurements behind its round-2 predecessor conv_xp apply): ONE workgroup of 4 wavefronts per CU with
// the whole register file, a fully unrolled K loop whose LDS reads are base register + immediate, the next tile's halo
// requested between the MFMAs.  Tile = 4(z) x TY(y) x 16(x) INPUT voxels, TY = 8 / 4 for Cout = 8 / 16; wavefront w
// owns input plane z0 + w with TY rows x NT tiles = 32 accumulator tiles.  The halo is one-sided (offsets 0 / -1 only):
// 5 x (TY+1) x 17 voxels.  The packed weights of every chunk stay in LDS for the whole launch.  Stores are buffer
// stores: rows outside the volume get an out-of-range offset and are dropped by the hardware's range check (measured on
// gfx950, tools_dev/micro/bufstore_semantics.hip: a store is dropped iff voffset + soffset reaches num_records) -- no
// branches in the epilogue, so that (Cout = 8, one chunk) it can be interleaved with the K loop.
#include <type_traits>
#include <utility>

#include "conv_common.h"

namespace {

constexpr int UP_TZ = 4, UP_TX = 16;
constexpr int UP_HZ = UP_TZ + 1, UP_HX = UP_TX + 1;
// LDS image: 64 bytes per voxel (16 channels), rows of 24 voxels (17 used): the row pitch is a multiple of 512 bytes, so
// every (dz, dy, row) displacement is an immediate that commutes with the bank swizzle (bit 5 ^= bit 8: the 16-lane
// groups of a 128-bit read -- lanes {0-3,12-15,20-27}, ... -- then fall into 16 distinct 16-byte bank slots whatever the
// x alignment; unswizzled, 37 % of the LDS cycles were conflict cycles).  The two x offsets get a base register each.
constexpr int UP_VB = 64;
constexpr int UP_PITCHV = 24;
constexpr int UP_ROWB = UP_PITCHV * UP_VB;             // bytes per image row
__device__ __forceinline__ int up_swz(int a) { return a ^ (((a >> 8) & 1) << 5); }

template <int COUT>
struct Up {
  static_assert(COUT == 8 || COUT == 16, "built for 8 and 16 output channels");
  static constexpr int NT = (COUT == 8) ? 4 : 8;       // 16-row tiles of the 8 x COUT virtual channels
  static constexpr int TY = 32 / NT;                   // input rows per wavefront (32 accumulator tiles)
  static constexpr int HY = TY + 1;
  static constexpr int IMG = UP_HZ * HY * UP_ROWB;     // bytes of the staged halo
  static constexpr int SLOTS = UP_HZ * HY * UP_HX * 4; // 16-byte halo slots
  static constexpr int MAXS = (SLOTS + 255) / 256;     // per thread
  static constexpr int SPO = (MAXS + 7) / 8;           // halo slots requested per offset step
  // tile m reads offset o = oz*4 + oy*2 + ox (bit = 1: offset -1) iff no axis pairs an odd parity with offset -1
  static constexpr bool active(int m, int o) {
    const int oz = o >> 2, oy = (o >> 1) & 1, ox = o & 1;
    if (COUT == 8) return ((m >> 1) == 0 || oz == 0) && ((m & 1) == 0 || oy == 0);
    return ((m >> 2) == 0 || oz == 0) && (((m >> 1) & 1) == 0 || oy == 0) && ((m & 1) == 0 || ox == 0);
  }
  // index of block (o, m) among the non-zero ones, in K-loop order; (8, 0) = their number
  static constexpr int pair_index(int o, int m) {
    int n = 0;
    for (int oo = 0; oo < 8; ++oo)
      for (int mm = 0; mm < NT; ++mm) {
        if (oo == o && mm == m) return n;
        if (active(mm, oo)) ++n;
      }
    return n;
  }
  static constexpr int PAIRS = pair_index(8, 0);
  static constexpr int WCH = PAIRS * 1024;             // bytes of packed weights per chunk
};
static_assert(Up<8>::PAIRS == 18 && Up<16>::PAIRS == 27, "non-zero (tile, offset) blocks");

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
  int wg;                // workgroups per sample (gridDim.x = groups * wg)
  long gx, gy;           // elements per sample of x / y
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int N>
using IC = std::integral_constant<int, N>;
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(IC<I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// SINGLE: Cin == 16 (one chunk per tile).  With Cout == 8 the tiles' accumulators are then stored as soon as their last
// MFMA is issued, interleaved with the MFMAs of the remaining offsets (tile (1,1) is complete after 2 of the 8 offsets,
// (1,0) after 4, (0,1) after 6): only a quarter of the epilogue is left behind the K loop.
template <int COUT, bool SINGLE, bool RELU>
__global__ __launch_bounds__(256, 1) void deconv_up_kernel(UpArgs p) {
  using U = Up<COUT>;
  constexpr int NT = U::NT, TY = U::TY, HY = U::HY, MAXS = U::MAXS;
  constexpr bool INTERLEAVE = SINGLE && COUT == 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;

  // packed weights of every chunk -> LDS, once (visible after the first stage's barriers)
  {
    const float4* src = reinterpret_cast<const float4*>(p.wp);
    float4* dst = reinterpret_cast<float4*>(smem + U::IMG);
    for (int i = tid; i < p.nchunk * (U::WCH / 16); i += 256) dst[i] = src[i];
  }

  // LDS read bases: this lane's fragment at halo voxel (wave, 0, r + 1 - ox) = offset (-1, -1, ox) of row 0 of the
  // wavefront's plane, swizzled; every (dz, dy, row) is a non-negative immediate from there
  int fbase[2];
#pragma unroll
  for (int ox = 0; ox < 2; ++ox) fbase[ox] = up_swz(((wave * HY) * UP_PITCHV + r + 1 - ox) * UP_VB + q * 16);
  const int wbase = U::IMG + lane * 16;

  // per-slot constants of this thread: global element offset from the halo origin and the packed halo coordinate
  // (zz | yy<<8 | xx<<16, each byte with its top bit set) for the bounds test; its swizzled LDS byte address
  int goff[MAXS], laddr[MAXS];
  unsigned pg[MAXS];
#pragma unroll
  for (int i = 0; i < MAXS; ++i) {
    int s = tid + i * 256;
    const bool live = s < U::SLOTS;
    s = min(s, U::SLOTS - 1);
    const int c4 = s & 3, v = s >> 2;
    const int xx = v % UP_HX, v2 = v / UP_HX;
    const int yy = v2 % HY, zz = v2 / HY;
    goff[i] = ((zz * p.Hi + yy) * p.Wi + xx) * p.Cin + c4 * 4;
    laddr[i] = up_swz(((zz * HY + yy) * UP_PITCHV + xx) * UP_VB + c4 * 16);
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
    *y0 = (rest % p.tiles_y) * TY;
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
  float4 pf[MAXS];
  auto pf_slot = [&](const PfTile& T, int i) __attribute__((always_inline)) {
    const unsigned t1 = pg[i] - T.lo;          // byte f keeps its top bit iff f >= lo_f
    const unsigned t2 = T.hi1 + ~pg[i];        // byte f has its top bit iff f <= hi_f
    const bool ok = ((t1 & t2) & 0x808080u) == 0x808080u;
    pf[i] = ld4(ok ? (T.xb + (T.org + goff[i])) : p.zeros);
  };

  // moments of this lane's 4 channels: Cout 8: (q&1)*4 + {0,1 | 2,3};  Cout 16: 4q + {0,1 | 2,3}
  f32x2 ssum2[2] = {{0.f, 0.f}, {0.f, 0.f}}, ssq2[2] = {{0.f, 0.f}, {0.f, 0.f}};
  f32x4 acc[TY][NT];
  // this sample's output as a buffer: stores at or beyond ybytes are dropped by the range check
  const unsigned ybytes = (unsigned)(p.gy * 4);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(yg, 0, ybytes, 0x00020000);

  if (nstage > 0) {
    const PfTile T0 = pf_tile(0);
#pragma unroll
    for (int i = 0; i < MAXS; ++i) pf_slot(T0, i);
  }

  for (int stage = 0; stage < nstage; ++stage) {
    const int k = stage / p.nchunk, ch = stage - k * p.nchunk;
    if (ch == 0) {
#pragma unroll
      for (int t = 0; t < TY; ++t)
#pragma unroll
        for (int m = 0; m < NT; ++m) acc[t][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();                       // every wavefront is done reading the previous stage's image
#pragma unroll
    for (int i = 0; i < MAXS; ++i)
      if (i < MAXS - 1 || tid + i * 256 < U::SLOTS) *reinterpret_cast<float4*>(smem + laddr[i]) = pf[i];
    __syncthreads();

    const PfTile T = pf_tile(min(stage + 1, nstage - 1));      // last stage: harmless re-read of its own halo
    const int wb = wbase + ch * U::WCH;

    // ---- output addressing of this tile (32-bit element offsets inside the sample).  This lane holds, of tile m,
    //   Cout  8: channels (q&1)*4..+3 of output voxel (2z+pz, 2(y0+t)+py, 2(x0+r) + (q>>1)), (pz,py) = m;
    //   Cout 16: channels 4q..+3      of output voxel (2z+pz, 2(y0+t)+py, 2(x0+r) + px),     (pz,py,px) = m.
    int tz0, ty0, tx0;
    tile_origin(k, &tz0, &ty0, &tx0);
    const int zo = tz0 + wave, xo = tx0 + r;
    const bool evox_ok = zo < p.Di && xo < p.Wi;
    const unsigned Hy = 2u * p.Hi, Wy = 2u * p.Wi;
    const unsigned erow = Wy * p.ldy;                                            // elements per output row
    const unsigned eplane = Hy * erow;
    const unsigned lane_c = (COUT == 8) ? (unsigned)((q >> 1) * p.ldy + (q & 1) * 4) : (unsigned)(q * 4);
    const unsigned eo = (((unsigned)(2 * zo) * Hy + 2 * ty0) * Wy + 2 * xo) * p.ldy + p.ycoff + lane_c;
    const unsigned vo_ok = evox_ok ? eo * 4u : ybytes;       // byte offset of this lane, or out of range (store dropped)
    // row t of tile m, branch-free: voffset = the lane's constant byte offset (out of range for rows outside the volume),
    // the uniform (tile, row) displacement travels in the scalar offset; the moments use packed adds / fused
    // multiply-adds (two channels per instruction).  NaN passes ReLU, as in tf.nn.relu
    auto flush = [&](auto MT, auto TT) __attribute__((always_inline)) {
      constexpr int m = decltype(MT)::value, t = decltype(TT)::value;
      constexpr int pz = (COUT == 8) ? (m >> 1) : (m >> 2), py = (COUT == 8) ? (m & 1) : ((m >> 1) & 1);
      constexpr int px = (COUT == 8) ? 0 : (m & 1);
      float a0 = acc[t][m][0], a1 = acc[t][m][1], a2 = acc[t][m][2], a3 = acc[t][m][3];
      if (RELU) {
        a0 = (a0 < 0.f) ? 0.f : a0; a1 = (a1 < 0.f) ? 0.f : a1;
        a2 = (a2 < 0.f) ? 0.f : a2; a3 = (a3 < 0.f) ? 0.f : a3;
      }
      const unsigned soff = (pz * eplane + (2 * t + py) * erow + px * p.ldy) * 4u;          // uniform: scalar offset
      // (bit casts of the scalars: __builtin_bit_cast of an ext-vector ELEMENT picked element 0 twice with this compiler)
      const u32x4 bits = {__builtin_bit_cast(unsigned, a0), __builtin_bit_cast(unsigned, a1),
                          __builtin_bit_cast(unsigned, a2), __builtin_bit_cast(unsigned, a3)};
      const bool row_ok = ty0 + t < p.Hi;                    // uniform
      __builtin_amdgcn_raw_buffer_store_b128(bits, yrsrc, row_ok ? vo_ok : ybytes, soff, 0);
      const bool ok = evox_ok && row_ok;
      f32x2 lo = {ok ? a0 : 0.f, ok ? a1 : 0.f}, hi = {ok ? a2 : 0.f, ok ? a3 : 0.f};
      ssum2[0] += lo;
      ssum2[1] += hi;
      ssq2[0] = __builtin_elementwise_fma(lo, lo, ssq2[0]);
      ssq2[1] = __builtin_elementwise_fma(hi, hi, ssq2[1]);
    };

    // ---- K loop: 8 offsets, fully unrolled; fragments and weights of offset o + 1 are requested in front of the MFMAs
    // of offset o, the next halo's slots SPO per offset behind them
    float4 B[2][TY], Wt[2][NT];
    auto request = [&](auto OT) __attribute__((always_inline)) {
      constexpr int o = decltype(OT)::value, oz = o >> 2, oy = (o >> 1) & 1, ox = o & 1;
      constexpr int disp = ((1 - oz) * HY + (1 - oy)) * UP_ROWB;
#pragma unroll
      for (int t = 0; t < TY; ++t)
        B[o & 1][t] = *reinterpret_cast<const float4*>(smem + fbase[ox] + (disp + t * UP_ROWB));
      static_for<NT>([&](auto MT) __attribute__((always_inline)) {
        constexpr int m = decltype(MT)::value;
        if constexpr (U::active(m, o))
          Wt[o & 1][m] = *reinterpret_cast<const float4*>(smem + wb + U::pair_index(o, m) * 1024);
      });
    };
    // one group = the TY MFMAs (rows t) of (tile m, k-substep s).  INTERLEAVE: behind group number gi of offset o goes one
    // row of a tile that is already complete: offsets 2,3 -> tile 3, offsets 4,5 -> tile 2 (a row behind every other of
    // their 8 groups), offsets 6,7 -> tile 1 (a row behind each of their 4 groups)
    auto group = [&](auto OT, auto MT, auto ST, auto GI) __attribute__((always_inline)) {
      constexpr int o = decltype(OT)::value, m = decltype(MT)::value, sub = decltype(ST)::value, gi = decltype(GI)::value;
#pragma unroll
      for (int t = 0; t < TY; ++t)
        acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(Wt[o & 1][m], sub), f4get(B[o & 1][t], sub), acc[t][m], 0, 0, 0);
      if constexpr (INTERLEAVE && o >= 2) {
        if constexpr (o < 6) {
          if constexpr ((gi & 1) == 0) flush(IC<(o < 4) ? 3 : 2>{}, IC<(o & 1) * 4 + gi / 2>{});
        } else {
          flush(IC<1>{}, IC<(o & 1) * 4 + gi>{});
        }
      }
    };
    auto step = [&](auto OT) __attribute__((always_inline)) {
      constexpr int o = decltype(OT)::value;
      if constexpr (o + 1 < 8) request(IC<o + 1>{});
      static_for<U::SPO>([&](auto JT) __attribute__((always_inline)) {
        constexpr int i = o * U::SPO + decltype(JT)::value;
        if constexpr (i < MAXS) pf_slot(T, i);
      });
      // compiler barrier (keeps the requests from sinking to their uses) + scheduling barrier (keeps them in front of
      // the MFMAs that cover their latency)
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      static_for<NT>([&](auto MT) __attribute__((always_inline)) {
        constexpr int m = decltype(MT)::value;
        if constexpr (U::active(m, o)) {
          constexpr int g0 = 4 * (U::pair_index(o, m) - U::pair_index(o, 0));     // groups of this offset so far
          group(OT, MT, IC<0>{}, IC<g0>{});
          group(OT, MT, IC<1>{}, IC<g0 + 1>{});
          group(OT, MT, IC<2>{}, IC<g0 + 2>{});
          group(OT, MT, IC<3>{}, IC<g0 + 3>{});
        }
      });
    };
    static_assert(8 * U::SPO >= MAXS, "every halo slot is requested inside the K loop");
    request(IC<0>{});
    asm volatile("" ::: "memory");
    static_for<8>([&](auto OT) __attribute__((always_inline)) { step(OT); });
    if (!SINGLE && ch != p.nchunk - 1) continue;

    // ---- the rest of the epilogue: INTERLEAVE -> tile 0; otherwise every tile
    static_for<NT>([&](auto MT) __attribute__((always_inline)) {
      if constexpr (!INTERLEAVE || decltype(MT)::value == 0)
        static_for<TY>([&](auto TT) __attribute__((always_inline)) { flush(MT, TT); });
    });
  }

  // ---- per-workgroup partial moments (sum, sum of squares) per output channel -> row blockIdx of stats: [2][16]
  // doubles (Cout 8: columns 8..15 = 0, the layout of conv_xw.hip)
  if (p.stats) {
    __syncthreads();
    double* s_red = reinterpret_cast<double*>(smem);   // [4 waves][2][16]
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      double a = (double)ssum2[kk >> 1][kk & 1], bq = (double)ssq2[kk >> 1][kk & 1];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        a += __shfl_xor(a, o);
        bq += __shfl_xor(bq, o);
      }
      if (COUT == 8) {
        a += __shfl_xor(a, 32);    // lanes q and q^2 hold the same channels (the two x parities)
        bq += __shfl_xor(bq, 32);
      }
      const int col = (COUT == 8) ? (q & 1) * 4 + kk : q * 4 + kk;
      if (r == 0 && (COUT == 16 || q < 2)) {
        s_red[(wave * 2 + 0) * 16 + col] = a;
        s_red[(wave * 2 + 1) * 16 + col] = bq;
      }
    }
    __syncthreads();
    if (tid < 32) {
      const int which = tid >> 4, col = tid & 15;
      double v = 0.0;
      if (col < COUT)
        v = (s_red[(0 * 2 + which) * 16 + col] + s_red[(1 * 2 + which) * 16 + col]) +
            (s_red[(2 * 2 + which) * 16 + col] + s_red[(3 * 2 + which) * 16 + col]);
      p.stats[((size_t)blockIdx.x * 2 + which) * 16 + col] = v;
    }
  }
}

int up_ty(int Cout) { return Cout == 8 ? Up<8>::TY : Up<16>::TY; }
int up_pairs(int Cout) { return Cout == 8 ? Up<8>::PAIRS : Up<16>::PAIRS; }

long up_ntiles(int D, int H, int W, int Cout) {
  const int ty = up_ty(Cout);
  return (long)((D + UP_TZ - 1) / UP_TZ) * ((H + ty - 1) / ty) * ((W + UP_TX - 1) / UP_TX);
}

bool up_shape_ok(int Cin, int Cout) { return (Cout == 8 || Cout == 16) && Cin > 0 && Cin % 16 == 0 && Cin <= 64; }

template <int COUT, bool SINGLE, bool RELU>
int launch_up(const UpArgs& a, long grid, atvs_stream_t stream) {
  const size_t lds = (size_t)Up<COUT>::IMG + (size_t)a.nchunk * Up<COUT>::WCH;
  // the attribute is per device: one flag per device ordinal of this process (and per instantiation)
  static AtvsAttrOnce lds_once;                   // per kernel instantiation (this function is a template / has one kernel)
  if (const int rc_ = atvs_set_max_lds_once(lds_once, reinterpret_cast<const void*>(deconv_up_kernel<COUT, SINGLE, RELU>), 160 * 1024)) return rc_;
  hipLaunchKernelGGL((deconv_up_kernel<COUT, SINGLE, RELU>), dim3((unsigned)grid), dim3(256), lds, as_stream(stream), a);
  return ATVS_OK;
}

template <int COUT>
int launch_up_cout(const UpArgs& a, long grid, int relu, atvs_stream_t stream) {
  if (a.nchunk == 1) return relu ? launch_up<COUT, true, true>(a, grid, stream) : launch_up<COUT, true, false>(a, grid, stream);
  return relu ? launch_up<COUT, false, true>(a, grid, stream) : launch_up<COUT, false, false>(a, grid, stream);
}

template <int COUT>
void pack_up(const float* w, int Cin, float* packed) {
  using U = Up<COUT>;
  auto kof = [](int par, int off) { return par ? (off ? -1 : 1) : (off ? 2 : 0); };
  for (int ch = 0; ch < Cin / 16; ++ch)
    for (int o = 0; o < 8; ++o)
      for (int m = 0; m < U::NT; ++m) {
        if (!U::active(m, o)) continue;
        const int oz = o >> 2, oy = (o >> 1) & 1, ox = o & 1;
        const int pi = U::pair_index(o, m);
        const int pz = (COUT == 8) ? (m >> 1) : (m >> 2), py = (COUT == 8) ? (m & 1) : ((m >> 1) & 1);
        const int kd = kof(pz, oz), kh = kof(py, oy);
        for (int q = 0; q < 4; ++q)
          for (int row = 0; row < 16; ++row) {
            const int px = (COUT == 8) ? (row >> 3) : (m & 1), co = (COUT == 8) ? (row & 7) : row;
            const int kw = kof(px, ox);
            if (kw < 0) continue;
            for (int s = 0; s < 4; ++s) {
              const int ci = ch * 16 + 4 * q + s;
              packed[((((size_t)ch * U::PAIRS + pi) * 64) + q * 16 + row) * 4 + s] =
                  w[((((size_t)kd * 3 + kh) * 3 + kw) * COUT + co) * Cin + ci];
            }
          }
      }
}

}  // namespace

// Floats of the packed form of a transposed-convolution kernel [3,3,3,Cout,Cin] (Cout 8 or 16, Cin % 16 == 0, Cin <= 64),
// including 4 trailing zeros.
extern "C" int atvs_deconv_up_pack_size(int Cin, int Cout, long* packed_floats) {
  if (!packed_floats) return ATVS_ERR_NULL;
  if (!up_shape_ok(Cin, Cout)) return ATVS_ERR_SHAPE;
  *packed_floats = (long)(Cin / 16) * up_pairs(Cout) * 256 + 4;
  return ATVS_OK;
}

// HOST function.  w: TF kernel of tf.layers.conv3d_transpose, [3,3,3,Cout,Cin].  packed[chunk][block][lane][s]: block =
// the non-zero (offset, tile) pairs in K-loop order; lane = q*16 + row (Cout 8: row = px*8 + co of tile (pz,py); Cout 16:
// row = co of tile (pz,py,px)); value = w[kd][kh][kw][co][chunk*16 + 4q + s] with k per axis from (parity, offset):
// (0,0) -> 0, (0,-1) -> 2, (1,0) -> 1, (1,-1) -> structural zero.
extern "C" int atvs_deconv_up_pack(const float* w, int Cin, int Cout, float* packed) {
  if (!w || !packed) return ATVS_ERR_NULL;
  long pf;
  int rc = atvs_deconv_up_pack_size(Cin, Cout, &pf);
  if (rc) return rc;
  for (long i = 0; i < pf; ++i) packed[i] = 0.f;
  if (Cout == 8) pack_up<8>(w, Cin, packed);
  else pack_up<16>(w, Cin, packed);
  return ATVS_OK;
}

// workgroups PER SAMPLE of a launch over `groups` independent samples (rows of the statistics buffer = groups * this):
// one workgroup per CU in all, shared out among the samples, a multiple of 8 each
extern "C" long atvs_deconv_up_grid(int D, int H, int W, int Cout, int groups) {
  if (groups < 1) groups = 1;
  if (Cout != 8 && Cout != 16) return 0;
  long nt = up_ntiles(D, H, W, Cout);
  long share = 256 / groups / 8 * 8;
  if (share < 8) share = 8;
  long g = nt < share ? nt : share;
  return (g + 7) / 8 * 8;
}

// y (2D, 2H, 2W, ldy)[..., y_coff : y_coff + Cout] = conv3d_transpose(x (D,H,W,Cin), w, stride 2, SAME) (+ ReLU),
// `groups` independent samples on the leading axis of x / y.  stats_partial: groups * atvs_deconv_up_grid rows of [2][16]
// doubles (partial sums / sums of squares per channel of the stored values), or NULL.
extern "C" int atvs_deconv_up_f32(const float* x, const float* packed_w, float* y, double* stats_partial, int groups, int D,
                                  int H, int W, int Cin, int Cout, int ldy, int y_coff, int relu, atvs_stream_t stream) {
  if (!x || !packed_w || !y) return ATVS_ERR_NULL;
  if (!up_shape_ok(Cin, Cout) || groups <= 0 || D <= 0 || H <= 0 || W <= 0) return ATVS_ERR_SHAPE;
  if (y_coff < 0 || y_coff + Cout > ldy || (ldy % 4) || (y_coff % 4)) return ATVS_ERR_SHAPE;
  if ((double)D * H * W * Cin >= 2147483648.0) return ATVS_ERR_SHAPE;            // 31-bit halo-relative element offsets
  if (8.0 * D * H * W * ldy * 4.0 >= 4294967296.0) return ATVS_ERR_SHAPE;        // 32-bit output BYTE offsets (buffer stores)
  UpArgs a;
  a.x = x; a.wp = packed_w; a.zeros = packed_w + (size_t)(Cin / 16) * up_pairs(Cout) * 256;
  a.y = y; a.stats = stats_partial;
  a.Di = D; a.Hi = H; a.Wi = W; a.Cin = Cin; a.ldy = ldy; a.ycoff = y_coff; a.nchunk = Cin / 16;
  const int ty = up_ty(Cout);
  a.tiles_y = (H + ty - 1) / ty; a.tiles_x = (W + UP_TX - 1) / UP_TX;
  a.ntiles = (int)up_ntiles(D, H, W, Cout);
  const long blocks = atvs_deconv_up_grid(D, H, W, Cout, groups);
  a.wg = (int)blocks;
  a.gx = (long)D * H * W * Cin; a.gy = 8L * D * H * W * ldy;
  if (blocks * groups > 0x7fffffffL) return ATVS_ERR_SHAPE;
  int rc = (Cout == 8) ? launch_up_cout<8>(a, blocks * groups, relu, stream) : launch_up_cout<16>(a, blocks * groups, relu, stream);
  if (rc) return rc;
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}
