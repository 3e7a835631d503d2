// 3x3x3 SAME stride-1 convolution to SIXTEEN output channels from 8, 16 or 32 input channels, one wavefront per SIMD
// (gfx950).
//
// These are the half-resolution layers of the stacked U-Nets / the refinement net (conv_b*_1_1, global_refine_3dconv1_1:
// 16 -> 16 channels on a (D/2, h/2, w/2) volume; cnn_wrapper/atvsnet.py StackedUNet / CostVolRefineNet, layer code
// /root/reference/cnn_wrapper/network.py:165-215) and the shared | unique convolution of the AANet modules (8 -> 8 + 8
// channels at full resolution, network.py:282-351 issued as one 8 -> 16 convolution): 16 output channels fill the 16 rows
// of an MFMA tile exactly (every issued MFMA is useful; 27 of 28 with 8 input channels, two taps per K step), and on the
// generic LDS-tiled kernel (conv_tiled.hip) they were its largest blocks of time at 55-57 % of the fp32 MFMA peak (MFMA
// pipe busy 58 %, 2.75 other VALU instructions per MFMA: run-time tap tables, rotating register copies, a halo burst in
// front of the K loop, two workgroups per CU on half the register file each).
//
// Built like conv_xw.hip / deconv_up.hip: ONE workgroup of 4 wavefronts per CU with the whole register file; tile =
// 4(z) x 8(y) x 16(x) output voxels, wavefront w owns plane z0 + w (8 accumulator tiles); the K loop is fully unrolled
// (27 taps x 4 MFMAs x 8 rows per 16-channel chunk), every LDS read is one of three swizzled base registers (the x
// displacement) + an immediate (the row pitch of the image is a multiple of 512 bytes); the fragments and the weights of
// tap j + 1 are requested in front of the MFMAs of tap j, the next tile's halo one slot per tap behind them; the packed
// weights of every chunk stay in LDS for the whole launch.
#include <type_traits>
#include <utility>

#include "conv_common.h"

namespace {

constexpr int C16_TZ = 4, C16_TY = 8, C16_TX = 16;
constexpr int C16_HZ = C16_TZ + 2, C16_HY = C16_TY + 2, C16_HX = C16_TX + 2;
constexpr int C16_TAPS = 27;

// C4 = float4 channel groups per voxel of a chunk: 4 (16-channel chunks; Cin 16 / 32) or 2 (Cin 8).
//   C4 == 4: one tap per K step, lane group q = channel group; the image is bank-swizzled (bit 5 ^= bit 8) and its row
//            pitch (24 voxels = 1536 B) a multiple of 512 B, so (kd, kh, row) displacements are immediates.
//   C4 == 2: two taps per K step, tap = 2 j + (q >> 1), channel group q & 1; 32-byte voxels need no swizzle (the 16-lane
//            groups of a 128-bit read already fall into distinct bank slots), every displacement is a plain addend.
template <int C4>
struct C16 {
  static_assert(C4 == 4 || C4 == 2, "8- or 16-channel chunks");
  static constexpr int VB = C4 * 16;                           // bytes per voxel in LDS
  static constexpr int PITCHV = (C4 == 4) ? 24 : C16_HX;       // voxels per image row
  static constexpr int ROWB = PITCHV * VB;
  static constexpr int IMG = C16_HZ * C16_HY * ROWB;           // 92,160 / 34,560 bytes
  static constexpr int SLOTS = C16_HZ * C16_HY * C16_HX * C4;  // 16-byte halo slots
  static constexpr int MAXS = (SLOTS + 255) / 256;             // 17 / 9 per thread
  static constexpr int JC = (C4 == 4) ? C16_TAPS : (C16_TAPS + 1) / 2;      // K steps per chunk
  static constexpr int WCH = JC * 1024;                        // bytes of packed weights per chunk
  static constexpr int CC = 4 * C4;                            // channels per chunk
  static_assert(MAXS <= JC, "one halo slot per K step");
  static_assert(((2 * C16_HY + 2) + C16_TY) * ROWB + 2 * VB < 65536, "ds_read immediate offset");
  // byte displacement of tap t from the halo voxel (wave, 0, r): (kd, kh) rows (+ kw voxels when it is not a base register)
  static constexpr int disp(int t, bool with_kw) {
    return ((t / 9) * C16_HY + (t / 3) % 3) * ROWB + (with_kw ? (t % 3) * VB : 0);
  }
};

struct C16Args {
  const float* x;
  const float* wp;       // packed weights (atvs_conv_c16_pack) + 4 trailing zeros
  const float* zeros;    // those 16 zero bytes: source of the zero padding
  const float* bias;     // 16 floats or nullptr
  float* y;
  double* stats;
  int Di, Hi, Wi, Cin;
  int ldy, ycoff;
  int nchunk;
  int tiles_y, tiles_x, ntiles;
  int wg;                // workgroups per sample (gridDim.x = groups * wg)
  long gx, gy;           // elements per sample of x / y
};

__device__ __forceinline__ int c16_swz(int a) { return a ^ (((a >> 8) & 1) << 5); }

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int N>
using IC = std::integral_constant<int, N>;
template <class F, int... I>
__device__ __forceinline__ void c16_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(IC<I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void c16_static_for(F&& f) {
  c16_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// NT = 16-channel output tiles: 1 (16 channels; packed weights of every chunk resident in LDS) or 2 (32 channels: the
// weights no longer fit next to the image and stream from L2, requested C16_LOOK steps ahead like conv_xw.hip's).
constexpr int C16_LOOK = 4;

template <int C4, int NT, bool RELU>
__global__ __launch_bounds__(256, 1) void conv_c16_kernel(C16Args p) {
  using K = C16<C4>;
  constexpr int TY = C16_TY, HY = C16_HY, MAXS = K::MAXS;
  constexpr bool WLDS = (NT == 1);
  static_assert(NT == 1 || C4 == 4, "32 output channels: 16-channel chunks only");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;

  // packed weights of every chunk -> LDS, once (visible after the first stage's barriers)
  if (WLDS) {
    const float4* src = reinterpret_cast<const float4*>(p.wp);
    float4* dst = reinterpret_cast<float4*>(smem + K::IMG);
    for (int i = tid; i < p.nchunk * (K::WCH / 16); i += 256) dst[i] = src[i];
  }

  // LDS read bases: this lane's fragment at halo voxel (wave, 0, r + kw) = tap (0, 0, kw) of row 0 of the wavefront's
  // plane.  C4 == 4: one swizzled base per kw, every (kd, kh, row) an immediate from there; C4 == 2: one base (kw = 0)
  int fbase[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int a = ((wave * HY) * K::PITCHV + r + kw) * K::VB + ((C4 == 4) ? q : (q & 1)) * 16;
    fbase[kw] = (C4 == 4) ? c16_swz(a) : a;
  }
  const int wbase = K::IMG + lane * 16;

  // per-slot constants of this thread: global element offset from the halo origin, swizzled LDS byte address and the
  // packed halo coordinate (zz | yy<<8 | xx<<16, each byte with its top bit set) for the bounds test
  int goff[MAXS], laddr[MAXS];
  unsigned pg[MAXS];
#pragma unroll
  for (int i = 0; i < MAXS; ++i) {
    int s = tid + i * 256;
    const bool live = s < K::SLOTS;
    s = min(s, K::SLOTS - 1);
    const int c4 = s % C4, v = s / C4;
    const int xx = v % C16_HX, v2 = v / C16_HX;
    const int yy = v2 % HY, zz = v2 / HY;
    goff[i] = ((zz * p.Hi + yy) * p.Wi + xx) * p.Cin + c4 * 4;
    const int la = ((zz * HY + yy) * K::PITCHV + xx) * K::VB + c4 * 16;
    laddr[i] = (C4 == 4) ? c16_swz(la) : la;
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
    *x0 = bx * C16_TX;
    *y0 = (rest % p.tiles_y) * TY;
    *z0 = (rest / p.tiles_y) * C16_TZ;
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
    T.xb = xg + ch * K::CC;
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

  // moments of this lane's channels 4q + {0,1 | 2,3}
  f32x2 ssum2[NT][2], ssq2[NT][2];
#pragma unroll
  for (int n = 0; n < NT; ++n) ssum2[n][0] = ssum2[n][1] = ssq2[n][0] = ssq2[n][1] = (f32x2){0.f, 0.f};
  f32x4 acc[TY][NT];
  float4 bv[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) bv[n] = p.bias ? ld4(p.bias + n * 16 + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  const float4* __restrict__ wpg = reinterpret_cast<const float4*>(p.wp);
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
        for (int n = 0; n < NT; ++n) acc[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // streamed weights: those of the first steps are on their way while the image is written
    const float4* wch = wpg + (size_t)ch * K::JC * NT * 64 + lane;
    float4 Wg[WLDS ? 1 : K::JC][NT];
    if (!WLDS) {
#pragma unroll
      for (int jj = 0; jj < C16_LOOK; ++jj)
#pragma unroll
        for (int n = 0; n < NT; ++n) Wg[jj][n] = wch[(jj * NT + n) * 64];
    }
    __syncthreads();                       // every wavefront is done reading the previous stage's image
#pragma unroll
    for (int i = 0; i < MAXS; ++i)
      if (i < MAXS - 1 || tid + i * 256 < K::SLOTS) *reinterpret_cast<float4*>(smem + laddr[i]) = pf[i];
    __syncthreads();

    const PfTile T = pf_tile(min(stage + 1, nstage - 1));      // last stage: harmless re-read of its own halo
    const int wb = wbase + ch * K::WCH;

    // ---- K loop, fully unrolled: 27 steps of one tap (C4 == 4) or 14 steps of two taps (C4 == 2; the 28th tap has zero
    // weights and re-reads the 27th's fragment)
    float4 B[2][TY], Wt[2];
    auto request = [&](auto JT) __attribute__((always_inline)) {
      constexpr int j = decltype(JT)::value;
      if constexpr (C4 == 4) {
        constexpr int disp = K::disp(j, false);
#pragma unroll
        for (int t = 0; t < TY; ++t)
          B[j & 1][t] = *reinterpret_cast<const float4*>(smem + fbase[j % 3] + (disp + t * K::ROWB));
      } else {
        constexpr int t0 = 2 * j, t1 = (2 * j + 1 < C16_TAPS) ? 2 * j + 1 : C16_TAPS - 1;
        const int a = fbase[0] + ((q >> 1) ? K::disp(t1, true) : K::disp(t0, true));
#pragma unroll
        for (int t = 0; t < TY; ++t) B[j & 1][t] = *reinterpret_cast<const float4*>(smem + a + t * K::ROWB);
      }
      if constexpr (WLDS) Wt[j & 1] = *reinterpret_cast<const float4*>(smem + wb + j * 1024);
    };
    request(IC<0>{});
    asm volatile("" ::: "memory");
    c16_static_for<K::JC>([&](auto JT) __attribute__((always_inline)) {
      constexpr int j = decltype(JT)::value;
      if constexpr (!WLDS && j + C16_LOOK < K::JC) {
#pragma unroll
        for (int n = 0; n < NT; ++n) Wg[(WLDS ? 0 : j + C16_LOOK)][n] = wch[((j + C16_LOOK) * NT + n) * 64];
      }
      if constexpr (j + 1 < K::JC) request(IC<j + 1>{});
      if constexpr (j < MAXS) pf_slot(T, j);
      // compiler barrier (keeps the requests from sinking to their uses) + scheduling barrier (keeps them in front of
      // the MFMAs that cover their latency)
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const float4 wv = WLDS ? Wt[j & 1] : Wg[WLDS ? 0 : j][n];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int t = 0; t < TY; ++t)
            acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(wv, s), f4get(B[j & 1][t], s), acc[t][n], 0, 0, 0);
      }
    });
    if (ch != p.nchunk - 1) continue;

    // ---- epilogue: this lane holds channels 4q..4q+3 of voxel (z0 + wave, y0 + t, x0 + r); 32-bit element offsets inside
    // the sample, branch-free buffer stores (rows outside the volume get an out-of-range offset and are dropped)
    int tz0, ty0, tx0;
    tile_origin(k, &tz0, &ty0, &tx0);
    const int zo = tz0 + wave, xo = tx0 + r;
    const bool evox_ok = zo < p.Di && xo < p.Wi;
    const unsigned erow = (unsigned)p.Wi * p.ldy;
    const unsigned eo = (((unsigned)zo * p.Hi + ty0) * p.Wi + xo) * p.ldy + p.ycoff + q * 4;
    const unsigned vo_ok = evox_ok ? eo * 4u : ybytes;
    c16_static_for<TY>([&](auto TT) __attribute__((always_inline)) {
      constexpr int t = decltype(TT)::value;
      const bool row_ok = ty0 + t < p.Hi;            // uniform
      const bool ok = evox_ok && row_ok;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        float a0 = acc[t][n][0] + bv[n].x, a1 = acc[t][n][1] + bv[n].y, a2 = acc[t][n][2] + bv[n].z, a3 = acc[t][n][3] + bv[n].w;
        if (RELU) {                                  // NaN passes through, as in tf.nn.relu
          a0 = (a0 < 0.f) ? 0.f : a0; a1 = (a1 < 0.f) ? 0.f : a1;
          a2 = (a2 < 0.f) ? 0.f : a2; a3 = (a3 < 0.f) ? 0.f : a3;
        }
        const u32x4 bits = {__builtin_bit_cast(unsigned, a0), __builtin_bit_cast(unsigned, a1),
                            __builtin_bit_cast(unsigned, a2), __builtin_bit_cast(unsigned, a3)};
        __builtin_amdgcn_raw_buffer_store_b128(bits, yrsrc, row_ok ? vo_ok : ybytes, (t * erow + n * 16) * 4u, 0);
        f32x2 lo = {ok ? a0 : 0.f, ok ? a1 : 0.f}, hi = {ok ? a2 : 0.f, ok ? a3 : 0.f};
        ssum2[n][0] += lo;
        ssum2[n][1] += hi;
        ssq2[n][0] = __builtin_elementwise_fma(lo, lo, ssq2[n][0]);
        ssq2[n][1] = __builtin_elementwise_fma(hi, hi, ssq2[n][1]);
      }
    });
  }

  // ---- per-workgroup partial moments (sum, sum of squares) per output channel -> row blockIdx of stats: [2][16 NT] doubles
  if (p.stats) {
    constexpr int CP = 16 * NT;
    __syncthreads();
    double* s_red = reinterpret_cast<double*>(smem);   // [4 waves][2][CP]
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        double a = (double)ssum2[n][kk >> 1][kk & 1], bq = (double)ssq2[n][kk >> 1][kk & 1];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          a += __shfl_xor(a, o);
          bq += __shfl_xor(bq, o);
        }
        if (r == 0) {
          s_red[(wave * 2 + 0) * CP + n * 16 + q * 4 + kk] = a;
          s_red[(wave * 2 + 1) * CP + n * 16 + q * 4 + kk] = bq;
        }
      }
    __syncthreads();
    if (tid < 2 * CP) {
      const int which = tid / CP, col = tid % CP;
      p.stats[((size_t)blockIdx.x * 2 + which) * CP + col] =
          (s_red[(0 * 2 + which) * CP + col] + s_red[(1 * 2 + which) * CP + col]) +
          (s_red[(2 * 2 + which) * CP + col] + s_red[(3 * 2 + which) * CP + col]);
    }
  }
}

long c16_ntiles(int D, int H, int W) {
  return (long)((D + C16_TZ - 1) / C16_TZ) * ((H + C16_TY - 1) / C16_TY) * ((W + C16_TX - 1) / C16_TX);
}

// 16 channels out: the weights of every chunk + the image must fit 160 KB of LDS: Cin 8 (one 8-channel chunk), 16 or 32
// (16-channel chunks); 32 channels out: streamed weights, Cin a multiple of 16 up to 64
bool c16_shape_ok(int Cin, int Cout) {
  if (Cout == 16) return Cin == 8 || Cin == 16 || Cin == 32;
  return Cout == 32 && Cin % 16 == 0 && Cin > 0 && Cin <= 64;
}
int c16_c4(int Cin) { return Cin == 8 ? 2 : 4; }
long c16_packed_floats(int Cin, int Cout) {
  return Cin == 8 ? (long)C16<2>::JC * 256 : (long)(Cin / 16) * C16<4>::JC * (Cout / 16) * 256;
}

template <int C4, int NT, bool RELU>
int launch_c16(const C16Args& a, long grid, atvs_stream_t stream) {
  const size_t lds = (size_t)C16<C4>::IMG + (NT == 1 ? (size_t)a.nchunk * C16<C4>::WCH : 0);
  // the attribute is per device: one flag per device ordinal of this process (and per instantiation)
  static AtvsAttrOnce lds_once;                   // per kernel instantiation (this function is a template / has one kernel)
  if (const int rc_ = atvs_set_max_lds_once(lds_once, reinterpret_cast<const void*>(conv_c16_kernel<C4, NT, RELU>), 160 * 1024)) return rc_;
  hipLaunchKernelGGL((conv_c16_kernel<C4, NT, RELU>), dim3((unsigned)grid), dim3(256), lds, as_stream(stream), a);
  return ATVS_OK;
}

}  // namespace

// Floats of the packed form of a kernel [3,3,3,Cin,Cout] (Cout 16: Cin 8, 16 or 32; Cout 32: Cin 16, 32, 48 or 64),
// including 4 trailing zeros.
extern "C" int atvs_conv_c16_pack_size(int Cin, int Cout, long* packed_floats) {
  if (!packed_floats) return ATVS_ERR_NULL;
  if (!c16_shape_ok(Cin, Cout)) return ATVS_ERR_SHAPE;
  *packed_floats = c16_packed_floats(Cin, Cout) + 4;
  return ATVS_OK;
}

// HOST function.  w: TF kernel [3,3,3,Cin,Cout].  Cin % 16 == 0: packed[chunk][tap][tile n][lane = q*16 + co][s] =
// w[tap][chunk*16 + 4q + s][n*16 + co];  Cin 8: packed[step][lane = q*16 + co][s] = w[tap = 2 step + (q>>1)][(q&1)*4 + s][co]
// (zero for tap 27).
extern "C" int atvs_conv_c16_pack(const float* w, int Cin, int Cout, float* packed) {
  if (!w || !packed) return ATVS_ERR_NULL;
  long pf;
  int rc = atvs_conv_c16_pack_size(Cin, Cout, &pf);
  if (rc) return rc;
  for (long i = 0; i < pf; ++i) packed[i] = 0.f;
  if (Cin == 8) {
    for (int j = 0; j < C16<2>::JC; ++j)
      for (int q = 0; q < 4; ++q) {
        const int tap = 2 * j + (q >> 1);
        if (tap >= C16_TAPS) continue;
        for (int co = 0; co < 16; ++co)
          for (int s = 0; s < 4; ++s)
            packed[(((size_t)j * 64) + q * 16 + co) * 4 + s] = w[((size_t)tap * 8 + (q & 1) * 4 + s) * 16 + co];
      }
    return ATVS_OK;
  }
  const int NT = Cout / 16;
  for (int ch = 0; ch < Cin / 16; ++ch)
    for (int j = 0; j < C16_TAPS; ++j)
      for (int n = 0; n < NT; ++n)
        for (int q = 0; q < 4; ++q)
          for (int co = 0; co < 16; ++co)
            for (int s = 0; s < 4; ++s)
              packed[(((((size_t)ch * C16_TAPS + j) * NT + n) * 64) + q * 16 + co) * 4 + s] =
                  w[((size_t)j * Cin + ch * 16 + 4 * q + s) * Cout + n * 16 + co];
  return ATVS_OK;
}

// workgroups PER SAMPLE of a launch over `groups` independent samples (rows of the statistics buffer = groups * this):
// one workgroup per CU in all, shared out among the samples, a multiple of 8 each
extern "C" long atvs_conv_c16_grid(int D, int H, int W, int groups) {
  if (groups < 1) groups = 1;
  long nt = c16_ntiles(D, H, W);
  long share = 256 / groups / 8 * 8;
  if (share < 8) share = 8;
  long g = nt < share ? nt : share;
  return (g + 7) / 8 * 8;
}

// y (D,H,W,ldy)[..., y_coff : y_coff + Cout] = conv3d(x (D,H,W,Cin), w, stride 1, SAME) (+ bias, ReLU), `groups`
// independent samples on the leading axis of x / y.  stats_partial: groups * atvs_conv_c16_grid rows of [2][Cout] doubles
// (partial sums / sums of squares per channel of the stored values), or NULL.
extern "C" int atvs_conv_c16_f32(const float* x, const float* packed_w, const float* bias, float* y, double* stats_partial,
                                 int groups, int D, int H, int W, int Cin, int Cout, int ldy, int y_coff, int relu,
                                 atvs_stream_t stream) {
  if (!x || !packed_w || !y) return ATVS_ERR_NULL;
  if (!c16_shape_ok(Cin, Cout) || groups <= 0 || D <= 0 || H <= 0 || W <= 0) return ATVS_ERR_SHAPE;
  if (y_coff < 0 || y_coff + Cout > ldy || (ldy % 4) || (y_coff % 4)) return ATVS_ERR_SHAPE;
  if ((double)D * H * W * Cin >= 2147483648.0) return ATVS_ERR_SHAPE;            // 31-bit halo-relative element offsets
  if ((double)D * H * W * ldy * 4.0 >= 4294967296.0) return ATVS_ERR_SHAPE;      // 32-bit output BYTE offsets (buffer stores)
  C16Args a;
  a.x = x; a.wp = packed_w; a.zeros = packed_w + c16_packed_floats(Cin, Cout);
  a.bias = bias; a.y = y; a.stats = stats_partial;
  a.Di = D; a.Hi = H; a.Wi = W; a.Cin = Cin; a.ldy = ldy; a.ycoff = y_coff; a.nchunk = (Cin == 8) ? 1 : Cin / 16;
  a.tiles_y = (H + C16_TY - 1) / C16_TY; a.tiles_x = (W + C16_TX - 1) / C16_TX;
  a.ntiles = (int)c16_ntiles(D, H, W);
  const long blocks = atvs_conv_c16_grid(D, H, W, groups);
  a.wg = (int)blocks;
  a.gx = (long)D * H * W * Cin; a.gy = (long)D * H * W * ldy;
  if (blocks * groups > 0x7fffffffL) return ATVS_ERR_SHAPE;
  const long grid = blocks * groups;
  int rc;
  if (Cout == 32) rc = relu ? launch_c16<4, 2, true>(a, grid, stream) : launch_c16<4, 2, false>(a, grid, stream);
  else if (c16_c4(Cin) == 2) rc = relu ? launch_c16<2, 1, true>(a, grid, stream) : launch_c16<2, 1, false>(a, grid, stream);
  else rc = relu ? launch_c16<4, 1, true>(a, grid, stream) : launch_c16<4, 1, false>(a, grid, stream);
  if (rc) return rc;
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}
