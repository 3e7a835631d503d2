// 3x3x3 stride-2 SAME transposed convolution to 8 or 16 output channels on the 16-bit matrix cores with SPLIT operands (gfx950):
// deconv_up.hip's layers (conv_b*_6_0, global_refine_3dconv6_0: 16 -> 8, full-resolution output; conv_b*_5_0,
// global_refine_3dconv5_0: 32 -> 16; /root/reference/cnn_wrapper/network.py:510-550) with every fp32 operand split into TWO fp16
// pieces, three products, fp32 accumulation (conv_xb.hip has the arithmetic: x = h0 + h1 / 2048, the cross terms in an
// accumulator of their own that is scaled once; atvs_split2_f16).  (Rounds 3 / 4 first half: three bf16 pieces, six products,
// one accumulator set of 32 tiles per wavefront -- a second set did not fit beside it; the tile is now half as tall.)
//
// Form of deconv_up.hip: per input voxel M = 8 parity classes x Cout rows, K = 8 offsets x Cin; here a K = 32 instruction
// covers the TWO x offsets of an (oz, oy) pair x 16 channels (lane group q: ox = q >> 1, channels 8 (q & 1) ..), so
//   Cout =  8: tile m = (pz, py), rows = (px, channel): 9 (tile, oz, oy) steps of three 16-cycle MFMAs;
//   Cout = 16: tile m = (pz, py, px), rows = channel: 18 steps (the odd-px tiles use only the ox = 0 half of K).
// Steps are issued in GROUPS of up to four tiles that share (oz, oy), hence the input fragments; a group runs in two phases
// (input piece h0 with both weight pieces, then h1 with g0), fragments requested one phase ahead, the group's weights (from LDS,
// resident for the launch) one group ahead.  Tile 4(z) x TY(y) x 16(x) input voxels, TY = 4 / 2 (16 output tiles x two accumulator
// sets per wavefront); one-sided halo 5 x (TY+1) x 17; two piece images of 32-byte voxels (no swizzle needed: 16 lanes read 16
// consecutive voxels, the channel half shifts by 16 bytes); branch-free buffer stores as in deconv_up.hip.
// At most 256 registers per wavefront and (for the product's shapes) under 80 KB of LDS: TWO workgroups share a CU, one's
// epilogue stores and staging run beside the other's K loop.  Scalar fp32 arithmetic only (-fno-slp-vectorize: wavefronts of
// the two workgroups share SIMDs).
#include <cstring>
#include <type_traits>
#include <utility>

#include "conv_common.h"


// Development build (-DATVS_UB_DEBUG): per-wavefront tick counts of the phases (tools_dev/phase_ub.py)
#ifdef ATVS_UB_DEBUG
__device__ unsigned long long atvs_dbg_ub[4096 * 8];
extern "C" int atvs_debug_read_ub(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(atvs_dbg_ub), sizeof(atvs_dbg_ub));
}
#define UDBG(i) { unsigned long long t_ = clock64(); dbg_acc[i] += t_ - dbg_t; dbg_t = t_; }
#else
#define UDBG(i)
#endif

namespace {

constexpr int UB_TZ = 4, UB_TX = 16;
constexpr int UB_NP = 2;                               // operand pieces
#ifndef ATVS_UB_WGS8
#define ATVS_UB_WGS8 1
#endif
#ifndef ATVS_UB_WGS16
#define ATVS_UB_WGS16 2
#endif
#define UB_WGS_PER_CU(COUT) ((COUT) == 8 ? ATVS_UB_WGS8 : ATVS_UB_WGS16)
#ifndef ATVS_UB_GRID8
#define ATVS_UB_GRID8 ATVS_UB_WGS8
#endif
#define UB_GRID_PER_CU(COUT) ((COUT) == 8 ? ATVS_UB_GRID8 : ATVS_UB_WGS16)
constexpr float UB_RS = 2048.f, UB_IRS = 1.f / 2048.f; // scale of the residual piece and its inverse
constexpr int UB_HZ = UB_TZ + 1, UB_HX = UB_TX + 1;
constexpr int UB_VB = 32;
constexpr int UB_ROWB = UB_HX * UB_VB;                 // 544

template <int COUT>
struct UpB {
  static_assert(COUT == 8 || COUT == 16, "built for 8 and 16 output channels");
  static constexpr int NT = (COUT == 8) ? 4 : 8;
  static constexpr int TY = 16 / NT;                   // 16 output tiles per wavefront x (main, cross) accumulators
  static constexpr int HY = TY + 1;
  static constexpr int IMG = UB_HZ * HY * UB_ROWB;     // bytes of one piece image
  static constexpr int SLOTS = UB_HZ * HY * UB_HX * 4;
  static constexpr int MAXS = (SLOTS + 255) / 256;     // 7 / 4
  static constexpr int NG = (COUT == 8) ? 4 : 5;       // groups
  // group g: its (oz, oy) pair o2 = oz * 2 + oy, its tiles
  static constexpr int o2_of(int g) { return (COUT == 8) ? g : (g < 2 ? 0 : g - 1); }
  static constexpr int ntg(int g) { return (COUT == 8) ? (g == 0 ? 4 : g == 3 ? 1 : 2) : (g == 4 ? 2 : 4); }
  static constexpr int tile(int g, int i) {
    if (COUT == 8) return g == 0 ? i : g == 1 ? 2 * i : g == 2 ? i : 0;       // o2 = 1 (oy = -1): py = 0; o2 = 2 (oz = -1): pz = 0
    // Cout 16, tile = pz * 4 + py * 2 + px
    return g == 0 ? i : g == 1 ? 4 + i : g == 2 ? (i < 2 ? i : 2 + i) : g == 3 ? i : i;      // g2: {0,1,4,5}; g3: {0,1,2,3}; g4: {0,1}
  }
  static constexpr int first_step(int g) {
    int n = 0;
    for (int k = 0; k < g; ++k) n += ntg(k);
    return n;
  }
  static constexpr int NSTEP = first_step(NG);
  static constexpr int WCH = NSTEP * UB_NP * 1024;     // bytes of packed weights per chunk (global memory: 1 KB per step and piece)
  // In LDS the steps of the odd-px tiles of the 16-channel form keep only lanes 0..31: lane groups q = 2, 3 carry the x offset
  // whose tap is a structural zero for px = 1 (kw < 0 in pack_upb) -- those lanes read one shared line of zeros instead.
  // 27 KB instead of 36 KB per chunk: the two chunks of the 32 -> 16 layers (conv_b*_5_0) stay RESIDENT beside the images with
  // two workgroups per CU (70 KB each); streamed (round 4) the copy of a chunk was 41 % of a stage (tools_dev/phase_ub.py).
  static constexpr int tile_of_step(int s) {
    int g = 0;
    while (g + 1 < NG && first_step(g + 1) <= s) ++g;
    return tile(g, s - first_step(g));
  }
  static constexpr bool half_step(int s) { return COUT == 16 && (tile_of_step(s) & 1) != 0; }
  static constexpr int step_bytes(int s) { return half_step(s) ? 512 : 1024; }      // per piece
  static constexpr int woff(int s, int piece) {        // LDS byte offset of (step, piece) inside a chunk
    int o = 0;
    for (int k = 0; k < s; ++k) o += UB_NP * step_bytes(k);
    return o + piece * step_bytes(s);
  }
  static constexpr int WLDS = woff(NSTEP, 0);          // bytes of a chunk's weights in LDS
  // the same as closed forms for run-time step indices (the copy loop): the half steps of the 16-channel form are the odd ones
  __host__ __device__ static constexpr int lanes_rt(int s) { return (COUT == 16 && (s & 1)) ? 32 : 64; }
  __host__ __device__ static constexpr int woff_rt(int s, int piece) {
    return COUT == 8 ? (s * UB_NP + piece) * 1024 : (s >> 1) * 3072 + ((s & 1) ? 2048 + piece * 512 : piece * 1024);
  }
  static constexpr bool closed_forms_ok() {
    for (int s = 0; s < NSTEP; ++s)
      for (int pc = 0; pc < UB_NP; ++pc)
        if (woff_rt(s, pc) != woff(s, pc) || lanes_rt(s) * 16 != step_bytes(s)) return false;
    return true;
  }
};
static_assert(UpB<8>::NSTEP == 9 && UpB<16>::NSTEP == 18, "(tile, oz, oy) steps");
static_assert(UpB<8>::closed_forms_ok() && UpB<16>::closed_forms_ok(), "closed forms of the LDS weight layout");

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct UpBArgs {
  const float* x;
  // summing form (deconv_up_b_sum_kernel): the input is the U-Net's skip SUM, formed while the halo is staged (the bn_add pass it replaces, norm.hip):
  //   x_in = t(x, pa, bit 0) + t(x2, pb, bit 1) [+ t(x3, pc, bit 2)],  t(v, par, relu) = par ? relu?((v - mean) * scale + beta) : v
  const float* x2;
  const float* x3;            // null: two terms
  const float* pa;            // (groups, 3, Cin) or null (that term is a finished tensor)
  const float* pb;
  const float* pc;
  int relu_mask;
  const unsigned char* wp;
  const float* zeros;
  float* y;
  double* stats;
  int Di, Hi, Wi, Cin;
  int ldy, ycoff;
  int nchunk;
  int tiles_y, tiles_x, ntiles;
  int wg;
  int relu;
  int stats_ld, stats_coff;   // doubles per half row of stats, first column of this launch's channels
  long gx, gy;
};

template <int N>
using IC = std::integral_constant<int, N>;
template <class F, int... I>
__device__ __forceinline__ void ub_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(IC<I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void ub_static_for(F&& f) {
  ub_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// STREAMW: Cin too large for all chunks to stay resident (the 64 -> 32 layers as two 16-channel launches): TWO weight buffers of
// one chunk each (compact layout, 2 x 27 KB: two workgroups per CU still fit), the NEXT stage's chunk fetched by LDS-DMA
// (global_load ... lds: no registers) while this stage's K loop runs.  (Round 4: one buffer, copied through registers between
// the barriers of every stage -- 3.8 k of an 8.3 k-cycle stage, tools_dev/phase_ub.py.)
// (The skip sum formed on load was a third form of this kernel in rounds 5 / 6; it is deconv_up_b_sum_kernel below: two roles.)
template <int COUT, bool STREAMW>
__global__ __launch_bounds__(256, UB_WGS_PER_CU(COUT)) void deconv_up_b_kernel(UpBArgs p) {
  using U = UpB<COUT>;
  constexpr int NT = U::NT, TY = U::TY, HY = U::HY, MAXS = U::MAXS, NG = U::NG, WCH = U::WCH;
  // the forms that run ONE workgroup per CU (one wavefront per SIMD) reserve the SIMD's whole register file like conv_c16b: no
  // wavefront of another kernel then runs beside their 16x16x32 MFMAs (such a neighbour with packed fp32
  // arithmetic computed wrong lane quarters; only reachable with co-residency switched on).  The two-per-CU forms cannot.
  if constexpr (UB_WGS_PER_CU(COUT) == 1) asm volatile("" ::: "v255", "a255");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
#ifdef ATVS_UB_DEBUG
  unsigned long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long dbg_t = clock64();
  const unsigned long long dbg_w0 = wall_clock64();
#endif

  const int lane16_ = lane * 16;
  // packed weights of every chunk -> LDS, once (visible after the first stage's barriers)
  // (step, piece) blocks of 64 lanes x 16 bytes in global memory -> 64 or 32 lanes in LDS (U::woff; for the 16-channel form the
  // half steps are the odd ones: U::woff_rt, checked against U::woff at compile time).  Wavefront w copies the blocks w, w + 4, ...
  auto copy_chunk = [&](int ch, int lds_chunk) __attribute__((always_inline)) {
    const float4* src = reinterpret_cast<const float4*>(p.wp + (size_t)ch * U::WCH);
    unsigned char* dstb = smem + UB_NP * U::IMG + lds_chunk * U::WLDS;
    for (int sp = wave; sp < U::NSTEP * UB_NP; sp += 4) {
      const int st = sp / UB_NP, pc = sp % UB_NP;
      if (lane < U::lanes_rt(st)) reinterpret_cast<float4*>(dstb + U::woff_rt(st, pc))[lane] = src[sp * 64 + lane];
    }
  };
  // the same copy by LDS-DMA into weight buffer `buf` (asynchronous: complete after the issuing wavefront's s_waitcnt vmcnt(0))
  // (a buffer descriptor over the packed weights: the per-lane part of the address is lane * 16 for every block, the rest scalar)
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p.wp), 0, p.nchunk * WCH, 0x00020000);
  auto dma_chunk = [&](int ch, int buf) __attribute__((always_inline)) {
    unsigned char* dstb = smem + UB_NP * U::IMG + buf * U::WLDS;
    for (int sp = wave; sp < U::NSTEP * UB_NP; sp += 4) {
      const int st = sp / UB_NP, pc = sp % UB_NP;
      if (lane < U::lanes_rt(st))
        // (WCH: a local constant -- with U::WCH inside this argument list the HOST pass of hipcc 7.2 emits no stub for the kernel,
        // without a diagnostic: the library then fails to load with an undefined symbol)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (void __attribute__((address_space(3)))*)(dstb + U::woff_rt(st, pc)), 16, lane16_,
                                                 ch * WCH + sp * 1024, 0, 0);
    }
  };
  if (!STREAMW)
    for (int ch = 0; ch < p.nchunk; ++ch) copy_chunk(ch, ch);
  else
    dma_chunk(0, 0);
  // the line of zeros the dropped lanes read (behind the weights)
  const int wzero = UB_NP * U::IMG + (STREAMW ? 2 : p.nchunk) * U::WLDS;
  if (tid == 0) *reinterpret_cast<float4*>(smem + wzero) = make_float4(0.f, 0.f, 0.f, 0.f);
  // this lane's fragment at halo voxel (wave, 0, r + 1 - ox), ox = q >> 1: offset (-1, -1, ox) of row 0 of the wavefront's plane
  const int fbase = ((wave * HY) * UB_HX + r + 1 - (q >> 1)) * UB_VB + (q & 1) * 16;
  // LDS address of this lane's weight fragment of (step, piece) of the chunk at wb: its own 16 bytes, or the zero line for the
  // dropped lanes of a half step
  const int lane16 = lane * 16;
  auto wfrag = [&](int wb, auto ST, auto PC) __attribute__((always_inline)) {
    constexpr int st = decltype(ST)::value, pc = decltype(PC)::value;
    if constexpr (U::half_step(st)) return (lane < 32) ? wb + U::woff(st, pc) + lane16 : wzero;
    else return wb + U::woff(st, pc) + lane16;
  };

  int goff[MAXS], laddr[MAXS];
  unsigned pg[MAXS];
#pragma unroll
  for (int i = 0; i < MAXS; ++i) {
    int s = tid + i * 256;
    const bool live = s < U::SLOTS;
    s = min(s, U::SLOTS - 1);
    const int c4 = s & 3, v = s >> 2;
    const int xx = v % UB_HX, v2 = v / UB_HX;
    const int yy = v2 % HY, zz = v2 / HY;
    goff[i] = ((zz * p.Hi + yy) * p.Wi + xx) * p.Cin + c4 * 4;
    laddr[i] = ((zz * HY + yy) * UB_HX + xx) * UB_VB + c4 * 8;
    pg[i] = 0x808080u | (unsigned)(live ? zz : 0x7f) | ((unsigned)yy << 8) | ((unsigned)xx << 16);
  }

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
    *x0 = bx * UB_TX;
    *y0 = (rest % p.tiles_y) * TY;
    *z0 = (rest / p.tiles_y) * UB_TZ;
  };
  struct PfTile {
    const float* xb;
    int org;
    unsigned lo, hi1;
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
    const unsigned t1 = pg[i] - T.lo;
    const unsigned t2 = T.hi1 + ~pg[i];
    const bool ok = ((t1 & t2) & 0x808080u) == 0x808080u;
    pf[i] = ld4(ok ? (T.xb + (T.org + goff[i])) : p.zeros);
  };
  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};       // scalar on purpose (two workgroups share a CU)
  f32x4 acc[TY][NT], accx[TY][NT];       // h0 g0 | (h0 g1 + h1 g0) * 2^11
  const unsigned ybytes = (unsigned)(p.gy * 4);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(yg, 0, ybytes, 0x00020000);

  if (nstage > 0) {
    const PfTile T0 = pf_tile(0);
#pragma unroll
    for (int i = 0; i < MAXS; ++i) pf_slot(T0, i);
  }

  UDBG(6)
  for (int stage = 0; stage < nstage; ++stage) {
    const int k = stage / p.nchunk, ch = stage - k * p.nchunk;
    UDBG(0)
    if (ch == 0) {
#pragma unroll
      for (int t = 0; t < TY; ++t)
#pragma unroll
        for (int m = 0; m < NT; ++m) acc[t][m] = accx[t][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();                       // every wavefront is done reading the previous stage's images (and weights)
    UDBG(1)
#pragma unroll
    for (int i = 0; i < MAXS; ++i)
      if (i < MAXS - 1 || tid + i * 256 < U::SLOTS) {
        uint2 p0, p1;
        atvs_split2_f16(pf[i].x, pf[i].y, UB_RS, &p0.x, &p1.x);
        atvs_split2_f16(pf[i].z, pf[i].w, UB_RS, &p0.y, &p1.y);
        *reinterpret_cast<uint2*>(smem + laddr[i]) = p0;
        *reinterpret_cast<uint2*>(smem + U::IMG + laddr[i]) = p1;
      }
    UDBG(2)
    if (STREAMW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wavefront's share of the stage's weights has landed
    __syncthreads();
    UDBG(3)

    const PfTile T = pf_tile(min(stage + 1, nstage - 1));      // last stage: harmless re-read of its own halo
    // this chunk's weights in LDS; streamed: buffer stage & 1, and the next stage's chunk on its way into the other one (free:
    // every wavefront has left the K loop that read it)
    const int wb = UB_NP * U::IMG + (STREAMW ? (stage & 1) : ch) * U::WLDS;
    if (STREAMW && stage + 1 < nstage) dma_chunk(ch + 1 == p.nchunk ? 0 : ch + 1, (stage + 1) & 1);

    // ---- K loop: groups of tiles sharing (oz, oy), two phases each: input piece h0 with the weight pieces g0 (main) and g1
    // (cross), then h1 with g0 (cross)
    // A0: the g0 weight fragments of a group, double-buffered (needed in both phases: the next group's are requested during
    // this group's first phase); A1: the g1 fragments, needed in the first phase only -- the next group's are requested during
    // the second phase into the same registers (a wavefront has 256: two workgroups share a CU)
    f16x8 Bq[2][TY], A0[2][4], A1[4];
    auto request_B = [&](auto PH) __attribute__((always_inline)) {
      constexpr int ph = decltype(PH)::value, g = ph / UB_NP, pc = ph % UB_NP, o2 = U::o2_of(g), oz = o2 >> 1, oy = o2 & 1;
      constexpr int disp = ((1 - oz) * HY + (1 - oy)) * UB_ROWB;
#pragma unroll
      for (int t = 0; t < TY; ++t) Bq[ph & 1][t] = *reinterpret_cast<const f16x8*>(smem + pc * U::IMG + fbase + (disp + t * UB_ROWB));
    };
    auto request_A0 = [&](auto GT) __attribute__((always_inline)) {
      constexpr int g = decltype(GT)::value;
      ub_static_for<U::ntg(g)>([&](auto IT) __attribute__((always_inline)) {
        constexpr int i = decltype(IT)::value;
        A0[g & 1][i] = *reinterpret_cast<const f16x8*>(smem + wfrag(wb, IC<U::first_step(g) + i>{}, IC<0>{}));
      });
    };
    auto request_A1 = [&](auto GT) __attribute__((always_inline)) {
      constexpr int g = decltype(GT)::value;
      ub_static_for<U::ntg(g)>([&](auto IT) __attribute__((always_inline)) {
        constexpr int i = decltype(IT)::value;
        A1[i] = *reinterpret_cast<const f16x8*>(smem + wfrag(wb, IC<U::first_step(g) + i>{}, IC<1>{}));
      });
    };
    request_A0(IC<0>{});
    request_A1(IC<0>{});
    request_B(IC<0>{});
    asm volatile("" ::: "memory");
    ub_static_for<UB_NP * NG>([&](auto PH) __attribute__((always_inline)) {
      constexpr int ph = decltype(PH)::value, g = ph / UB_NP, pc = ph % UB_NP;
      if constexpr (ph + 1 < UB_NP * NG) request_B(IC<ph + 1>{});
      if constexpr (pc == 0 && g + 1 < NG) request_A0(IC<g + 1>{});
      if constexpr (ph < MAXS) pf_slot(T, ph);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      ub_static_for<U::ntg(g)>([&](auto IT) __attribute__((always_inline)) {
        constexpr int i = decltype(IT)::value, m = U::tile(g, i);
        if constexpr (pc == 0) {
#pragma unroll
          for (int t = 0; t < TY; ++t) acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A0[g & 1][i], Bq[ph & 1][t], acc[t][m], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < TY; ++t) accx[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1[i], Bq[ph & 1][t], accx[t][m], 0, 0, 0);
        } else {
#pragma unroll
          for (int t = 0; t < TY; ++t) accx[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A0[g & 1][i], Bq[ph & 1][t], accx[t][m], 0, 0, 0);
        }
      });
      if constexpr (pc == 0 && g + 1 < NG) {     // behind the MFMAs that read A1: the next group's g1 fragments
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        request_A1(IC<g + 1>{});
      }
    });
    static_assert(UB_NP * NG >= MAXS, "every halo slot is requested inside the K loop");
    UDBG(4)
    if (ch != p.nchunk - 1) continue;

    // ---- epilogue (deconv_up.hip): this lane holds, of tile m,
    //   Cout  8: channels (q&1)*4..+3 of output voxel (2z+pz, 2(y0+t)+py, 2(x0+r) + (q>>1)), (pz,py) = m;
    //   Cout 16: channels 4q..+3      of output voxel (2z+pz, 2(y0+t)+py, 2(x0+r) + px),     (pz,py,px) = m.
    int tz0, ty0, tx0;
    tile_origin(k, &tz0, &ty0, &tx0);
    const int zo = tz0 + wave, xo = tx0 + r;
    const bool evox_ok = zo < p.Di && xo < p.Wi;
    const unsigned Hy = 2u * p.Hi, Wy = 2u * p.Wi;
    const unsigned erow = Wy * p.ldy;
    const unsigned eplane = Hy * erow;
    const unsigned lane_c = (COUT == 8) ? (unsigned)((q >> 1) * p.ldy + (q & 1) * 4) : (unsigned)(q * 4);
    const unsigned eo = (((unsigned)(2 * zo) * Hy + 2 * ty0) * Wy + 2 * xo) * p.ldy + p.ycoff + lane_c;
    const unsigned vo_ok = evox_ok ? eo * 4u : ybytes;
    ub_static_for<NT>([&](auto MT) __attribute__((always_inline)) {
      ub_static_for<TY>([&](auto TT) __attribute__((always_inline)) {
        constexpr int m = decltype(MT)::value, t = decltype(TT)::value;
        constexpr int pz = (COUT == 8) ? (m >> 1) : (m >> 2), py = (COUT == 8) ? (m & 1) : ((m >> 1) & 1);
        constexpr int px = (COUT == 8) ? 0 : (m & 1);
        // fmaf(cross, 2^-11, main): the product by a power of two is exact
        float a0 = __builtin_fmaf(accx[t][m][0], UB_IRS, acc[t][m][0]), a1 = __builtin_fmaf(accx[t][m][1], UB_IRS, acc[t][m][1]);
        float a2 = __builtin_fmaf(accx[t][m][2], UB_IRS, acc[t][m][2]), a3 = __builtin_fmaf(accx[t][m][3], UB_IRS, acc[t][m][3]);
        if (p.relu) {
          a0 = (a0 < 0.f) ? 0.f : a0; a1 = (a1 < 0.f) ? 0.f : a1;
          a2 = (a2 < 0.f) ? 0.f : a2; a3 = (a3 < 0.f) ? 0.f : a3;
        }
        const unsigned soff = (pz * eplane + (2 * t + py) * erow + px * p.ldy) * 4u;
        const u32x4 bits = {__builtin_bit_cast(unsigned, a0), __builtin_bit_cast(unsigned, a1),
                            __builtin_bit_cast(unsigned, a2), __builtin_bit_cast(unsigned, a3)};
        const bool row_ok = ty0 + t < p.Hi;
        __builtin_amdgcn_raw_buffer_store_b128(bits, yrsrc, row_ok ? vo_ok : ybytes, soff, ATVS_BUF_NT);
        const bool ok = evox_ok && row_ok;
        const float b0 = ok ? a0 : 0.f, b1 = ok ? a1 : 0.f, b2 = ok ? a2 : 0.f, b3 = ok ? a3 : 0.f;
        ssum[0] += b0; ssum[1] += b1; ssum[2] += b2; ssum[3] += b3;
        ssq[0] = __builtin_fmaf(b0, b0, ssq[0]); ssq[1] = __builtin_fmaf(b1, b1, ssq[1]);
        ssq[2] = __builtin_fmaf(b2, b2, ssq[2]); ssq[3] = __builtin_fmaf(b3, b3, ssq[3]);
      });
    });
    UDBG(5)
  }

#ifdef ATVS_UB_DEBUG
  UDBG(5)
  if (lane == 0 && blockIdx.x < 1024) {
    // slot 7: stages | wall-clock ticks (100 MHz) from entry to here << 16
    dbg_acc[7] = (unsigned long long)nstage | ((wall_clock64() - dbg_w0) << 16);
    dbg_acc[0] = dbg_w0;                     // absolute start (100 MHz): dispatch skew across the workgroups
    for (int i = 0; i < 8; ++i) atvs_dbg_ub[(blockIdx.x * 4 + wave) * 8 + i] = dbg_acc[i];
  }
#endif
  // ---- per-workgroup partial moments -> row blockIdx of stats: [2][16] doubles (Cout 8: columns 8..15 = 0)
  if (p.stats) {
    __syncthreads();
    double* s_red = reinterpret_cast<double*>(smem);   // [4 waves][2][16]
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      double a = (double)ssum[kk], bq = (double)ssq[kk];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        a += __shfl_xor(a, o);
        bq += __shfl_xor(bq, o);
      }
      if (COUT == 8) {
        a += __shfl_xor(a, 32);
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
      if (col < COUT || p.stats_ld == 16) p.stats[((size_t)blockIdx.x * 2 + which) * p.stats_ld + p.stats_coff + col] = v;
    }
  }
}

// ---- The summing decoder in TWO ROLES (round 6, second half) ------------------------------------------------------------------
// The full-resolution decoder conv_b*_6_0 / global_refine_3dconv6_0 (16 -> 8) whose input is the U-Net's skip SUM, formed while the halo
// is staged (the bn_add pass it replaces, norm.hip):  x_in = t(x, pa, bit 0) + t(x2, pb, bit 1) [+ t(x3, pc, bit 2)],
// t(v, par, relu) = par ? relu?(bn(v)) : v.  Rounds 5 / 6 first half: a form of the kernel above with ONE workgroup of four wavefronts
// per CU (the three sources of a stage in flight in 84 registers), every stage a chain -- barrier, the pieces into LDS, barrier, the K
// loop (the next halo's 21 loads per thread issued and four slots transformed in its shadow), the epilogue's stores (three more slots
// transformed behind them): 13.4 k cycles per stage against 9.1 k of the plain form on a materialised sum (phase timers at the bench's
// shape, three terms, cold inputs; 459 / 407 us per launch with three / two terms, 360 us inside a depth map).
// Here wavefronts 0-3 MULTIPLY (LDS fragment reads, MFMAs, the epilogue's stores and moments: the instruction stream of the plain form
// without its halo requests) and wavefronts 4-7 STAGE the next tile (its 21 loads per thread stay in flight across a whole stage, then
// the batch norms, the sum, the split, the LDS writes into the OTHER image pair): one LDS-only barrier per stage, global loads and
// stores stay in flight across it.  400 / 317 us cold, 318 us inside a depth map (-0.17 ms per map).  What a stage now waits for is
// memory on BOTH sides (tools_dev/phase_ub_sum.py, -DATVS_UB_DEBUG): with three terms the staging role spends 5.6 k of its 10.6 k
// cycles ISSUING the next 21 loads (the queue is full) and the multiply role 4.7 k in the epilogue's stores; with two terms the
// epilogue's stores are 6.1 k of 9.4 k -- a stage moves 82 KB of halo reads (1.66 x its tile: 5 x 5 x 17 voxels for 4 x 4 x 16) and
// 66 KB of stores per CU, 6 TB/s over the 256 CUs counting the halo's re-reads.
// Same tiles, same lanes, same order per value as the one-role form: bit for bit bn_add followed by the plain decoder, moments included
// (tests/test_gpu_groups.py::test_deconv_sums_its_inputs_on_load_bitwise).
__device__ __forceinline__ void ub_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(512, 1) void deconv_up_b_sum_kernel(UpBArgs p) {
  using U = UpB<8>;
  constexpr int NT = U::NT, TY = U::TY, HY = U::HY, MAXS = U::MAXS, NG = U::NG;
  constexpr int IMG2 = UB_NP * U::IMG;                  // one image pair (h0 | h1)
  // two wavefronts per SIMD, 256 registers each: the workgroup fills its SIMDs' register file like conv_xb / aanet_b, so that no
  // wavefront of another kernel runs beside its 16x16x32 MFMAs (only reachable with co-residency switched on)
  asm volatile("" ::: "v255");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const bool staging = wave >= 4;                       // wave-uniform
  const int mw = wave & 3;                              // multiply: the z plane of the tile; staging: rank among the staging wavefronts
  const int stid = tid & 255;
  const int r = lane & 15, q = lane >> 4;
#ifdef ATVS_UB_DEBUG
  // multiply: 0 = K loop, 1 = epilogue, 2 = waiting at the barrier; staging: 0 = batch norms + sum + split + LDS writes (behind the wait
  // for the loads), 1 = the next requests, 2 = waiting at the barrier
  unsigned long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long dbg_t = clock64();
  const unsigned long long dbg_w0 = wall_clock64();
#endif

  // packed weights of the one chunk -> LDS behind the two image pairs, once, by all eight wavefronts
  {
    const float4* src = reinterpret_cast<const float4*>(p.wp);
    unsigned char* dstb = smem + 2 * IMG2;
    for (int sp = wave; sp < U::NSTEP * UB_NP; sp += 8) reinterpret_cast<float4*>(dstb + sp * 1024)[lane] = src[sp * 64 + lane];
  }
  const int wb = 2 * IMG2;

  const int G = p.wg;
  const int grp = blockIdx.x / p.wg, lbk = blockIdx.x - grp * p.wg;
  const int xcd = lbk & 7, tslot = lbk >> 3;
  const float* __restrict__ xg = p.x + (size_t)grp * p.gx;
  const float* __restrict__ xg2 = p.x2 + (size_t)grp * p.gx;
  const float* __restrict__ xg3 = p.x3 ? p.x3 + (size_t)grp * p.gx : nullptr;
  float* __restrict__ yg = p.y + (size_t)grp * p.gy;
  const int per_xcd = (p.ntiles + 7) >> 3;
  const int slots_per_xcd = G >> 3;
  int my_tiles = 0;
  {
    int last = min(per_xcd, p.ntiles - xcd * per_xcd);
    if (tslot < last) my_tiles = (last - tslot + slots_per_xcd - 1) / slots_per_xcd;
  }
  const int nstage = my_tiles;
  auto tile_origin = [&](int k, int* z0, int* y0, int* x0) __attribute__((always_inline)) {
    int tl = xcd * per_xcd + tslot + k * slots_per_xcd;
    int bx = tl % p.tiles_x;
    int rest = tl / p.tiles_x;
    *x0 = bx * UB_TX;
    *y0 = (rest % p.tiles_y) * TY;
    *z0 = (rest / p.tiles_y) * UB_TZ;
  };
  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};

  if (staging) {
    // ======== staging role: halo slots stid, stid + 256, ... (4 channels of a halo voxel each) of the NEXT stage
    __builtin_amdgcn_s_setprio(3);
    int goff[MAXS], laddr[MAXS];
    unsigned pg[MAXS];
#pragma unroll
    for (int i = 0; i < MAXS; ++i) {
      int sl = stid + i * 256;
      const bool live = sl < U::SLOTS;
      sl = min(sl, U::SLOTS - 1);
      const int c4 = sl & 3, v = sl >> 2;
      const int xx = v % UB_HX, v2 = v / UB_HX;
      const int yy = v2 % HY, zz = v2 / HY;
      goff[i] = ((zz * p.Hi + yy) * p.Wi + xx) * p.Cin + c4 * 4;
      laddr[i] = ((zz * HY + yy) * UB_HX + xx) * UB_VB + c4 * 8;
      pg[i] = 0x808080u | (unsigned)(live ? zz : 0x7f) | ((unsigned)yy << 8) | ((unsigned)xx << 16);
    }
    // the batch-norm rows of this thread's four channels (c4 = stid & 3 in every slot): scale and shift (atvs_bn_shift4: the one-fma
    // form every site of a batch norm uses)
    float4 bns[3], bsh[3];
    {
      const float* pr[3] = {p.pa, p.pb, p.pc};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float* q3 = pr[k] ? pr[k] + (size_t)grp * 3 * p.Cin + (stid & 3) * 4 : nullptr;
        const float4 m = q3 ? ld4(q3) : make_float4(0.f, 0.f, 0.f, 0.f);
        bns[k] = q3 ? ld4(q3 + p.Cin) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 b = q3 ? ld4(q3 + 2 * p.Cin) : make_float4(0.f, 0.f, 0.f, 0.f);
        bsh[k] = atvs_bn_shift4(m, bns[k], b);
      }
    }
    float4 pf[MAXS], pf2[MAXS], pf3[MAXS];
    unsigned okm = 0;                        // which slots of the stage in flight lie inside the volume
    auto request = [&](int stage) __attribute__((always_inline)) {
      int z0, y0, x0;
      tile_origin(stage, &z0, &y0, &x0);
      const int gz0 = z0 - 1, gy0 = y0 - 1, gx0 = x0 - 1;
      const int org = ((gz0 * p.Hi + gy0) * p.Wi + gx0) * p.Cin;
      const unsigned lo = (unsigned)(gz0 < 0) | ((unsigned)(gy0 < 0) << 8) | ((unsigned)(gx0 < 0) << 16);
      const unsigned hi1 = (unsigned)(min(p.Di - 1 - gz0, 0x7e) + 1) | ((unsigned)(min(p.Hi - 1 - gy0, 0x7e) + 1) << 8) |
                           ((unsigned)(min(p.Wi - 1 - gx0, 0x7e) + 1) << 16);
      okm = 0;
#pragma unroll
      for (int i = 0; i < MAXS; ++i) {
        const unsigned t1 = pg[i] - lo;
        const unsigned t2 = hi1 + ~pg[i];
        const bool ok = ((t1 & t2) & 0x808080u) == 0x808080u;
        pf[i] = ld4(ok ? (xg + (org + goff[i])) : p.zeros);
        pf2[i] = ld4(ok ? (xg2 + (org + goff[i])) : p.zeros);
        if (xg3) pf3[i] = ld4(ok ? (xg3 + (org + goff[i])) : p.zeros);
        okm |= (ok ? 1u : 0u) << i;
      }
    };
    // one term of the sum: bn_add_kernel's arithmetic (norm.hip)
    auto term = [&](const float4& v, int k, const float* par) __attribute__((always_inline)) {
      float4 o = v;
      if (par) {
        o = atvs_bn4(v, bns[k], bsh[k]);
        if ((p.relu_mask >> k) & 1) {
          o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
      }
      return o;
    };
    // the stage in flight -> its two pieces, into image pair `img`
    auto deliver = [&](int img) __attribute__((always_inline)) {
      unsigned char* base = smem + img * IMG2;
#pragma unroll
      for (int i = 0; i < MAXS; ++i) {
        float4 a = term(pf[i], 0, p.pa);
        const float4 b = term(pf2[i], 1, p.pb);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        if (xg3) {
          const float4 d = term(pf3[i], 2, p.pc);
          a.x += d.x; a.y += d.y; a.z += d.z; a.w += d.w;
        }
        const bool in = (okm >> i) & 1u;
        uint2 p0, p1;
        atvs_split2_f16(in ? a.x : 0.f, in ? a.y : 0.f, UB_RS, &p0.x, &p1.x);
        atvs_split2_f16(in ? a.z : 0.f, in ? a.w : 0.f, UB_RS, &p0.y, &p1.y);
        if (i < MAXS - 1 || stid + i * 256 < U::SLOTS) {
          *reinterpret_cast<uint2*>(base + laddr[i]) = p0;
          *reinterpret_cast<uint2*>(base + U::IMG + laddr[i]) = p1;
        }
      }
    };
    if (nstage > 0) {
      request(0);
      deliver(0);
      if (nstage > 1) request(1);
      ub_lds_barrier();                      // image pair 0 (and the weights) are in LDS; stage 1's loads stay in flight
      UDBG(6)
      for (int stage = 0; stage < nstage; ++stage) {
        if (stage + 1 < nstage) {
          deliver((stage + 1) & 1);          // the other pair: its last readers left it at the previous barrier
          UDBG(0)
          if (stage + 2 < nstage) request(stage + 2);
          UDBG(1)
        }
        ub_lds_barrier();
        UDBG(2)
      }
    }
  } else {
    // ======== multiply role: the plain form's K loop and epilogue
    if (tid == 0) *reinterpret_cast<float4*>(smem + wb + U::WLDS) = make_float4(0.f, 0.f, 0.f, 0.f);      // (unused by the 8-channel form; keeps the layout of ub_lds)
    const int fbase = ((mw * HY) * UB_HX + r + 1 - (q >> 1)) * UB_VB + (q & 1) * 16;
    const int lane16 = lane * 16;
    f32x4 acc[TY][NT], accx[TY][NT];
    const unsigned ybytes = (unsigned)(p.gy * 4);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(yg, 0, ybytes, 0x00020000);
    if (nstage > 0) ub_lds_barrier();
    UDBG(6)
    for (int stage = 0; stage < nstage; ++stage) {
#pragma unroll
      for (int t = 0; t < TY; ++t)
#pragma unroll
        for (int m = 0; m < NT; ++m) acc[t][m] = accx[t][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int fb = fbase + (stage & 1) * IMG2;
      f16x8 Bq[2][TY], A0[2][4], A1[4];
      auto request_B = [&](auto PH) __attribute__((always_inline)) {
        constexpr int ph = decltype(PH)::value, g = ph / UB_NP, pc = ph % UB_NP, o2 = U::o2_of(g), oz = o2 >> 1, oy = o2 & 1;
        constexpr int disp = ((1 - oz) * HY + (1 - oy)) * UB_ROWB;
#pragma unroll
        for (int t = 0; t < TY; ++t) Bq[ph & 1][t] = *reinterpret_cast<const f16x8*>(smem + pc * U::IMG + fb + (disp + t * UB_ROWB));
      };
      auto request_A0 = [&](auto GT) __attribute__((always_inline)) {
        constexpr int g = decltype(GT)::value;
        ub_static_for<U::ntg(g)>([&](auto IT) __attribute__((always_inline)) {
          constexpr int i = decltype(IT)::value;
          A0[g & 1][i] = *reinterpret_cast<const f16x8*>(smem + wb + U::woff(U::first_step(g) + i, 0) + lane16);
        });
      };
      auto request_A1 = [&](auto GT) __attribute__((always_inline)) {
        constexpr int g = decltype(GT)::value;
        ub_static_for<U::ntg(g)>([&](auto IT) __attribute__((always_inline)) {
          constexpr int i = decltype(IT)::value;
          A1[i] = *reinterpret_cast<const f16x8*>(smem + wb + U::woff(U::first_step(g) + i, 1) + lane16);
        });
      };
      request_A0(IC<0>{});
      request_A1(IC<0>{});
      request_B(IC<0>{});
      asm volatile("" ::: "memory");
      ub_static_for<UB_NP * NG>([&](auto PH) __attribute__((always_inline)) {
        constexpr int ph = decltype(PH)::value, g = ph / UB_NP, pc = ph % UB_NP;
        if constexpr (ph + 1 < UB_NP * NG) request_B(IC<ph + 1>{});
        if constexpr (pc == 0 && g + 1 < NG) request_A0(IC<g + 1>{});
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        ub_static_for<U::ntg(g)>([&](auto IT) __attribute__((always_inline)) {
          constexpr int i = decltype(IT)::value, m = U::tile(g, i);
          if constexpr (pc == 0) {
#pragma unroll
            for (int t = 0; t < TY; ++t) acc[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A0[g & 1][i], Bq[ph & 1][t], acc[t][m], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < TY; ++t) accx[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1[i], Bq[ph & 1][t], accx[t][m], 0, 0, 0);
          } else {
#pragma unroll
            for (int t = 0; t < TY; ++t) accx[t][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A0[g & 1][i], Bq[ph & 1][t], accx[t][m], 0, 0, 0);
          }
        });
        if constexpr (pc == 0 && g + 1 < NG) {
          asm volatile("" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          request_A1(IC<g + 1>{});
        }
      });
      UDBG(0)
      // ---- epilogue: this lane holds, of tile m = (pz, py), channels (q&1)*4..+3 of output voxel (2z+pz, 2(y0+t)+py, 2(x0+r) + (q>>1))
      int tz0, ty0, tx0;
      tile_origin(stage, &tz0, &ty0, &tx0);
      const int zo = tz0 + mw, xo = tx0 + r;
      const bool evox_ok = zo < p.Di && xo < p.Wi;
      const unsigned Hy = 2u * p.Hi, Wy = 2u * p.Wi;
      const unsigned erow = Wy * p.ldy;
      const unsigned eplane = Hy * erow;
      const unsigned lane_c = (unsigned)((q >> 1) * p.ldy + (q & 1) * 4);
      const unsigned eo = (((unsigned)(2 * zo) * Hy + 2 * ty0) * Wy + 2 * xo) * p.ldy + p.ycoff + lane_c;
      const unsigned vo_ok = evox_ok ? eo * 4u : ybytes;
      ub_static_for<NT>([&](auto MT) __attribute__((always_inline)) {
        ub_static_for<TY>([&](auto TT) __attribute__((always_inline)) {
          constexpr int m = decltype(MT)::value, t = decltype(TT)::value;
          constexpr int pz = m >> 1, py = m & 1;
          float a0 = __builtin_fmaf(accx[t][m][0], UB_IRS, acc[t][m][0]), a1 = __builtin_fmaf(accx[t][m][1], UB_IRS, acc[t][m][1]);
          float a2 = __builtin_fmaf(accx[t][m][2], UB_IRS, acc[t][m][2]), a3 = __builtin_fmaf(accx[t][m][3], UB_IRS, acc[t][m][3]);
          if (p.relu) {
            a0 = (a0 < 0.f) ? 0.f : a0; a1 = (a1 < 0.f) ? 0.f : a1;
            a2 = (a2 < 0.f) ? 0.f : a2; a3 = (a3 < 0.f) ? 0.f : a3;
          }
          const unsigned soff = (pz * eplane + (2 * t + py) * erow) * 4u;
          const u32x4 bits = {__builtin_bit_cast(unsigned, a0), __builtin_bit_cast(unsigned, a1),
                              __builtin_bit_cast(unsigned, a2), __builtin_bit_cast(unsigned, a3)};
          const bool row_ok = ty0 + t < p.Hi;
          __builtin_amdgcn_raw_buffer_store_b128(bits, yrsrc, row_ok ? vo_ok : ybytes, soff, ATVS_BUF_NT);
          const bool ok = evox_ok && row_ok;
          const float b0 = ok ? a0 : 0.f, b1 = ok ? a1 : 0.f, b2 = ok ? a2 : 0.f, b3 = ok ? a3 : 0.f;
          ssum[0] += b0; ssum[1] += b1; ssum[2] += b2; ssum[3] += b3;
          ssq[0] = __builtin_fmaf(b0, b0, ssq[0]); ssq[1] = __builtin_fmaf(b1, b1, ssq[1]);
          ssq[2] = __builtin_fmaf(b2, b2, ssq[2]); ssq[3] = __builtin_fmaf(b3, b3, ssq[3]);
        });
      });
      UDBG(1)
      ub_lds_barrier();                      // this pair is free for the stage after next; the stores stay in flight
      UDBG(2)
    }
  }
#ifdef ATVS_UB_DEBUG
  if (lane == 0 && blockIdx.x < 512) {       // rows (block, wave 0..7): wavefronts 4..7 are the staging role
    dbg_acc[7] = (unsigned long long)nstage | ((wall_clock64() - dbg_w0) << 16);
    for (int i = 0; i < 8; ++i) atvs_dbg_ub[(blockIdx.x * 8 + wave) * 8 + i] = dbg_acc[i];
  }
#endif

  // ---- per-workgroup partial moments -> row blockIdx of stats: [2][16] doubles (columns 8..15 = 0): the one-role form's reduction
  if (p.stats) {
    __syncthreads();
    double* s_red = reinterpret_cast<double*>(smem);   // [4 multiply waves][2][16]
    if (!staging) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        double a = (double)ssum[kk], bq = (double)ssq[kk];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          a += __shfl_xor(a, o);
          bq += __shfl_xor(bq, o);
        }
        a += __shfl_xor(a, 32);
        bq += __shfl_xor(bq, 32);
        const int col = (q & 1) * 4 + kk;
        if (r == 0 && q < 2) {
          s_red[(mw * 2 + 0) * 16 + col] = a;
          s_red[(mw * 2 + 1) * 16 + col] = bq;
        }
      }
    }
    __syncthreads();
    if (tid < 32) {
      const int which = tid >> 4, col = tid & 15;
      double v = 0.0;
      if (col < 8)
        v = (s_red[(0 * 2 + which) * 16 + col] + s_red[(1 * 2 + which) * 16 + col]) +
            (s_red[(2 * 2 + which) * 16 + col] + s_red[(3 * 2 + which) * 16 + col]);
      if (col < 8 || p.stats_ld == 16) p.stats[((size_t)blockIdx.x * 2 + which) * p.stats_ld + p.stats_coff + col] = v;
    }
  }
}

// HOST: the two fp16 pieces of v (round to nearest even; the kernel's atvs_split2_f16) at out[base + piece * 64 * 8]; false if v
// does not fit fp16's range
bool ub_put(uint16_t* out, size_t base, float v) {
  const _Float16 h0 = (_Float16)v;
  const _Float16 h1 = (_Float16)((v - (float)h0) * UB_RS);
  std::memcpy(&out[base], &h0, 2);
  std::memcpy(&out[base + 64 * 8], &h1, 2);
  const float back = (float)h0;
  return back - back == 0.f;                             // finite
}

size_t ub_lds(int Cin, int Cout) {      // images + all chunks' weights + the zero line
  return (Cout == 8 ? UB_NP * (size_t)UpB<8>::IMG + (size_t)(Cin / 16) * UpB<8>::WLDS
                    : UB_NP * (size_t)UpB<16>::IMG + (size_t)(Cin / 16) * UpB<16>::WLDS) + 16;
}
// Cout 16: all chunks resident only while TWO workgroups still fit the CU's 160 KB -- with 88 KB (two chunks of whole 1 KB
// blocks, the 32 -> 16 layer until round 5) one workgroup ran per CU whatever __launch_bounds__ says (measured: the second half
// of the grid started when the first had finished).  The 32 -> 16 layers now keep both chunks resident in the compact layout
// (UpB::woff: 70 KB with the images; phase timers: the per-stage copy of a chunk had been 4.4 k of a 10.8 k-cycle stage, 0.7 k
// now); more input channels: one chunk at a time, whole blocks, re-read from L2 at every stage
#ifndef ATVS_UB_RESIDENT_MAX
#define ATVS_UB_RESIDENT_MAX (80 * 1024)
#endif
bool ub_stream(int Cin, int Cout) { return Cout == 16 && ub_lds(Cin, Cout) > ATVS_UB_RESIDENT_MAX; }

template <int COUT, bool STREAMW>
int launch_upb(const UpBArgs& a, long grid, size_t lds, hipStream_t s) {
  static AtvsAttrOnce lds_once;                   // per kernel instantiation (this function is a template / has one kernel)
  if (const int rc_ = atvs_set_max_lds_once(lds_once, reinterpret_cast<const void*>(deconv_up_b_kernel<COUT, STREAMW>), 160 * 1024)) return rc_;
  hipLaunchKernelGGL((deconv_up_b_kernel<COUT, STREAMW>), dim3((unsigned)grid), dim3(256), lds, s, a);
  return ATVS_OK;
}

// the summing decoder in two roles: 512 threads, two image pairs + the chunk's weights + the zero line
int launch_upb_sum2(const UpBArgs& a, long grid, hipStream_t s) {
  static AtvsAttrOnce lds_once;
  if (const int rc_ = atvs_set_max_lds_once(lds_once, reinterpret_cast<const void*>(deconv_up_b_sum_kernel), 160 * 1024)) return rc_;
  const size_t lds = 2 * UB_NP * (size_t)UpB<8>::IMG + UpB<8>::WLDS + 16;
  hipLaunchKernelGGL(deconv_up_b_sum_kernel, dim3((unsigned)grid), dim3(512), lds, s, a);
  return ATVS_OK;
}

template <int COUT>
bool pack_upb(const float* w, int Cin, uint16_t* out) {
  bool fits = true;
  using U = UpB<COUT>;
  auto kof = [](int par, int off) { return par ? (off ? -1 : 1) : (off ? 2 : 0); };
  for (int ch = 0; ch < Cin / 16; ++ch)
    for (int g = 0; g < U::NG; ++g)
      for (int i = 0; i < U::ntg(g); ++i) {
        const int m = U::tile(g, i), o2 = U::o2_of(g), oz = o2 >> 1, oy = o2 & 1, step = U::first_step(g) + i;
        const int pz = (COUT == 8) ? (m >> 1) : (m >> 2), py = (COUT == 8) ? (m & 1) : ((m >> 1) & 1);
        const int kd = kof(pz, oz), kh = kof(py, oy);
        if (kd < 0 || kh < 0) continue;                   // never: the groups hold active (tile, oz, oy) only
        for (int q = 0; q < 4; ++q)
          for (int row = 0; row < 16; ++row) {
            const int px = (COUT == 8) ? (row >> 3) : (m & 1), co = (COUT == 8) ? (row & 7) : row;
            const int kw = kof(px, q >> 1);
            if (kw < 0) continue;
            for (int e = 0; e < 8; ++e) {
              const int ci = ch * 16 + (q & 1) * 8 + e;
              const float v = w[((((size_t)kd * 3 + kh) * 3 + kw) * COUT + co) * Cin + ci];
              fits &= ub_put(out, ((((size_t)ch * U::NSTEP + step) * UB_NP) * 64 + q * 16 + row) * 8 + e, v);
            }
          }
      }
  return fits;
}

}  // namespace

// the packed weights of all chunks stay in LDS beside the three piece images where they fit (Cout 8: Cin <= 48; Cout 16:
// Cin <= 32); Cout 16 with more input channels re-reads one chunk's weights per stage
extern "C" int atvs_deconv_up_b_supported(int Cin, int Cout) {
  if ((Cout != 8 && Cout != 16) || Cin <= 0 || Cin % 16 || Cin > 256) return 0;
  return (ub_lds(Cin, Cout) <= 160 * 1024 || Cout == 16) ? 1 : 0;
}

extern "C" int atvs_deconv_up_b_pack_size(int Cin, int Cout, long* packed_bytes) {
  if (!packed_bytes) return ATVS_ERR_NULL;
  if (!atvs_deconv_up_b_supported(Cin, Cout)) return ATVS_ERR_SHAPE;
  *packed_bytes = (long)(Cin / 16) * (Cout == 8 ? UpB<8>::WCH : UpB<16>::WCH) + 16;
  return ATVS_OK;
}

// HOST function.  w: TF kernel of tf.layers.conv3d_transpose, [3,3,3,Cout,Cin].  packed[chunk][step][piece][lane = q*16 + row]
// [8 bf16]: step = the (tile, oz, oy) steps in group order; row as in atvs_deconv_up_pack; lane group q: ox = q >> 1, input
// channels chunk*16 + 8 (q & 1) + e; value = piece of w[kd][kh][kw][co][ci], k per axis from (parity, offset): (0,0) -> 0,
// (0,-1) -> 2, (1,0) -> 1, (1,-1) -> structural zero.
extern "C" int atvs_deconv_up_b_pack(const float* w, int Cin, int Cout, unsigned char* packed) {
  if (!w || !packed) return ATVS_ERR_NULL;
  long pb;
  int rc = atvs_deconv_up_b_pack_size(Cin, Cout, &pb);
  if (rc) return rc;
  std::memset(packed, 0, (size_t)pb);
  const bool fits = (Cout == 8) ? pack_upb<8>(w, Cin, reinterpret_cast<uint16_t*>(packed)) : pack_upb<16>(w, Cin, reinterpret_cast<uint16_t*>(packed));
  return fits ? ATVS_OK : ATVS_ERR_ARG;                  // a weight beyond fp16's range (|w| > 65504)
}

// workgroups PER SAMPLE of a launch over `groups` independent samples (rows of the statistics buffer = groups * this): the
// workgroups that fit the GPU at once (UB_WGS_PER_CU per CU), shared out among the samples, a multiple of 8 each
extern "C" long atvs_deconv_up_b_grid(int D, int H, int W, int Cout, int groups) {
  if (groups < 1) groups = 1;
  if (Cout != 8 && Cout != 16) return 0;
  const int ty = (Cout == 8) ? UpB<8>::TY : UpB<16>::TY;
  const long nt = (long)((D + UB_TZ - 1) / UB_TZ) * ((H + ty - 1) / ty) * ((W + UB_TX - 1) / UB_TX);
  long share = 256 * UB_GRID_PER_CU(Cout) / groups / 8 * 8;
  if (share < 8) share = 8;
  const long g = nt < share ? nt : share;
  return (g + 7) / 8 * 8;
}

// Contract of atvs_deconv_up_f32 with split operands (fp32-class results); grid / statistics rows = atvs_deconv_up_b_grid.
// stats_ld / stats_coff: a statistics row is [2][stats_ld] doubles and this launch's channels start at column stats_coff
// (16 / 0 = atvs_deconv_up_f32's layout; a wider layer computed 16 channels per launch passes its width and 0, 16, ...).
namespace {
int upb_launch(const float* x, const float* x2, const float* x3, const float* pa, const float* pb, const float* pc, int relu_mask,
               bool pro, const unsigned char* packed_w, float* y, double* stats_partial, int groups, int D, int H, int W, int Cin,
               int Cout, int ldy, int y_coff, int relu, int stats_ld, int stats_coff, atvs_stream_t stream);
}

extern "C" int atvs_deconv_up_b_f32(const float* x, const unsigned char* packed_w, float* y, double* stats_partial, int groups,
                                    int D, int H, int W, int Cin, int Cout, int ldy, int y_coff, int relu, int stats_ld,
                                    int stats_coff, atvs_stream_t stream) {
  return upb_launch(x, nullptr, nullptr, nullptr, nullptr, nullptr, 0, false, packed_w, y, stats_partial, groups, D, H, W, Cin, Cout,
                    ldy, y_coff, relu, stats_ld, stats_coff, stream);
}

// The same transposed convolution of the SUM of two or three volumes that is never written: x_in = t(x0, params0, bit 0) +
// t(x1, params1, bit 1) [+ t(x2, params2, bit 2)] with t(v, par, relu) = par ? relu?((v - mean) * scale + beta) : v -- exactly
// atvs_bn_add's arithmetic and order, formed per staged halo voxel (the U-Net's skip adds in front of conv_b*_6_0 /
// global_refine_3dconv6_0, reference cnn_wrapper/atvsnet.py:156-158,186-188,332-334).  params_i: (groups,3,Cin) or NULL (a
// finished tensor); x2 NULL: two terms.  Built for Cin = 16, Cout = 8 (atvs_deconv_up_b_sum_supported); results bit for bit
// those of atvs_bn_add followed by atvs_deconv_up_b_f32.
extern "C" int atvs_deconv_up_b_sum_supported(int Cin, int Cout) { return (Cin == 16 && Cout == 8) ? 1 : 0; }

extern "C" int atvs_deconv_up_b_sum_f32(const float* x0, const float* params0, const float* x1, const float* params1,
                                        const float* x2, const float* params2, int relu_mask, const unsigned char* packed_w,
                                        float* y, double* stats_partial, int groups, int D, int H, int W, int Cin, int Cout,
                                        int ldy, int y_coff, int relu, int stats_ld, int stats_coff, atvs_stream_t stream) {
  if (!x0 || !x1) return ATVS_ERR_NULL;
  if (!atvs_deconv_up_b_sum_supported(Cin, Cout)) return ATVS_ERR_SHAPE;
  if (!x2 && params2) return ATVS_ERR_ARG;
  return upb_launch(x0, x1, x2, params0, params1, params2, relu_mask, true, packed_w, y, stats_partial, groups, D, H, W, Cin, Cout,
                    ldy, y_coff, relu, stats_ld, stats_coff, stream);
}

namespace {
int upb_launch(const float* x, const float* x2, const float* x3, const float* pa, const float* pb, const float* pc, int relu_mask,
               bool pro, const unsigned char* packed_w, float* y, double* stats_partial, int groups, int D, int H, int W, int Cin,
               int Cout, int ldy, int y_coff, int relu, int stats_ld, int stats_coff, atvs_stream_t stream) {
  if (!x || !packed_w || !y) return ATVS_ERR_NULL;
  if (stats_ld < 16 || stats_coff < 0 || stats_coff + (stats_ld == 16 ? 16 : Cout) > stats_ld) return ATVS_ERR_ARG;
  if (!atvs_deconv_up_b_supported(Cin, Cout) || groups <= 0 || D <= 0 || H <= 0 || W <= 0) return ATVS_ERR_SHAPE;
  if (y_coff < 0 || y_coff + Cout > ldy || (ldy % 4) || (y_coff % 4)) return ATVS_ERR_SHAPE;
  if ((double)D * H * W * Cin >= 2147483648.0) return ATVS_ERR_SHAPE;
  if (8.0 * D * H * W * ldy * 4.0 >= 4294967296.0) return ATVS_ERR_SHAPE;
  UpBArgs a;
  long pbytes;
  atvs_deconv_up_b_pack_size(Cin, Cout, &pbytes);
  a.x = x; a.x2 = x2; a.x3 = x3; a.pa = pa; a.pb = pb; a.pc = pc; a.relu_mask = relu_mask;
  a.wp = packed_w; a.zeros = reinterpret_cast<const float*>(packed_w + (pbytes - 16));
  a.y = y; a.stats = stats_partial;
  a.Di = D; a.Hi = H; a.Wi = W; a.Cin = Cin; a.ldy = ldy; a.ycoff = y_coff; a.nchunk = Cin / 16; a.relu = relu;
  a.stats_ld = stats_ld; a.stats_coff = stats_coff;
  const int ty = (Cout == 8) ? UpB<8>::TY : UpB<16>::TY;
  a.tiles_y = (H + ty - 1) / ty; a.tiles_x = (W + UB_TX - 1) / UB_TX;
  a.ntiles = ((D + UB_TZ - 1) / UB_TZ) * a.tiles_y * a.tiles_x;
  const long blocks = atvs_deconv_up_b_grid(D, H, W, Cout, groups);
  a.wg = (int)blocks;
  a.gx = (long)D * H * W * Cin; a.gy = 8L * D * H * W * ldy;
  if (blocks * groups > 0x7fffffffL) return ATVS_ERR_SHAPE;
  hipStream_t st = as_stream(stream);
  const bool stream_w = ub_stream(Cin, Cout);
  const size_t lds = stream_w ? UB_NP * (size_t)UpB<16>::IMG + 2 * (size_t)UpB<16>::WLDS + 16 : ub_lds(Cin, Cout);
  int rc = pro ? launch_upb_sum2(a, blocks * groups, st)
               : (Cout == 8) ? launch_upb<8, false>(a, blocks * groups, lds, st)
                             : stream_w ? launch_upb<16, true>(a, blocks * groups, lds, st) : launch_upb<16, false>(a, blocks * groups, lds, st);
  if (rc) return rc;
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}
}  // namespace
