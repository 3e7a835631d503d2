// AANet aggregation over the source views in ONE launch (gfx950, split fp16 operands): the shared | unique 3x3x3 score
// convolutions of every view AND the cross-view softmax + weighted sum.
//
// Reference: Network.attention_activation / attention_aggregation, /root/reference/cnn_wrapper/network.py:282-351, 378-408 with
// second_weight=True, relu=True, biased=False (call sites cnn_wrapper/atvsnet.py:202,234):
//   S_n = relu(conv3d(X_n, W_shared)),  R_n = relu(conv3d(X_n, W_unique))           (8 -> 8 channels each, per view n)
//   S_sum = sum_n S_n;  U_n = (R_n - S_n) + S_sum;  score = softmax_n(U);  out = sum_n score_n * X_n
// Until round 4 this was two launches per module: conv_c16b<8> wrote [S|R] of every view (960 MB at 4 views of configs[2]) and
// aanet_combine read it back with the X_n (295 us of pure traffic at the HBM rate).  Round 5 made it one launch of four wavefronts
// that did everything in turn (608 us, matrix pipe 24 % busy: split + LDS writes, barriers, 64 expf + 64 divisions per lane and
// tile with nothing beside them).
//
// Round 6: ONE workgroup of EIGHT wavefronts per CU, two roles (conv_xb's structure).  A workgroup owns a 4(z) x 8(y) x 16(x)
// tile and walks the views; a STAGE is one (tile, view).
//   * wavefronts 0..3 (one per SIMD) MULTIPLY: LDS fragment reads + MFMAs only (7 K steps of four taps x 8 channels, three
//     products: the K order atvs_tap8 shared with conv_c16b's Cin = 8 form -- every halo row fetched once per column group), then conv_c16b's epilogue arithmetic (zero bias, ReLU) and 8 ds_write_b128 that hand the
//     view's [S|R] of the wavefront's plane to its partner -- no vector-memory instruction at all;
//   * wavefronts 4..7 (the second wavefront of each SIMD) STAGE and COMBINE: they fetch the fp32 halo of the stage after next,
//     split the next stage's into its two fp16 pieces and write the OTHER of two image buffers; they read the previous stage's
//     hand-off ([S|R] of one view) and run the cross-view softmax + weighted sum beside the multiplying wavefronts' K loops:
//       1 .. 4 views: the reference's literal arithmetic (aanet_combine_kernel's, in its order: S_sum accumulates in view order from
//         0.f, U_n = (R_n - S_n) + S_sum) on S_sum and R_n - S_n kept in registers -- (NV + 1) x 16 values instead of the 2 NV x 16
//         of [S|R], twice (tiles alternate): the previous tile is combined row group by row group, one group per stage.  Bit for
//         bit the two-launch form;
//       5 .. 8 views: a RUNNING softmax over the arriving views -- (max, sum e, sum e x) per value, 48 registers whatever NV, the
//         same work in every stage -- of R_n - S_n alone: the S_sum term shifts every view's score alike and cancels in the
//         softmax.  Within 1e-6 of the two-launch form, tolerance-tested.
// ONE LDS-only barrier per stage (s_waitcnt lgkmcnt(0); s_barrier -- __syncthreads() would drain the halo requests in flight):
// it publishes the next image buffer and this stage's hand-off and retires the buffers both will be overwritten in.
// The 16 MFMA rows are ordered (S[2q], S[2q+1], R[2q], R[2q+1]) for lane group q (atvs_aanet_b_pack), so a lane holds S AND R of
// its two channels -- no cross-lane exchange -- and reads the matching two channels of every X_n's centre voxel (L2 hits: the
// halo just came through).
// Scalar fp32 arithmetic (-fno-slp-vectorize): the staging wavefronts compute beside the kernel's own 16x16x32 MFMA wavefronts
// (DESIGN.md appendix B); two wavefronts of 256 registers per SIMD: no other kernel's wavefront fits beside them.
#include <cstring>
#include <type_traits>
#include <utility>

#include "conv_common.h"

// buffer_load_dwordx4 (offen).  hipcc 7.2's __builtin_amdgcn_raw_buffer_load_b128 compiles to a ONE-dword load whose value is
// splat over the four components (geometry.hip), so the LLVM intrinsic is bound by name instead.
typedef float ab_f32x4 __attribute__((ext_vector_type(4)));
__device__ ab_f32x4 ab_buffer_load_x4(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.ptr.buffer.load.v4f32");
typedef float ab_f32x2 __attribute__((ext_vector_type(2)));
__device__ ab_f32x2 ab_buffer_load_x2(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.ptr.buffer.load.v2f32");
__device__ void ab_buffer_store_x2(ab_f32x2 v, __amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.ptr.buffer.store.v2f32");

namespace {

constexpr int AB_TZ = 4, AB_TY = 8, AB_TX = 16;
constexpr int AB_HZ = AB_TZ + 2, AB_HY = AB_TY + 2, AB_HX = AB_TX + 2;
constexpr int AB_VB = 16;                                      // bytes per voxel of one piece image (8 channels x 2 B)
constexpr int AB_ROWB = AB_HX * AB_VB;
constexpr int AB_IMG = AB_HZ * AB_HY * AB_ROWB;                // 17,280 bytes per piece
constexpr int AB_SLOTS = AB_HZ * AB_HY * AB_HX * 2;            // float4 slots of the fp32 halo
constexpr int AB_MAXS = (AB_SLOTS + 255) / 256;                // 9 per staging thread
constexpr int AB_JC = 7;                                       // K steps of four taps x 8 channels: atvs_tap8 (conv_common.h)
constexpr int AB_NP = 2;
constexpr int AB_WSTEP = AB_NP * 1024;
constexpr int AB_MAXV = 8;                                     // views per launch
constexpr float AB_RS = 2048.f, AB_IRS = 1.f / 2048.f;
// LDS map: two image buffers (two piece images each) | packed weights | two hand-off buffers ([wavefront][row][lane] float4)
constexpr int AB_BUF = AB_NP * AB_IMG;                         // 34,560
constexpr int AB_WOFF = 2 * AB_BUF;                            // 69,120
constexpr int AB_HOFF = AB_WOFF + AB_JC * AB_WSTEP;            // 83,456
constexpr int AB_HBUF = 4 * AB_TY * 1024;                      // 32,768
constexpr int AB_LDS = AB_HOFF + 2 * AB_HBUF;                  // 148,992 of 163,840

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

struct AbArgs {
  const float* x[AB_MAXV];     // the views' volumes (D,H,W,8)
  const unsigned char* wp;     // atvs_aanet_b_pack
  const float* zeros;
  float* out;                  // (D,H,W,8)
  int Di, Hi, Wi;
  int tiles_y, tiles_x, ntiles;
  int wg;
};

template <int N>
using IC = std::integral_constant<int, N>;
template <class F, int... I>
__device__ __forceinline__ void ab_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(IC<I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void ab_static_for(F&& f) {
  ab_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ void ab_split(const float4& v, f16x4* p0, f16x4* p1) {      // conv_c16b's b16_split
  // atvs_split2_f16 (common.h): five vector instructions per two values instead of the 8-9 of the C form, the same values
  uint2 a, b;
  atvs_split4_f16(v, &a, &b);
  *p0 = __builtin_bit_cast(f16x4, a);
  *p1 = __builtin_bit_cast(f16x4, b);
}

#ifdef ATVS_AB_DEBUG
// development build (tools_dev/phase_ab.py): per-wavefront tick counts of the phases of a stage
__device__ unsigned long long atvs_dbg_ab[2048 * 8];
extern "C" int atvs_debug_read_ab(void* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(atvs_dbg_ab), sizeof(atvs_dbg_ab));
}
#define ABDBG(i) { unsigned long long t_ = clock64(); dbg_acc[i] += t_ - dbg_t; dbg_t = t_; }
#else
#define ABDBG(i)
#endif

// LDS-only barrier: the staging wavefronts' halo requests stay in flight across it
__device__ __forceinline__ void ab_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// NV: the number of views (compile time: no branches in the combine, exactly NV + 1 register slots per lane value)
template <int NV>
__global__ __launch_bounds__(512, 1) void aanet_b_kernel(AbArgs p) {
  // the workgroup's two wavefronts per SIMD take the SIMD's whole register file (256 each): no wavefront of ANOTHER kernel runs
  // beside this one's 16-bit MFMAs
  asm volatile("" ::: "v255");
  constexpr int TY = AB_TY, HY = AB_HY, MAXS = AB_MAXS, JC = AB_JC;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave: a scalar (uniform branches)
  const int r = lane & 15, q = lane >> 4;

  {
    const float4* src = reinterpret_cast<const float4*>(p.wp);
    float4* dst = reinterpret_cast<float4*>(smem + AB_WOFF);
    for (int i = tid; i < JC * (AB_WSTEP / 16); i += 512) dst[i] = src[i];
  }

  const int G = p.wg;
  const int lbk = blockIdx.x;
  const int xcd = lbk & 7, tslot = lbk >> 3;
  const int per_xcd = (p.ntiles + 7) >> 3;
  const int slots_per_xcd = G >> 3;
  int my_tiles = 0;
  {
    int last = min(per_xcd, p.ntiles - xcd * per_xcd);
    if (tslot < last) my_tiles = (last - tslot + slots_per_xcd - 1) / slots_per_xcd;
  }
  const int nstages = my_tiles * NV;
#ifdef ATVS_AB_DEBUG
  unsigned long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long dbg_t = clock64();
#endif
  auto tile_origin = [&](int k, int* z0, int* y0, int* x0) __attribute__((always_inline)) {
    int tl = xcd * per_xcd + tslot + k * slots_per_xcd;
    int bx = tl % p.tiles_x;
    int rest = tl / p.tiles_x;
    *x0 = bx * AB_TX;
    *y0 = (rest % p.tiles_y) * TY;
    *z0 = (rest / p.tiles_y) * AB_TZ;
  };

  if (wave < 4) {
    // ======================= MULTIPLYING wavefronts: wavefront w owns plane z0 + w of the tile =======================
    const int fbase = ((wave * HY) * AB_HX + r) * AB_VB;
    // K order atvs_tap8 (conv_common.h): steps 3 G .. 3 G + 2 (G = 0, 1) carry four (kd, kw) columns at kh = 0, 1, 2 -- row t of step kh
    // reads halo row t + kh of the lane group's column, so each of a group's ten halo rows is fetched ONCE and multiplied by the (up to)
    // three steps that want it; step 6 carries the ninth column at kh = q.  56 fragment reads per stage instead of 112: the LDS array
    // (51 % busy, shared with the staging role's writes and hand-off) was what both roles waited for -- 0.45 -> 0.37 ms at 4 views
    // in a development build that simply skipped half the reads.
    int cb[2];                                                 // this lane group's column of groups 0 | 1: (kd, kw) of combination 4 G + q
#pragma unroll
    for (int G2 = 0; G2 < 2; ++G2) {
      const int c = G2 * 4 + q;
      cb[G2] = fbase + ((c / 3) * HY) * AB_ROWB + (c % 3) * AB_VB;
    }
    const int c8 = fbase + (2 * HY + min(q, 2)) * AB_ROWB + 2 * AB_VB;      // step 6: column (2, 2) at kh = q (q = 3: zero weights)
    const int wbase = AB_WOFF + lane * 16;
    const int hbase = AB_HOFF + (wave * TY) * 1024 + lane * 16;
    __syncthreads();                                           // weights + stage 0 are in LDS
    f32x4 acc[TY], accx[TY];
#pragma unroll 1
    for (int s = 0; s < nstages; ++s) {
      const int ib = (s & 1) * AB_BUF;
#pragma unroll
      for (int t = 0; t < TY; ++t) acc[t] = accx[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // a UNIT = one halo row y of a group (20 units: two fragment reads -- both pieces -- and up to nine MFMAs) or one output row
      // of step 6 (8 units of three MFMAs).  F[u % 3]: a group unit's fragments, requested TWO units ahead (the short units at a
      // group's ends -- 3 and 6 MFMAs -- do not cover an LDS round trip under load); F6[t]: step 6's fragments, requested during
      // group 1's units (its 24 MFMAs then run without a wait); A[set][kh][piece]: a group's weight fragments (group 0: set 0,
      // group 1: set 1, step 6: set 0 again), requested during the previous group.
      constexpr int NG = 20, NU = 28;
      f16x8 F[3][AB_NP], F6[TY][AB_NP], A[2][3][AB_NP];
      auto request_F = [&](auto UU, auto PC) __attribute__((always_inline)) {
        constexpr int u = decltype(UU)::value, pc = decltype(PC)::value;
        F[u % 3][pc] = *reinterpret_cast<const f16x8*>(smem + ib + cb[u / 10] + ((u % 10) * AB_ROWB + pc * AB_IMG));
      };
      auto request_F6 = [&](auto TT, auto PC) __attribute__((always_inline)) {
        constexpr int t = decltype(TT)::value, pc = decltype(PC)::value;
        F6[t][pc] = *reinterpret_cast<const f16x8*>(smem + ib + c8 + (t * AB_ROWB + pc * AB_IMG));
      };
      auto request_A = [&](auto SS, auto PC) __attribute__((always_inline)) {       // K step S -> set (S / 3) & 1, slot S % 3
        constexpr int S = decltype(SS)::value, pc = decltype(PC)::value;
        A[(S / 3) & 1][S % 3][pc] = *reinterpret_cast<const f16x8*>(smem + wbase + S * AB_WSTEP + pc * 1024);
      };
      ab_static_for<3>([&](auto SS) __attribute__((always_inline)) { request_A(SS, IC<0>{}); request_A(SS, IC<1>{}); });
      ab_static_for<2>([&](auto UU) __attribute__((always_inline)) { request_F(UU, IC<0>{}); request_F(UU, IC<1>{}); });
      asm volatile("" ::: "memory");
      ABDBG(0)
      ab_static_for<NU>([&](auto UU) __attribute__((always_inline)) {
        constexpr int u = decltype(UU)::value;
        constexpr int G = u < NG ? u / 10 : 2, y = u < NG ? u % 10 : u - NG;
        constexpr int set = G == 1 ? 1 : 0;
        // the rows this unit feeds: group units row t = y - kh for every step kh with 0 <= t < TY; a step-6 unit its own row
        constexpr int kh_lo = G == 2 ? 0 : (y - (TY - 1) > 0 ? y - (TY - 1) : 0), kh_hi = G == 2 ? 0 : (y < 2 ? y : 2);
        constexpr int nr = kh_hi - kh_lo + 1;
        // every LDS read behind ONE MFMA: MFMA m of the unit is followed by hook(m)
        auto hook = [&](auto MM) __attribute__((always_inline)) {
          constexpr int m = decltype(MM)::value;
          if constexpr (u + 2 < NG) {
            if constexpr (m == 0) request_F(IC<u + 2>{}, IC<0>{});
            if constexpr (m == 1) request_F(IC<u + 2>{}, IC<1>{});
          }
          // the next group's weights during this group's 9-MFMA units y = 3, 4, 5 (two fragments each); step 6's during group 1's y = 3
          if constexpr (G < 2 && nr == 3 && (m == 2 || m == 3)) {
            if constexpr (G == 0 && y >= 3 && y <= 5) request_A(IC<3 + (y - 3)>{}, IC<m - 2>{});
            if constexpr (G == 1 && y == 3) request_A(IC<6>{}, IC<m - 2>{});
          }
          // step 6's fragments of row y - 1 during group 1's units y = 1 .. 8 (six MFMAs or more each)
          if constexpr (G == 1 && y >= 1 && y <= TY && (m == 4 || m == 5)) request_F6(IC<y - 1>{}, IC<m - 4>{});
          asm volatile("" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        };
        // input piece h0 with both weight pieces (main | cross), row by row; then h1 with g0 (cross): per accumulator the order of
        // conv_c16b's phases -- main: g0 h0; cross: g1 h0, g0 h1 -- and steps in ascending order
        ab_static_for<nr>([&](auto RR) __attribute__((always_inline)) {
          constexpr int i = decltype(RR)::value, kh = G == 2 ? 0 : kh_lo + i, t = G == 2 ? y : y - kh;
          const f16x8& f0 = G == 2 ? F6[y][0] : F[u % 3][0];
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[set][kh][0], f0, acc[t], 0, 0, 0);
          hook(IC<2 * i>{});
          accx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[set][kh][1], f0, accx[t], 0, 0, 0);
          hook(IC<2 * i + 1>{});
        });
        ab_static_for<nr>([&](auto RR) __attribute__((always_inline)) {
          constexpr int i = decltype(RR)::value, kh = G == 2 ? 0 : kh_lo + i, t = G == 2 ? y : y - kh;
          const f16x8& f1 = G == 2 ? F6[y][1] : F[u % 3][1];
          accx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[set][kh][0], f1, accx[t], 0, 0, 0);
          hook(IC<2 * nr + i>{});
        });
      });
      ABDBG(1)
      // this view's [S|R] of the lane's voxels: conv_c16b's epilogue arithmetic (zero bias, ReLU) -> the partner wavefront
      const int hb = hbase + (s & 1) * AB_HBUF;
#pragma unroll
      for (int t = 0; t < TY; ++t) {
        // conv_c16b's (main + cross * 2^-11) + bias(0), ReLU in THREE instructions per value instead of five: the product by a power
        // of two is exact, so the fused multiply-add rounds what the separate add rounds; "+ 0.f" only turns -0 into +0, which the
        // partner's arithmetic cannot tell apart (S enters as 0.f + S and as R - S).  These instructions share the SIMD's issue with
        // the partner: every one of them is stage time.
        float a0 = __builtin_fmaf(accx[t][0], AB_IRS, acc[t][0]), a1 = __builtin_fmaf(accx[t][1], AB_IRS, acc[t][1]);
        float a2 = __builtin_fmaf(accx[t][2], AB_IRS, acc[t][2]), a3 = __builtin_fmaf(accx[t][3], AB_IRS, acc[t][3]);
        a0 = (a0 < 0.f) ? 0.f : a0; a1 = (a1 < 0.f) ? 0.f : a1;      // compare + select, not v_max: a NaN stays a NaN
        a2 = (a2 < 0.f) ? 0.f : a2; a3 = (a3 < 0.f) ? 0.f : a3;
        *reinterpret_cast<float4*>(smem + hb + t * 1024) = make_float4(a0, a1, a2, a3);
      }
      ABDBG(2)
      ab_lds_barrier();
      ABDBG(3)
    }
#ifdef ATVS_AB_DEBUG
    if (lane == 0 && blockIdx.x < 256) {
      dbg_acc[7] = (unsigned long long)nstages;
      for (int i = 0; i < 8; ++i) atvs_dbg_ab[(blockIdx.x * 8 + wave) * 8 + i] = dbg_acc[i];
    }
#endif
    return;
  }

  // ========================== STAGING / COMBINING wavefronts: wavefront 4 + w is the partner of w ==========================
  // (their vector instructions share the SIMD's issue with the partner's MFMAs -- an MFMA holds it for 8 of its 16 cycles -- so
  // this role is written for FEW instructions: halo requests through a buffer descriptor at the tile's origin with per-lane
  // constant offsets and a per-tile validity mask (2 instructions per request), 2^(x log2 e) and one reciprocal in the softmax)
  const int stid = tid - 256, sw = wave - 4;
  // the multiplying partner always has an MFMA ready and, being older, wins every arbitration for the SIMD's vector issue: this
  // role then advances one instruction per MFMA.  Raised priority lets its instructions go first when they are ready (the
  // partner's MFMAs fill what is left): split + LDS writes 2.1k -> 1.3k cycles per stage, launch -5 %.
#ifndef ATVS_AB_PRIO
#define ATVS_AB_PRIO 3
#endif
  __builtin_amdgcn_s_setprio(ATVS_AB_PRIO);
  int goff[MAXS], laddr[MAXS];
  // slot i of this lane = float4 c4 of halo voxel (zz, yy, xx); packed coordinates for the border tiles' validity test (recomputed
  // there instead of kept: nine registers this role does not have)
  auto slot_pg = [&](int i, int stid_now) __attribute__((always_inline)) {
    int s = stid_now + i * 256;
    const bool live = s < AB_SLOTS;
    s = min(s, AB_SLOTS - 1);
    const int v = s / 2;
    const int xx = v % AB_HX, v2 = v / AB_HX;
    const int yy = v2 % HY, zz = v2 / HY;
    return 0x808080u | (unsigned)(live ? zz : 0x7f) | ((unsigned)yy << 8) | ((unsigned)xx << 16);
  };
#pragma unroll
  for (int i = 0; i < MAXS; ++i) {
    int s = stid + i * 256;
    s = min(s, AB_SLOTS - 1);
    const int c4 = s % 2, v = s / 2;
    const int xx = v % AB_HX, v2 = v / AB_HX;
    const int yy = v2 % HY, zz = v2 / HY;
    goff[i] = (((zz * p.Hi + yy) * p.Wi + xx) * 8 + c4 * 4) * 4;       // BYTES from the halo's origin
    laddr[i] = ((zz * HY + yy) * AB_HX + xx) * AB_VB + c4 * 8;
  }
  struct PfTile {
    int org;                   // element offset of the halo's origin (may lie outside the volume: masked lanes never read)
    unsigned vmask;            // per lane: bit i = slot i lies outside the volume
    int z0, y0, x0;
  };
  auto pf_tile = [&](int k) __attribute__((always_inline)) {
    PfTile T;
    tile_origin(min(k, max(my_tiles - 1, 0)), &T.z0, &T.y0, &T.x0);      // past the last tile: a harmless re-read
    const int gz0 = T.z0 - 1, gy0 = T.y0 - 1, gx0 = T.x0 - 1;
    T.org = ((gz0 * p.Hi + gy0) * p.Wi + gx0) * 8;
    const unsigned lo = (unsigned)(gz0 < 0) | ((unsigned)(gy0 < 0) << 8) | ((unsigned)(gx0 < 0) << 16);
    const unsigned hi1 = (unsigned)(min(p.Di - 1 - gz0, 0x7e) + 1) | ((unsigned)(min(p.Hi - 1 - gy0, 0x7e) + 1) << 8) |
                         ((unsigned)(min(p.Wi - 1 - gx0, 0x7e) + 1) << 16);
    T.vmask = 0;
    const bool interior = lo == 0 && gz0 + AB_HZ <= p.Di && gy0 + HY <= p.Hi && gx0 + AB_HX <= p.Wi;      // uniform
    if (!interior) {
      int stid_now = stid;
      asm volatile("" : "+v"(stid_now));                       // opaque: or the nine coordinates are hoisted out of the tile loop (and spilled)
#pragma unroll
      for (int i = 0; i < MAXS; ++i) {
        const unsigned pgi = slot_pg(i, stid_now);
        const unsigned t1 = pgi - lo;
        const unsigned t2 = hi1 + ~pgi;
        const bool ok = ((t1 & t2) & 0x808080u) == 0x808080u;
        T.vmask |= ok ? 0u : (1u << i);
      }
    } else if (stid + (MAXS - 1) * 256 >= AB_SLOTS) {
      T.vmask = 1u << (MAXS - 1);                                       // the lanes beyond the last slot
    }
    return T;
  };
  float4 pf[MAXS];
  auto pf_request = [&](const PfTile& T, const float* xg) __attribute__((always_inline)) {
    // descriptor at the halo's origin, no bounds but 2 GB: lanes outside the volume get an offset beyond it and read zeros
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xg + T.org), 0, 0x7ffffff0, 0x00020000);
#pragma unroll
    for (int i = 0; i < MAXS; ++i) {
      const int voff = goff[i] | __builtin_amdgcn_sbfe((int)T.vmask, i, 1);      // all ones when outside
      const ab_f32x4 v = ab_buffer_load_x4(rs, voff, 0, 0);
      pf[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
  };
  auto pf_write = [&](int ib) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MAXS; ++i) {
      if (i < MAXS - 1 || stid + i * 256 < AB_SLOTS) {
        f16x4 p0, p1;
        ab_split(pf[i], &p0, &p1);
        *reinterpret_cast<f16x4*>(smem + ib + laddr[i]) = p0;
        *reinterpret_cast<f16x4*>(smem + ib + AB_IMG + laddr[i]) = p1;
      }
    }
  };

  // stage 0 -> image buffer 0; the halo of stage 1 requested
  PfTile Tk = pf_tile(0), Tk1 = pf_tile(1), Tp = Tk, Tk2 = Tk1;
  if (my_tiles > 0) {
    pf_request(Tk, p.x[0]);
    pf_write(0);
    pf_request(NV > 1 ? Tk : Tk1, p.x[NV > 1 ? 1 : 0]);
  }
  __syncthreads();

  const int hread = AB_HOFF + (sw * TY) * 1024 + lane * 16;
  int s = 0;
  if constexpr (NV > 4) {
    // ---- 5 .. 8 views: the softmax over views as a RUNNING one.  softmax_n((R_n - S_n) + S_sum) = softmax_n(R_n - S_n): the sum of
    // the shared scores is the same shift for every view and cancels (the oracle and the two-launch form keep the reference's
    // literal formula; the difference is the rounding of d_n + S_sum, ~|S_sum| 2^-24 relative in a weight).  Per lane value only
    // (max, sum e, sum e x) live on -- 48 registers whatever NV, where S_sum and R_n - S_n of eight views are 144 and spilled
    // (1.91 ms at configs[3]) -- and every stage does the same work: the arriving view's hand-off and centre voxels update the
    // running triple, e' = 2^((max - max') log2 e) rescales it; after the last view out = sum e x / sum e.
    // Tolerance-tested against the oracle and the two-launch form (2e-5 of the output maximum), not bit for bit.
    constexpr float L2E = 1.44269504088896340736f;
    float mx[TY], my[TY], dnx[TY], dny[TY], nmx[TY], nmy[TY];
    const int lane_xoff = (r * 8 + 2 * q) * 4;
    auto stage_body = [&](auto VV, int k) __attribute__((always_inline)) {
      constexpr int v = decltype(VV)::value;
      constexpr int vp = (v + NV - 1) % NV;                    // the view whose hand-off arrives: of this tile (v > 0) or the previous one
      const bool live = k < my_tiles;
      const bool have = s > 0;
      const PfTile& Tt = v == 0 ? Tp : Tk;                     // the tile that view belongs to
      const int zo = Tt.z0 + sw, ty0 = Tt.y0;
      // the lane's voxel column of the tile's plane: a buffer descriptor at (zo, ty0, x0) of the view (scalar), the lane's constant
      // byte offset (or all ones = beyond the descriptor: reads zeros, stores nothing), the row as the scalar offset -- no per-row
      // 64-bit addresses (hoisted out of the tile loop they were 16 spilled registers per view, reloaded in every stage)
      const size_t tbase = (((size_t)zo * p.Hi + ty0) * p.Wi + Tt.x0) * 8;
      const int voff = (zo < p.Di && Tt.x0 + r < p.Wi) ? lane_xoff : -1;
      const int vrowb = p.Wi * 32;
      const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x[vp] + tbase), 0, 0x7ffffff0, 0x00020000);
      // the hand-off and the view's centre voxels, requested FIRST (older than the halo requests below: vector-memory results
      // return in order)
      float4 h[TY];
      ab_f32x2 xv[TY];
      if (have) {
        const int hb = hread + ((s - 1) & 1) * AB_HBUF;
#pragma unroll
        for (int t = 0; t < TY; ++t) h[t] = *reinterpret_cast<const float4*>(smem + hb + t * 1024);
#pragma unroll
        for (int t = 0; t < TY; ++t) xv[t] = ab_buffer_load_x2(xrs, (ty0 + t < p.Hi) ? voff : -1, t * vrowb, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (live) {
#ifdef ATVS_AB_DEBUG
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ABDBG(5)
#endif
        pf_write(((s + 1) & 1) * AB_BUF);
        ABDBG(0)
        constexpr int idx = v + 2, dk = idx / NV, vn = idx % NV;
        pf_request(dk == 0 ? Tk : Tk1, p.x[vn]);
        ABDBG(1)
      }
      if (have) {
#pragma unroll
        for (int t = 0; t < TY; ++t) {
          const float dx = h[t].z - h[t].x, dy = h[t].w - h[t].y;           // R - S
          if constexpr (vp == 0) {
            mx[t] = dx; my[t] = dy; dnx[t] = 1.f; dny[t] = 1.f; nmx[t] = xv[t][0]; nmy[t] = xv[t][1];
          } else {
            const float m2x = fmaxf(mx[t], dx), m2y = fmaxf(my[t], dy);
            const float lx = m2x * L2E, ly = m2y * L2E;
            const float ax = __builtin_amdgcn_exp2f(__builtin_fmaf(mx[t], L2E, -lx)), ay = __builtin_amdgcn_exp2f(__builtin_fmaf(my[t], L2E, -ly));
            const float bx = __builtin_amdgcn_exp2f(__builtin_fmaf(dx, L2E, -lx)), by = __builtin_amdgcn_exp2f(__builtin_fmaf(dy, L2E, -ly));
            dnx[t] = __builtin_fmaf(dnx[t], ax, bx); dny[t] = __builtin_fmaf(dny[t], ay, by);
            nmx[t] = __builtin_fmaf(nmx[t], ax, bx * xv[t][0]); nmy[t] = __builtin_fmaf(nmy[t], ay, by * xv[t][1]);
            mx[t] = m2x; my[t] = m2y;
          }
          if constexpr (vp == NV - 1) {
            const ab_f32x2 o = {nmx[t] * __builtin_amdgcn_rcpf(dnx[t]), nmy[t] * __builtin_amdgcn_rcpf(dny[t])};
            const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(p.out + tbase, 0, 0x7ffffff0, 0x00020000);
            ab_buffer_store_x2(o, ors, (ty0 + t < p.Hi) ? voff : -1, t * vrowb, 0);
          }
        }
        ABDBG(2)
      }
      ABDBG(3)
      if (live) {
        ab_lds_barrier();
        ++s;
      }
      ABDBG(4)
    };
#pragma unroll 1
    for (int k = 0; k <= my_tiles; ++k) {
      if (k < my_tiles) {
        ab_static_for<NV>([&](auto VV) __attribute__((always_inline)) { stage_body(VV, k); });
      } else {
        stage_body(IC<0>{}, k);                                  // the last view of the last tile
      }
      Tp = Tk;
      Tk = Tk1;
      Tk1 = pf_tile(k + 2);
    }
  } else {
    // ---- 1 .. 4 views: the reference's literal arithmetic, bit for bit the two-launch form.
    // S_sum and R_n - S_n of channels 2q, 2q+1 of voxels (z0 + sw, y0 + t, x0 + r), t < TY: TWO sets, tiles alternate.  The set of
    // tile k fills view by view during the stages of tile k (+ the first of tile k + 1) while the set of tile k - 1 is combined ROW
    // GROUP by row group, one group per stage of tile k (rows [v TY / NV, (v + 1) TY / NV) in stage v): the 16 NV exponentials of a
    // tile in ONE stage made that stage twice as long as the multiplying wavefronts' (they waited 1.7k cycles per stage on average).
    float ssx[2][TY], ssy[2][TY], dvx[2][NV][TY], dvy[2][NV][TY];
    const int lane_xoff = (r * 8 + 2 * q) * 4;
    auto stage_body = [&](auto PP, auto VV, int k) __attribute__((always_inline)) {
      constexpr int P = decltype(PP)::value, v = decltype(VV)::value;
      constexpr int vp = (v + NV - 1) % NV;                    // the view whose hand-off arrives: of this tile (v > 0) or the previous one
      constexpr int HS = v == 0 ? 1 - P : P;                   // ... and the set it belongs to
      constexpr int CS = 1 - P;                                // the set of tile k - 1
      constexpr int r0 = v * TY / NV, r1 = (v + 1) * TY / NV;  // the row group of tile k - 1 combined in this stage
      constexpr int NR = r1 - r0 > 0 ? r1 - r0 : 1;
      const bool live = k < my_tiles;                          // k == my_tiles: only the last tile's hand-off and combine are left
      const bool comb = k > 0 && r1 > r0;
      // (1) the hand-off of stage s - 1 and (2) the centre voxels of this stage's rows, requested first: both are consumed at the
      // end of the stage, and the X loads must be OLDER than the halo requests below (vector-memory results return in order: a wait
      // for a younger load would wait for the whole halo of the stage after next).
      // (The hand-off UNCONDITIONALLY: in the very first stage and in the drain stages behind the last tile the buffer holds stale
      // values, which land in slots that are assigned again before they are read -- a conditional update would keep both sets alive
      // across the loop.)
      float4 h[TY];
      const int hb = hread + ((s - 1) & 1) * AB_HBUF;
#pragma unroll
      for (int t = 0; t < TY; ++t) h[t] = *reinterpret_cast<const float4*>(smem + hb + t * 1024);
      // the lane's voxel column of tile k - 1's plane: a buffer descriptor per view at (zo, ty0, x0) (scalar), the lane's constant
      // byte offset (all ones outside the volume: reads zeros, stores nothing), the row as the scalar offset
      const int zo = Tp.z0 + sw, ty0 = Tp.y0;
      const size_t tbase = (((size_t)zo * p.Hi + ty0) * p.Wi + Tp.x0) * 8;
      const int voff = (zo < p.Di && Tp.x0 + r < p.Wi) ? lane_xoff : -1;
      const int vrowb = p.Wi * 32;
      ab_f32x2 xv[NR][NV];
      if (comb) {
#pragma unroll
        for (int n = 0; n < NV; ++n) {
          const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x[n] + tbase), 0, 0x7ffffff0, 0x00020000);
#pragma unroll
          for (int t = r0; t < r1; ++t) xv[t - r0][n] = ab_buffer_load_x2(xrs, (ty0 + t < p.Hi) ? voff : -1, t * vrowb, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (live) {
        // the image of stage s + 1 (requested one stage ago) -> the buffer the multiplying wavefronts read LAST stage
#ifdef ATVS_AB_DEBUG
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ABDBG(5)
#endif
        pf_write(((s + 1) & 1) * AB_BUF);
        ABDBG(0)
        // the halo of stage s + 2
        constexpr int idx = v + 2, dk = idx / NV, vn = idx % NV;
        pf_request(dk == 0 ? Tk : dk == 1 ? Tk1 : Tk2, p.x[vn]);
        ABDBG(1)
      }
#pragma unroll
      for (int t = 0; t < TY; ++t) {
        if constexpr (vp == 0) { ssx[HS][t] = 0.f + h[t].x; ssy[HS][t] = 0.f + h[t].y; }
        else { ssx[HS][t] += h[t].x; ssy[HS][t] += h[t].y; }
        dvx[HS][vp][t] = h[t].z - h[t].x; dvy[HS][vp][t] = h[t].w - h[t].y;
      }
      ABDBG(2)
      if (comb) {
        // ---- combine (aanet_combine_kernel's arithmetic and order) of rows r0..r1-1 of tile k - 1, channels 2q, 2q+1
        const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(p.out + tbase, 0, 0x7ffffff0, 0x00020000);
#pragma unroll
        for (int t = r0; t < r1; ++t) {
          float ux[NV], uy[NV];
#pragma unroll
          for (int n = 0; n < NV; ++n) { ux[n] = dvx[CS][n][t] + ssx[CS][t]; uy[n] = dvy[CS][n][t] + ssy[CS][t]; }
          ab_f32x2 o;
          float ox, oy;
          atvs_aanet_softmax_sum<NV>(ux, [&](int n) __attribute__((always_inline)) { return xv[t - r0][n][0]; }, &ox);
          atvs_aanet_softmax_sum<NV>(uy, [&](int n) __attribute__((always_inline)) { return xv[t - r0][n][1]; }, &oy);
          o[0] = ox; o[1] = oy;
          ab_buffer_store_x2(o, ors, (ty0 + t < p.Hi) ? voff : -1, t * vrowb, 0);
        }
      }
      ABDBG(3)
      if (live) {
        ab_lds_barrier();
        ++s;
      }
      ABDBG(4)
    };
    auto tile_body = [&](auto PP, int k) __attribute__((always_inline)) {
      if (k > my_tiles) return;
      if (NV == 1) Tk2 = pf_tile(k + 2);                       // the stage after next lies two tiles ahead only with one view
      else Tk2 = Tk1;
      ab_static_for<NV>([&](auto VV) __attribute__((always_inline)) { stage_body(PP, VV, k); });
      Tp = Tk;
      Tk = Tk1;
      Tk1 = pf_tile(k + 2);
    };
#pragma unroll 1
    for (int k = 0; k <= my_tiles; k += 2) {
      tile_body(IC<0>{}, k);
      tile_body(IC<1>{}, k + 1);
    }
  }
#ifdef ATVS_AB_DEBUG
  if (lane == 0 && blockIdx.x < 256) {
    dbg_acc[7] = (unsigned long long)nstages;
    for (int i = 0; i < 8; ++i) atvs_dbg_ab[(blockIdx.x * 8 + wave) * 8 + i] = dbg_acc[i];
  }
#endif
}

template <int NV>
int launch_ab(const AbArgs& a, long grid, hipStream_t s) {
  static AtvsAttrOnce lds_once;                   // per kernel instantiation
  if (const int rc_ = atvs_set_max_lds_once(lds_once, reinterpret_cast<const void*>(&aanet_b_kernel<NV>), AB_LDS)) return rc_;
  hipLaunchKernelGGL((aanet_b_kernel<NV>), dim3((unsigned)grid), dim3(512), AB_LDS, s, a);
  return ATVS_OK;
}

}  // namespace

extern "C" int atvs_aanet_b_pack_size(long* packed_bytes) {
  if (!packed_bytes) return ATVS_ERR_NULL;
  *packed_bytes = (long)AB_JC * AB_WSTEP + 16;
  return ATVS_OK;
}

// HOST function.  w_shared, w_unique: TF kernels [3,3,3,8,8] (attention_activation/weight_shared, weight_unique).
// packed[step j][piece][lane = q*16 + row][8 fp16] = piece of w[tap atvs_tap8(j, q)][ci = e][row -> (S | R, channel)]: row 4 g + i of lane group
// g is channel 2 g + (i & 1) of W_shared (i < 2) or W_unique (i >= 2) -- a lane of the kernel then holds S and R of ITS two channels.
// Pieces as atvs_conv_c16b_pack; ATVS_ERR_ARG for a weight beyond fp16's range.
extern "C" int atvs_aanet_b_pack(const float* w_shared, const float* w_unique, unsigned char* packed) {
  if (!w_shared || !w_unique || !packed) return ATVS_ERR_NULL;
  long pb;
  atvs_aanet_b_pack_size(&pb);
  std::memset(packed, 0, (size_t)pb);
  uint16_t* out = reinterpret_cast<uint16_t*>(packed);
  bool fits = true;
  for (int j = 0; j < AB_JC; ++j)
    for (int q = 0; q < 4; ++q) {
      const int tap = atvs_tap8(j, q);
      if (tap > 26) continue;
      for (int row = 0; row < 16; ++row) {
        const int g = row >> 2, i = row & 3, c = 2 * g + (i & 1);
        const float* w = (i < 2) ? w_shared : w_unique;
        for (int e = 0; e < 8; ++e) {
          const float v = w[((size_t)tap * 8 + e) * 8 + c];
          const _Float16 g0 = (_Float16)v, g1 = (_Float16)((v - (float)g0) * AB_RS);
          std::memcpy(&out[(((size_t)j * AB_NP + 0) * 64 + q * 16 + row) * 8 + e], &g0, 2);
          std::memcpy(&out[(((size_t)j * AB_NP + 1) * 64 + q * 16 + row) * 8 + e], &g1, 2);
          const float back = (float)g0;
          fits &= (back - back == 0.f);
        }
      }
    }
  return fits ? ATVS_OK : ATVS_ERR_ARG;
}

extern "C" int atvs_aanet_b_supported(int C, int nv) { return (C == 8 && nv >= 1 && nv <= AB_MAXV) ? 1 : 0; }

// out (D,H,W,8) = sum_n softmax_n((R_n - S_n) + sum_m S_m) * X_n with S_n | R_n = relu(conv3d(X_n, W_shared | W_unique, SAME)):
// the AANet module (reference cnn_wrapper/network.py:282-351,378-408) over nv <= 8 views in one launch (more views: the two-launch form).  x: HOST array of nv
// device pointers, each (D,H,W,8); packed_w: atvs_aanet_b_pack.  Bit for bit atvs_conv_c16b_f32 (shared | unique, ReLU) per
// view followed by atvs_aanet_combine.
extern "C" int atvs_aanet_b_f32(const float* const* x, int nv, const unsigned char* packed_w, float* out, int D, int H, int W,
                                atvs_stream_t stream) {
  if (!x || !packed_w || !out) return ATVS_ERR_NULL;
  if (!atvs_aanet_b_supported(8, nv) || D <= 0 || H <= 0 || W <= 0) return ATVS_ERR_SHAPE;
  if ((double)D * H * W * 8 >= 2147483648.0) return ATVS_ERR_SHAPE;
  AbArgs a;
  for (int n = 0; n < AB_MAXV; ++n) {
    a.x[n] = x[n < nv ? n : 0];
    if (!a.x[n]) return ATVS_ERR_NULL;
  }
  long pb;
  atvs_aanet_b_pack_size(&pb);
  a.wp = packed_w; a.zeros = reinterpret_cast<const float*>(packed_w + (pb - 16));
  a.out = out;
  a.Di = D; a.Hi = H; a.Wi = W;
  a.tiles_y = (H + AB_TY - 1) / AB_TY; a.tiles_x = (W + AB_TX - 1) / AB_TX;
  a.ntiles = ((D + AB_TZ - 1) / AB_TZ) * a.tiles_y * a.tiles_x;
  long grid = a.ntiles < 256 ? a.ntiles : 256;
  grid = (grid + 7) / 8 * 8;
  a.wg = (int)grid;
  hipStream_t st = as_stream(stream);
  int rc = ATVS_ERR_SHAPE;
  switch (nv) {
    case 1: rc = launch_ab<1>(a, grid, st); break;
    case 2: rc = launch_ab<2>(a, grid, st); break;
    case 3: rc = launch_ab<3>(a, grid, st); break;
    case 4: rc = launch_ab<4>(a, grid, st); break;
    case 5: rc = launch_ab<5>(a, grid, st); break;
    case 6: rc = launch_ab<6>(a, grid, st); break;
    case 7: rc = launch_ab<7>(a, grid, st); break;
    case 8: rc = launch_ab<8>(a, grid, st); break;
  }
  if (rc) return rc;
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}
