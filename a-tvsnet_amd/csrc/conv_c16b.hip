// The 8 / 16 -> 16 channel 3x3x3 convolutions of conv_c16.hip on the 16-bit matrix cores with SPLIT operands -- BASELINE.json
// configs[1] names "bf16 conv3d MFMA"; plain bf16 operands miss the 1e-3 depth bar by a factor 200.  Round 4:
// every fp32 operand = TWO fp16 pieces,
//     x = h0 + h1 / 2048,   h0 = f16(x), h1 = f16((x - h0) * 2048)        (22 significant bits for |x| >= 2^-14; the residual is
//                                                                            scaled towards fp16's normal range -- below 2^-14 both
//                                                                            pieces are fp16 subnormals: 2^-36 absolute)
// and THREE products on v_mfma_f32_16x16x32_f16 with fp32 accumulation: h0 g0 into the main accumulator, h0 g1 + h1 g0 into a
// second one scaled by 2^-11 in the epilogue (conv_xb.hip has the measurements: per-layer error against a double sum below
// the fp32 matrix cores' and below round 3's three bf16 pieces / six products).  One K = 32 instruction (16 cycles) covers two
// taps x 16 channels, which takes eight 16x16x4 fp32 instructions (256 cycles); three of them = 48 cycles.
//
// Structure = conv_c16.hip's 8-channel form (32-byte voxels, two taps per K step, lane half q >> 1 picks the tap): here a
// voxel of one PIECE image is 16 channels x 2 bytes = 32 bytes, the two piece images lie IMG bytes apart, the split is
// done once per staged element on its way into LDS, the packed weights (two pieces per K step, split on the host) are
// resident in LDS.  Cout = 16, Cin = 16 (conv_b*_1_1, global_refine_3dconv1_1: two taps per K step) or Cin = 8 (the AANet
// modules' shared | unique convolution: four taps per K step, 16-byte voxels).
#include <cstring>
#include <type_traits>
#include <utility>

#include "conv_common.h"

namespace {

constexpr int B16_TZ = 4, B16_TY = 8, B16_TX = 16;
constexpr int B16_HZ = B16_TZ + 2, B16_HY = B16_TY + 2, B16_HX = B16_TX + 2;

// CIN input channels (8 or 16): a K = 32 instruction covers TPS = 32 / CIN taps; lane group q carries tap TPS*j + q / (4/TPS)
// and channels 8 * (q % (4/TPS)) .. + 7 of it.
template <int CIN>
struct B16 {
  static_assert(CIN == 8 || CIN == 16, "8 or 16 input channels");
  static constexpr int TPS = 32 / CIN;                          // taps per K step: 4 / 2
  static constexpr int LPT = 4 / TPS;                           // lane groups per tap: 1 / 2
  static constexpr int VB = CIN * 2;                            // bytes per voxel of one piece image: 16 / 32
  static constexpr int ROWB = B16_HX * VB;
  static constexpr int IMG = B16_HZ * B16_HY * ROWB;            // 17,280 / 34,560 bytes per piece
  static constexpr int C4 = CIN / 4;                            // float4 slots per voxel of the fp32 halo
  static constexpr int SLOTS = B16_HZ * B16_HY * B16_HX * C4;
  static constexpr int MAXS = (SLOTS + 255) / 256;              // 9 / 17 per thread
  static constexpr int JC = (27 + TPS - 1) / TPS;               // K steps: 7 / 14 (taps past 26 = zero weights)
  static_assert(MAXS <= 2 * JC, "two halo slots per K step");
  // byte displacement of tap t from halo voxel (wave, 0, r): (kd, kh) rows + kw voxels
  static constexpr int clamp26(int t) { return t < 26 ? t : 26; }
  // the tap of lane-group slot i of K step j (fragment ADDRESS: a tap past 26 has zero weights and re-reads tap 26): Cin = 16 walks the
  // taps in order, two per step; Cin = 8 uses the shared order of aanet_b.hip (atvs_tap8, conv_common.h)
  static constexpr int tap(int j, int i) { return clamp26(CIN == 8 ? atvs_tap8(j, i) : TPS * j + i); }
  static constexpr int disp(int t) { return ((t / 9) * B16_HY + (t / 3) % 3) * ROWB + (t % 3) * VB; }
};
constexpr int B16_NP = 2;                                     // operand pieces
constexpr int B16_WSTEP = B16_NP * 1024;                      // bytes of packed weights per K step (2 pieces x 64 lanes x 16 B)
constexpr float B16_RS = 2048.f, B16_IRS = 1.f / 2048.f;      // scale of the residual piece and its inverse

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct B16Args {
  const float* x;
  // PRO forms: the input is formed while the halo is staged, never written --
  //   1: x_in = t(x, pa, bit 0)                     (a pending batch norm [+ ReLU]: atvs_bn_apply's arithmetic)
  //   2: x_in = t(x, pa, bit 0) + t(x2, pb, bit 1)  (the U-Net's skip sum: atvs_bn_add's arithmetic and order)
  // with t(v, par, relu) = par ? relu?((v - mean) * scale + beta) : v
  const float* x2;
  const float* pa;             // (groups, 3, Cin) or null (that term is a finished tensor)
  const float* pb;
  int relu_mask;
  const unsigned char* wp;     // packed fp16 pieces (atvs_conv_c16b_pack)
  const float* zeros;          // 16 zero bytes
  const float* bias;
  float* y;
  double* stats;
  int Di, Hi, Wi;
  int ldy, ycoff;
  int tiles_y, tiles_x, ntiles;
  int wg;
  long gx, gy;
};

template <int N>
using IC = std::integral_constant<int, N>;
template <class F, int... I>
__device__ __forceinline__ void b16_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(IC<I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void b16_static_for(F&& f) {
  b16_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// the two fp16 pieces of four fp32 values: h0 = f16(x), h1 = f16((x - h0) * 2^11)
__device__ __forceinline__ void b16_split(const float4& v, f16x4* p0, f16x4* p1) {
  // atvs_split2_f16 (common.h): five vector instructions per two values instead of the 8-9 of the C form, the same values
  uint2 a, b;
  atvs_split4_f16(v, &a, &b);
  *p0 = __builtin_bit_cast(f16x4, a);
  *p1 = __builtin_bit_cast(f16x4, b);
}

template <int CIN, bool RELU, int PRO>
__global__ __launch_bounds__(256, 1) void conv_c16b_kernel(B16Args p) {
  // own the SIMD's whole register file (512 per lane): no wavefront of ANOTHER kernel runs beside this one's bf16 MFMAs --
  // beside them other kernels' wavefronts computed wrong lane quarters (tools_dev/micro/pk_beside_mfma.hip)
  asm volatile("" ::: "v255", "a255");
  using K = B16<CIN>;
  constexpr int TY = B16_TY, HY = B16_HY, MAXS = K::MAXS, JC = K::JC;
  constexpr int B16_IMG = K::IMG, B16_VB = K::VB, B16_ROWB = K::ROWB, B16_SLOTS = K::SLOTS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;

  // packed weights -> LDS, once
  {
    const float4* src = reinterpret_cast<const float4*>(p.wp);
    float4* dst = reinterpret_cast<float4*>(smem + B16_NP * B16_IMG);
    for (int i = tid; i < JC * (B16_WSTEP / 16); i += 256) dst[i] = src[i];
  }
  // this lane's B fragment (8 consecutive channels of a voxel) at halo voxel (wave, 0, r), tap (0,0,0)
  const int fbase = ((wave * HY) * B16_HX + r) * B16_VB + (q % K::LPT) * 16;
  const int wbase = B16_NP * B16_IMG + lane * 16;

  // halo slots: float4 = channels 4 c4 .. of a voxel -> 8 bytes at (voxel, c4) of each piece image
  int goff[MAXS], laddr[MAXS];
  unsigned pg[MAXS];
#pragma unroll
  for (int i = 0; i < MAXS; ++i) {
    int s = tid + i * 256;
    const bool live = s < B16_SLOTS;
    s = min(s, B16_SLOTS - 1);
    const int c4 = s % K::C4, v = s / K::C4;
    const int xx = v % B16_HX, v2 = v / B16_HX;
    const int yy = v2 % HY, zz = v2 / HY;
    goff[i] = ((zz * p.Hi + yy) * p.Wi + xx) * CIN + c4 * 4;
    laddr[i] = ((zz * HY + yy) * B16_HX + xx) * B16_VB + c4 * 8;
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
  auto tile_origin = [&](int k, int* z0, int* y0, int* x0) __attribute__((always_inline)) {
    int tl = xcd * per_xcd + tslot + k * slots_per_xcd;
    int bx = tl % p.tiles_x;
    int rest = tl / p.tiles_x;
    *x0 = bx * B16_TX;
    *y0 = (rest % p.tiles_y) * TY;
    *z0 = (rest / p.tiles_y) * B16_TZ;
  };
  struct PfTile {
    int org;
    unsigned lo, hi1;
  };
  auto pf_tile = [&](int k) __attribute__((always_inline)) {
    PfTile T;
    int z0, y0, x0;
    tile_origin(k, &z0, &y0, &x0);
    const int gz0 = z0 - 1, gy0 = y0 - 1, gx0 = x0 - 1;
    T.org = ((gz0 * p.Hi + gy0) * p.Wi + gx0) * CIN;
    T.lo = (unsigned)(gz0 < 0) | ((unsigned)(gy0 < 0) << 8) | ((unsigned)(gx0 < 0) << 16);
    T.hi1 = (unsigned)(min(p.Di - 1 - gz0, 0x7e) + 1) | ((unsigned)(min(p.Hi - 1 - gy0, 0x7e) + 1) << 8) |
            ((unsigned)(min(p.Wi - 1 - gx0, 0x7e) + 1) << 16);
    return T;
  };
  float4 pf[MAXS], pf2[PRO == 2 ? MAXS : 1];
  const float* __restrict__ xg2 = (PRO == 2) ? p.x2 + (size_t)grp * p.gx : nullptr;
  auto pf_slot = [&](const PfTile& T, int i) __attribute__((always_inline)) {
    const unsigned t1 = pg[i] - T.lo;
    const unsigned t2 = T.hi1 + ~pg[i];
    const bool ok = ((t1 & t2) & 0x808080u) == 0x808080u;
    pf[i] = ld4(ok ? (xg + (T.org + goff[i])) : p.zeros);
    if (PRO == 2) pf2[i] = ld4(ok ? (xg2 + (T.org + goff[i])) : p.zeros);
  };
  // PRO: the batch-norm rows of this thread's four channels (c4 = tid % C4 in every slot)
  float4 bnm[2], bns[2], bnb[2];
  if (PRO) {
    const float* pr[2] = {p.pa, p.pb};
#pragma unroll
    for (int k = 0; k < (PRO == 2 ? 2 : 1); ++k) {
      const float* q3 = pr[k] ? pr[k] + (size_t)grp * 3 * CIN + (tid % K::C4) * 4 : nullptr;
      bnm[k] = q3 ? ld4(q3) : make_float4(0.f, 0.f, 0.f, 0.f);
      bns[k] = q3 ? ld4(q3 + CIN) : make_float4(1.f, 1.f, 1.f, 1.f);
      bnb[k] = q3 ? ld4(q3 + 2 * CIN) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  auto pro_term = [&](const float4& v, int k, const float* par) __attribute__((always_inline)) {
    float4 o = v;
    if (par) {
      o = atvs_bn4(v, bns[k], atvs_bn_shift4(bnm[k], bns[k], bnb[k]));
      const float fl = ((p.relu_mask >> k) & 1) ? 0.f : -INFINITY;      // ReLU or nothing, branch-free
      o.x = fmaxf(o.x, fl); o.y = fmaxf(o.y, fl); o.z = fmaxf(o.z, fl); o.w = fmaxf(o.w, fl);
    }
    return o;
  };

  // slot i of a tile -> its two fp16 pieces, kept in pf[i] (p0 | p1) until the tile's LDS writes.  The slots of the tile in flight are
  // transformed behind MFMAs of the K loop, LAG slots (LAG / 2 steps) after their loads were issued -- in front of the LDS writes
  // all four wavefronts wait for it (conv3d_s2b.hip)
  constexpr int LAG = 8, NIN = (2 * JC - LAG) < MAXS ? (2 * JC - LAG) : MAXS;
  auto xform_slot = [&](int i, const PfTile& TT) __attribute__((always_inline)) {
    float4 a = pf[i];
    if (PRO) {
      a = pro_term(a, 0, p.pa);
      if (PRO == 2) {
        const float4 b = pro_term(pf2[i], 1, p.pb);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      }
      // a halo slot outside the volume: its padding stays zero (it is not zero after a batch norm)
      const unsigned t1 = pg[i] - TT.lo;
      const unsigned t2 = TT.hi1 + ~pg[i];
      const bool in = ((t1 & t2) & 0x808080u) == 0x808080u;
      a = make_float4(in ? a.x : 0.f, in ? a.y : 0.f, in ? a.z : 0.f, in ? a.w : 0.f);
    }
    uint2 h0, h1;
    atvs_split4_f16(a, &h0, &h1);
    pf[i] = make_float4(__uint_as_float(h0.x), __uint_as_float(h0.y), __uint_as_float(h1.x), __uint_as_float(h1.y));
  };

  f32x2 ssum2[2], ssq2[2];
  ssum2[0] = ssum2[1] = ssq2[0] = ssq2[1] = (f32x2){0.f, 0.f};
  f32x4 acc[TY], accx[TY];             // h0 g0 | (h0 g1 + h1 g0) * 2^11
  const float4 bv = p.bias ? ld4(p.bias + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  const unsigned ybytes = (unsigned)(p.gy * 4);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(yg, 0, ybytes, 0x00020000);

  PfTile Tn = pf_tile(0);                  // the tile whose halo is in flight
  if (my_tiles > 0) {
#pragma unroll
    for (int i = 0; i < MAXS; ++i) pf_slot(Tn, i);
#pragma unroll
    for (int i = 0; i < NIN; ++i) xform_slot(i, Tn);      // (later tiles: inside the previous tile's K loop)
  }

  for (int k = 0; k < my_tiles; ++k) {
#pragma unroll
    for (int t = 0; t < TY; ++t) acc[t] = accx[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();                       // every wavefront is done reading the previous tile's images
    // the staged halo's two fp16 pieces into LDS
    const PfTile Tc = Tn;
#pragma unroll
    for (int i = NIN; i < MAXS; ++i) xform_slot(i, Tc);
#pragma unroll
    for (int i = 0; i < MAXS; ++i) {
      if (i < MAXS - 1 || tid + i * 256 < B16_SLOTS) {
        *reinterpret_cast<uint2*>(smem + laddr[i]) = make_uint2(__float_as_uint(pf[i].x), __float_as_uint(pf[i].y));
        *reinterpret_cast<uint2*>(smem + B16_IMG + laddr[i]) = make_uint2(__float_as_uint(pf[i].z), __float_as_uint(pf[i].w));
      }
    }
    __syncthreads();

    Tn = pf_tile(min(k + 1, my_tiles - 1));
    const PfTile T = Tn;      // last tile: harmless re-read of its own halo

    // ---- K loop: 14 steps of two taps x 16 channels, each in two phases -- input piece h0 with both weight pieces (16 MFMAs),
    // h1 with g0 (8): the fragments of ONE input piece are live at a time (the next phase's are requested behind this
    // phase's first MFMAs), the two weight pieces of a step are requested one step ahead
    f16x8 Bq[2][TY], A[2][B16_NP];
    auto request_B = [&](auto PH) __attribute__((always_inline)) {
      constexpr int ph = decltype(PH)::value, j = ph / B16_NP, pc = ph % B16_NP;
      // this lane group's tap of the step (taps past 26 have zero weights: re-read tap 26's fragment)
      constexpr int tA = K::tap(j, 0), tB = K::tap(j, 1), tC = K::tap(j, 2), tD = K::tap(j, 3);
      int a;
      if constexpr (K::TPS == 2) a = fbase + ((q >> 1) ? K::disp(tB) : K::disp(tA));
      else a = fbase + ((q & 2) ? ((q & 1) ? K::disp(tD) : K::disp(tC)) : ((q & 1) ? K::disp(tB) : K::disp(tA)));
#pragma unroll
      for (int t = 0; t < TY; ++t)
        Bq[ph & 1][t] = *reinterpret_cast<const f16x8*>(smem + pc * B16_IMG + a + t * B16_ROWB);
    };
    auto request_B1 = [&](auto PH, auto TT) __attribute__((always_inline)) {      // row t of phase ph
      constexpr int ph = decltype(PH)::value, j = ph / B16_NP, pc = ph % B16_NP, t = decltype(TT)::value;
      constexpr int tA = K::tap(j, 0), tB = K::tap(j, 1), tC = K::tap(j, 2), tD = K::tap(j, 3);
      int a;
      if constexpr (K::TPS == 2) a = fbase + ((q >> 1) ? K::disp(tB) : K::disp(tA));
      else a = fbase + ((q & 2) ? ((q & 1) ? K::disp(tD) : K::disp(tC)) : ((q & 1) ? K::disp(tB) : K::disp(tA)));
      Bq[ph & 1][t] = *reinterpret_cast<const f16x8*>(smem + pc * B16_IMG + a + t * B16_ROWB);
    };
    auto request_A = [&](auto JT) __attribute__((always_inline)) {
      constexpr int j = decltype(JT)::value;
#pragma unroll
      for (int pc = 0; pc < B16_NP; ++pc) A[j & 1][pc] = *reinterpret_cast<const f16x8*>(smem + wbase + j * B16_WSTEP + pc * 1024);
    };
    request_A(IC<0>{});
    request_B(IC<0>{});
    asm volatile("" ::: "memory");
    // every memory instruction behind ONE MFMA (tools_dev/micro/mfma_bf16_rate.hip: issued in a bunch at the top of a phase they
    // cost matrix-core time): MFMA m of a phase = (weight piece jw, row t); behind the first eight the next phase's fragments,
    // then (first phase of a step) the next step's weights, then a halo slot of the next stage
    b16_static_for<B16_NP * JC>([&](auto PH) __attribute__((always_inline)) {
      constexpr int ph = decltype(PH)::value, j = ph / B16_NP, pc = ph % B16_NP;
      b16_static_for<(2 - pc) * TY>([&](auto M) __attribute__((always_inline)) {
        constexpr int m = decltype(M)::value, jw = m / TY, t = m % TY;
        if constexpr (pc == 0 && jw == 0) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[j & 1][0], Bq[ph & 1][t], acc[t], 0, 0, 0);
        else accx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[j & 1][pc == 0 ? 1 : 0], Bq[ph & 1][t], accx[t], 0, 0, 0);
        if constexpr (m < TY) {
          if constexpr (ph + 1 < B16_NP * JC) request_B1(IC<ph + 1>{}, IC<m>{});
        } else if constexpr (m < TY + B16_NP) {
          if constexpr (j + 1 < JC) A[(j + 1) & 1][m - TY] = *reinterpret_cast<const f16x8*>(smem + wbase + (j + 1) * B16_WSTEP + (m - TY) * 1024);
        } else if constexpr (m == TY + 3 || m == TY + 6) {
          constexpr int slot = 2 * j + (m == TY + 6 ? 1 : 0);
          if constexpr (slot < MAXS) pf_slot(T, slot);
        } else if constexpr (m == TY + 4 || m == TY + 7) {
          constexpr int slot = 2 * j + (m == TY + 7 ? 1 : 0) - LAG;
          if constexpr (slot >= 0 && slot < NIN) xform_slot(slot, T);
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    static_assert(2 * JC >= MAXS, "two halo slots per K step");

    // ---- epilogue (conv_c16.hip): this lane holds channels 4q..4q+3 of voxel (z0 + wave, y0 + t, x0 + r)
    int tz0, ty0, tx0;
    tile_origin(k, &tz0, &ty0, &tx0);
    const int zo = tz0 + wave, xo = tx0 + r;
    const bool evox_ok = zo < p.Di && xo < p.Wi;
    const unsigned erow = (unsigned)p.Wi * p.ldy;
    const unsigned eo = (((unsigned)zo * p.Hi + ty0) * p.Wi + xo) * p.ldy + p.ycoff + q * 4;
    const unsigned vo_ok = evox_ok ? eo * 4u : ybytes;
    b16_static_for<TY>([&](auto TT) __attribute__((always_inline)) {
      constexpr int t = decltype(TT)::value;
      const bool row_ok = ty0 + t < p.Hi;
      const bool ok = evox_ok && row_ok;
      float a0 = (acc[t][0] + accx[t][0] * B16_IRS) + bv.x, a1 = (acc[t][1] + accx[t][1] * B16_IRS) + bv.y;
      float a2 = (acc[t][2] + accx[t][2] * B16_IRS) + bv.z, a3 = (acc[t][3] + accx[t][3] * B16_IRS) + bv.w;
      if (RELU) {
        a0 = (a0 < 0.f) ? 0.f : a0; a1 = (a1 < 0.f) ? 0.f : a1;
        a2 = (a2 < 0.f) ? 0.f : a2; a3 = (a3 < 0.f) ? 0.f : a3;
      }
      const u32x4 bits = {__builtin_bit_cast(unsigned, a0), __builtin_bit_cast(unsigned, a1),
                          __builtin_bit_cast(unsigned, a2), __builtin_bit_cast(unsigned, a3)};
      __builtin_amdgcn_raw_buffer_store_b128(bits, yrsrc, row_ok ? vo_ok : ybytes, (t * erow) * 4u, ATVS_BUF_NT);
      f32x2 lo = {ok ? a0 : 0.f, ok ? a1 : 0.f}, hi = {ok ? a2 : 0.f, ok ? a3 : 0.f};
      ssum2[0] += lo;
      ssum2[1] += hi;
      ssq2[0] = __builtin_elementwise_fma(lo, lo, ssq2[0]);
      ssq2[1] = __builtin_elementwise_fma(hi, hi, ssq2[1]);
    });
  }

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
      if (r == 0) {
        s_red[(wave * 2 + 0) * 16 + q * 4 + kk] = a;
        s_red[(wave * 2 + 1) * 16 + q * 4 + kk] = bq;
      }
    }
    __syncthreads();
    if (tid < 32) {
      const int which = tid / 16, col = tid % 16;
      p.stats[((size_t)blockIdx.x * 2 + which) * 16 + col] =
          (s_red[(0 * 2 + which) * 16 + col] + s_red[(1 * 2 + which) * 16 + col]) +
          (s_red[(2 * 2 + which) * 16 + col] + s_red[(3 * 2 + which) * 16 + col]);
    }
  }
}


}  // namespace

// Bytes of the packed form of a TF kernel [3,3,3,Cin,16] (Cin 8 or 16) for atvs_conv_c16b_f32 (+ 16 trailing zero bytes).
extern "C" int atvs_conv_c16b_pack_size(int Cin, long* packed_bytes) {
  if (!packed_bytes) return ATVS_ERR_NULL;
  if (Cin != 8 && Cin != 16) return ATVS_ERR_SHAPE;
  *packed_bytes = (long)(Cin == 8 ? B16<8>::JC : B16<16>::JC) * B16_WSTEP + 16;
  return ATVS_OK;
}

// HOST function.  packed[step j][piece][lane = q*16 + co][8 fp16] = piece of w[tap][ci = (q % LPT)*8 + e][co], tap = TPS*j + q / LPT
// for Cin = 16 (TPS = 32 / Cin taps per step, LPT = 4 / TPS lane groups per tap) and atvs_tap8(j, q) for Cin = 8 (conv_common.h); zero
// for taps past 26; pieces g0 = f16(w),
// g1 = f16((w - g0) * 2048), round to nearest even.  ATVS_ERR_ARG if a weight does not fit fp16's range (|w| > 65504).
extern "C" int atvs_conv_c16b_pack(const float* w, int Cin, unsigned char* packed) {
  if (!w || !packed) return ATVS_ERR_NULL;
  long pb;
  int rc = atvs_conv_c16b_pack_size(Cin, &pb);
  if (rc) return rc;
  std::memset(packed, 0, (size_t)pb);
  uint16_t* out = reinterpret_cast<uint16_t*>(packed);
  const int TPS = 32 / Cin, LPT = 4 / TPS, JC = (27 + TPS - 1) / TPS;
  bool fits = true;
  for (int j = 0; j < JC; ++j)
    for (int q = 0; q < 4; ++q) {
      const int tap = Cin == 8 ? atvs_tap8(j, q) : TPS * j + q / LPT;
      if (tap > 26) continue;
      for (int co = 0; co < 16; ++co)
        for (int e = 0; e < 8; ++e) {
          const int ci = (q % LPT) * 8 + e;
          const float v = w[((size_t)tap * Cin + ci) * 16 + co];
          const _Float16 g0 = (_Float16)v, g1 = (_Float16)((v - (float)g0) * B16_RS);
          std::memcpy(&out[(((size_t)j * B16_NP + 0) * 64 + q * 16 + co) * 8 + e], &g0, 2);
          std::memcpy(&out[(((size_t)j * B16_NP + 1) * 64 + q * 16 + co) * 8 + e], &g1, 2);
          const float back = (float)g0;
          fits &= (back - back == 0.f);
        }
    }
  return fits ? ATVS_OK : ATVS_ERR_ARG;
}

namespace {
template <int CIN, bool RELU, int PRO = 0>
int launch_c16b(const B16Args& a, long grid, hipStream_t s) {
  const size_t lds = B16_NP * (size_t)B16<CIN>::IMG + (size_t)B16<CIN>::JC * B16_WSTEP;
  static AtvsAttrOnce lds_once;                   // per kernel instantiation (this function is a template / has one kernel)
  if (const int rc_ = atvs_set_max_lds_once(lds_once, reinterpret_cast<const void*>(conv_c16b_kernel<CIN, RELU, PRO>), 160 * 1024)) return rc_;
  hipLaunchKernelGGL((conv_c16b_kernel<CIN, RELU, PRO>), dim3((unsigned)grid), dim3(256), lds, s, a);
  return ATVS_OK;
}
}  // namespace

namespace {
int c16b_launch(const float* x, const float* x2, const float* pa, const float* pb, int relu_mask, int pro,
                const unsigned char* packed_w, const float* bias, float* y, double* stats_partial, int groups, int D, int H,
                int W, int Cin, int ldy, int y_coff, int relu, atvs_stream_t stream) {
  if (!x || !packed_w || !y) return ATVS_ERR_NULL;
  if (groups <= 0 || D <= 0 || H <= 0 || W <= 0 || (Cin != 8 && Cin != 16)) return ATVS_ERR_SHAPE;
  if (y_coff < 0 || y_coff + 16 > ldy || (ldy % 4) || (y_coff % 4)) return ATVS_ERR_SHAPE;
  if ((double)D * H * W * Cin >= 2147483648.0 || (double)D * H * W * ldy * 4.0 >= 4294967296.0) return ATVS_ERR_SHAPE;
  B16Args a;
  long pb_;
  atvs_conv_c16b_pack_size(Cin, &pb_);
  a.x = x; a.x2 = x2; a.pa = pa; a.pb = pb; a.relu_mask = relu_mask;
  a.wp = packed_w; a.zeros = reinterpret_cast<const float*>(packed_w + (pb_ - 16));
  a.bias = bias; a.y = y; a.stats = stats_partial;
  a.Di = D; a.Hi = H; a.Wi = W; a.ldy = ldy; a.ycoff = y_coff;
  a.tiles_y = (H + B16_TY - 1) / B16_TY; a.tiles_x = (W + B16_TX - 1) / B16_TX;
  a.ntiles = ((D + B16_TZ - 1) / B16_TZ) * a.tiles_y * a.tiles_x;
  const long blocks = atvs_conv_c16_grid(D, H, W, groups);
  a.wg = (int)blocks;
  a.gx = (long)D * H * W * Cin; a.gy = (long)D * H * W * ldy;
  const long grid = blocks * groups;
  if (grid > 0x7fffffffL) return ATVS_ERR_SHAPE;
  hipStream_t st = as_stream(stream);
  int rc;
  if (pro == 1) rc = relu ? launch_c16b<16, true, 1>(a, grid, st) : launch_c16b<16, false, 1>(a, grid, st);
  else if (pro == 2) rc = relu ? launch_c16b<16, true, 2>(a, grid, st) : launch_c16b<16, false, 2>(a, grid, st);
  else if (Cin == 8) rc = relu ? launch_c16b<8, true>(a, grid, st) : launch_c16b<8, false>(a, grid, st);
  else rc = relu ? launch_c16b<16, true>(a, grid, st) : launch_c16b<16, false>(a, grid, st);
  if (rc) return rc;
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}
}  // namespace

// y (G,D,H,W,ldy)[..., y_coff : y_coff + 16] = conv3d(x (G,D,H,W,Cin), w [3,3,3,Cin,16], stride 1, SAME) (+ bias, ReLU), Cin 8
// or 16, with split-fp16 operands (fp32-class results; rounding differs from atvs_conv_c16_f32).  Grid / statistics rows =
// atvs_conv_c16_grid.
extern "C" int atvs_conv_c16b_f32(const float* x, const unsigned char* packed_w, const float* bias, float* y,
                                  double* stats_partial, int groups, int D, int H, int W, int Cin, int ldy, int y_coff, int relu,
                                  atvs_stream_t stream) {
  return c16b_launch(x, nullptr, nullptr, nullptr, 0, 0, packed_w, bias, y, stats_partial, groups, D, H, W, Cin, ldy, y_coff, relu,
                     stream);
}

// The same convolution (Cin = 16) of an input that is never written: x_in = t(x0, params0, bit 0) [+ t(x1, params1, bit 1)] with
// t(v, par, relu) = par ? relu?((v - mean) * scale + beta) : v -- atvs_bn_apply's arithmetic for one term (a pending batch norm
// applied while the halo is staged), atvs_bn_add's arithmetic and order for two (the U-Net's skip sum in front of
// conv_b{1,2}_1_1: conv_b*_1_1_concat, reference cnn_wrapper/atvsnet.py:45-46,75-76,140-141,170-171).  params_i:
// (groups,3,16) or NULL (that term is a finished tensor); x1 NULL: one term.  Results bit for bit those of atvs_bn_apply /
// atvs_bn_add followed by atvs_conv_c16b_f32.
extern "C" int atvs_conv_c16b_sum_supported(int Cin) { return Cin == 16 ? 1 : 0; }

extern "C" int atvs_conv_c16b_sum_f32(const float* x0, const float* params0, const float* x1, const float* params1, int relu_mask,
                                      const unsigned char* packed_w, const float* bias, float* y, double* stats_partial,
                                      int groups, int D, int H, int W, int Cin, int ldy, int y_coff, int relu,
                                      atvs_stream_t stream) {
  if (!x0) return ATVS_ERR_NULL;
  if (!atvs_conv_c16b_sum_supported(Cin)) return ATVS_ERR_SHAPE;
  if (!x1 && params1) return ATVS_ERR_ARG;
  if (!x1 && !params0) return ATVS_ERR_ARG;          // nothing to form: atvs_conv_c16b_f32
  return c16b_launch(x0, x1, params0, params1, relu_mask, x1 ? 2 : 1, packed_w, bias, y, stats_partial, groups, D, H, W, Cin, ldy,
                     y_coff, relu, stream);
}
