// 3x3x3 SAME stride-1 convolutions with 16 k input and 32 / 64 output channels on the bf16 matrix cores with SPLIT operands
// (gfx950): the quarter- and eighth-resolution layers of the 3-D U-Nets (conv_b*_2_1: 32 -> 32, conv_b*_3_1: 64 -> 64,
// global_refine_3dconv{2,3}_1; /root/reference/cnn_wrapper/atvsnet.py StackedUNet / CostVolRefineNet, layer code
// network.py:172-215), which ran on the fp32 matrix cores at 60-90 TFLOP/s (conv_c16.hip / conv_mfma.hip).  Arithmetic of
// conv_c16b.hip (round 4): every fp32 operand = two fp16 pieces (h0, h1 = f16((x - h0) * 2048)), three products on
// v_mfma_f32_16x16x32_f16, the cross terms in a second accumulator scaled by 2^-11 in the epilogue.
//
// Structure = conv_c16b.hip's 16-channel form generalised: tile 4(z) x 8(y) x 16(x), wavefront w owns plane z0 + w, the input
// is staged in 16-channel chunks as two piece images (34.5 KB each, single-buffered, the next stage's halo waits in
// registers), a K = 32 step = two taps x 16 channels (14 steps per chunk); the wavefront holds ALL output channels of its
// voxels (2 or 4 accumulator tiles per row), computed two tiles at a time over the same staged image; weight pieces stream
// from L2 one step ahead (they do not fit in LDS beside the images: 84 KB per chunk and pair of tiles).
#include <cstring>
#include <type_traits>
#include <utility>

#include "conv_common.h"

extern "C" long atvs_conv_c16_grid(int D, int H, int W, int groups);

namespace {

constexpr int C3B_TZ = 4, C3B_TY = 8, C3B_TX = 16;
constexpr int C3B_HZ = C3B_TZ + 2, C3B_HY = C3B_TY + 2, C3B_HX = C3B_TX + 2;
constexpr int C3B_VB = 32;                                     // bytes per voxel of one piece image: 16 channels
constexpr int C3B_ROWB = C3B_HX * C3B_VB;
constexpr int C3B_IMG = C3B_HZ * C3B_HY * C3B_ROWB;            // 34,560 bytes per piece
constexpr int C3B_SLOTS = C3B_HZ * C3B_HY * C3B_HX * 4;        // float4 slots of the fp32 halo of a chunk
constexpr int C3B_MAXS = (C3B_SLOTS + 255) / 256;              // 17 per thread
constexpr int C3B_JC = 14;                                     // K steps per chunk: taps 2 j, 2 j + 1 (tap 27 = zero weights)
constexpr int C3B_NP = 2;                                      // operand pieces
constexpr float C3B_RS = 2048.f, C3B_IRS = 1.f / 2048.f;
static_assert(C3B_MAXS <= C3B_NP * C3B_JC, "one halo slot per phase of the K loop");
constexpr int c3b_clamp26(int t) { return t < 26 ? t : 26; }
constexpr int c3b_disp(int t) { return ((t / 9) * C3B_HY + (t / 3) % 3) * C3B_ROWB + (t % 3) * C3B_VB; }

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct C3bArgs {
  const float* x;
  // NORM form: x is a raw convolution output whose batch norm [+ ReLU] is applied while the halo is staged (atvs_bn_apply's
  // arithmetic: relu?((v - mean) * scale + beta)); in_par (groups, 3, Cin).  (A two-term skip sum on load as conv_c16b.hip has
  // it -- a second halo in flight -- spilled 158 registers here: two output tiles' accumulators fill the file.)
  const float* in_par;
  int in_relu;
  const f16x8* wp;             // packed fp16 pieces (atvs_conv3d_b_pack)
  const float* zeros;          // 16 zero bytes
  const float* bias;
  float* y;
  double* stats;
  int Di, Hi, Wi, Cin;
  int ldy, ycoff;
  int nchunk;
  int tiles_y, tiles_x, ntiles;
  int wg;
  int relu;
  long gx, gy;
  int nhalf;                   // 1: Cout = 32; 2: Cout = 64 as two 32-channel halves, blocks [0, bh) and [bh, 2 bh) of one launch
  int bh;                      // blocks per half = wg * groups
};

template <int N>
using IC = std::integral_constant<int, N>;
template <class F, int... I>
__device__ __forceinline__ void c3b_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(IC<I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void c3b_static_for(F&& f) {
  c3b_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ void c3b_split(const float4& v, f16x4* p0, f16x4* p1) {
  // atvs_split2_f16 (common.h): five vector instructions per two values instead of the 8-9 of the C form, the same values
  uint2 a, b;
  atvs_split4_f16(v, &a, &b);
  *p0 = __builtin_bit_cast(f16x4, a);
  *p1 = __builtin_bit_cast(f16x4, b);
}

// NT = 2 output tiles (32 channels) per workgroup.  Cout = 64 runs as TWO 32-channel halves in one launch (blocks [bh, 2 bh)
// compute channels 32..63: the input is staged twice, but four tiles' main + cross accumulators -- 256 registers -- do not
// fit a wavefront (294 spilled registers), and at eighth resolution the launch had 192 tiles for 256 CUs anyway).
template <int NT, bool NORM>
__global__ __launch_bounds__(256, 1) void conv3d_b_kernel(C3bArgs p) {
  // own the SIMD's whole register file (512 per lane): no wavefront of ANOTHER kernel runs beside this one's bf16 MFMAs --
  // beside them other kernels' wavefronts computed wrong lane quarters (tools_dev/micro/pk_beside_mfma.hip)
  asm volatile("" ::: "v255", "a255");
  constexpr int TY = C3B_TY, HY = C3B_HY, MAXS = C3B_MAXS, JC = C3B_JC, NH = NT / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;

  // this lane's B fragment (channels 8 (q & 1) .. of a voxel) at halo voxel (wave, 0, r), tap (0,0,0); lane half q >> 1 = tap
  const int fbase = ((wave * HY) * C3B_HX + r) * C3B_VB + (q & 1) * 16;

  int goff[MAXS], laddr[MAXS];
  unsigned pg[MAXS];
#pragma unroll
  for (int i = 0; i < MAXS; ++i) {
    int s = tid + i * 256;
    const bool live = s < C3B_SLOTS;
    s = min(s, C3B_SLOTS - 1);
    const int c4 = s & 3, v = s >> 2;
    const int xx = v % C3B_HX, v2 = v / C3B_HX;
    const int yy = v2 % HY, zz = v2 / HY;
    goff[i] = ((zz * p.Hi + yy) * p.Wi + xx) * p.Cin + c4 * 4;
    laddr[i] = ((zz * HY + yy) * C3B_HX + xx) * C3B_VB + c4 * 8;
    pg[i] = 0x808080u | (unsigned)(live ? zz : 0x7f) | ((unsigned)yy << 8) | ((unsigned)xx << 16);
  }

  const int G = p.wg;
  const int hf = (int)blockIdx.x / p.bh, bidx = (int)blockIdx.x - hf * p.bh;      // output-channel half, block within it
  const int NTT = NT * p.nhalf;                                                    // tiles in the packed weights
  const int grp = bidx / p.wg, lbk = bidx - grp * p.wg;
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
    *x0 = bx * C3B_TX;
    *y0 = (rest % p.tiles_y) * TY;
    *z0 = (rest / p.tiles_y) * C3B_TZ;
  };
  struct PfTile {
    int org;
    unsigned lo, hi1;
  };
  auto pf_tile = [&](int stage) __attribute__((always_inline)) {
    PfTile T;
    const int k = stage / p.nchunk, ch = stage - k * p.nchunk;
    int z0, y0, x0;
    tile_origin(k, &z0, &y0, &x0);
    const int gz0 = z0 - 1, gy0 = y0 - 1, gx0 = x0 - 1;
    T.org = ((gz0 * p.Hi + gy0) * p.Wi + gx0) * p.Cin + ch * 16;
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
    pf[i] = ld4(ok ? (xg + (T.org + goff[i])) : p.zeros);
  };
  // NORM: the (3, Cin) batch-norm rows of this sample behind the images in LDS (visible after the first stage's barrier); a
  // thread reads the rows of its four channels (c4 = tid & 3 in every slot) of the chunk being staged
  float* s_par = reinterpret_cast<float*>(smem + C3B_NP * C3B_IMG);
  if (NORM)
    for (int i = tid; i < 3 * p.Cin; i += 256) s_par[i] = p.in_par[(size_t)grp * 3 * p.Cin + i];

  f32x4 acc[NT][TY], accx[NT][TY];     // h0 g0 | (h0 g1 + h1 g0) * 2^11
  f32x2 ssum2[NT][2], ssq2[NT][2];
#pragma unroll
  for (int n = 0; n < NT; ++n) ssum2[n][0] = ssum2[n][1] = ssq2[n][0] = ssq2[n][1] = (f32x2){0.f, 0.f};
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
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int t = 0; t < TY; ++t) acc[n][t] = accx[n][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // weight pieces of this chunk: [step][tile][piece][lane]; the first step's are on their way while the images are written
    const f16x8* wch = p.wp + (((size_t)ch * JC * NTT + hf * NT) * C3B_NP) * 64 + lane;
    f16x8 A[2][2][C3B_NP];
#pragma unroll
    for (int nn = 0; nn < 2; ++nn)
#pragma unroll
      for (int pc = 0; pc < C3B_NP; ++pc) A[0][nn][pc] = wch[(nn * C3B_NP + pc) * 64];

    __syncthreads();                       // every wavefront is done reading the previous stage's images
    const PfTile Tc = pf_tile(stage);      // NORM: which slots lie inside the volume (a padding zero is not zero after its batch norm)
    float4 bnm, bns, bnb;
    const float nfloor = p.in_relu ? 0.f : -INFINITY;
    if (NORM) {
      const float* q3 = s_par + ch * 16 + (tid & 3) * 4;
      bnm = *reinterpret_cast<const float4*>(q3);
      bns = *reinterpret_cast<const float4*>(q3 + p.Cin);
      bnb = *reinterpret_cast<const float4*>(q3 + 2 * p.Cin);
    }
#pragma unroll
    for (int i = 0; i < MAXS; ++i) {
      if (i < MAXS - 1 || tid + i * 256 < C3B_SLOTS) {
        if (NORM) {
          float4 a = pf[i];
          a = atvs_bn4(a, bns, atvs_bn_shift4(bnm, bns, bnb));
          a.x = fmaxf(a.x, nfloor); a.y = fmaxf(a.y, nfloor); a.z = fmaxf(a.z, nfloor); a.w = fmaxf(a.w, nfloor);     // ReLU or nothing
          const unsigned t1 = pg[i] - Tc.lo;
          const unsigned t2 = Tc.hi1 + ~pg[i];
          const bool in = ((t1 & t2) & 0x808080u) == 0x808080u;
          pf[i] = make_float4(in ? a.x : 0.f, in ? a.y : 0.f, in ? a.z : 0.f, in ? a.w : 0.f);
        }
        f16x4 p0, p1;
        c3b_split(pf[i], &p0, &p1);
        *reinterpret_cast<f16x4*>(smem + laddr[i]) = p0;
        *reinterpret_cast<f16x4*>(smem + C3B_IMG + laddr[i]) = p1;
      }
    }
    __syncthreads();

    const PfTile T = pf_tile(min(stage + 1, nstage - 1));      // last stage: harmless re-read of its own halo

    // ---- K loops: per pair of output tiles 14 steps of two taps x 16 channels, each in two phases -- input piece h0 with both
    // weight pieces (32 MFMAs), h1 with g0 (16); fragments requested one phase ahead, weights one step ahead
    f16x8 Bq[2][TY];
    c3b_static_for<NH>([&](auto HF) __attribute__((always_inline)) {
      constexpr int half = decltype(HF)::value;
      auto request_B = [&](auto PH) __attribute__((always_inline)) {
        constexpr int ph = decltype(PH)::value, j = ph / C3B_NP, pc = ph % C3B_NP;
        constexpr int tA = c3b_clamp26(2 * j), tB = c3b_clamp26(2 * j + 1);
        const int a = fbase + ((q >> 1) ? c3b_disp(tB) : c3b_disp(tA));
#pragma unroll
        for (int t = 0; t < TY; ++t)
          Bq[ph & 1][t] = *reinterpret_cast<const f16x8*>(smem + pc * C3B_IMG + a + t * C3B_ROWB);
      };
      auto request_B1 = [&](auto PH, auto TT) __attribute__((always_inline)) {      // row t of phase ph
        constexpr int ph = decltype(PH)::value, j = ph / C3B_NP, pc = ph % C3B_NP, t = decltype(TT)::value;
        constexpr int tA = c3b_clamp26(2 * j), tB = c3b_clamp26(2 * j + 1);
        const int a = fbase + ((q >> 1) ? c3b_disp(tB) : c3b_disp(tA));
        Bq[ph & 1][t] = *reinterpret_cast<const f16x8*>(smem + pc * C3B_IMG + a + t * C3B_ROWB);
      };
      static_assert(JC % 2 == 0, "step 0 of the next tile pair goes to slot 0 while the last step reads slot 1");
      request_B(IC<0>{});
      asm volatile("" ::: "memory");
      if constexpr (NT == 2) {
      // every memory instruction behind ONE MFMA (tools_dev/micro/mfma_bf16_rate.hip): MFMA m of a phase = (weight piece jw,
        // tile nn, row t); behind the first eight the next phase's fragments, then (first phase of a step) the six weight
        // registers of the next step, then a halo slot of the next stage
        c3b_static_for<C3B_NP * JC>([&](auto PH) __attribute__((always_inline)) {
          constexpr int ph = decltype(PH)::value, j = ph / C3B_NP, pc = ph % C3B_NP;
          c3b_static_for<(2 - pc) * 2 * TY>([&](auto M) __attribute__((always_inline)) {
            constexpr int m = decltype(M)::value, jw = m / (2 * TY), nn = (m / TY) % 2, t = m % TY;
            if constexpr (pc == 0 && jw == 0)
              acc[half * 2 + nn][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[j & 1][nn][0], Bq[ph & 1][t], acc[half * 2 + nn][t], 0, 0, 0);
            else
              accx[half * 2 + nn][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[j & 1][nn][pc == 0 ? 1 : 0], Bq[ph & 1][t], accx[half * 2 + nn][t], 0, 0, 0);
            if constexpr (m < TY) {
              if constexpr (ph + 1 < C3B_NP * JC) request_B1(IC<ph + 1>{}, IC<m>{});
            } else if constexpr (pc == 0 && m < TY + 2 * C3B_NP) {
              constexpr int e = m - TY, en = e / C3B_NP, ep = e % C3B_NP;
              if constexpr (j + 1 < JC) A[(j + 1) & 1][en][ep] = wch[(((j + 1) * NTT + half * 2 + en) * C3B_NP + ep) * 64];
              else if constexpr (half + 1 < NH) A[0][en][ep] = wch[((0 * NTT + (half + 1) * 2 + en) * C3B_NP + ep) * 64];      // step 0 of the next pair
            } else if constexpr (half == 0 && pc == 0 && (m == TY + 6 || m == TY + 10)) {
              constexpr int slot = 2 * j + (m == TY + 10 ? 1 : 0);
              if constexpr (slot < MAXS) pf_slot(T, slot);
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
          });
        });
      } else {
        // four tiles per wavefront: the interleaved form spills (74 registers against 36) and is slower -- requests at the top
        // of each phase
        auto request_A = [&](auto JT, auto HH) __attribute__((always_inline)) {      // step j of tile pair hh -> slot j & 1
          constexpr int j = decltype(JT)::value, hh = decltype(HH)::value;
#pragma unroll
          for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int pc = 0; pc < C3B_NP; ++pc) A[j & 1][nn][pc] = wch[((j * NTT + hh * 2 + nn) * C3B_NP + pc) * 64];
        };
        c3b_static_for<C3B_NP * JC>([&](auto PH) __attribute__((always_inline)) {
          constexpr int ph = decltype(PH)::value, j = ph / C3B_NP, pc = ph % C3B_NP;
          if constexpr (ph + 1 < C3B_NP * JC) request_B(IC<ph + 1>{});
          if constexpr (pc == 0 && j + 1 < JC) request_A(IC<j + 1>{}, IC<half>{});
          if constexpr (pc == 0 && j + 1 == JC && half + 1 < NH) request_A(IC<0>{}, IC<half + 1>{});    // step 0 of the next pair
          if constexpr (half == 0 && ph < MAXS) pf_slot(T, ph);
          asm volatile("" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int nn = 0; nn < 2; ++nn) {
            if constexpr (pc == 0) {
#pragma unroll
              for (int t = 0; t < TY; ++t)
                acc[half * 2 + nn][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[j & 1][nn][0], Bq[ph & 1][t], acc[half * 2 + nn][t], 0, 0, 0);
#pragma unroll
              for (int t = 0; t < TY; ++t)
                accx[half * 2 + nn][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[j & 1][nn][1], Bq[ph & 1][t], accx[half * 2 + nn][t], 0, 0, 0);
            } else {
#pragma unroll
              for (int t = 0; t < TY; ++t)
                accx[half * 2 + nn][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[j & 1][nn][0], Bq[ph & 1][t], accx[half * 2 + nn][t], 0, 0, 0);
            }
          }
        });
      }
      static_assert(2 * JC >= MAXS, "two halo slots per K step");
    });
    if (ch != p.nchunk - 1) continue;

    // ---- epilogue: this lane holds channels 16 n + 4 q .. + 3 of voxel (z0 + wave, y0 + t, x0 + r)
    int tz0, ty0, tx0;
    tile_origin(k, &tz0, &ty0, &tx0);
    const int zo = tz0 + wave, xo = tx0 + r;
    const bool evox_ok = zo < p.Di && xo < p.Wi;
    const unsigned erow = (unsigned)p.Wi * p.ldy;
    const unsigned eo = (((unsigned)zo * p.Hi + ty0) * p.Wi + xo) * p.ldy + p.ycoff + hf * (NT * 16) + q * 4;
    const unsigned vo_ok = evox_ok ? eo * 4u : ybytes;
    c3b_static_for<NT>([&](auto NN) __attribute__((always_inline)) {
      constexpr int n = decltype(NN)::value;
      const float4 bv = p.bias ? ld4(p.bias + hf * (NT * 16) + n * 16 + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      c3b_static_for<TY>([&](auto TT) __attribute__((always_inline)) {
        constexpr int t = decltype(TT)::value;
        const bool row_ok = ty0 + t < p.Hi;
        const bool ok = evox_ok && row_ok;
        float a0 = (acc[n][t][0] + accx[n][t][0] * C3B_IRS) + bv.x, a1 = (acc[n][t][1] + accx[n][t][1] * C3B_IRS) + bv.y;
        float a2 = (acc[n][t][2] + accx[n][t][2] * C3B_IRS) + bv.z, a3 = (acc[n][t][3] + accx[n][t][3] * C3B_IRS) + bv.w;
        if (p.relu) {
          a0 = (a0 < 0.f) ? 0.f : a0; a1 = (a1 < 0.f) ? 0.f : a1;
          a2 = (a2 < 0.f) ? 0.f : a2; a3 = (a3 < 0.f) ? 0.f : a3;
        }
        const u32x4 bits = {__builtin_bit_cast(unsigned, a0), __builtin_bit_cast(unsigned, a1),
                            __builtin_bit_cast(unsigned, a2), __builtin_bit_cast(unsigned, a3)};
        __builtin_amdgcn_raw_buffer_store_b128(bits, yrsrc, row_ok ? vo_ok : ybytes, (t * erow + n * 16) * 4u, ATVS_BUF_NT);
        f32x2 lo = {ok ? a0 : 0.f, ok ? a1 : 0.f}, hi = {ok ? a2 : 0.f, ok ? a3 : 0.f};
        ssum2[n][0] += lo;
        ssum2[n][1] += hi;
        ssq2[n][0] = __builtin_elementwise_fma(lo, lo, ssq2[n][0]);
        ssq2[n][1] = __builtin_elementwise_fma(hi, hi, ssq2[n][1]);
      });
    });
  }

  if (p.stats) {
    constexpr int CO = NT * 16;
    __syncthreads();
    double* s_red = reinterpret_cast<double*>(smem);   // [4 waves][2][CO]
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
          s_red[(wave * 2 + 0) * CO + n * 16 + q * 4 + kk] = a;
          s_red[(wave * 2 + 1) * CO + n * 16 + q * 4 + kk] = bq;
        }
      }
    __syncthreads();
    if (tid < 2 * CO) {
      const int which = tid / CO, col = tid % CO;
      p.stats[((size_t)bidx * 2 + which) * (CO * p.nhalf) + hf * CO + col] =
          (s_red[(0 * 2 + which) * CO + col] + s_red[(1 * 2 + which) * CO + col]) +
          (s_red[(2 * 2 + which) * CO + col] + s_red[(3 * 2 + which) * CO + col]);
    }
  }
}


template <int NT, bool NORM>
int launch_c3b(const C3bArgs& a, long grid, hipStream_t s) {
  const size_t lds = C3B_NP * (size_t)C3B_IMG + (NORM ? 3 * (size_t)a.Cin * 4 : 0);
  static AtvsAttrOnce lds_once;                   // per kernel instantiation (this function is a template / has one kernel)
  if (const int rc_ = atvs_set_max_lds_once(lds_once, reinterpret_cast<const void*>(conv3d_b_kernel<NT, NORM>), 160 * 1024)) return rc_;
  hipLaunchKernelGGL((conv3d_b_kernel<NT, NORM>), dim3((unsigned)(grid * a.nhalf)), dim3(256), lds, s, a);
  return ATVS_OK;
}

}  // namespace

extern "C" int atvs_conv3d_b_supported(int Cin, int Cout) {
  return (Cin >= 16 && Cin % 16 == 0 && Cin <= 256 && (Cout == 32 || Cout == 64)) ? 1 : 0;
}

// Bytes of the packed form of a TF kernel [3,3,3,Cin,Cout] for atvs_conv3d_b_f32 (+ 16 trailing zero bytes).
extern "C" int atvs_conv3d_b_pack_size(int Cin, int Cout, long* packed_bytes) {
  if (!packed_bytes) return ATVS_ERR_NULL;
  if (!atvs_conv3d_b_supported(Cin, Cout)) return ATVS_ERR_SHAPE;
  *packed_bytes = (long)(Cin / 16) * C3B_JC * (Cout / 16) * C3B_NP * 1024 + 16;
  return ATVS_OK;
}

// HOST function.  packed[chunk][step j][tile n][piece][lane = q*16 + co16][8 bf16] = piece of
// w[tap = 2 j + (q >> 1)][ci = 16 chunk + 8 (q & 1) + e][co = 16 n + co16] (zero for tap 27); pieces g0 = f16(w),
// g1 = f16((w - g0) * 2048), round to nearest even; ATVS_ERR_ARG for a weight beyond fp16's range.
extern "C" int atvs_conv3d_b_pack(const float* w, int Cin, int Cout, unsigned char* packed) {
  if (!w || !packed) return ATVS_ERR_NULL;
  long pb;
  int rc = atvs_conv3d_b_pack_size(Cin, Cout, &pb);
  if (rc) return rc;
  std::memset(packed, 0, (size_t)pb);
  uint16_t* out = reinterpret_cast<uint16_t*>(packed);
  const int NT = Cout / 16;
  bool fits = true;
  for (int ch = 0; ch < Cin / 16; ++ch)
    for (int j = 0; j < C3B_JC; ++j)
      for (int n = 0; n < NT; ++n)
        for (int q = 0; q < 4; ++q) {
          const int tap = 2 * j + (q >> 1);
          if (tap > 26) continue;
          for (int co = 0; co < 16; ++co)
            for (int e = 0; e < 8; ++e) {
              const int ci = ch * 16 + (q & 1) * 8 + e;
              const float v = w[((size_t)tap * Cin + ci) * Cout + n * 16 + co];
              const _Float16 g0 = (_Float16)v, g1 = (_Float16)((v - (float)g0) * C3B_RS);
              std::memcpy(&out[(((((size_t)ch * C3B_JC + j) * NT + n) * C3B_NP + 0) * 64 + q * 16 + co) * 8 + e], &g0, 2);
              std::memcpy(&out[(((((size_t)ch * C3B_JC + j) * NT + n) * C3B_NP + 1) * 64 + q * 16 + co) * 8 + e], &g1, 2);
              const float back = (float)g0;
              fits &= (back - back == 0.f);
            }
        }
  return fits ? ATVS_OK : ATVS_ERR_ARG;
}

namespace {
int c3b_launch(const float* x, const float* in_par, int in_relu, const unsigned char* packed_w, const float* bias, float* y,
               double* stats_partial, int groups, int D, int H, int W, int Cin, int Cout, int ldy, int y_coff, int relu,
               atvs_stream_t stream) {
  if (!x || !packed_w || !y) return ATVS_ERR_NULL;
  if (groups <= 0 || D <= 0 || H <= 0 || W <= 0 || !atvs_conv3d_b_supported(Cin, Cout)) return ATVS_ERR_SHAPE;
  if (y_coff < 0 || y_coff + Cout > ldy || (ldy % 4) || (y_coff % 4)) return ATVS_ERR_SHAPE;
  if ((double)D * H * W * Cin >= 2147483648.0 || (double)D * H * W * ldy * 4.0 >= 4294967296.0) return ATVS_ERR_SHAPE;
  C3bArgs a;
  long pbytes;
  atvs_conv3d_b_pack_size(Cin, Cout, &pbytes);
  a.x = x; a.in_par = in_par; a.in_relu = in_relu;
  a.wp = reinterpret_cast<const f16x8*>(packed_w); a.zeros = reinterpret_cast<const float*>(packed_w + (pbytes - 16));
  a.bias = bias; a.y = y; a.stats = stats_partial;
  a.Di = D; a.Hi = H; a.Wi = W; a.Cin = Cin; a.ldy = ldy; a.ycoff = y_coff; a.nchunk = Cin / 16; a.relu = relu;
  a.tiles_y = (H + C3B_TY - 1) / C3B_TY; a.tiles_x = (W + C3B_TX - 1) / C3B_TX;
  a.ntiles = ((D + C3B_TZ - 1) / C3B_TZ) * a.tiles_y * a.tiles_x;
  const long blocks = atvs_conv_c16_grid(D, H, W, groups);
  a.wg = (int)blocks;
  a.gx = (long)D * H * W * Cin; a.gy = (long)D * H * W * ldy;
  const long grid = blocks * groups;
  if (grid > 0x7fffffffL) return ATVS_ERR_SHAPE;
  hipStream_t st = as_stream(stream);
  a.nhalf = Cout / 32; a.bh = (int)grid;
  if (grid * a.nhalf > 0x7fffffffL) return ATVS_ERR_SHAPE;
  int rc = in_par ? launch_c3b<2, true>(a, grid, st) : launch_c3b<2, false>(a, grid, st);
  if (rc) return rc;
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}
}  // namespace

// y (G,D,H,W,ldy)[..., y_coff : y_coff + Cout] = conv3d(x (G,D,H,W,Cin), w [3,3,3,Cin,Cout], stride 1, SAME) (+ bias, ReLU),
// Cin % 16 == 0, Cout 32 or 64, split-fp16 operands (fp32-class results).  Grid / statistics rows = atvs_conv_c16_grid
// ([2][Cout] doubles per row).
extern "C" int atvs_conv3d_b_f32(const float* x, const unsigned char* packed_w, const float* bias, float* y,
                                 double* stats_partial, int groups, int D, int H, int W, int Cin, int Cout, int ldy, int y_coff,
                                 int relu, atvs_stream_t stream) {
  return c3b_launch(x, nullptr, 0, packed_w, bias, y, stats_partial, groups, D, H, W, Cin, Cout, ldy, y_coff, relu, stream);
}

// The same convolution of relu?((x - mean) * scale + beta), x a raw convolution output and in_params (groups,3,Cin) its pending
// batch norm (conv_b*_{2,3}_1 read conv_b*_{2,3}_0, reference cnn_wrapper/atvsnet.py:20-26): formed per staged halo voxel, the
// normalised tensor is never written.  Bit for bit atvs_bn_apply followed by atvs_conv3d_b_f32.
extern "C" int atvs_conv3d_b_norm_f32(const float* x, const float* in_params, int in_relu, const unsigned char* packed_w,
                                      const float* bias, float* y, double* stats_partial, int groups, int D, int H, int W, int Cin,
                                      int Cout, int ldy, int y_coff, int relu, atvs_stream_t stream) {
  if (!in_params) return ATVS_ERR_NULL;
  return c3b_launch(x, in_params, in_relu, packed_w, bias, y, stats_partial, groups, D, H, W, Cin, Cout, ldy, y_coff, relu, stream);
}
