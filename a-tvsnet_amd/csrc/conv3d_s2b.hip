// 3x3x3 SAME STRIDE-2 convolutions with 16 k input and 32 / 64 output channels on the bf16 matrix cores with SPLIT operands
// (gfx950): the encoders of the 3-D U-Nets below half resolution (conv_b*_2_0: 16 -> 32, conv_b*_3_0: 32 -> 64,
// global_refine_3dconv{2,3}_0; /root/reference/cnn_wrapper/atvsnet.py StackedUNet / CostVolRefineNet, layer code
// network.py:172-215), which ran on the gather kernel (conv.hip) at ~60 TFLOP/s.  Arithmetic of conv_c16b.hip.
//
// Output tile 2(z) x 4(y) x 16(x); its input halo 5 x 9 x 33 voxels is staged in 16-channel chunks as three piece images whose
// rows hold the EVEN voxels first, then the odd ones: output x reads input 2 x + kw, so a tap's 16 lanes read 16 consecutive
// 32-byte voxels of one parity run (conflict-free), and every (kd, kh, kw, row) displacement is an immediate.  The wavefronts
// split (z plane, output-channel half): 4 rows x NT/2 tiles each; a K = 32 step = two taps x 16 channels, one phase per step,
// all three pieces' fragments requested one step ahead; weight pieces stream from L2.  Every memory instruction of the K loop
// sits behind one MFMA (tools_dev/micro/mfma_bf16_rate.hip).  TF SAME padding: pad_before = (2 (out - 1) + 3 - in) / 2 per axis.
#include <cstring>
#include <type_traits>
#include <utility>

#include "conv_common.h"

// Development build (-DATVS_S2_DEBUG): per-wavefront tick counts of the phases (tools_dev/phase_s2.py)
#ifdef ATVS_S2_DEBUG
__device__ unsigned long long atvs_dbg_s2[4096 * 8];
extern "C" int atvs_debug_read_s2(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(atvs_dbg_s2), sizeof(atvs_dbg_s2));
}
#define SDBG(i) { unsigned long long t_ = clock64(); dbg_acc[i] += t_ - dbg_t; dbg_t = t_; }
#else
#define SDBG(i)
#endif

namespace {

constexpr int S2_TZ = 2, S2_TY = 4, S2_TX = 16;
constexpr int S2_HZ = 2 * S2_TZ + 1, S2_HY = 2 * S2_TY + 1, S2_HX = 2 * S2_TX + 1;
constexpr int S2_VB = 32;
constexpr int S2_ROWB = S2_HX * S2_VB;                         // 1,056: 17 even voxels, 16 odd voxels
constexpr int S2_IMG = S2_HZ * S2_HY * S2_ROWB;                // 47,520 bytes per piece
constexpr int S2_SLOTS = S2_HZ * S2_HY * S2_HX * 4;
constexpr int S2_MAXS = (S2_SLOTS + 255) / 256;                // 24 per thread
constexpr int S2_JC = 14;
static_assert(S2_MAXS <= 2 * (S2_JC - 1), "two halo slots per K step");
constexpr int s2_clamp26(int t) { return t < 26 ? t : 26; }
constexpr int s2_xsel(int kw) { return kw == 0 ? 0 : kw == 1 ? (S2_TX + 1) * S2_VB : S2_VB; }
constexpr int s2_disp(int t) { return ((t / 9) * S2_HY + (t / 3) % 3) * S2_ROWB + s2_xsel(t % 3); }
__device__ __forceinline__ constexpr int s2_xcol(int xx) { return ((xx & 1) ? (S2_TX + 1) + (xx >> 1) : (xx >> 1)) * S2_VB; }

constexpr int S2_NP = 2;                                       // operand pieces: x = h0 + h1 / 2048 (conv_c16b.hip, round 4)
constexpr float S2_RS = 2048.f, S2_IRS = 1.f / 2048.f;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct S2Args {
  const float* x;
  // NORM form: x is a raw convolution output whose batch norm [+ ReLU] is applied while the halo is staged (atvs_bn_apply's
  // arithmetic: relu?((v - mean) * scale + beta)); in_par (groups, 3, Cin)
  const float* in_par;
  int in_relu;
  const f16x8* wp;
  const float* zeros;
  const float* bias;
  float* y;
  double* stats;
  int Di, Hi, Wi, Cin;
  int Do, Ho, Wo;
  int pbz, pby, pbx;
  int ldy, ycoff;
  int nchunk;
  int tiles_y, tiles_x, ntiles;
  int wg;
  int relu;
  long gx, gy;
};

template <int N>
using IC = std::integral_constant<int, N>;
template <class F, int... I>
__device__ __forceinline__ void s2_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(IC<I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void s2_static_for(F&& f) {
  s2_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ void s2_split(const float4& v, f16x4* p0, f16x4* p1) {
  // atvs_split2_f16 (common.h): five vector instructions per two values instead of the 8-9 of the C form, the same values
  uint2 a, b;
  atvs_split4_f16(v, &a, &b);
  *p0 = __builtin_bit_cast(f16x4, a);
  *p1 = __builtin_bit_cast(f16x4, b);
}

// NT = Cout / 16 (2 or 4); a wavefront owns NTW = NT / 2 output tiles of one z plane.
// WLDS: the packed weights of every chunk are copied into LDS once (16 -> 32: 56 KB beside the 95 KB of images) and the K loop
// takes its weight fragments from there.  Streamed from L2 one step ahead (rounds 3-4) a step's 12 MFMAs = 192 cycles could not
// cover the L2 round trip: every one of the 14 steps of a stage waited for its weights (round 5).  Where the weights do not fit
// (32 -> 64) they are requested TWO steps ahead (three register slots).
template <int NT, bool WLDS, bool NORM>
__global__ __launch_bounds__(256, 1) void conv3d_s2b_kernel(S2Args p) {
  // own the SIMD's whole register file (512 per lane): no wavefront of ANOTHER kernel runs beside this one's bf16 MFMAs --
  // beside them other kernels' wavefronts computed wrong lane quarters (tools_dev/micro/pk_beside_mfma.hip)
  asm volatile("" ::: "v255", "a255");
  constexpr int TY = S2_TY, HY = S2_HY, MAXS = S2_MAXS, JC = S2_JC, NTW = NT / 2, NM = 3 * NTW * TY;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int wz = wave >> 1, wn = wave & 1;

  // this lane's fragment (channels 8 (q & 1) ..) of output (wz, row 0, r) at tap (0,0,0); lane half q >> 1 = tap of the step
  const int fbase = (2 * wz * HY) * S2_ROWB + r * S2_VB + (q & 1) * 16;
  const f16x8* wsrc = p.wp;
  if (WLDS) {
    const float4* src = reinterpret_cast<const float4*>(p.wp);
    float4* dst = reinterpret_cast<float4*>(smem + S2_NP * S2_IMG);
    for (int i = tid; i < p.nchunk * JC * NT * S2_NP * 64; i += 256) dst[i] = src[i];
    __syncthreads();
    wsrc = reinterpret_cast<const f16x8*>(smem + S2_NP * S2_IMG);
  }

  int goff[MAXS], laddr[MAXS];
  unsigned pg[MAXS];
#pragma unroll
  for (int i = 0; i < MAXS; ++i) {
    int s = tid + i * 256;
    const bool live = s < S2_SLOTS;
    s = min(s, S2_SLOTS - 1);
    const int c4 = s & 3, v = s >> 2;
    const int xx = v % S2_HX, v2 = v / S2_HX;
    const int yy = v2 % HY, zz = v2 / HY;
    goff[i] = ((zz * p.Hi + yy) * p.Wi + xx) * p.Cin + c4 * 4;
    laddr[i] = (zz * HY + yy) * S2_ROWB + s2_xcol(xx) + c4 * 8;
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
  auto tile_origin = [&](int k, int* z0, int* y0, int* x0) __attribute__((always_inline)) {      // OUTPUT coordinates
    int tl = xcd * per_xcd + tslot + k * slots_per_xcd;
    int bx = tl % p.tiles_x;
    int rest = tl / p.tiles_x;
    *x0 = bx * S2_TX;
    *y0 = (rest % p.tiles_y) * TY;
    *z0 = (rest / p.tiles_y) * S2_TZ;
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
    const int gz0 = 2 * z0 - p.pbz, gy0 = 2 * y0 - p.pby, gx0 = 2 * x0 - p.pbx;       // -1 at most
    T.org = ((gz0 * p.Hi + gy0) * p.Wi + gx0) * p.Cin + ch * 16;
    T.lo = (unsigned)(gz0 < 0) | ((unsigned)(gy0 < 0) << 8) | ((unsigned)(gx0 < 0) << 16);
    T.hi1 = (unsigned)(max(min(p.Di - 1 - gz0, 0x7e), -1) + 1) | ((unsigned)(max(min(p.Hi - 1 - gy0, 0x7e), -1) + 1) << 8) |
            ((unsigned)(max(min(p.Wi - 1 - gx0, 0x7e), -1) + 1) << 16);
    return T;
  };
  float4 pf[MAXS];
  auto pf_slot = [&](const PfTile& T, int i) __attribute__((always_inline)) {
    const unsigned t1 = pg[i] - T.lo;
    const unsigned t2 = T.hi1 + ~pg[i];
    const bool ok = ((t1 & t2) & 0x808080u) == 0x808080u;
    pf[i] = ld4(ok ? (xg + (T.org + goff[i])) : p.zeros);
  };

  // NORM: the batch-norm rows of this thread's four channels (c4 = tid & 3 in every slot) of the chunk being staged, and of
  // the next stage's chunk (requested in front of the K loop)
  float4 bnm, bns, bnb, nbnm, nbns, nbnb;
  auto norm_rows = [&](int ch) __attribute__((always_inline)) {
    const float* q3 = p.in_par + (size_t)grp * 3 * p.Cin + ch * 16 + (tid & 3) * 4;
    nbnm = ld4(q3); nbns = ld4(q3 + p.Cin); nbnb = ld4(q3 + 2 * p.Cin);
  };
  if (NORM) norm_rows(0);
  // slot i of a stage -> its two fp16 pieces, kept in pf[i] (p0 | p1) until the stage's LDS writes.  Slots [0, NIN) of the stage in
  // flight are transformed behind MFMAs of the K loop, LAG slots (LAG / 2 steps) after their loads were issued -- in front of the
  // LDS writes all four wavefronts wait for it (tools_dev/phase_s2.py: split + LDS write 26 % of a stage).  rm / rs / rb: the rows
  // of that stage's chunk.
  constexpr int LAG = 8, NIN = (2 * JC - LAG) < MAXS ? (2 * JC - LAG) : MAXS;
  auto xform_slot = [&](int i, const PfTile& TT, const float4& rm, const float4& rs, const float4& rb) __attribute__((always_inline)) {
    float4 a = pf[i];
    if (NORM) {
      const float nfloor = p.in_relu ? 0.f : -INFINITY;
      a = atvs_bn4(a, rs, atvs_bn_shift4(rm, rs, rb));
      a.x = fmaxf(a.x, nfloor); a.y = fmaxf(a.y, nfloor); a.z = fmaxf(a.z, nfloor); a.w = fmaxf(a.w, nfloor);     // ReLU or nothing
      // a halo slot outside the volume: its padding stays zero
      const unsigned t1 = pg[i] - TT.lo;
      const unsigned t2 = TT.hi1 + ~pg[i];
      const bool in = ((t1 & t2) & 0x808080u) == 0x808080u;
      a = make_float4(in ? a.x : 0.f, in ? a.y : 0.f, in ? a.z : 0.f, in ? a.w : 0.f);
    }
    uint2 h0, h1;
    atvs_split4_f16(a, &h0, &h1);
    pf[i] = make_float4(__uint_as_float(h0.x), __uint_as_float(h0.y), __uint_as_float(h1.x), __uint_as_float(h1.y));
  };

  f32x4 acc[NTW][TY], accx[NTW][TY];   // h0 g0 | (h0 g1 + h1 g0) * 2^11
  f32x2 ssum2[NTW][2], ssq2[NTW][2];
#pragma unroll
  for (int n = 0; n < NTW; ++n) ssum2[n][0] = ssum2[n][1] = ssq2[n][0] = ssq2[n][1] = (f32x2){0.f, 0.f};
  const unsigned ybytes = (unsigned)(p.gy * 4);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(yg, 0, ybytes, 0x00020000);

  PfTile Tn = pf_tile(0);                  // the stage whose halo is in flight
  if (nstage > 0) {
#pragma unroll
    for (int i = 0; i < MAXS; ++i) pf_slot(Tn, i);
#pragma unroll
    for (int i = 0; i < NIN; ++i) xform_slot(i, Tn, nbnm, nbns, nbnb);      // (later stages: inside the previous stage's K loop)
  }

#ifdef ATVS_S2_DEBUG
  unsigned long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long dbg_t = clock64();
#endif
  for (int stage = 0; stage < nstage; ++stage) {
    const int k = stage / p.nchunk, ch = stage - k * p.nchunk;
    SDBG(0)
    if (ch == 0) {
#pragma unroll
      for (int n = 0; n < NTW; ++n)
#pragma unroll
        for (int t = 0; t < TY; ++t) acc[n][t] = accx[n][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // weight pieces of this chunk: [step][tile][piece][lane]
    const f16x8* wch = wsrc + ((size_t)ch * JC * NT * S2_NP + (size_t)wn * NTW * S2_NP) * 64 + lane;
    constexpr int AHEAD = WLDS ? 1 : 2, NA = AHEAD + 1;          // weight fragments requested AHEAD steps early, NA register slots
    f16x8 A[NA][NTW][S2_NP], B[2][S2_NP][TY];
#pragma unroll
    for (int j0 = 0; j0 < AHEAD; ++j0)
#pragma unroll
      for (int nn = 0; nn < NTW; ++nn)
#pragma unroll
        for (int pc = 0; pc < S2_NP; ++pc) A[j0][nn][pc] = wch[((j0 * NT + nn) * S2_NP + pc) * 64];

    __syncthreads();                       // every wavefront is done reading the previous stage's images
    SDBG(1)
#ifdef ATVS_S2_DEBUG
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SDBG(2)
#endif
    const PfTile Tc = Tn;                  // the stage being staged
    if (NORM) { bnm = nbnm; bns = nbns; bnb = nbnb; }
#pragma unroll
    for (int i = NIN; i < MAXS; ++i) xform_slot(i, Tc, bnm, bns, bnb);
#pragma unroll
    for (int i = 0; i < MAXS; ++i) {
      if (i < MAXS - 1 || tid + i * 256 < S2_SLOTS) {
        *reinterpret_cast<uint2*>(smem + laddr[i]) = make_uint2(__float_as_uint(pf[i].x), __float_as_uint(pf[i].y));
        *reinterpret_cast<uint2*>(smem + S2_IMG + laddr[i]) = make_uint2(__float_as_uint(pf[i].z), __float_as_uint(pf[i].w));
      }
    }
    SDBG(3)
    __syncthreads();
    SDBG(4)

    Tn = pf_tile(min(stage + 1, nstage - 1));
    const PfTile T = Tn;      // last stage: harmless re-read of its own halo
    if (NORM && p.nchunk > 1) norm_rows((stage + 1) % p.nchunk);

    auto fragment = [&](auto JT, auto PC, auto TT) __attribute__((always_inline)) {       // step j, piece pc, row t
      constexpr int j = decltype(JT)::value, pc = decltype(PC)::value, t = decltype(TT)::value;
      constexpr int tA = s2_clamp26(2 * j), tB = s2_clamp26(2 * j + 1);
      const int a = fbase + ((q >> 1) ? s2_disp(tB) : s2_disp(tA));
      B[j & 1][pc][t] = *reinterpret_cast<const f16x8*>(smem + pc * S2_IMG + a + 2 * t * S2_ROWB);
    };
    s2_static_for<S2_NP * TY>([&](auto M) __attribute__((always_inline)) {
      constexpr int m = decltype(M)::value;
      fragment(IC<0>{}, IC<m / TY>{}, IC<m % TY>{});
    });
    asm volatile("" ::: "memory");
    // ---- K loop: 14 steps of two taps x 16 channels; MFMA m of a step = (product pr: h0 g0 | h0 g1 | h1 g0, tile nn, row t);
    // behind MFMA m: m < 8 the next step's fragments, then its weights, then two halo slots of the next stage
    s2_static_for<JC>([&](auto JT) __attribute__((always_inline)) {
      constexpr int j = decltype(JT)::value;
      s2_static_for<NM>([&](auto M) __attribute__((always_inline)) {
        constexpr int m = decltype(M)::value, pr = m / (NTW * TY), nn = (m / TY) % NTW, t = m % TY;
        constexpr int pc = pr < 2 ? 0 : 1, jw = pr == 1 ? 1 : 0;
        if constexpr (pr == 0) acc[nn][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[j % NA][nn][0], B[j & 1][0][t], acc[nn][t], 0, 0, 0);
        else accx[nn][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[j % NA][nn][jw], B[j & 1][pc][t], accx[nn][t], 0, 0, 0);
        if constexpr (m < S2_NP * TY) {
          if constexpr (j + 1 < JC) fragment(IC<j + 1>{}, IC<m / TY>{}, IC<m % TY>{});
        } else if constexpr (m < S2_NP * TY + S2_NP * NTW) {
          constexpr int e = m - S2_NP * TY;
          if constexpr (j + AHEAD < JC) A[(j + AHEAD) % NA][e / S2_NP][e % S2_NP] = wch[(((j + AHEAD) * NT + e / S2_NP) * S2_NP + e % S2_NP) * 64];
        } else if constexpr (m == S2_NP * TY + S2_NP * NTW || m == S2_NP * TY + S2_NP * NTW + 1) {
          constexpr int s = 2 * j + (m - S2_NP * TY - S2_NP * NTW);
          if constexpr (s < MAXS) pf_slot(T, s);
          if constexpr (s >= LAG && s - LAG < NIN) xform_slot(s - LAG, T, nbnm, nbns, nbnb);
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    SDBG(5)
    if (ch != p.nchunk - 1) continue;

    // ---- epilogue: this lane holds channels 16 (wn NTW + nn) + 4 q .. + 3 of output voxel (z0 + wz, y0 + t, x0 + r)
    int tz0, ty0, tx0;
    tile_origin(k, &tz0, &ty0, &tx0);
    const int zo = tz0 + wz, xo = tx0 + r;
    const bool evox_ok = zo < p.Do && xo < p.Wo;
    const unsigned erow = (unsigned)p.Wo * p.ldy;
    const unsigned eo = (((unsigned)zo * p.Ho + ty0) * p.Wo + xo) * p.ldy + p.ycoff + (wn * NTW) * 16 + q * 4;
    const unsigned vo_ok = evox_ok ? eo * 4u : ybytes;
    s2_static_for<NTW>([&](auto NN) __attribute__((always_inline)) {
      constexpr int n = decltype(NN)::value;
      const float4 bv = p.bias ? ld4(p.bias + (wn * NTW + n) * 16 + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      s2_static_for<TY>([&](auto TT) __attribute__((always_inline)) {
        constexpr int t = decltype(TT)::value;
        const bool row_ok = ty0 + t < p.Ho;
        const bool ok = evox_ok && row_ok;
        float a0 = (acc[n][t][0] + accx[n][t][0] * S2_IRS) + bv.x, a1 = (acc[n][t][1] + accx[n][t][1] * S2_IRS) + bv.y;
        float a2 = (acc[n][t][2] + accx[n][t][2] * S2_IRS) + bv.z, a3 = (acc[n][t][3] + accx[n][t][3] * S2_IRS) + bv.w;
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
    SDBG(6)
  }
#ifdef ATVS_S2_DEBUG
  if (lane == 0 && blockIdx.x < 1024) {
    dbg_acc[7] = (unsigned long long)nstage;
    for (int i = 0; i < 8; ++i) atvs_dbg_s2[(blockIdx.x * 4 + wave) * 8 + i] = dbg_acc[i];
  }
#endif

  if (p.stats) {
    constexpr int CO = NT * 16;
    __syncthreads();
    double* s_red = reinterpret_cast<double*>(smem);   // [2 z planes][2][CO]
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        double a = (double)ssum2[n][kk >> 1][kk & 1], bq = (double)ssq2[n][kk >> 1][kk & 1];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          a += __shfl_xor(a, o);
          bq += __shfl_xor(bq, o);
        }
        if (r == 0) {
          const int c = (wn * NTW + n) * 16 + q * 4 + kk;
          s_red[(wz * 2 + 0) * CO + c] = a;
          s_red[(wz * 2 + 1) * CO + c] = bq;
        }
      }
    __syncthreads();
    if (tid < 2 * CO) {
      const int which = tid / CO, col = tid % CO;
      p.stats[((size_t)blockIdx.x * 2 + which) * CO + col] = s_red[(0 * 2 + which) * CO + col] + s_red[(1 * 2 + which) * CO + col];
    }
  }
}

template <int NT, bool WLDS, bool NORM>
int launch_s2b(const S2Args& a, long grid, hipStream_t s) {
  const size_t lds = S2_NP * (size_t)S2_IMG + (WLDS ? (size_t)a.nchunk * S2_JC * NT * S2_NP * 1024 : 0);
  static AtvsAttrOnce lds_once;                   // per kernel instantiation (this function is a template / has one kernel)
  if (const int rc_ = atvs_set_max_lds_once(lds_once, reinterpret_cast<const void*>(conv3d_s2b_kernel<NT, WLDS, NORM>), 160 * 1024)) return rc_;
  hipLaunchKernelGGL((conv3d_s2b_kernel<NT, WLDS, NORM>), dim3((unsigned)grid), dim3(256), lds, s, a);
  return ATVS_OK;
}

long s2_ntiles(int Do, int Ho, int Wo) {
  return (long)((Do + S2_TZ - 1) / S2_TZ) * ((Ho + S2_TY - 1) / S2_TY) * ((Wo + S2_TX - 1) / S2_TX);
}

}  // namespace

extern "C" int atvs_conv3d_s2b_supported(int Cin, int Cout) {
  return (Cin >= 16 && Cin % 16 == 0 && Cin <= 256 && (Cout == 32 || Cout == 64)) ? 1 : 0;
}

// workgroups PER SAMPLE of a launch over `groups` samples of OUTPUT size (Do,Ho,Wo) (rows of the statistics buffer = groups *
// this): one workgroup per CU in all, shared out among the samples, a multiple of 8 each
extern "C" long atvs_conv3d_s2b_grid(int Do, int Ho, int Wo, int groups) {
  if (groups < 1) groups = 1;
  const long nt = s2_ntiles(Do, Ho, Wo);
  long share = 256 / groups / 8 * 8;
  if (share < 8) share = 8;
  const long g = nt < share ? nt : share;
  return (g + 7) / 8 * 8;
}

extern "C" int atvs_conv3d_s2b_pack_size(int Cin, int Cout, long* packed_bytes) {
  if (!packed_bytes) return ATVS_ERR_NULL;
  if (!atvs_conv3d_s2b_supported(Cin, Cout)) return ATVS_ERR_SHAPE;
  *packed_bytes = (long)(Cin / 16) * S2_JC * (Cout / 16) * S2_NP * 1024 + 16;
  return ATVS_OK;
}

// HOST function; the layout of atvs_conv3d_b_pack: packed[chunk][step j][tile n][piece][lane = q*16 + co16][8 bf16] = piece of
// w[tap = 2 j + (q >> 1)][ci = 16 chunk + 8 (q & 1) + e][co = 16 n + co16] (zero for tap 27).
extern "C" int atvs_conv3d_s2b_pack(const float* w, int Cin, int Cout, unsigned char* packed) {
  if (!w || !packed) return ATVS_ERR_NULL;
  long pb;
  int rc = atvs_conv3d_s2b_pack_size(Cin, Cout, &pb);
  if (rc) return rc;
  std::memset(packed, 0, (size_t)pb);
  uint16_t* out = reinterpret_cast<uint16_t*>(packed);
  const int NT = Cout / 16;
  bool fits = true;
  for (int ch = 0; ch < Cin / 16; ++ch)
    for (int j = 0; j < S2_JC; ++j)
      for (int n = 0; n < NT; ++n)
        for (int q = 0; q < 4; ++q) {
          const int tap = 2 * j + (q >> 1);
          if (tap > 26) continue;
          for (int co = 0; co < 16; ++co)
            for (int e = 0; e < 8; ++e) {
              const int ci = ch * 16 + (q & 1) * 8 + e;
              const float v = w[((size_t)tap * Cin + ci) * Cout + n * 16 + co];
              const _Float16 g0 = (_Float16)v, g1 = (_Float16)((v - (float)g0) * S2_RS);
              std::memcpy(&out[(((((size_t)ch * S2_JC + j) * NT + n) * S2_NP + 0) * 64 + q * 16 + co) * 8 + e], &g0, 2);
              std::memcpy(&out[(((((size_t)ch * S2_JC + j) * NT + n) * S2_NP + 1) * 64 + q * 16 + co) * 8 + e], &g1, 2);
              const float back = (float)g0;
              fits &= (back - back == 0.f);
            }
        }
  return fits ? ATVS_OK : ATVS_ERR_ARG;
}

namespace {
int s2b_launch(const float* x, const float* in_par, int in_relu, const unsigned char* packed_w, const float* bias, float* y,
               double* stats_partial, int groups, int D, int H, int W, int Cin, int Cout, int ldy, int y_coff, int relu,
               atvs_stream_t stream) {
  if (!x || !packed_w || !y) return ATVS_ERR_NULL;
  if (groups <= 0 || D <= 0 || H <= 0 || W <= 0 || !atvs_conv3d_s2b_supported(Cin, Cout)) return ATVS_ERR_SHAPE;
  if (y_coff < 0 || y_coff + Cout > ldy || (ldy % 4) || (y_coff % 4)) return ATVS_ERR_SHAPE;
  const int Do = (D + 1) / 2, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  if ((double)D * H * W * Cin >= 2147483648.0 || (double)Do * Ho * Wo * ldy * 4.0 >= 4294967296.0) return ATVS_ERR_SHAPE;
  S2Args a;
  long pb;
  atvs_conv3d_s2b_pack_size(Cin, Cout, &pb);
  a.x = x; a.in_par = in_par; a.in_relu = in_relu;
  a.wp = reinterpret_cast<const f16x8*>(packed_w); a.zeros = reinterpret_cast<const float*>(packed_w + (pb - 16));
  a.bias = bias; a.y = y; a.stats = stats_partial;
  a.Di = D; a.Hi = H; a.Wi = W; a.Cin = Cin; a.Do = Do; a.Ho = Ho; a.Wo = Wo;
  a.pbz = (2 * (Do - 1) + 3 - D) / 2; a.pby = (2 * (Ho - 1) + 3 - H) / 2; a.pbx = (2 * (Wo - 1) + 3 - W) / 2;
  a.ldy = ldy; a.ycoff = y_coff; a.nchunk = Cin / 16; a.relu = relu;
  a.tiles_y = (Ho + S2_TY - 1) / S2_TY; a.tiles_x = (Wo + S2_TX - 1) / S2_TX;
  a.ntiles = (int)s2_ntiles(Do, Ho, Wo);
  const long blocks = atvs_conv3d_s2b_grid(Do, Ho, Wo, groups);
  a.wg = (int)blocks;
  a.gx = (long)D * H * W * Cin; a.gy = (long)Do * Ho * Wo * ldy;
  const long grid = blocks * groups;
  if (grid > 0x7fffffffL) return ATVS_ERR_SHAPE;
  hipStream_t st = as_stream(stream);
  // all chunks' weights resident in LDS where they fit beside the two piece images
  const bool wlds = S2_NP * (size_t)S2_IMG + (size_t)a.nchunk * S2_JC * (Cout / 16) * S2_NP * 1024 <= 160 * 1024;
  int rc;
  if (in_par)
    rc = (Cout == 32) ? (wlds ? launch_s2b<2, true, true>(a, grid, st) : launch_s2b<2, false, true>(a, grid, st))
                      : (wlds ? launch_s2b<4, true, true>(a, grid, st) : launch_s2b<4, false, true>(a, grid, st));
  else
    rc = (Cout == 32) ? (wlds ? launch_s2b<2, true, false>(a, grid, st) : launch_s2b<2, false, false>(a, grid, st))
                      : (wlds ? launch_s2b<4, true, false>(a, grid, st) : launch_s2b<4, false, false>(a, grid, st));
  if (rc) return rc;
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}
}  // namespace

// y (G,Do,Ho,Wo,ldy)[..., y_coff : y_coff + Cout] = conv3d(x (G,D,H,W,Cin), w [3,3,3,Cin,Cout], stride 2, SAME) (+ bias, ReLU),
// Do = ceil(D / 2) ...; Cin % 16 == 0, Cout 32 or 64, split-fp16 operands (fp32-class results).  stats_partial: groups *
// atvs_conv3d_s2b_grid(Do,Ho,Wo,groups) rows of [2][Cout] doubles or NULL.
extern "C" int atvs_conv3d_s2b_f32(const float* x, const unsigned char* packed_w, const float* bias, float* y,
                                   double* stats_partial, int groups, int D, int H, int W, int Cin, int Cout, int ldy, int y_coff,
                                   int relu, atvs_stream_t stream) {
  return s2b_launch(x, nullptr, 0, packed_w, bias, y, stats_partial, groups, D, H, W, Cin, Cout, ldy, y_coff, relu, stream);
}

// The same convolution of relu?((x - mean) * scale + beta), x a raw convolution output and in_params (groups,3,Cin) its pending
// batch norm (the encoders conv_b*_{2,3}_0 read conv_b*_{1,2}_0, reference cnn_wrapper/atvsnet.py:10-12): formed per staged halo
// voxel, the normalised tensor is never written.  Bit for bit atvs_bn_apply followed by atvs_conv3d_s2b_f32.
extern "C" int atvs_conv3d_s2b_norm_f32(const float* x, const float* in_params, int in_relu, const unsigned char* packed_w,
                                        const float* bias, float* y, double* stats_partial, int groups, int D, int H, int W,
                                        int Cin, int Cout, int ldy, int y_coff, int relu, atvs_stream_t stream) {
  if (!in_params) return ATVS_ERR_NULL;
  return s2b_launch(x, in_params, in_relu, packed_w, bias, y, stats_partial, groups, D, H, W, Cin, Cout, ldy, y_coff, relu, stream);
}
