// AANet aggregation over the source views in ONE launch (gfx950, split fp16 operands): the shared | unique 3x3x3 score
// convolutions of every view AND the cross-view softmax + weighted sum.
//
// Reference: Network.attention_activation / attention_aggregation, /root/reference/cnn_wrapper/network.py:282-351, 378-408 with
// second_weight=True, relu=True, biased=False (call sites cnn_wrapper/atvsnet.py:202,234):
//   S_n = relu(conv3d(X_n, W_shared)),  R_n = relu(conv3d(X_n, W_unique))           (8 -> 8 channels each, per view n)
//   S_sum = sum_n S_n;  U_n = (R_n - S_n) + S_sum;  score = softmax_n(U);  out = sum_n score_n * X_n
// Until round 4 this was two launches per module: conv_c16b<8> wrote [S|R] of every view (960 MB at 4 views of configs[2]) and
// aanet_combine read it back with the X_n (295 us of pure traffic at the HBM rate).  Here a workgroup owns a 4(z) x 8(y) x 16(x)
// tile and walks the views: per view conv_c16b's stage (halo of the next stage fetched during the K loop, split on the way into
// LDS, 7 K steps of four taps x 8 channels, three products) whose results stay in REGISTERS; after the last view the lanes combine.
// The 16 MFMA rows are ordered (S[2q], S[2q+1], R[2q], R[2q+1]) for lane group q (atvs_aanet_b_pack), so a lane holds S AND R of
// its two channels -- no cross-lane exchange -- and reads the matching two channels of every X_n's centre voxel (L2 hits: the
// halo just came through).  The arithmetic and its order are conv_c16b's and aanet_combine_kernel's: bit for bit the two launches.
// One workgroup per CU, register file reserved, scalar fp32 arithmetic.
#include <cstring>
#include <type_traits>
#include <utility>

#include "conv_common.h"

namespace {

constexpr int AB_TZ = 4, AB_TY = 8, AB_TX = 16;
constexpr int AB_HZ = AB_TZ + 2, AB_HY = AB_TY + 2, AB_HX = AB_TX + 2;
constexpr int AB_VB = 16;                                      // bytes per voxel of one piece image (8 channels x 2 B)
constexpr int AB_ROWB = AB_HX * AB_VB;
constexpr int AB_IMG = AB_HZ * AB_HY * AB_ROWB;                // 17,280 bytes per piece
constexpr int AB_SLOTS = AB_HZ * AB_HY * AB_HX * 2;            // float4 slots of the fp32 halo
constexpr int AB_MAXS = (AB_SLOTS + 255) / 256;                // 9 per thread
constexpr int AB_JC = 7;                                       // K steps: taps 4 j + q (tap 27 = zero weights)
constexpr int AB_NP = 2;
constexpr int AB_WSTEP = AB_NP * 1024;
constexpr int AB_MAXV = 4;                                     // views per launch: [S|R] of 4 views x 8 rows = 128 registers (8 views spill)
constexpr float AB_RS = 2048.f, AB_IRS = 1.f / 2048.f;
static_assert(AB_MAXS <= 2 * AB_JC, "two halo slots per K step");

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

constexpr int ab_clamp26(int t) { return t < 26 ? t : 26; }
constexpr int ab_disp(int t) { return ((t / 9) * AB_HY + (t / 3) % 3) * AB_ROWB + (t % 3) * AB_VB; }

// NV: the number of views (compile time: no branches in the combine, exactly NV register slots for [S|R])
template <int NV>
__global__ __launch_bounds__(256, 1) void aanet_b_kernel(AbArgs p) {
  asm volatile("" ::: "v255", "a255");                         // own the SIMD's register file (conv_c16b.hip)
  constexpr int TY = AB_TY, HY = AB_HY, MAXS = AB_MAXS, JC = AB_JC;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;

  {
    const float4* src = reinterpret_cast<const float4*>(p.wp);
    float4* dst = reinterpret_cast<float4*>(smem + AB_NP * AB_IMG);
    for (int i = tid; i < JC * (AB_WSTEP / 16); i += 256) dst[i] = src[i];
  }
  const int fbase = ((wave * HY) * AB_HX + r) * AB_VB;
  const int wbase = AB_NP * AB_IMG + lane * 16;

  int goff[MAXS], laddr[MAXS];
  unsigned pg[MAXS];
#pragma unroll
  for (int i = 0; i < MAXS; ++i) {
    int s = tid + i * 256;
    const bool live = s < AB_SLOTS;
    s = min(s, AB_SLOTS - 1);
    const int c4 = s % 2, v = s / 2;
    const int xx = v % AB_HX, v2 = v / AB_HX;
    const int yy = v2 % HY, zz = v2 / HY;
    goff[i] = ((zz * p.Hi + yy) * p.Wi + xx) * 8 + c4 * 4;
    laddr[i] = ((zz * HY + yy) * AB_HX + xx) * AB_VB + c4 * 8;
    pg[i] = 0x808080u | (unsigned)(live ? zz : 0x7f) | ((unsigned)yy << 8) | ((unsigned)xx << 16);
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
  auto tile_origin = [&](int k, int* z0, int* y0, int* x0) __attribute__((always_inline)) {
    int tl = xcd * per_xcd + tslot + k * slots_per_xcd;
    int bx = tl % p.tiles_x;
    int rest = tl / p.tiles_x;
    *x0 = bx * AB_TX;
    *y0 = (rest % p.tiles_y) * TY;
    *z0 = (rest / p.tiles_y) * AB_TZ;
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
    T.org = ((gz0 * p.Hi + gy0) * p.Wi + gx0) * 8;
    T.lo = (unsigned)(gz0 < 0) | ((unsigned)(gy0 < 0) << 8) | ((unsigned)(gx0 < 0) << 16);
    T.hi1 = (unsigned)(min(p.Di - 1 - gz0, 0x7e) + 1) | ((unsigned)(min(p.Hi - 1 - gy0, 0x7e) + 1) << 8) |
            ((unsigned)(min(p.Wi - 1 - gx0, 0x7e) + 1) << 16);
    return T;
  };
  float4 pf[MAXS];
  auto pf_slot = [&](const PfTile& T, const float* xg, int i) __attribute__((always_inline)) {
    const unsigned t1 = pg[i] - T.lo;
    const unsigned t2 = T.hi1 + ~pg[i];
    const bool ok = ((t1 & t2) & 0x808080u) == 0x808080u;
    pf[i] = ld4(ok ? (xg + (T.org + goff[i])) : p.zeros);
  };

  // [S(2q) S(2q+1) R(2q) R(2q+1)] of voxel (z0 + wave, y0 + t, x0 + r) per view
  float4 sr[NV][TY];
  f32x4 acc[TY], accx[TY];

  if (my_tiles > 0) {
    const PfTile T0 = pf_tile(0);
#pragma unroll
    for (int i = 0; i < MAXS; ++i) pf_slot(T0, p.x[0], i);
  }

  for (int k = 0; k < my_tiles; ++k) {
    // the views as a ROLLED loop (unrolled -- [S|R] into static register slots, no selects -- the launch was SLOWER: 663 instead
    // of 628 us, four copies of the stage body)
#pragma unroll 1
    for (int v = 0; v < NV; ++v) {
#pragma unroll
      for (int t = 0; t < TY; ++t) acc[t] = accx[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      __syncthreads();                       // every wavefront is done reading the previous stage's images
#pragma unroll
      for (int i = 0; i < MAXS; ++i) {
        if (i < MAXS - 1 || tid + i * 256 < AB_SLOTS) {
          f16x4 p0, p1;
          ab_split(pf[i], &p0, &p1);
          *reinterpret_cast<f16x4*>(smem + laddr[i]) = p0;
          *reinterpret_cast<f16x4*>(smem + AB_IMG + laddr[i]) = p1;
        }
      }
      __syncthreads();

      // the next stage: the next view of this tile, else view 0 of the next tile (last stage: a harmless re-read)
      const bool last_view = v + 1 == NV;
      const int vn = last_view ? 0 : v + 1;
      const PfTile T = pf_tile(last_view ? min(k + 1, my_tiles - 1) : k);
      const float* __restrict__ xn = p.x[vn];

      f16x8 Bq[2][TY], A[2][AB_NP];
      auto request_B = [&](auto PH) __attribute__((always_inline)) {
        constexpr int ph = decltype(PH)::value, j = ph / AB_NP, pc = ph % AB_NP;
        constexpr int tA = ab_clamp26(4 * j), tB = ab_clamp26(4 * j + 1), tC = ab_clamp26(4 * j + 2), tD = ab_clamp26(4 * j + 3);
        const int a = fbase + ((q & 2) ? ((q & 1) ? ab_disp(tD) : ab_disp(tC)) : ((q & 1) ? ab_disp(tB) : ab_disp(tA)));
#pragma unroll
        for (int t = 0; t < TY; ++t) Bq[ph & 1][t] = *reinterpret_cast<const f16x8*>(smem + pc * AB_IMG + a + t * AB_ROWB);
      };
      auto request_B1 = [&](auto PH, auto TT) __attribute__((always_inline)) {
        constexpr int ph = decltype(PH)::value, j = ph / AB_NP, pc = ph % AB_NP, t = decltype(TT)::value;
        constexpr int tA = ab_clamp26(4 * j), tB = ab_clamp26(4 * j + 1), tC = ab_clamp26(4 * j + 2), tD = ab_clamp26(4 * j + 3);
        const int a = fbase + ((q & 2) ? ((q & 1) ? ab_disp(tD) : ab_disp(tC)) : ((q & 1) ? ab_disp(tB) : ab_disp(tA)));
        Bq[ph & 1][t] = *reinterpret_cast<const f16x8*>(smem + pc * AB_IMG + a + t * AB_ROWB);
      };
#pragma unroll
      for (int pc = 0; pc < AB_NP; ++pc) A[0][pc] = *reinterpret_cast<const f16x8*>(smem + wbase + pc * 1024);
      request_B(IC<0>{});
      asm volatile("" ::: "memory");
      ab_static_for<AB_NP * JC>([&](auto PH) __attribute__((always_inline)) {
        constexpr int ph = decltype(PH)::value, j = ph / AB_NP, pc = ph % AB_NP;
        ab_static_for<(2 - pc) * TY>([&](auto M) __attribute__((always_inline)) {
          constexpr int m = decltype(M)::value, jw = m / TY, t = m % TY;
          if constexpr (pc == 0 && jw == 0) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[j & 1][0], Bq[ph & 1][t], acc[t], 0, 0, 0);
          else accx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[j & 1][pc == 0 ? 1 : 0], Bq[ph & 1][t], accx[t], 0, 0, 0);
          if constexpr (m < TY) {
            if constexpr (ph + 1 < AB_NP * JC) request_B1(IC<ph + 1>{}, IC<m>{});
          } else if constexpr (m < TY + AB_NP) {
            if constexpr (j + 1 < JC) A[(j + 1) & 1][m - TY] = *reinterpret_cast<const f16x8*>(smem + wbase + (j + 1) * AB_WSTEP + (m - TY) * 1024);
          } else if constexpr (m == TY + 3 || m == TY + 6) {
            constexpr int slot = 2 * j + (m == TY + 6 ? 1 : 0);
            if constexpr (slot < MAXS) pf_slot(T, xn, slot);
          }
          asm volatile("" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        });
      });

      // this view's [S|R] of the lane's voxels: conv_c16b's epilogue arithmetic (zero bias, ReLU), kept in registers
#pragma unroll
      for (int t = 0; t < TY; ++t) {
        float a0 = (acc[t][0] + accx[t][0] * AB_IRS) + 0.f, a1 = (acc[t][1] + accx[t][1] * AB_IRS) + 0.f;
        float a2 = (acc[t][2] + accx[t][2] * AB_IRS) + 0.f, a3 = (acc[t][3] + accx[t][3] * AB_IRS) + 0.f;
        a0 = (a0 < 0.f) ? 0.f : a0; a1 = (a1 < 0.f) ? 0.f : a1;
        a2 = (a2 < 0.f) ? 0.f : a2; a3 = (a3 < 0.f) ? 0.f : a3;
#pragma unroll
        for (int n = 0; n < NV; ++n) {                         // v is a run-time index: a register array takes it as selects
          const bool here = v == n;
          sr[n][t].x = here ? a0 : sr[n][t].x;
          sr[n][t].y = here ? a1 : sr[n][t].y;
          sr[n][t].z = here ? a2 : sr[n][t].z;
          sr[n][t].w = here ? a3 : sr[n][t].w;
        }
      }
      if (!last_view) continue;

      // ---- combine (aanet_combine_kernel's arithmetic and order) for channels 2q, 2q+1 of the lane's TY voxels
      int tz0, ty0, tx0;
      tile_origin(k, &tz0, &ty0, &tx0);
      const int zo = tz0 + wave, xo = tx0 + r;
      const bool evox_ok = zo < p.Di && xo < p.Wi;
      const size_t vo0 = (((size_t)zo * p.Hi + ty0) * p.Wi + xo) * 8 + 2 * q;
      const size_t vrow = (size_t)p.Wi * 8;
      // the centre voxels of the views (L2 hits: the halo just came through), one row AHEAD of the arithmetic: requested where
      // they are used (round 5, first form) every row waited a full round trip -- 8 per tile.  (All rows in front of the last view's
      // K loop: 64 more live registers, 15 spilled, 628 -> 721 us.)  Rows outside the volume read the zero line.
      float2 xv[2][NV];
      auto request_x = [&](int t) __attribute__((always_inline)) {
        const bool ok = evox_ok && ty0 + t < p.Hi;
#pragma unroll
        for (int n = 0; n < NV; ++n)
          xv[t & 1][n] = *reinterpret_cast<const float2*>(ok ? (p.x[n] + (vo0 + t * vrow)) : p.zeros);
      };
      request_x(0);
#pragma unroll
      for (int t = 0; t < TY; ++t) {
        if (t + 1 < TY) request_x(t + 1);
        const bool ok = evox_ok && ty0 + t < p.Hi;
        float sx = 0.f, sy = 0.f;
        float ux[NV], uy[NV];
#pragma unroll
        for (int n = 0; n < NV; ++n) {
          sx += sr[n][t].x; sy += sr[n][t].y;
          ux[n] = sr[n][t].z - sr[n][t].x; uy[n] = sr[n][t].w - sr[n][t].y;
        }
        float mx = -INFINITY, my = -INFINITY;
#pragma unroll
        for (int n = 0; n < NV; ++n) {
          ux[n] += sx; uy[n] += sy;
          mx = fmaxf(mx, ux[n]); my = fmaxf(my, uy[n]);
        }
        float dx = 0.f, dy = 0.f;
#pragma unroll
        for (int n = 0; n < NV; ++n) {
          ux[n] = expf(ux[n] - mx); uy[n] = expf(uy[n] - my);
          dx += ux[n]; dy += uy[n];
        }
        float ox = 0.f, oy = 0.f;
#pragma unroll
        for (int n = 0; n < NV; ++n) {
          ox += (ux[n] / dx) * xv[t & 1][n].x;
          oy += (uy[n] / dy) * xv[t & 1][n].y;
        }
        if (ok) *reinterpret_cast<float2*>(p.out + (vo0 + t * vrow)) = make_float2(ox, oy);
      }
    }
  }
}

template <int NV>
int launch_ab(const AbArgs& a, long grid, hipStream_t s) {
  const size_t lds = AB_NP * (size_t)AB_IMG + (size_t)AB_JC * AB_WSTEP;
  hipLaunchKernelGGL((aanet_b_kernel<NV>), dim3((unsigned)grid), dim3(256), lds, s, a);
  return ATVS_OK;
}

}  // namespace

extern "C" int atvs_aanet_b_pack_size(long* packed_bytes) {
  if (!packed_bytes) return ATVS_ERR_NULL;
  *packed_bytes = (long)AB_JC * AB_WSTEP + 16;
  return ATVS_OK;
}

// HOST function.  w_shared, w_unique: TF kernels [3,3,3,8,8] (attention_activation/weight_shared, weight_unique).
// packed[step j][piece][lane = q*16 + row][8 fp16] = piece of w[tap 4 j + q][ci = e][row -> (S | R, channel)]: row 4 g + i of lane group
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
      const int tap = 4 * j + q;
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
// the AANet module (reference cnn_wrapper/network.py:282-351,378-408) over nv <= 4 views in one launch (more views: the two-launch form).  x: HOST array of nv
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
  const int rc = nv == 1 ? launch_ab<1>(a, grid, st) : nv == 2 ? launch_ab<2>(a, grid, st) : nv == 3 ? launch_ab<3>(a, grid, st) : launch_ab<4>(a, grid, st);
  if (rc) return rc;
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}
