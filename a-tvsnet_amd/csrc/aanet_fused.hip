// AANet aggregation over source views in ONE launch (gfx950): the shared | unique 3x3x3 8 -> 16 channel convolution of every
// view (conv_c16b.hip, split-bf16 operands, ReLU) and the cross-view softmax + weighted sum (aanet.hip) without the
// [S | R] tensors ever reaching memory.  /root/reference/cnn_wrapper/network.py:282-351, 378-408 (attention_activation /
// attention_aggregation, second_weight=True, relu=True, biased=False; call sites cnn_wrapper/atvsnet.py:202,234):
//   S_n | R_n = relu(conv(X_n, shared | unique));  U_n = (R_n - S_n) + sum_m S_m;  out = sum_n softmax_n(U)_n X_n.
// Separately the scores cost a 1 GB write + read per module at configs[2] (4 views x 3.9 M voxels x 16 channels) and a second
// read of X.  Here a workgroup keeps the accumulators of ALL views of its 4 x 8 x 16 tile (NV x 8 tiles per wavefront) and
// combines them in the epilogue: lanes q and q + 2 hold S and R of the same channels (one cross-lane exchange), X comes from
// L2.  Arithmetic and its order are those of atvs_conv_c16b_f32 followed by atvs_aanet_combine: results are bit-identical to
// the two launches (tests/test_gpu_conv.py::test_aanet_fused_equals_two_launches).
#include <cstring>
#include <type_traits>
#include <utility>

#include "conv_common.h"

extern "C" long atvs_conv_c16_grid(int D, int H, int W, int groups);

namespace {

constexpr int AF_TZ = 4, AF_TY = 8, AF_TX = 16;
constexpr int AF_HZ = AF_TZ + 2, AF_HY = AF_TY + 2, AF_HX = AF_TX + 2;
constexpr int AF_VB = 16;                                      // 8 channels x 2 bytes per voxel of one piece image
constexpr int AF_ROWB = AF_HX * AF_VB;
constexpr int AF_IMG = AF_HZ * AF_HY * AF_ROWB;                // 17,280
constexpr int AF_SLOTS = AF_HZ * AF_HY * AF_HX * 2;
constexpr int AF_MAXS = (AF_SLOTS + 255) / 256;                // 9
constexpr int AF_JC = 7;                                       // K steps: four taps x 8 channels (tap 27 = zero weights)
constexpr int AF_WSTEP = 3 * 1024;
constexpr int af_clamp26(int t) { return t < 26 ? t : 26; }
constexpr int af_disp(int t) { return ((t / 9) * AF_HY + (t / 3) % 3) * AF_ROWB + (t % 3) * AF_VB; }

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct AfArgs {
  const float* x;              // (NV, D, H, W, 8)
  const unsigned char* wp;     // atvs_conv_c16b_pack(Cin = 8) of [shared | unique]
  const float* zeros;
  float* out;                  // (D, H, W, 8)
  int Di, Hi, Wi;
  int tiles_y, tiles_x, ntiles;
  int wg;
  long gx;                     // floats per view
};

template <int N>
using IC = std::integral_constant<int, N>;
template <class F, int... I>
__device__ __forceinline__ void af_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(IC<I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void af_static_for(F&& f) {
  af_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

__device__ __forceinline__ void af_split(const float4& v, bf16x4* p0, bf16x4* p1, bf16x4* p2) {
  const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const __bf16 a = (__bf16)x[i];
    const float r1 = x[i] - (float)a;
    const __bf16 b = (__bf16)r1;
    const float r2 = r1 - (float)b;
    (*p0)[i] = a;
    (*p1)[i] = b;
    (*p2)[i] = (__bf16)r2;
  }
}

template <int NV>
__global__ __launch_bounds__(256, 1) void aanet_fused_kernel(AfArgs p) {
  // own the SIMD's whole register file (512 per lane): no wavefront of ANOTHER kernel runs beside this one's bf16 MFMAs --
  // beside them other kernels' wavefronts computed wrong lane quarters (tools_dev/micro/pk_beside_mfma.hip)
  asm volatile("" ::: "v255", "a255");
  constexpr int TY = AF_TY, HY = AF_HY, MAXS = AF_MAXS, JC = AF_JC;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;

  {
    const float4* src = reinterpret_cast<const float4*>(p.wp);
    float4* dst = reinterpret_cast<float4*>(smem + 3 * AF_IMG);
    for (int i = tid; i < JC * (AF_WSTEP / 16); i += 256) dst[i] = src[i];
  }
  const int fbase = ((wave * HY) * AF_HX + r) * AF_VB;
  const int wbase = 3 * AF_IMG + lane * 16;

  int goff[MAXS], laddr[MAXS];
  unsigned pg[MAXS];
#pragma unroll
  for (int i = 0; i < MAXS; ++i) {
    int s = tid + i * 256;
    const bool live = s < AF_SLOTS;
    s = min(s, AF_SLOTS - 1);
    const int c4 = s & 1, v = s >> 1;
    const int xx = v % AF_HX, v2 = v / AF_HX;
    const int yy = v2 % HY, zz = v2 / HY;
    goff[i] = ((zz * p.Hi + yy) * p.Wi + xx) * 8 + c4 * 4;
    laddr[i] = ((zz * HY + yy) * AF_HX + xx) * AF_VB + c4 * 8;
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
    *x0 = bx * AF_TX;
    *y0 = (rest % p.tiles_y) * TY;
    *z0 = (rest / p.tiles_y) * AF_TZ;
  };
  struct PfTile {
    const float* xb;
    int org;
    unsigned lo, hi1;
  };
  auto pf_tile = [&](int k, int v) __attribute__((always_inline)) {      // tile k of view v
    PfTile T;
    int z0, y0, x0;
    tile_origin(k, &z0, &y0, &x0);
    const int gz0 = z0 - 1, gy0 = y0 - 1, gx0 = x0 - 1;
    T.xb = p.x + (size_t)v * p.gx;
    T.org = ((gz0 * p.Hi + gy0) * p.Wi + gx0) * 8;
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

  f32x4 acc[NV][TY];
  const size_t vol = (size_t)p.Di * p.Hi * p.Wi;
  const unsigned obytes = (unsigned)(vol * 8 * 4);
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, obytes, 0x00020000);

  if (my_tiles > 0) {
    const PfTile T0 = pf_tile(0, 0);
#pragma unroll
    for (int i = 0; i < MAXS; ++i) pf_slot(T0, i);
  }

  for (int k = 0; k < my_tiles; ++k) {
    af_static_for<NV>([&](auto VT) __attribute__((always_inline)) {
      constexpr int v = decltype(VT)::value;
#pragma unroll
      for (int t = 0; t < TY; ++t) acc[v][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      __syncthreads();                       // every wavefront is done reading the previous stage's images
#pragma unroll
      for (int i = 0; i < MAXS; ++i) {
        if (i < MAXS - 1 || tid + i * 256 < AF_SLOTS) {
          bf16x4 p0, p1, p2;
          af_split(pf[i], &p0, &p1, &p2);
          *reinterpret_cast<bf16x4*>(smem + laddr[i]) = p0;
          *reinterpret_cast<bf16x4*>(smem + AF_IMG + laddr[i]) = p1;
          *reinterpret_cast<bf16x4*>(smem + 2 * AF_IMG + laddr[i]) = p2;
        }
      }
      __syncthreads();

      // the next stage: the next view of this tile, or the first view of the next tile (last stage: harmless re-read)
      const PfTile T = (v + 1 < NV) ? pf_tile(k, v + 1) : pf_tile(min(k + 1, my_tiles - 1), 0);

      // ---- K loop of conv_c16b.hip (Cin = 8): 7 steps of four taps x 8 channels, three phases each
      bf16x8 Bq[2][TY], A[2][3];
      auto request_B = [&](auto PH) __attribute__((always_inline)) {
        constexpr int ph = decltype(PH)::value, j = ph / 3, pc = ph % 3;
        constexpr int tA = af_clamp26(4 * j), tB = af_clamp26(4 * j + 1), tC = af_clamp26(4 * j + 2), tD = af_clamp26(4 * j + 3);
        const int a = fbase + ((q & 2) ? ((q & 1) ? af_disp(tD) : af_disp(tC)) : ((q & 1) ? af_disp(tB) : af_disp(tA)));
#pragma unroll
        for (int t = 0; t < TY; ++t) Bq[ph & 1][t] = *reinterpret_cast<const bf16x8*>(smem + pc * AF_IMG + a + t * AF_ROWB);
      };
      auto request_A = [&](auto JT) __attribute__((always_inline)) {
        constexpr int j = decltype(JT)::value;
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) A[j & 1][pc] = *reinterpret_cast<const bf16x8*>(smem + wbase + j * AF_WSTEP + pc * 1024);
      };
      request_A(IC<0>{});
      request_B(IC<0>{});
      asm volatile("" ::: "memory");
      af_static_for<3 * JC>([&](auto PH) __attribute__((always_inline)) {
        constexpr int ph = decltype(PH)::value, j = ph / 3, pc = ph % 3;
        if constexpr (ph + 1 < 3 * JC) request_B(IC<ph + 1>{});
        if constexpr (pc == 0 && j + 1 < JC) request_A(IC<j + 1>{});
        if constexpr (ph < MAXS) pf_slot(T, ph);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jw = 0; jw <= 2 - pc; ++jw)
#pragma unroll
          for (int t = 0; t < TY; ++t)
            acc[v][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[j & 1][jw], Bq[ph & 1][t], acc[v][t], 0, 0, 0);
      });
    });

    // ---- epilogue: this lane holds channels 4q..4q+3 of [S | R] (q < 2: S of channels 4q.., q >= 2: R of channels 4(q-2)..) of
    // voxel (z0 + wave, y0 + t, x0 + r) for every view.  Rows are combined in PAIRS so that every lane works: lanes q < 2 take
    // row t (S their own, R from lane + 32), lanes q >= 2 row t + 1 (R their own, S from lane - 32) -- one exchange per value;
    // X comes from L2 (requested one row pair ahead).  Arithmetic of aanet_combine_kernel (aanet.hip).
    int tz0, ty0, tx0;
    tile_origin(k, &tz0, &ty0, &tx0);
    const int zo = tz0 + wave, xo = tx0 + r;
    const bool hi = q >= 2;
    const bool evox_ok = zo < p.Di && xo < p.Wi;
    const unsigned erow = (unsigned)p.Wi * 8;
    const unsigned eo = (((unsigned)zo * p.Hi + ty0 + (hi ? 1 : 0)) * p.Wi + xo) * 8 + (q & 1) * 4;      // this lane's row of pair 0
    float4 xv[2][NV];
    auto load_x = [&](int t, int buf) __attribute__((always_inline)) {       // pair starting at row t
      const bool ok = evox_ok && ty0 + t + (hi ? 1 : 0) < p.Hi;
#pragma unroll
      for (int n = 0; n < NV; ++n) xv[buf][n] = ld4(ok ? p.x + (size_t)n * p.gx + (eo + t * erow) : p.zeros);
    };
    load_x(0, 0);
    af_static_for<TY / 2>([&](auto TT) __attribute__((always_inline)) {
      constexpr int t = 2 * decltype(TT)::value;
      if constexpr (t + 2 < TY) load_x(t + 2, (t / 2 + 1) & 1);
      float4 u[NV];
      float4 ssum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int n = 0; n < NV; ++n) {
        float4 a0, a1;                             // ReLU of the convolution (conv_c16b.hip's epilogue): NaN passes
        a0.x = (acc[n][t][0] < 0.f) ? 0.f : acc[n][t][0];
        a0.y = (acc[n][t][1] < 0.f) ? 0.f : acc[n][t][1];
        a0.z = (acc[n][t][2] < 0.f) ? 0.f : acc[n][t][2];
        a0.w = (acc[n][t][3] < 0.f) ? 0.f : acc[n][t][3];
        a1.x = (acc[n][t + 1][0] < 0.f) ? 0.f : acc[n][t + 1][0];
        a1.y = (acc[n][t + 1][1] < 0.f) ? 0.f : acc[n][t + 1][1];
        a1.z = (acc[n][t + 1][2] < 0.f) ? 0.f : acc[n][t + 1][2];
        a1.w = (acc[n][t + 1][3] < 0.f) ? 0.f : acc[n][t + 1][3];
        // lanes q < 2 keep row t and send row t + 1; lanes q >= 2 keep row t + 1 and send row t
        const float4 own = hi ? a1 : a0, snd = hi ? a0 : a1;
        float4 rcv;
        rcv.x = __shfl_xor(snd.x, 32); rcv.y = __shfl_xor(snd.y, 32); rcv.z = __shfl_xor(snd.z, 32); rcv.w = __shfl_xor(snd.w, 32);
        const float4 sv = hi ? rcv : own, rv = hi ? own : rcv;
        ssum.x += sv.x; ssum.y += sv.y; ssum.z += sv.z; ssum.w += sv.w;
        u[n] = make_float4(rv.x - sv.x, rv.y - sv.y, rv.z - sv.z, rv.w - sv.w);
      }
      float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
      for (int n = 0; n < NV; ++n) {
        u[n].x += ssum.x; u[n].y += ssum.y; u[n].z += ssum.z; u[n].w += ssum.w;      // (R - S) + S_sum
        m.x = fmaxf(m.x, u[n].x); m.y = fmaxf(m.y, u[n].y); m.z = fmaxf(m.z, u[n].z); m.w = fmaxf(m.w, u[n].w);
      }
      float4 den = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int n = 0; n < NV; ++n) {
        u[n].x = expf(u[n].x - m.x); u[n].y = expf(u[n].y - m.y); u[n].z = expf(u[n].z - m.z); u[n].w = expf(u[n].w - m.w);
        den.x += u[n].x; den.y += u[n].y; den.z += u[n].z; den.w += u[n].w;
      }
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int n = 0; n < NV; ++n) {
        const float4 x = xv[(t / 2) & 1][n];
        o.x += (u[n].x / den.x) * x.x;
        o.y += (u[n].y / den.y) * x.y;
        o.z += (u[n].z / den.z) * x.z;
        o.w += (u[n].w / den.w) * x.w;
      }
      const u32x4 bits = {__builtin_bit_cast(unsigned, o.x), __builtin_bit_cast(unsigned, o.y),
                          __builtin_bit_cast(unsigned, o.z), __builtin_bit_cast(unsigned, o.w)};
      const bool row_ok = evox_ok && ty0 + t + (hi ? 1 : 0) < p.Hi;
      __builtin_amdgcn_raw_buffer_store_b128(bits, orsrc, row_ok ? eo * 4u : obytes, (t * erow) * 4u, 0);
    });
  }
}

template <int NV>
int launch_af(const AfArgs& a, long grid, hipStream_t s) {
  const size_t lds = 3 * (size_t)AF_IMG + (size_t)AF_JC * AF_WSTEP;
  static bool attr_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return ATVS_ERR_LAUNCH;
  if (!attr_set[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(aanet_fused_kernel<NV>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return ATVS_ERR_LAUNCH;
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL((aanet_fused_kernel<NV>), dim3((unsigned)grid), dim3(256), lds, s, a);
  return ATVS_OK;
}

}  // namespace

extern "C" int atvs_aanet_fused_supported(int nv) { return (nv == 2 || nv == 3 || nv == 4 || nv == 8) ? 1 : 0; }

// out (D,H,W,8) = AANet aggregation of the nv views x (nv,D,H,W,8): [S_n | R_n] = relu(conv3x3x3(x_n, w)) with packed_w =
// atvs_conv_c16b_pack(Cin = 8) of the [3,3,3,8,16] kernel [shared | unique]; U_n = (R_n - S_n) + sum_m S_m; out = sum_n
// softmax_n(U)_n x_n.  Bit-identical to atvs_conv_c16b_f32 (relu) + atvs_aanet_combine.  nv: atvs_aanet_fused_supported.
extern "C" int atvs_aanet_fused_f32(const float* x, const unsigned char* packed_w, float* out, int nv, int D, int H, int W,
                                    atvs_stream_t stream) {
  if (!x || !packed_w || !out) return ATVS_ERR_NULL;
  if (!atvs_aanet_fused_supported(nv) || D <= 0 || H <= 0 || W <= 0) return ATVS_ERR_SHAPE;
  if ((double)D * H * W * 8 >= 2147483648.0 / 4) return ATVS_ERR_SHAPE;             // 32-bit output byte offsets
  AfArgs a;
  a.x = x; a.wp = packed_w; a.zeros = reinterpret_cast<const float*>(packed_w + (size_t)AF_JC * AF_WSTEP);
  a.out = out; a.Di = D; a.Hi = H; a.Wi = W;
  a.tiles_y = (H + AF_TY - 1) / AF_TY; a.tiles_x = (W + AF_TX - 1) / AF_TX;
  a.ntiles = ((D + AF_TZ - 1) / AF_TZ) * a.tiles_y * a.tiles_x;
  const long blocks = atvs_conv_c16_grid(D, H, W, 1);
  a.wg = (int)blocks;
  a.gx = (long)D * H * W * 8;
  hipStream_t st = as_stream(stream);
  int rc;
  switch (nv) {
    case 2: rc = launch_af<2>(a, blocks, st); break;
    case 3: rc = launch_af<3>(a, blocks, st); break;
    case 4: rc = launch_af<4>(a, blocks, st); break;
    default: rc = launch_af<8>(a, blocks, st); break;
  }
  if (rc) return rc;
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}
