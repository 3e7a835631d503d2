// One launch per pre-activation residual unit of the 2-D feature towers (gfx950, split fp16 operands).
//
// Network.bottleneck (/root/reference/cnn_wrapper/network.py:552-602) with depth == depth_in, stride 1 (identity shortcut):
//   preact = relu(slim.batch_norm(x))                 -- the unit's ONE global reduction: the moments of x, known at launch
//   r1 = relu(conv1 1x1 (preact) + b1)                -- nothing global from here on
//   r2 = relu(conv2 3x3 dilation d (r1, SAME) + b2)
//   y  = x + conv3 1x1 (r2) + b3
// Before round 5 this was three launches (conv1x1_b, conv2d_b, conv1x1_b: 24-38 us each at 3-9 % of the matrix pipe, r1 and r2
// through HBM).  Here a workgroup owns a TH x 16 tile of output pixels of one image and never leaves the CU:
//   P1  conv1 over the tile + its dilation halo ((TH + 2d) x (16 + 2d) pixels).  The waves split the PIXELS: a lane loads the
//       eight channels of its fragment straight from x (two float4), applies the pre-activation and splits in registers --
//       x never sits in LDS.  r1 (zero outside the image: conv2 pads r1, not x) leaves as fp16 pieces into LDS.
//   P2  conv2 from LDS, the waves split channels x rows (WN x WR), weights streamed from L2 one K step ahead.
//   P3  r2 -> fp16 pieces over the (dead) r1 image: the cross-wave exchange conv3 needs (all channels of a pixel as K).
//   P4  conv3 + b3 + x, the moments of y (the next unit's pre-activation) per workgroup.
// The K order of every product chain is the unfused kernels' (conv1x1_b: 32-channel chunks; conv2d_b: K step = two taps x 16
// channels, five steps per chunk) and the packed weights ARE theirs (atvs_conv1x1_b_pack, atvs_conv2d_b_pack): y is bit for bit
// what the three launches produce (tests/test_gpu_conv.py::test_bottleneck_fused_is_bitwise_the_three_launches); only the
// grouping of the statistics rows differs (per tile instead of per 128 pixels).
#include <cstring>
#include <type_traits>

#include "conv_common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
constexpr float BT_RS = 2048.f, BT_IRS = 1.f / 2048.f;

// Development build (-DATVS_BT_DEBUG): per-wavefront tick counts of the phases, read back with atvs_debug_read_bt
// (tools_dev/phase_bt.py).
#ifdef ATVS_BT_DEBUG
__device__ unsigned long long atvs_dbg_bt[4096 * 8];
#define BDBG(i) { unsigned long long t_ = clock64(); dbg_acc[i] += t_ - dbg_t; dbg_t = t_; }
#else
#define BDBG(i)
#endif

struct BtArgs {
  const float* x;
  const float* in_params;     // (G, 3, C): mean | 1 / sqrt(var + eps) | beta of the pre-activation batch norm
  const f16x8* w1;            // atvs_conv1x1_b_pack(conv1)
  const f16x8* w2;            // atvs_conv2d_b_pack(conv2)
  const f16x8* w3;            // atvs_conv1x1_b_pack(conv3)
  const float* b1;
  const float* b2;
  const float* b3;
  float* y;
  double* stats;              // (G, tiles, 2, C) doubles or null
  int G, H, W;
  int tiles_x, tiles;
  long total;
};

// the two fp16 pieces of four / eight values: atvs_split2_f16 (common.h: five vector instructions per two values, the values of
// the C form `h0 = f16(x); h1 = f16((x - h0) * 2048)` the unfused kernels use)
union BtQ { unsigned u[2]; f16x4 h; };
union BtO { unsigned u[4]; f16x8 h; };
__device__ __forceinline__ void bt_split(const float v[4], f16x4* p0, f16x4* p1) {
  BtQ a, b;
  atvs_split2_f16(v[0], v[1], BT_RS, &a.u[0], &b.u[0]);
  atvs_split2_f16(v[2], v[3], BT_RS, &a.u[1], &b.u[1]);
  *p0 = a.h;
  *p1 = b.h;
}
// ... of eight values that go STRAIGHT into a matrix instruction (conv1's B operand, never through LDS): the C form, not the
// inline assembly.  The hazard recogniser does not look inside an asm statement: with atvs_split2_f16 here the compiler placed a
// v_mfma_f32_16x16x32_f16 directly behind the v_cvt_pk_f16_f32 that wrote its B registers, without the wait states a
// compiler-visible VALU write gets, and on gfx950 the multiply then read stale registers (round 6: garbage in the last row group of
// the 64-channel unit as soon as the one-fma batch norm removed the few instructions that used to sit between the two).  Eight to
// nine instructions per two values instead of five, ~100 per tile; the values are the same (every step is exact before its one
// rounding), so the unit stays bitwise the three launches.
typedef float bt_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 bt_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void bt_split8(const float v[8], f16x8* p0, f16x8* p1) {
  f16x8 a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bt_f32x2 x = {v[2 * i], v[2 * i + 1]};
    const bt_f16x2 h = __builtin_convertvector(x, bt_f16x2);
    const bt_f32x2 r = {(x[0] - (float)h[0]) * BT_RS, (x[1] - (float)h[1]) * BT_RS};
    const bt_f16x2 l = __builtin_convertvector(r, bt_f16x2);
    a[2 * i] = h[0];
    a[2 * i + 1] = h[1];
    b[2 * i] = l[0];
    b[2 * i + 1] = l[1];
  }
  *p0 = a;
  *p1 = b;
}

// Workgroup barrier for LDS hand-offs that leaves global loads IN FLIGHT: __syncthreads() drains vmcnt too, which would turn every
// prefetch issued in front of it into an exposed round trip.  LDS traffic is all the barriers of this kernel order.
__device__ __forceinline__ void bt_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Sum of a double over the 16 lanes of a row (every lane ends with the row's sum), by DPP instead of four ds_bpermute round trips:
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror -- the partners hold equal partial sums, so the values are
// those of the xor butterfly.
template <int CTRL>
__device__ __forceinline__ double bt_dpp_add(double a) {
  const long long b = __builtin_bit_cast(long long, a);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
  return a + __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ double bt_row_sum(double a) {
  a = bt_dpp_add<0xB1>(a);
  a = bt_dpp_add<0x4E>(a);
  a = bt_dpp_add<0x141>(a);
  return bt_dpp_add<0x140>(a);
}

// C channels, dilation DIL, TH x 16 output pixels per workgroup; P2 / P4: WN waves across the channels x WR across the rows.
template <int C, int DIL, int TH, int WN, int WR>
__global__ __launch_bounds__(256, 2) void bottleneck_b_kernel(BtArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  static_assert(WN * WR == 4, "four waves");
  constexpr int NT = C / 16;                 // 16-channel output tiles
  constexpr int NTW = NT / WN;               // ... per wave (P2, P4)
  constexpr int TY = TH / WR;                // rows per wave (P2, P4)
  constexpr int HC = 16 + 2 * DIL, HR = TH + 2 * DIL;
  constexpr int NPH = HR * HC;               // pixels of the tile + halo
  constexpr int NG1 = (NPH + 15) / 16;       // 16-pixel groups of conv1
  constexpr int G1W = (NG1 + 3) / 4;         // ... per wave
  constexpr int PITCH = C * 2 + 16;          // bytes per pixel of a piece image: 16 consecutive pixels x 16 B cover all banks
  constexpr int PIMG = NG1 * 16 * PITCH;     // bytes of one piece image (r1; r2 = the first TH * 16 pixels of it)
  constexpr int NCH32 = C / 32, NCH16 = C / 16, JS = 5;
  static_assert(NT % WN == 0 && TH % WR == 0, "wave grid");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int wn = wave % WN, wr = wave / WN;

  const long per = (p.total + 7) >> 3;       // consecutive tiles stay on one XCD (its L2 holds their shared halo rows)
  const long lin = (long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (lin >= p.total) return;
  const int g = (int)(lin / p.tiles), tile = (int)(lin % p.tiles);
  const int y0 = (tile / p.tiles_x) * TH, x0 = (tile % p.tiles_x) * 16;
  const float* __restrict__ xg = p.x + (size_t)g * p.H * p.W * C;
#ifdef ATVS_BT_DEBUG
  unsigned long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long dbg_t = clock64();
#endif

  // ------------------------------------------------------------------ P1: conv1 over tile + halo, pixels split over the waves
  {
    // this lane's pixel of each of the wave's groups (group = wave + 4 i) and its fragment loads: channels ch * 32 + q * 8 ..
    float4 xa[G1W][NCH32], xb[G1W][NCH32];
    int hp[G1W];
    bool in[G1W];
#pragma unroll
    for (int i = 0; i < G1W; ++i) {
      const int grp = wave + 4 * i;
      hp[i] = grp * 16 + r;
      const int hy = hp[i] / HC, hx = hp[i] - hy * HC;
      const int gy = y0 - DIL + hy, gx = x0 - DIL + hx;
      in[i] = grp < NG1 && hp[i] < NPH && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
      const float* px = xg + ((size_t)gy * p.W + gx) * C + q * 8;
#pragma unroll
      for (int ch = 0; ch < NCH32; ++ch) {
        xa[i][ch] = in[i] ? ld4(px + ch * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
        xb[i][ch] = in[i] ? ld4(px + ch * 32 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    // conv1's weight pieces: [chunk][tile][piece][lane]
    f16x8 A1[NCH32][NT][2];
#pragma unroll
    for (int ch = 0; ch < NCH32; ++ch)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int w = 0; w < 2; ++w) A1[ch][n][w] = p.w1[((size_t)(ch * NT + n) * 2 + w) * 64 + lane];
    // the pre-activation's parameters of this lane's eight channels per chunk and conv1's bias: ONCE, in front of the groups (inside
    // the loop the compiler re-issued them per group and chunk: 72 loads and as many waits per tile)
    const float* ip = p.in_params + (size_t)g * 3 * C + q * 8;
    float mm[NCH32][8], ss[NCH32][8], cc[NCH32][8];
#pragma unroll
    for (int ch = 0; ch < NCH32; ++ch) {
      const float4 m0 = ld4(ip + ch * 32), m1 = ld4(ip + ch * 32 + 4);
      const float4 s0 = ld4(ip + C + ch * 32), s1 = ld4(ip + C + ch * 32 + 4);
      const float4 c0 = ld4(ip + 2 * C + ch * 32), c1 = ld4(ip + 2 * C + ch * 32 + 4);
      const float m_[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
      const float s_[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
      const float c_[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) { mm[ch][e] = m_[e]; ss[ch][e] = s_[e]; cc[ch][e] = c_[e]; }
    }
    float4 b1v[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) b1v[n] = ld4(p.b1 + n * 16 + 4 * q);
#ifdef ATVS_BT_DEBUG
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BDBG(0)
#endif
#pragma unroll
    for (int i = 0; i < G1W; ++i) {
      if (wave + 4 * i >= NG1) continue;                 // wave-uniform
      f32x4 acc[NT], accx[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[n] = accx[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ch = 0; ch < NCH32; ++ch) {
        // pre-activation (conv1x1_b's normalise-on-load: (v - mean) * scale + beta, ReLU), then the split
        float v[8] = {xa[i][ch].x, xa[i][ch].y, xa[i][ch].z, xa[i][ch].w, xb[i][ch].x, xb[i][ch].y, xb[i][ch].z, xb[i][ch].w};
        f16x8 h0, h1;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = in[i] ? fmaxf(atvs_bn1(v[e], ss[ch][e], atvs_bn_shift(mm[ch][e], ss[ch][e], cc[ch][e])), 0.f) : 0.f;
        bt_split8(v, &h0, &h1);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1[ch][n][0], h0, acc[n], 0, 0, 0);
          accx[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1[ch][n][1], h0, accx[n], 0, 0, 0);
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) accx[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A1[ch][n][0], h1, accx[n], 0, 0, 0);
      }
      // r1 = relu(. + b1), zero outside the image; pieces of channels n * 16 + 4 q .. + 3 of pixel hp
      unsigned char* d = smem + hp[i] * PITCH + q * 8;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const float bv[4] = {b1v[n].x, b1v[n].y, b1v[n].z, b1v[n].w};
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float t = acc[n][k] + accx[n][k] * BT_IRS;
          t += bv[k];
          v[k] = in[i] ? fmaxf(t, 0.f) : 0.f;
        }
        f16x4 p0, p1;
        bt_split(v, &p0, &p1);
        *reinterpret_cast<f16x4*>(d + n * 32) = p0;
        *reinterpret_cast<f16x4*>(d + n * 32 + PIMG) = p1;
      }
    }
  }
  BDBG(1)
  bt_lds_barrier();
  BDBG(2)

  // ------------------------------------------------------------------ P2: conv2 from the r1 image
  f32x4 acc[TY][NTW], accx[TY][NTW];
#pragma unroll
  for (int t = 0; t < TY; ++t)
#pragma unroll
    for (int n = 0; n < NTW; ++n) acc[t][n] = accx[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 b2v[NTW];                   // in front of P2: vector-memory results return in order, a load issued behind P3's prefetches
#pragma unroll                       // would wait for all of them
  for (int n = 0; n < NTW; ++n) b2v[n] = ld4(p.b2 + (wn * NTW + n) * 16 + 4 * q);
  {
    // fragment of step j: tap 2 j + (q >> 1) (the 10th tap has zero weights: the 9th tap's data), channels (q & 1) * 8 .. of the
    // 16-channel chunk; row 0 of this wave's rows
    int bd[JS];
#pragma unroll
    for (int j = 0; j < JS; ++j) {
      const int tap = min(2 * j + (q >> 1), 8);
      const int ky = tap / 3, kx = tap - 3 * ky;
      bd[j] = ((wr * TY + ky * DIL) * HC + r + kx * DIL) * PITCH + (q & 1) * 16;
    }
    // packed weights: [K step = chunk * 5 + j][NT tiles][2 pieces][64 lanes], one zero step of padding at the end
    const f16x8* __restrict__ wl = p.w2 + (size_t)(wn * NTW) * 2 * 64 + lane;
    constexpr int WSTEP = NT * 2 * 64;
    // weights two K steps ahead (three register slots: an L2 round trip is longer than one step's 12 * NTW MFMAs), fragments one
    // step ahead (two slots); the requests of a step are pinned in front of its MFMAs
    constexpr int S2 = NCH16 * JS;
    f16x8 Aw[3][NTW][2];
    auto request_a = [&](int slot, int step) __attribute__((always_inline)) {
#pragma unroll
      for (int n = 0; n < NTW; ++n)
#pragma unroll
        for (int w = 0; w < 2; ++w) Aw[slot][n][w] = wl[(size_t)step * WSTEP + (n * 2 + w) * 64];
    };
    f16x8 Bq[2][2][TY];
    auto request_b = [&](int slot, int step) __attribute__((always_inline)) {
      const int ch = step / JS, j = step % JS;
#pragma unroll
      for (int t = 0; t < TY; ++t) {
        const unsigned char* a = smem + bd[j] + t * (HC * PITCH) + ch * 32;
        Bq[slot][0][t] = *reinterpret_cast<const f16x8*>(a);
        Bq[slot][1][t] = *reinterpret_cast<const f16x8*>(a + PIMG);
      }
    };
    request_a(0, 0);
    request_a(1, 1);
    request_b(0, 0);
#pragma unroll
    for (int s = 0; s < S2; ++s) {
      const int cur = s % 3, bc = s & 1;
      if (s + 2 <= S2) request_a((s + 2) % 3, s + 2);          // step S2 = the pack's zero padding step: never multiplied
      if (s + 1 < S2) request_b(bc ^ 1, s + 1);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < NTW; ++n) {
#pragma unroll
        for (int t = 0; t < TY; ++t) acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aw[cur][n][0], Bq[bc][0][t], acc[t][n], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TY; ++t) accx[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aw[cur][n][1], Bq[bc][0][t], accx[t][n], 0, 0, 0);
      }
#pragma unroll
      for (int n = 0; n < NTW; ++n)
#pragma unroll
        for (int t = 0; t < TY; ++t) accx[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aw[cur][n][0], Bq[bc][1][t], accx[t][n], 0, 0, 0);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  BDBG(3)
  bt_lds_barrier();                             // every wave has read its last r1 fragment

  // ------------------------------------------------------------------ P3: r2 = relu(. + b2) as pieces over the r1 image
  // conv3's weight pieces, the identity shortcut and b3: requested HERE, a whole phase in front of their use (inside P4 every one
  // of them cost an exposed L2 / HBM round trip)
  const int xo = x0 + r;
  float4 rr[TY][NTW], b3v[NTW];
#pragma unroll
  for (int n = 0; n < NTW; ++n) b3v[n] = ld4(p.b3 + (wn * NTW + n) * 16 + 4 * q);
#pragma unroll
  for (int t = 0; t < TY; ++t) {
    const int yo = y0 + wr * TY + t;
    const bool ok = yo < p.H && xo < p.W;
#pragma unroll
    for (int n = 0; n < NTW; ++n)
      rr[t][n] = ok ? ld4(xg + ((size_t)yo * p.W + xo) * C + (wn * NTW + n) * 16 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  f16x8 A3[NCH32][NTW][2];
#pragma unroll
  for (int ch = 0; ch < NCH32; ++ch)
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
      for (int w = 0; w < 2; ++w) A3[ch][n][w] = p.w3[((size_t)(ch * NT + wn * NTW + n) * 2 + w) * 64 + lane];
#pragma unroll
  for (int t = 0; t < TY; ++t) {
    unsigned char* d = smem + ((wr * TY + t) * 16 + r) * PITCH + q * 8;
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
      const float bv[4] = {b2v[n].x, b2v[n].y, b2v[n].z, b2v[n].w};
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float s = acc[t][n][k] + accx[t][n][k] * BT_IRS;
        s += bv[k];
        v[k] = fmaxf(s, 0.f);
      }
      f16x4 p0, p1;
      bt_split(v, &p0, &p1);
      *reinterpret_cast<f16x4*>(d + (wn * NTW + n) * 32) = p0;
      *reinterpret_cast<f16x4*>(d + (wn * NTW + n) * 32 + PIMG) = p1;
    }
  }
  bt_lds_barrier();
  BDBG(4)

  // ------------------------------------------------------------------ P4: conv3 + b3 + x, moments of y
#pragma unroll
  for (int t = 0; t < TY; ++t)
#pragma unroll
    for (int n = 0; n < NTW; ++n) acc[t][n] = accx[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ch = 0; ch < NCH32; ++ch) {
    f16x8 B0[TY], B1[TY];
#pragma unroll
    for (int t = 0; t < TY; ++t) {
      const unsigned char* a = smem + ((wr * TY + t) * 16 + r) * PITCH + ch * 64 + q * 16;
      B0[t] = *reinterpret_cast<const f16x8*>(a);
      B1[t] = *reinterpret_cast<const f16x8*>(a + PIMG);
    }
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
#pragma unroll
      for (int t = 0; t < TY; ++t) acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A3[ch][n][0], B0[t], acc[t][n], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < TY; ++t) accx[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A3[ch][n][1], B0[t], accx[t][n], 0, 0, 0);
    }
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
      for (int t = 0; t < TY; ++t) accx[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A3[ch][n][0], B1[t], accx[t][n], 0, 0, 0);
  }

  BDBG(5)
  float ssum[NTW][4], ssq[NTW][4];
#pragma unroll
  for (int n = 0; n < NTW; ++n)
#pragma unroll
    for (int k = 0; k < 4; ++k) ssum[n][k] = ssq[n][k] = 0.f;
  float* yg = p.y + (size_t)g * p.H * p.W * C;
#pragma unroll
  for (int t = 0; t < TY; ++t) {
    const int yo = y0 + wr * TY + t;
    if (yo >= p.H || xo >= p.W) continue;
    const size_t rowb = ((size_t)yo * p.W + xo) * C;
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
      const int co = (wn * NTW + n) * 16 + 4 * q;
      float4 v = make_float4(acc[t][n][0] + accx[t][n][0] * BT_IRS, acc[t][n][1] + accx[t][n][1] * BT_IRS,
                             acc[t][n][2] + accx[t][n][2] * BT_IRS, acc[t][n][3] + accx[t][n][3] * BT_IRS);
      const float4 bb = b3v[n];
      v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
      v.x += rr[t][n].x; v.y += rr[t][n].y; v.z += rr[t][n].z; v.w += rr[t][n].w;       // the identity shortcut
      st4(yg + rowb + co, v);
      ssum[n][0] += v.x; ssum[n][1] += v.y; ssum[n][2] += v.z; ssum[n][3] += v.w;
      ssq[n][0] += v.x * v.x; ssq[n][1] += v.y * v.y; ssq[n][2] += v.z * v.z; ssq[n][3] += v.w * v.w;
    }
  }
  if (p.stats) {
    // row (image, tile): [2][C] doubles; a channel belongs to the WR waves of one column of the wave grid
    double* row = p.stats + (size_t)lin * 2 * C;
    double* s_red = reinterpret_cast<double*>(smem);           // [wr][2][C], the r2 image is dead behind the barrier
    if (WR > 1) __syncthreads();
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double a = bt_row_sum((double)ssum[n][k]), bq = bt_row_sum((double)ssq[n][k]);
        if (r == 0) {
          const int c = (wn * NTW + n) * 16 + 4 * q + k;
          if (WR == 1) {
            row[c] = a;
            row[C + c] = bq;
          } else {
            s_red[(wr * 2 + 0) * C + c] = a;
            s_red[(wr * 2 + 1) * C + c] = bq;
          }
        }
      }
    if (WR > 1) {
      __syncthreads();
      for (int i = tid; i < 2 * C; i += 256) {
        double v = 0.0;
#pragma unroll
        for (int a = 0; a < WR; ++a) v += s_red[a * 2 * C + i];
        row[i] = v;
      }
    }
  }
#ifdef ATVS_BT_DEBUG
  BDBG(6)
  if (lane == 0 && blockIdx.x < 1024) {
    dbg_acc[7] = 1;
    for (int i = 0; i < 8; ++i) atvs_dbg_bt[(blockIdx.x * 4 + wave) * 8 + i] = dbg_acc[i];
  }
#endif
}

template <int C, int DIL, int TH, int WN, int WR>
int launch_bt(BtArgs a, hipStream_t s) {
  constexpr int NPH = (TH + 2 * DIL) * (16 + 2 * DIL), NG1 = (NPH + 15) / 16;
  constexpr size_t lds = (size_t)2 * NG1 * 16 * (C * 2 + 16);
  static_assert(lds >= (size_t)WR * 2 * C * 8, "statistics rows fit the image");
  a.tiles_x = (a.W + 15) / 16;
  a.tiles = ((a.H + TH - 1) / TH) * a.tiles_x;
  a.total = (long)a.G * a.tiles;
  const long blocks = ((a.total + 7) / 8) * 8;
  if (blocks > 0x7fffffffL) return ATVS_ERR_SHAPE;
  hipLaunchKernelGGL((bottleneck_b_kernel<C, DIL, TH, WN, WR>), dim3((unsigned)blocks), dim3(256), lds, s, a);
  return ATVS_OK;
}

constexpr int bt_th(int C) { return C == 32 ? 16 : 8; }      // rows of a tile: 16 x 16 pixels at 32 channels, 8 x 16 at 64

}  // namespace

#ifdef ATVS_BT_DEBUG
extern "C" int atvs_debug_read_bt(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(atvs_dbg_bt), sizeof(atvs_dbg_bt));
}
#endif

// Shapes the fused unit is built for: depth 32 / 64 (conv0_x, conv1_x of ResNetDS2SPP), dilation 1.  The 128-channel units
// (dilation 2 / 4) keep their three launches: their r1 tile + halo does not fit LDS next to a useful tile.
extern "C" int atvs_bottleneck_b_supported(int C, int dilation) { return ((C == 32 || C == 64) && dilation == 1) ? 1 : 0; }

// workgroups per image = rows per image of stats_partial ([2][C] doubles each): tiles of 8 x 16 (C = 64) / 16 x 16 (C = 32) pixels
extern "C" long atvs_bottleneck_b_rows(int C, int H, int W) { return (long)((H + bt_th(C) - 1) / bt_th(C)) * ((W + 15) / 16); }

// y = x + conv3(relu(conv2(relu(conv1(relu(bn(x))) + b1)) + b2)) + b3 for G images (H, W, C), channel-last fp32.
// in_params (G,3,C): the pre-activation batch norm (atvs_bn_finalize rows mean | scale | beta).  w1 / w3: atvs_conv1x1_b_pack of
// the 1x1 kernels [C][C]; w2: atvs_conv2d_b_pack of the 3x3 kernel [3][3][C][C].  stats_partial: null or (G, atvs_bottleneck_b_rows,
// 2, C) doubles, the per-workgroup moments of y.  y may not alias x (halo pixels of x are read by other workgroups).
extern "C" int atvs_bottleneck_b_f32(const float* x, const float* in_params, const unsigned char* w1, const float* b1,
                                     const unsigned char* w2, const float* b2, const unsigned char* w3, const float* b3, float* y,
                                     double* stats_partial, int G, int H, int W, int C, int dilation, atvs_stream_t stream) {
  if (!x || !in_params || !w1 || !w2 || !w3 || !b1 || !b2 || !b3 || !y) return ATVS_ERR_NULL;
  if (x == y) return ATVS_ERR_ARG;
  if (G <= 0 || H <= 0 || W <= 0 || !atvs_bottleneck_b_supported(C, dilation)) return ATVS_ERR_SHAPE;
  if ((double)H * W * C >= 2147483648.0) return ATVS_ERR_SHAPE;
  BtArgs a;
  a.x = x; a.in_params = in_params;
  a.w1 = reinterpret_cast<const f16x8*>(w1); a.w2 = reinterpret_cast<const f16x8*>(w2); a.w3 = reinterpret_cast<const f16x8*>(w3);
  a.b1 = b1; a.b2 = b2; a.b3 = b3; a.y = y; a.stats = stats_partial;
  a.G = G; a.H = H; a.W = W;
  a.tiles_x = a.tiles = 0; a.total = 0;
  hipStream_t s = as_stream(stream);
  int rc;
  if (C == 64) rc = launch_bt<64, 1, bt_th(64), 2, 2>(a, s);
  else rc = launch_bt<32, 1, bt_th(32), 1, 4>(a, s);
  if (rc) return rc;
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}
