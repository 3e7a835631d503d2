// 1x1 convolution of feature maps = a tall GEMM  Y[p, co] = sum_ci X[p, ci] W[ci, co]  on the fp32 matrix cores (gfx950).
//
// The bottlenecks of the 2-D feature towers (/root/reference/cnn_wrapper/network.py:552-602: slim.conv2d 1x1 `conv1`,
// `conv3` (+ shortcut add), `shortcut`; cnn_wrapper/atvsnet.py:254-292) and fusion1: Cin, Cout in {32, 64, 128}, 20 480 or
// 81 920 pixels per image.  2 Cin Cout FLOP per 4 (Cin + Cout) bytes: at 128 -> 128 the layer is as much HBM- as
// MFMA-bound, and on the gather kernel (one pixel tile per wave, weights re-fetched from L2 by every wave) it was
// latency-bound (37 TF/s, 1.2 TB/s).  Here:
//   * the packed weights (<= 64 KB) are staged ONCE per workgroup in LDS (A operand, conflict-free ds_read_b128);
//   * a wave owns 64 consecutive pixels (TM = 4 tiles) and ALL output channels (NT tiles): the pixels stream from global
//     memory exactly once (B operand: lane = (pixel, 4-channel group), 16-byte loads), prefetched one K step ahead;
//   * prologue: the slim.batch_norm + ReLU pre-activation of the bottleneck (network.py:570-571) applied to the pixels as
//     they are loaded (normalise-on-load, per-image parameters) -- the pre-activated tensor is never written;
//   * epilogue: bias, residual (the shortcut), ReLU, channel-last stores, per-(image, workgroup) batch-norm moments.
#include "conv_common.h"

namespace {

struct C1Args {
  const float* x;
  const float* wp;          // packed [K step][NT][64 lanes] float4
  const float* bias;
  const float* res;
  const float* in_params;   // (G, 3, Cin) or nullptr
  float* y;
  double* stats;
  int Cin, Cout, ldy, ycoff;
  int relu, in_relu;
  long rows;                // pixels per image
  int wgs;                  // workgroups per image
};

template <int NT>
__global__ __launch_bounds__(256, 2) void conv1x1_kernel(C1Args p) {
  extern __shared__ __attribute__((aligned(16))) float4 s_w[];      // [J][NT][64]
  constexpr int TM = 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int J = p.Cin / 16;
  const float4* __restrict__ wg = reinterpret_cast<const float4*>(p.wp);
  for (int i = tid; i < J * NT * 64; i += 256) s_w[i] = wg[i];
  const int grp = blockIdx.x / p.wgs, wb = blockIdx.x - grp * p.wgs;
  const long pix0 = ((long)wb * 4 + wave) * (TM * 16);               // first pixel of this wave inside the image
  const float* __restrict__ xg = p.x + (size_t)grp * p.rows * p.Cin;
  const float* __restrict__ ip = p.in_params ? p.in_params + (size_t)grp * 3 * p.Cin : nullptr;

  auto load = [&](int j, float4* b) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const long px = pix0 + t * 16 + r;
      b[t] = (px < p.rows) ? ld4(xg + (size_t)px * p.Cin + j * 16 + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto norm = [&](int j, float4* b) __attribute__((always_inline)) {
    if (!ip) return;
    const int c = j * 16 + q * 4;
    const float4 m = ld4(ip + c), s = ld4(ip + p.Cin + c), be = ld4(ip + 2 * p.Cin + c);
    const float4 sh = atvs_bn_shift4(m, s, be);
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      float4 v = b[t];
      v = atvs_bn4(v, s, sh);
      if (p.in_relu) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
      b[t] = v;
    }
  };

  f32x4 acc[TM][NT];
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 b_cur[TM], b_nxt[TM];
#pragma unroll
  for (int t = 0; t < TM; ++t) b_nxt[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  load(0, b_cur);
  __syncthreads();                                                   // weights staged
  for (int j = 0; j < J; ++j) {
    if (j + 1 < J) load(j + 1, b_nxt);
    norm(j, b_cur);
    float4 a[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) a[n] = s_w[(j * NT + n) * 64 + lane];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int t = 0; t < TM; ++t)
          acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(a[n], s), f4get(b_cur[t], s), acc[t][n], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < TM; ++t) b_cur[t] = b_nxt[t];
  }

  // ---- epilogue: lane holds channels n*16 + 4q .. +3 of pixel pix0 + t*16 + r
  float ssum[NT][4], ssq[NT][4];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int k = 0; k < 4; ++k) ssum[n][k] = ssq[n][k] = 0.f;
  float* yg = p.y + (size_t)grp * p.rows * p.ldy;
  const float* rg = p.res ? p.res + (size_t)grp * p.rows * p.ldy : nullptr;
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const long px = pix0 + t * 16 + r;
    if (px >= p.rows) continue;
    const size_t rowb = (size_t)px * p.ldy + p.ycoff;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int co = n * 16 + 4 * q;
      float4 v = make_float4(acc[t][n][0], acc[t][n][1], acc[t][n][2], acc[t][n][3]);
      if (p.bias) {
        const float4 bb = ld4(p.bias + co);
        v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
      }
      if (rg) {
        const float4 rr = ld4(rg + rowb + co);
        v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
      }
      if (p.relu) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
      st4(yg + rowb + co, v);
      ssum[n][0] += v.x; ssum[n][1] += v.y; ssum[n][2] += v.z; ssum[n][3] += v.w;
      ssq[n][0] += v.x * v.x; ssq[n][1] += v.y * v.y; ssq[n][2] += v.z * v.z; ssq[n][3] += v.w * v.w;
    }
  }
  if (p.stats) {
    __syncthreads();                                                 // the weights are dead: reuse their LDS
    double* s_red = reinterpret_cast<double*>(s_w);                  // [4 waves][2][NT*16]
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        double a = (double)ssum[n][k], bq = (double)ssq[n][k];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          a += __shfl_xor(a, o);
          bq += __shfl_xor(bq, o);
        }
        if (r == 0) {
          s_red[(wave * 2 + 0) * (NT * 16) + n * 16 + 4 * q + k] = a;
          s_red[(wave * 2 + 1) * (NT * 16) + n * 16 + 4 * q + k] = bq;
        }
      }
    __syncthreads();
    for (int i = tid; i < 2 * NT * 16; i += 256) {
      const int which = i / (NT * 16), c = i % (NT * 16);
      p.stats[((size_t)blockIdx.x * 2 + which) * (NT * 16) + c] =
          (s_red[(0 * 2 + which) * (NT * 16) + c] + s_red[(1 * 2 + which) * (NT * 16) + c]) +
          (s_red[(2 * 2 + which) * (NT * 16) + c] + s_red[(3 * 2 + which) * (NT * 16) + c]);
    }
  }
}

template <int NT>
int launch_c1(const C1Args& a, int groups, hipStream_t s) {
  size_t lds = (size_t)(a.Cin / 16) * NT * 64 * sizeof(float4);
  const size_t red = (size_t)4 * 2 * NT * 16 * sizeof(double);
  if (lds < red) lds = red;
  static AtvsAttrOnce lds_once;                   // per kernel instantiation (this function is a template / has one kernel)
  if (const int rc_ = atvs_set_max_lds_once(lds_once, reinterpret_cast<const void*>(conv1x1_kernel<NT>), 80 * 1024)) return rc_;
  const long blocks = (long)a.wgs * groups;
  if (blocks > 0x7fffffffL) return ATVS_ERR_SHAPE;
  hipLaunchKernelGGL((conv1x1_kernel<NT>), dim3((unsigned)blocks), dim3(256), lds, s, a);
  return ATVS_OK;
}

}  // namespace

extern "C" int atvs_conv1x1_supported(int Cin, int Cout) {
  return (Cin > 0 && Cin % 16 == 0 && Cin <= 128 && (Cout == 32 || Cout == 64 || Cout == 128)) ? 1 : 0;
}

// workgroups per image = rows per image of stats_partial ([2][Cout] doubles each): 256 pixels per workgroup
extern "C" long atvs_conv1x1_rows(long pixels) { return (pixels + 255) / 256; }

extern "C" int atvs_conv1x1_pack_size(int Cin, int Cout, long* packed_floats) {
  if (!packed_floats) return ATVS_ERR_NULL;
  if (!atvs_conv1x1_supported(Cin, Cout)) return ATVS_ERR_SHAPE;
  *packed_floats = (long)(Cin / 16) * (Cout / 16) * 64 * 4;
  return ATVS_OK;
}

// HOST function.  w: TF kernel [1][1][Cin][Cout] (= [Cin][Cout]).  packed[K step j][tile n][lane = q*16 + co16][s] =
// w[16 j + 4 q + s][16 n + co16].
extern "C" int atvs_conv1x1_pack(const float* w, int Cin, int Cout, float* packed) {
  if (!w || !packed) return ATVS_ERR_NULL;
  long pf;
  int rc = atvs_conv1x1_pack_size(Cin, Cout, &pf);
  if (rc) return rc;
  const int NT = Cout / 16;
  for (int j = 0; j < Cin / 16; ++j)
    for (int n = 0; n < NT; ++n)
      for (int q = 0; q < 4; ++q)
        for (int co16 = 0; co16 < 16; ++co16)
          for (int s = 0; s < 4; ++s)
            packed[((((size_t)j * NT + n) * 64) + q * 16 + co16) * 4 + s] = w[(size_t)(16 * j + 4 * q + s) * Cout + 16 * n + co16];
  return ATVS_OK;
}

// y (G, pixels, ldy)[..., y_coff + co] = x (G, pixels, Cin) W (+ bias, + residual, ReLU); in_params (G,3,Cin) != NULL: the
// batch norm (+ ReLU if in_relu) of x applied on load.  residual: same addressing as y (y_coff must be 0).
// stats_partial: G * atvs_conv1x1_rows(pixels) rows of [2][Cout] doubles or NULL.
extern "C" int atvs_conv1x1_f32(const float* x, const float* packed_w, const float* bias, const float* residual,
                                const float* in_params, int in_relu, float* y, double* stats_partial, int groups,
                                long pixels, int Cin, int Cout, int ldy, int y_coff, int relu, atvs_stream_t stream) {
  if (!x || !packed_w || !y) return ATVS_ERR_NULL;
  if (groups <= 0 || pixels <= 0 || !atvs_conv1x1_supported(Cin, Cout)) return ATVS_ERR_SHAPE;
  if (y_coff < 0 || y_coff + Cout > ldy || (ldy % 4) || (y_coff % 4)) return ATVS_ERR_SHAPE;
  if (residual && y_coff != 0) return ATVS_ERR_ARG;
  C1Args a;
  a.x = x; a.wp = packed_w; a.bias = bias; a.res = residual; a.in_params = in_params; a.y = y; a.stats = stats_partial;
  a.Cin = Cin; a.Cout = Cout; a.ldy = ldy; a.ycoff = y_coff; a.relu = relu; a.in_relu = in_relu;
  a.rows = pixels; a.wgs = (int)atvs_conv1x1_rows(pixels);
  hipStream_t s = as_stream(stream);
  int rc;
  if (Cout == 128) rc = launch_c1<8>(a, groups, s);
  else if (Cout == 64) rc = launch_c1<4>(a, groups, s);
  else rc = launch_c1<2>(a, groups, s);
  if (rc) return rc;
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}
