// Spatial-pyramid-pooling helpers of the 2-D feature tower (gfx950): SAME average
// pooling, align_corners bilinear resize, channel-slice copy (tf.concat).
//
// Reference: tf.layers.average_pooling2d at /root/reference/cnn_wrapper/network.py:665-671
// (SAME: mean over the VALID window elements only), tf.image.resize_images(BILINEAR,
// align_corners=True) at :649-655, tf.concat at :691-693; used by ResNetDS2SPP,
// /root/reference/cnn_wrapper/atvsnet.py:269-290.  All trivially small next to the volumes.
#include "common.h"

// Stage 1: one workgroup per (output pixel, slice of the window's PIXELS): partial sums -> ws (Ho, Wo, SL, C).  Stage 2: fixed-
// order sum of the slices / valid count.  Deterministic.  SL (pool_slices) is chosen so that an image has ~640 workgroups whatever the
// window: the pyramid's 64 x 64 windows of a 128 x 160 map (2 x 3 outputs) get 64 slices of 64 pixels, the 8 x 8 windows (16 x 20
// outputs) two.  C % 4 == 0: a thread owns four channels of every 256 / (C / 4)-th pixel of its slice (16-byte loads, whole rows
// of the channel-last map per wavefront); its float4 partial sums meet in LDS.  (Round 5's form -- 16 row slices, 4-byte loads -- took
// 28 us per pool at configs[2]; the grouping of the partial sums, hence the last bit of the means, differs from it.)
static inline int pool_slices(int Ho, int Wo) {
  long s = 640 / ((long)Ho * Wo);
  return (int)(s < 1 ? 1 : s > 64 ? 64 : s);
}

__global__ __launch_bounds__(256) void avg_pool_partial_kernel(const float* __restrict__ x, float* __restrict__ ws, int H,
                                                               int W, int C, int Wo, int k, int s, int pad_t, int pad_l, int SL) {
  __shared__ float4 sm[256];
  const int oy = blockIdx.y, ox = blockIdx.x, sl = blockIdx.z % SL, grp = blockIdx.z / SL;
  x += (size_t)grp * H * W * C;                               // independent image
  ws += (size_t)grp * gridDim.y * Wo * SL * C;
  const int y0 = max(oy * s - pad_t, 0), y1 = min(oy * s - pad_t + k, H);
  const int x0 = max(ox * s - pad_l, 0), x1 = min(ox * s - pad_l + k, W);
  const int ww = x1 - x0, n = (y1 - y0) * ww;
  const int i0 = (int)(((long)n * sl) / SL), i1 = (int)(((long)n * (sl + 1)) / SL);
  const int t = threadIdx.x;
  if ((C & 3) == 0) {
    const int c4n = C >> 2;
    for (int cb = 0; cb < c4n; cb += 256) {
      const int cw = min(c4n - cb, 256);                      // channel groups of this block
      const int lanes = 256 / cw;                             // pixel lanes
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t < lanes * cw) {
        const int cg = cb + t % cw;
        int i = i0 + t / cw;
        int yy = y0 + i / ww, xx = x0 + i % ww;
        for (; i < i1; i += lanes) {
          const float4 v = ld4(x + ((size_t)yy * W + xx) * C + cg * 4);
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
          xx += lanes;
          while (xx >= x1) { xx -= ww; ++yy; }
        }
      }
      sm[t] = acc;
      __syncthreads();
      if (t < cw) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < lanes; ++j) {
          const float4 u = sm[j * cw + t];
          v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        st4(ws + (((size_t)oy * Wo + ox) * SL + sl) * C + (cb + t) * 4, v);
      }
      __syncthreads();
    }
    return;
  }
  float* smf = reinterpret_cast<float*>(sm);
  for (int cb = 0; cb < C; cb += 256) {
    const int cw = min(C - cb, 256);
    const int lanes = 256 / cw;
    float acc = 0.f;
    if (t < lanes * cw) {
      const int c = cb + t % cw;
      for (int i = i0 + t / cw; i < i1; i += lanes) {
        int yy = y0 + i / ww, xx = x0 + i % ww;
        acc += x[((size_t)yy * W + xx) * C + c];
      }
    }
    smf[t] = acc;
    __syncthreads();
    if (t < cw) {
      float v = 0.f;
      for (int j = 0; j < lanes; ++j) v += smf[j * cw + t];
      ws[(((size_t)oy * Wo + ox) * SL + sl) * C + cb + t] = v;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void avg_pool_finish_kernel(const float* __restrict__ ws, float* __restrict__ y, int H, int W,
                                                              int C, int Ho, int Wo, int k, int s, int pad_t, int pad_l,
                                                              int groups, int SL) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)groups * Ho * Wo * C) return;
  int c = (int)(i % C);
  long pix = i / C;                                           // (image, oy, ox)
  int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho);
  const int y0 = max(oy * s - pad_t, 0), y1 = min(oy * s - pad_t + k, H);
  const int x0 = max(ox * s - pad_l, 0), x1 = min(ox * s - pad_l + k, W);
  float v = 0.f;
  for (int sl = 0; sl < SL; ++sl) v += ws[((size_t)pix * SL + sl) * C + c];
  y[i] = v / (float)((y1 - y0) * (x1 - x0));
}

extern "C" long atvs_avg_pool_ws_floats(int H, int W, int C, int stride) {      // per image
  long Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
  return Ho * Wo * pool_slices((int)Ho, (int)Wo) * C;
}

extern "C" int atvs_avg_pool_same(const float* x, float* y, float* ws, int groups, int H, int W, int C, int pool, int stride,
                                  atvs_stream_t stream) {
  if (!x || !y || !ws) return ATVS_ERR_NULL;
  if (groups <= 0 || H <= 0 || W <= 0 || C <= 0 || pool <= 0 || stride <= 0) return ATVS_ERR_SHAPE;
  int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
  const int SL = pool_slices(Ho, Wo);
  if ((long)groups * SL > 65535 || Ho > 65535) return ATVS_ERR_SHAPE;
  if ((C & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(ws)) & 15)) return ATVS_ERR_ARG;
  int ph = max((Ho - 1) * stride + pool - H, 0), pw = max((Wo - 1) * stride + pool - W, 0);
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(avg_pool_partial_kernel, dim3(Wo, Ho, SL * groups), dim3(256), 0, s, x, ws, H, W, C, Wo, pool,
                     stride, ph / 2, pw / 2, SL);
  hipLaunchKernelGGL(avg_pool_finish_kernel, dim3(cdiv((long)groups * Ho * Wo * C, 256)), dim3(256), 0, s, ws, y, H, W, C, Ho, Wo,
                     pool, stride, ph / 2, pw / 2, groups, SL);
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}

// out written into a channel slice [c_off, c_off+C) of rows of width ld
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ x, float* __restrict__ y, int H,
                                                              int W, int C, int Ho, int Wo, float sy, float sx, int ld,
                                                              int c_off, int groups) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long n = (long)groups * Ho * Wo * C;
  if (i >= n) return;
  int c = (int)(i % C);
  long pix = i / C;                                           // (image, oy, ox)
  int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho);
  x += (size_t)(pix / ((long)Ho * Wo)) * H * W * C;
  float fy = (float)oy * sy, fx = (float)ox * sx;
  int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
  int y1 = min((int)ceilf(fy), H - 1), x1 = min((int)ceilf(fx), W - 1);
  float ly = fy - (float)y0, lx = fx - (float)x0;
  float tl = x[((size_t)y0 * W + x0) * C + c], tr = x[((size_t)y0 * W + x1) * C + c];
  float bl = x[((size_t)y1 * W + x0) * C + c], br = x[((size_t)y1 * W + x1) * C + c];
  float t = tl + (tr - tl) * lx;
  float b = bl + (br - bl) * lx;
  y[(size_t)pix * ld + c_off + c] = t + (b - t) * ly;
}

extern "C" int atvs_resize_bilinear(const float* x, float* y, int groups, int H, int W, int C, int Ho, int Wo, int ld_out,
                                    int c_off, atvs_stream_t stream) {
  if (!x || !y) return ATVS_ERR_NULL;
  if (groups <= 0 || H <= 0 || W <= 0 || C <= 0 || Ho <= 0 || Wo <= 0 || c_off < 0 || c_off + C > ld_out) return ATVS_ERR_SHAPE;
  float sy = (Ho > 1) ? (float)((double)(H - 1) / (double)(Ho - 1)) : 0.f;
  float sx = (Wo > 1) ? (float)((double)(W - 1) / (double)(Wo - 1)) : 0.f;
  long n = (long)groups * Ho * Wo * C;
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), x, y, H, W, C, Ho, Wo,
                     sy, sx, ld_out, c_off, groups);
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}

// dst[r, dst_off + c] = src[r, src_off + c], c < C  (tf.concat / un-stacking a trailing axis)
__global__ __launch_bounds__(256) void copy_channels_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                            long rows, int C, int ld_src, int src_off, int ld_dst,
                                                            int dst_off) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * C) return;
  long r = i / C;
  int c = (int)(i % C);
  dst[(size_t)r * ld_dst + dst_off + c] = src[(size_t)r * ld_src + src_off + c];
}

extern "C" int atvs_copy_channels(const float* src, float* dst, long rows, int C, int ld_src, int src_off, int ld_dst,
                                  int dst_off, atvs_stream_t stream) {
  if (!src || !dst) return ATVS_ERR_NULL;
  if (rows <= 0 || C <= 0 || src_off < 0 || dst_off < 0 || src_off + C > ld_src || dst_off + C > ld_dst)
    return ATVS_ERR_SHAPE;
  long n = rows * C;
  hipLaunchKernelGGL(copy_channels_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), src, dst, rows, C,
                     ld_src, src_off, ld_dst, dst_off);
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}

// dst[i] = *srcs[i] (i < n), each `elems` floats: torch.stack along a new leading axis for up to 16 tensors in ONE launch (the
// per-pair reference features of a batch of cost volumes, model.py:186; the two initial depth maps of the refinement, :289)
struct StackPtrs { const float* p[16]; };
__global__ __launch_bounds__(256) void stack_kernel(StackPtrs s, float* __restrict__ dst, long elems4) {
  const float4* __restrict__ src = reinterpret_cast<const float4*>(s.p[blockIdx.y]);
  float4* __restrict__ d = reinterpret_cast<float4*>(dst) + (size_t)blockIdx.y * elems4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < elems4; i += (long)gridDim.x * blockDim.x) d[i] = src[i];
}

extern "C" int atvs_stack(const float* const* srcs, int n, long elems, float* dst, atvs_stream_t stream) {
  if (!srcs || !dst) return ATVS_ERR_NULL;
  if (n <= 0 || n > 16 || elems <= 0 || (elems % 4)) return ATVS_ERR_SHAPE;
  StackPtrs s;
  for (int i = 0; i < 16; ++i) {
    s.p[i] = srcs[i < n ? i : 0];
    if (!s.p[i] || (reinterpret_cast<uintptr_t>(s.p[i]) & 15)) return ATVS_ERR_ARG;
  }
  if (reinterpret_cast<uintptr_t>(dst) & 15) return ATVS_ERR_ARG;
  long blocks = (elems / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(stack_kernel, dim3((unsigned)blocks, (unsigned)n), dim3(256), 0, as_stream(stream), s, dst, elems / 4);
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}
