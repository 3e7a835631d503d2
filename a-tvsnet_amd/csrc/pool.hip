// Spatial-pyramid-pooling helpers of the 2-D feature tower (gfx950): SAME average
// pooling, align_corners bilinear resize, channel-slice copy (tf.concat).
//
// Reference: tf.layers.average_pooling2d at /root/reference/cnn_wrapper/network.py:665-671
// (SAME: mean over the VALID window elements only), tf.image.resize_images(BILINEAR,
// align_corners=True) at :649-655, tf.concat at :691-693; used by ResNetDS2SPP,
// /root/reference/cnn_wrapper/atvsnet.py:269-290.  All trivially small next to the volumes.
#include "common.h"

// Stage 1: one workgroup per (output pixel, slice of the window rows): partial sums -> ws
// (Ho, Wo, POOL_SLICES, C).  Stage 2: fixed-order sum of the slices / valid count.  Deterministic.
#define POOL_SLICES 16
__global__ __launch_bounds__(256) void avg_pool_partial_kernel(const float* __restrict__ x, float* __restrict__ ws, int H,
                                                               int W, int C, int Wo, int k, int s, int pad_t, int pad_l) {
  __shared__ float sm[256];
  const int oy = blockIdx.y, ox = blockIdx.x, sl = blockIdx.z % POOL_SLICES, grp = blockIdx.z / POOL_SLICES;
  x += (size_t)grp * H * W * C;                               // independent image
  ws += (size_t)grp * gridDim.y * Wo * POOL_SLICES * C;
  const int y0 = max(oy * s - pad_t, 0), y1 = min(oy * s - pad_t + k, H);
  const int x0 = max(ox * s - pad_l, 0), x1 = min(ox * s - pad_l + k, W);
  const int rows = y1 - y0;
  const int ra = y0 + (rows * sl) / POOL_SLICES, rb = y0 + (rows * (sl + 1)) / POOL_SLICES;
  const int ww = x1 - x0, n = (rb - ra) * ww;
  for (int cb = 0; cb < C; cb += 256) {
    const int cw = min(C - cb, 256);
    const int lanes = 256 / cw;
    const int t = threadIdx.x;
    float acc = 0.f;
    if (t < lanes * cw) {
      const int c = cb + t % cw;
      for (int i = t / cw; i < n; i += lanes) {
        int yy = ra + i / ww, xx = x0 + i % ww;
        acc += x[((size_t)yy * W + xx) * C + c];
      }
    }
    sm[t] = acc;
    __syncthreads();
    if (t < cw) {
      float v = 0.f;
      for (int j = 0; j < lanes; ++j) v += sm[j * cw + t];
      ws[(((size_t)oy * Wo + ox) * POOL_SLICES + sl) * C + cb + t] = v;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void avg_pool_finish_kernel(const float* __restrict__ ws, float* __restrict__ y, int H, int W,
                                                              int C, int Ho, int Wo, int k, int s, int pad_t, int pad_l,
                                                              int groups) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)groups * Ho * Wo * C) return;
  int c = (int)(i % C);
  long pix = i / C;                                           // (image, oy, ox)
  int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho);
  const int y0 = max(oy * s - pad_t, 0), y1 = min(oy * s - pad_t + k, H);
  const int x0 = max(ox * s - pad_l, 0), x1 = min(ox * s - pad_l + k, W);
  float v = 0.f;
  for (int sl = 0; sl < POOL_SLICES; ++sl) v += ws[((size_t)pix * POOL_SLICES + sl) * C + c];
  y[i] = v / (float)((y1 - y0) * (x1 - x0));
}

extern "C" long atvs_avg_pool_ws_floats(int H, int W, int C, int stride) {      // per image
  long Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
  return Ho * Wo * POOL_SLICES * C;
}

extern "C" int atvs_avg_pool_same(const float* x, float* y, float* ws, int groups, int H, int W, int C, int pool, int stride,
                                  atvs_stream_t stream) {
  if (!x || !y || !ws) return ATVS_ERR_NULL;
  if (groups <= 0 || groups * POOL_SLICES > 65535 || H <= 0 || W <= 0 || C <= 0 || pool <= 0 || stride <= 0) return ATVS_ERR_SHAPE;
  int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
  int ph = max((Ho - 1) * stride + pool - H, 0), pw = max((Wo - 1) * stride + pool - W, 0);
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(avg_pool_partial_kernel, dim3(Wo, Ho, POOL_SLICES * groups), dim3(256), 0, s, x, ws, H, W, C, Wo, pool,
                     stride, ph / 2, pw / 2);
  hipLaunchKernelGGL(avg_pool_finish_kernel, dim3(cdiv((long)groups * Ho * Wo * C, 256)), dim3(256), 0, s, ws, y, H, W, C, Ho, Wo,
                     pool, stride, ph / 2, pw / 2, groups);
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
