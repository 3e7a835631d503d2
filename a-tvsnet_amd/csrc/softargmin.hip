// Soft-argmin depth regression (gfx950).
//
// Reference: /root/reference/atvsnet/model.py:80-109 (prob2depth: softmax(-cost)
// over D, expectation of tf.linspace(start, end, D)) and :68-76,113-129
// (prob2depth_upsample: bilinear x4, align_corners, of the PRE-softmax cost per
// depth plane, then the same soft-argmin at full resolution; quirk C10).
//
// HBM-bound: the (D, h, w) cost volume is read once.  Lanes run along w (the
// contiguous axis) so every plane read is coalesced; the depth axis is split
// over the four wavefronts of a workgroup and the online-softmax partials
// (max, sum, weighted sum) are merged through LDS.  The x4 variant never
// materialises the (D, 4h, 4w) volume the reference builds.
#include "common.h"

struct Osm {  // online softmax state over x = -cost
  float m, s, t;
};

__device__ __forceinline__ void osm_push(Osm& a, float x, float val) {
  if (x > a.m) {
    float r = expf(a.m - x);
    a.s = a.s * r + 1.f;
    a.t = a.t * r + val;
    a.m = x;
  } else {
    float e = expf(x - a.m);
    a.s += e;
    a.t += e * val;
  }
}

__device__ __forceinline__ float linspace_step(float start, float interval, int D, float* end_out) {
  float end = start + ((float)D - 1.0f) * interval;
  *end_out = end;
  return (D > 1) ? (end - start) / (float)(D - 1) : 0.f;
}

__global__ __launch_bounds__(256) void softargmin_kernel(const float* __restrict__ cost, const float* __restrict__ depth_start,
                                                         const float* __restrict__ depth_interval, float* __restrict__ depth_out,
                                                         int D, long npix) {
  __shared__ float sm[3][4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  cost += (size_t)blockIdx.y * D * npix;                       // blockIdx.y = independent volume (same depth sweep)
  depth_out += (size_t)blockIdx.y * npix;
  long pix = (long)blockIdx.x * 64 + lane;
  float start = depth_start[0], end;
  float step = linspace_step(start, depth_interval[0], D, &end);
  Osm a = {-INFINITY, 0.f, 0.f};
  int d0 = (int)(((long)D * wv) / 4), d1 = (int)(((long)D * (wv + 1)) / 4);
  if (pix < npix) {
    const float* p = cost + pix;
    for (int d = d0; d < d1; ++d) {
      float c = p[(size_t)d * npix];
      osm_push(a, -1.0f * c, start + step * (float)d);
    }
  }
  sm[0][wv][lane] = a.m;
  sm[1][wv][lane] = a.s;
  sm[2][wv][lane] = a.t;
  __syncthreads();
  if (wv == 0 && pix < npix) {
    float m = fmaxf(fmaxf(sm[0][0][lane], sm[0][1][lane]), fmaxf(sm[0][2][lane], sm[0][3][lane]));
    float s = 0.f, t = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float mk = sm[0][k][lane];
      float r = (mk == -INFINITY) ? 0.f : expf(mk - m);
      s += sm[1][k][lane] * r;
      t += sm[2][k][lane] * r;
    }
    depth_out[pix] = t / s;
  }
}

extern "C" int atvs_softargmin(const float* cost, const float* depth_start, const float* depth_interval, float* depth_out,
                               int groups, int D, int h, int w, atvs_stream_t stream) {
  if (!cost || !depth_start || !depth_interval || !depth_out) return ATVS_ERR_NULL;
  if (groups <= 0 || groups > 65535 || D <= 0 || h <= 0 || w <= 0) return ATVS_ERR_SHAPE;
  long npix = (long)h * w;
  hipLaunchKernelGGL(softargmin_kernel, dim3(cdiv(npix, 64), groups), dim3(256), 0, as_stream(stream), cost, depth_start,
                     depth_interval, depth_out, D, npix);
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}

// One thread per full-resolution pixel; its four low-resolution taps are shared with
// its neighbours and stay in L1/L2 (the whole (D,h,w) volume is a few MB).
__global__ __launch_bounds__(256) void upsample_softargmin_kernel(const float* __restrict__ cost, const float* __restrict__ depth_start,
                                                                  const float* __restrict__ depth_interval, float* __restrict__ depth_out,
                                                                  int D, int h, int w, int H, int W, float sy, float sx) {
  int ox = blockIdx.x * 64 + (threadIdx.x & 63);
  int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (ox >= W || oy >= H) return;
  float fy = (float)oy * sy, fx = (float)ox * sx;
  int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
  int y1 = min((int)ceilf(fy), h - 1), x1 = min((int)ceilf(fx), w - 1);
  float ly = fy - (float)y0, lx = fx - (float)x0;
  float start = depth_start[0], end;
  float step = linspace_step(start, depth_interval[0], D, &end);
  long npix = (long)h * w;
  const float* p00 = cost + (size_t)y0 * w + x0;
  const float* p01 = cost + (size_t)y0 * w + x1;
  const float* p10 = cost + (size_t)y1 * w + x0;
  const float* p11 = cost + (size_t)y1 * w + x1;
  Osm a = {-INFINITY, 0.f, 0.f};
  for (int d = 0; d < D; ++d) {
    size_t o = (size_t)d * npix;
    float tl = p00[o], tr = p01[o], bl = p10[o], br = p11[o];
    float t = tl + (tr - tl) * lx;
    float b = bl + (br - bl) * lx;
    float c = t + (b - t) * ly;
    osm_push(a, -1.0f * c, start + step * (float)d);
  }
  depth_out[(size_t)oy * W + ox] = a.t / a.s;
}

// The same regression with the low-resolution taps of a 16 x 16 tile of output pixels staged in LDS (round 5): a tile touches at
// most UW x UW low-resolution columns / rows (up-scale >= 3: 16 output pixels span <= 16 / 3 + 2 inputs) of every plane, 6.9 k
// floats at D = 192, loaded once with coalesced rows instead of 4 x D gathers per output pixel through L1 (1 GB of L1 traffic
// per call at 640 x 512 x 192: 92 us).  The interpolation and the online soft-max are the scalar form's operations in its order:
// identical bits.
constexpr int UW = 8;
__global__ __launch_bounds__(256) void upsample_softargmin_tile_kernel(const float* __restrict__ cost, const float* __restrict__ depth_start,
                                                                       const float* __restrict__ depth_interval, float* __restrict__ depth_out,
                                                                       int D, int h, int w, int H, int W, float sy, float sx, int dchunk) {
  extern __shared__ float tile[];                 // [dchunk][UW][UW]
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int ox0 = blockIdx.x * 16, oy0 = blockIdx.y * 16;
  const int ox = min(ox0 + tx, W - 1), oy = min(oy0 + ty, H - 1);
  // the window's origin: the first tap of the tile's first pixel
  const int wy0 = (int)floorf((float)oy0 * sy), wx0 = (int)floorf((float)ox0 * sx);
  const float fy = (float)oy * sy, fx = (float)ox * sx;
  const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
  const int y1 = min((int)ceilf(fy), h - 1), x1 = min((int)ceilf(fx), w - 1);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const int a00 = (y0 - wy0) * UW + (x0 - wx0), a01 = (y0 - wy0) * UW + (x1 - wx0);
  const int a10 = (y1 - wy0) * UW + (x0 - wx0), a11 = (y1 - wy0) * UW + (x1 - wx0);
  float start = depth_start[0], end;
  const float step = linspace_step(start, depth_interval[0], D, &end);
  const long npix = (long)h * w;
  Osm a = {-INFINITY, 0.f, 0.f};
  for (int d0 = 0; d0 < D; d0 += dchunk) {
    const int nd = min(dchunk, D - d0);
    __syncthreads();
    for (int i = threadIdx.x; i < nd * UW * UW; i += 256) {
      const int dd = i / (UW * UW), rem = i - dd * (UW * UW), yy = rem / UW, xx = rem - yy * UW;
      const int gy = min(wy0 + yy, h - 1), gx = min(wx0 + xx, w - 1);
      tile[i] = cost[(size_t)(d0 + dd) * npix + (size_t)gy * w + gx];
    }
    __syncthreads();
    for (int dd = 0; dd < nd; ++dd) {
      const float* t = tile + dd * (UW * UW);
      const float tl = t[a00], tr = t[a01], bl = t[a10], br = t[a11];
      const float tp = tl + (tr - tl) * lx;
      const float b = bl + (br - bl) * lx;
      const float c = tp + (b - tp) * ly;
      osm_push(a, -1.0f * c, start + step * (float)(d0 + dd));
    }
  }
  if (ox0 + tx < W && oy0 + ty < H) depth_out[(size_t)oy * W + ox] = a.t / a.s;
}

extern "C" int atvs_upsample_softargmin(const float* cost, const float* depth_start, const float* depth_interval,
                                        float* depth_up_out, int D, int h, int w, int up_scale, atvs_stream_t stream) {
  if (!cost || !depth_start || !depth_interval || !depth_up_out) return ATVS_ERR_NULL;
  if (D <= 0 || h <= 0 || w <= 0 || up_scale <= 0) return ATVS_ERR_SHAPE;
  int H = h * up_scale, W = w * up_scale;
  // tf.image.resize_images(align_corners=True): scale = (in-1)/(out-1), computed in double then rounded
  float sy = (H > 1) ? (float)((double)(h - 1) / (double)(H - 1)) : 0.f;
  float sx = (W > 1) ? (float)((double)(w - 1) / (double)(W - 1)) : 0.f;
  if (up_scale >= 3) {
    // 16 output pixels span at most 16 / 3 + 2 <= UW low-resolution taps; planes in chunks of 64 (16 KB of LDS)
    const int dchunk = D < 64 ? D : 64;
    hipLaunchKernelGGL(upsample_softargmin_tile_kernel, dim3(cdiv(W, 16), cdiv(H, 16)), dim3(256), (size_t)dchunk * UW * UW * 4,
                       as_stream(stream), cost, depth_start, depth_interval, depth_up_out, D, h, w, H, W, sy, sx, dchunk);
  } else {
    hipLaunchKernelGGL(upsample_softargmin_kernel, dim3(cdiv(W, 64), cdiv(H, 4)), dim3(256), 0, as_stream(stream), cost,
                       depth_start, depth_interval, depth_up_out, D, h, w, H, W, sy, sx);
  }
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}

// ---------------------------------------------------------------------------
// Probability (confidence) map: get_propability_map, /root/reference/atvsnet/model.py:13-65, as used by
// prob2depth(out_prob_map=True) :104-107 and prob2depth_upsample(out_prob_map=True) :122-125 (the ETH3D batch
// driver eval_pointcloud.py:232,269).  Per pixel, with d = (depth - depth_start) / depth_interval:
//   l0 = clip(floor(d)), l1 = clip(l0 - 1), r0 = clip(ceil(d)), r1 = clip(r0 + 1)   (clip to [0, D-1])
//   prob = P[l0] + P[l1] + P[r0] + P[r1]        (an integral d counts its plane twice, as the reference does)
// SOFTMAX: P = softmax(-cost) over D computed on the fly (max, then sum, then the four terms) instead of the
// reference's materialised probability volume; otherwise P is the given volume.  UP: the volume is read through
// the x4 align_corners bilinear interpolation of upsample_prob_vol (:66-75) instead of being materialised.
// One thread per output pixel; the (D,h,w) volume is a few MB and stays in L2.
template <bool SOFTMAX, bool UP>
__global__ __launch_bounds__(256) void probability_map_kernel(const float* __restrict__ vol, const float* __restrict__ depth_map,
                                                              const float* __restrict__ depth_start,
                                                              const float* __restrict__ depth_interval, float* __restrict__ prob_out,
                                                              int D, int h, int w, int H, int W, float sy, float sx) {
  int ox = blockIdx.x * 64 + (threadIdx.x & 63);
  int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (ox >= W || oy >= H) return;
  const long npix = (long)h * w;
  const float *p00, *p01 = nullptr, *p10 = nullptr, *p11 = nullptr;
  float lx = 0.f, ly = 0.f;
  if (UP) {
    float fy = (float)oy * sy, fx = (float)ox * sx;
    int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    int y1 = min((int)ceilf(fy), h - 1), x1 = min((int)ceilf(fx), w - 1);
    ly = fy - (float)y0;
    lx = fx - (float)x0;
    p00 = vol + (size_t)y0 * w + x0;
    p01 = vol + (size_t)y0 * w + x1;
    p10 = vol + (size_t)y1 * w + x0;
    p11 = vol + (size_t)y1 * w + x1;
  } else {
    p00 = vol + (size_t)oy * w + ox;
  }
  auto fetch = [&](int d) {
    size_t o = (size_t)d * npix;
    if (!UP) return p00[o];
    float tl = p00[o], tr = p01[o], bl = p10[o], br = p11[o];
    float t = tl + (tr - tl) * lx;
    float b = bl + (br - bl) * lx;
    return t + (b - t) * ly;
  };
  float m = 0.f, ssum = 1.f;
  if (SOFTMAX) {
    m = -INFINITY;
    for (int d = 0; d < D; ++d) m = fmaxf(m, -1.0f * fetch(d));
    ssum = 0.f;
    for (int d = 0; d < D; ++d) ssum += expf(-1.0f * fetch(d) - m);
  }
  const float dc = (depth_map[(size_t)oy * W + ox] - depth_start[0]) / depth_interval[0];
  const int l0 = min(max((int)floorf(dc), 0), D - 1);
  const int l1 = min(max(l0 - 1, 0), D - 1);
  const int r0 = min(max((int)ceilf(dc), 0), D - 1);
  const int r1 = min(max(r0 + 1, 0), D - 1);
  auto prob = [&](int d) { return SOFTMAX ? expf(-1.0f * fetch(d) - m) / ssum : fetch(d); };
  prob_out[(size_t)oy * W + ox] = ((prob(l0) + prob(l1)) + prob(r0)) + prob(r1);
}

// vol (D,h,w); depth_map / prob_out (h*up, w*up) (up_scale 1 = same resolution).  softmax != 0: vol is the
// pre-softmax cost (P = softmax(-vol)); softmax == 0: vol is the probability volume itself.
extern "C" int atvs_probability_map(const float* vol, const float* depth_map, const float* depth_start,
                                    const float* depth_interval, float* prob_out, int D, int h, int w, int up_scale,
                                    int softmax, atvs_stream_t stream) {
  if (!vol || !depth_map || !depth_start || !depth_interval || !prob_out) return ATVS_ERR_NULL;
  if (D <= 0 || h <= 0 || w <= 0 || up_scale <= 0) return ATVS_ERR_SHAPE;
  int H = h * up_scale, W = w * up_scale;
  float sy = (H > 1) ? (float)((double)(h - 1) / (double)(H - 1)) : 0.f;
  float sx = (W > 1) ? (float)((double)(w - 1) / (double)(W - 1)) : 0.f;
  dim3 grid(cdiv(W, 64), cdiv(H, 4)), block(256);
  hipStream_t s = as_stream(stream);
  if (up_scale == 1) {
    if (softmax) hipLaunchKernelGGL((probability_map_kernel<true, false>), grid, block, 0, s, vol, depth_map, depth_start, depth_interval, prob_out, D, h, w, H, W, sy, sx);
    else hipLaunchKernelGGL((probability_map_kernel<false, false>), grid, block, 0, s, vol, depth_map, depth_start, depth_interval, prob_out, D, h, w, H, W, sy, sx);
  } else {
    if (softmax) hipLaunchKernelGGL((probability_map_kernel<true, true>), grid, block, 0, s, vol, depth_map, depth_start, depth_interval, prob_out, D, h, w, H, W, sy, sx);
    else hipLaunchKernelGGL((probability_map_kernel<false, true>), grid, block, 0, s, vol, depth_map, depth_start, depth_interval, prob_out, D, h, w, H, W, sy, sx);
  }
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}
