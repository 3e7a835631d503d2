// Training-mode batch normalisation (batch statistics at inference, reference
// quirk C1) and the element-wise glue around the convolutions (gfx950).
//
// Reference: tf.layers.batch_normalization(center=False, scale=False, training=True)
// at /root/reference/cnn_wrapper/network.py:206-212, 541-547; slim.batch_norm
// (center=True) at :570-571; tf.add_n at :695-697.  y = (x - mean) * rsqrt(var + 1e-3) (as x * scale + (beta - mean * scale): atvs_bn1, common.h)
// [+ beta], biased variance over every axis but the channel.
//
// Statistics are deterministic: fixed-shape per-workgroup partial sums (written by
// the convolution epilogue or by channel_stats below) are tree-reduced in double.
#include "common.h"

// partials: [nblocks][2][cpad] doubles (sum, sum of squares).  params out: [3][C]
// floats = mean, rsqrt(var+eps), beta.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ partials, long nblocks, int cpad,
                                                          double count, const float* __restrict__ beta, float eps,
                                                          float* __restrict__ params, int C, int fold, int* __restrict__ flag) {
  __shared__ double sm[2][256];
  const int c = blockIdx.x;
  // blockIdx.y = independent sample (group): its own rows of partial sums, its own (3, C) parameter block
  partials += (size_t)blockIdx.y * nblocks * 2 * cpad;
  params += (size_t)blockIdx.y * 3 * C;
  double s = 0.0, q = 0.0;
  for (long b = threadIdx.x; b < nblocks; b += 256)
    for (int f = 0; f < fold; ++f) {      // columns c, c+C, ... hold the same channel (fused transposed conv)
      s += partials[(b * 2 + 0) * cpad + c + f * C];
      q += partials[(b * 2 + 1) * cpad + c + f * C];
    }
  sm[0][threadIdx.x] = s;
  sm[1][threadIdx.x] = q;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      sm[0][threadIdx.x] += sm[0][threadIdx.x + o];
      sm[1][threadIdx.x] += sm[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double mean = sm[0][0] / count;
    double var = sm[1][0] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    // a non-finite moment = an activation left the fp16 range of a split-operand kernel upstream (or the input was not finite): the
    // sticky device flag lets the host name the cause even where a later ReLU (fmaxf) swallows the NaN
    if (flag && !(mean - mean == 0.0 && var - var == 0.0)) atomicOr(flag, 1);
    params[c] = (float)mean;
    params[C + c] = (float)(1.0 / sqrt(var + (double)eps));
    params[2 * C + c] = beta ? beta[c] : 0.f;
  }
}

extern "C" int atvs_bn_finalize(const double* stats_partial, int groups, long num_blocks, int cpad, int fold, long count,
                                const float* beta, float eps, float* params, int C, int* nonfinite_flag, atvs_stream_t stream) {
  if (!stats_partial || !params) return ATVS_ERR_NULL;
  if (groups <= 0 || groups > 65535 || num_blocks <= 0 || C <= 0 || fold < 1 || cpad < C * fold || count <= 0)
    return ATVS_ERR_SHAPE;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C, groups), dim3(256), 0, as_stream(stream), stats_partial, num_blocks, cpad,
                     (double)count, beta, eps, params, C, fold, nonfinite_flag);
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}

// per-channel partial sums of an arbitrary (rows, C) tensor, C <= 256
#define STATS_ROWS_PER_BLOCK 512
__global__ __launch_bounds__(256) void channel_stats_kernel(const float* __restrict__ x, long rows, int C,
                                                            double* __restrict__ partials) {
  __shared__ double sm[2][256];
  x += (size_t)blockIdx.y * rows * C;                        // blockIdx.y = independent sample (group)
  partials += (size_t)blockIdx.y * gridDim.x * 2 * C;
  const int tpb = (256 / C) * C;        // active threads, multiple of C
  const int rstride = tpb / C;
  const int t = threadIdx.x;
  double s = 0.0, q = 0.0;
  if (t < tpb) {
    const int c = t % C;
    long r0 = (long)blockIdx.x * STATS_ROWS_PER_BLOCK;
    long r1 = min(rows, r0 + STATS_ROWS_PER_BLOCK);
    float fs = 0.f, fq = 0.f;
    int k = 0;
    for (long r = r0 + t / C; r < r1; r += rstride) {
      float v = x[(size_t)r * C + c];
      fs += v;
      fq += v * v;
      if (++k == 32) {   // bound the fp32 chains
        s += fs; q += fq; fs = fq = 0.f; k = 0;
      }
    }
    s += fs;
    q += fq;
  }
  sm[0][t] = s;
  sm[1][t] = q;
  __syncthreads();
  if (t < C) {
    double a = 0.0, b = 0.0;
    for (int k = t; k < tpb; k += C) {
      a += sm[0][k];
      b += sm[1][k];
    }
    partials[((size_t)blockIdx.x * 2 + 0) * C + t] = a;
    partials[((size_t)blockIdx.x * 2 + 1) * C + t] = b;
  }
}

extern "C" long atvs_channel_stats_num_blocks(long rows) { return (rows + STATS_ROWS_PER_BLOCK - 1) / STATS_ROWS_PER_BLOCK; }

extern "C" int atvs_channel_stats(const float* x, int groups, long rows, int C, double* stats_partial, atvs_stream_t stream) {
  if (!x || !stats_partial) return ATVS_ERR_NULL;
  if (groups <= 0 || groups > 65535 || rows <= 0 || C <= 0 || C > 256) return ATVS_ERR_SHAPE;
  long nb = atvs_channel_stats_num_blocks(rows);
  hipLaunchKernelGGL(channel_stats_kernel, dim3((unsigned)nb, groups), dim3(256), 0, as_stream(stream), x, rows, C, stats_partial);
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}

// rows of width ld; the C channels starting at c_off are normalised (a channel slice of a concat buffer)
template <int VEC>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ params,
                                                       float* __restrict__ y, long n, int C, int ld, int c_off, int relu,
                                                       long group_n) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  if (i >= n) return;
  int c = (int)(i % C);
  params += (i / group_n) * (3 * C);                          // group_n = elements per independent sample
  long a = (ld == C) ? i : (i / C) * ld + c_off + c;
  if (VEC == 4) {
    float4 v = ld4(x + a), m = ld4(params + c), s = ld4(params + C + c), b = ld4(params + 2 * C + c);
    v = atvs_bn4(v, s, atvs_bn_shift4(m, s, b));
    if (relu) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    st4(y + a, v);
  } else {
    float v = atvs_bn1(x[a], params[C + c], atvs_bn_shift(params[c], params[C + c], params[2 * C + c]));
    y[a] = relu ? fmaxf(v, 0.f) : v;
  }
}

extern "C" int atvs_bn_apply(const float* x, const float* params, float* y, int groups, long rows, int C, int ld, int c_off,
                             int relu, atvs_stream_t stream) {
  if (!x || !params || !y) return ATVS_ERR_NULL;
  if (groups <= 0 || rows <= 0 || C <= 0 || ld < C || c_off < 0 || c_off + C > ld) return ATVS_ERR_SHAPE;
  long n = (long)groups * rows * C;          // rows = rows per independent sample; params (groups, 3, C)
  const long gn = rows * C;
  if (C % 4 == 0 && ld % 4 == 0 && c_off % 4 == 0)
    hipLaunchKernelGGL((bn_apply_kernel<4>), dim3(cdiv(n / 4, 256)), dim3(256), 0, as_stream(stream), x, params, y, n, C, ld, c_off, relu, gn);
  else
    hipLaunchKernelGGL((bn_apply_kernel<1>), dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), x, params, y, n, C, ld, c_off, relu, gn);
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}

// tf.add_n of 2 or 3 tensors: (a + b) + c
__global__ __launch_bounds__(256) void add_n_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                    const float* __restrict__ c, float* __restrict__ y, long n) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    float4 u = ld4(a + i), v = ld4(b + i);
    u.x += v.x; u.y += v.y; u.z += v.z; u.w += v.w;
    if (c) {
      float4 w = ld4(c + i);
      u.x += w.x; u.y += w.y; u.z += w.z; u.w += w.w;
    }
    st4(y + i, u);
  } else {
    for (; i < n; ++i) {
      float u = a[i] + b[i];
      if (c) u += c[i];
      y[i] = u;
    }
  }
}

extern "C" int atvs_add_n(const float* a, const float* b, const float* c, float* y, long n, atvs_stream_t stream) {
  if (!a || !b || !y) return ATVS_ERR_NULL;
  if (n <= 0) return ATVS_ERR_SHAPE;
  hipLaunchKernelGGL(add_n_kernel, dim3(cdiv((n + 3) / 4, 256)), dim3(256), 0, as_stream(stream), a, b, c, y, n);
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}

// Sum of up to three tensors, each optionally a raw convolution output with pending batch norm + ReLU:
//   y = sum_i  (p_i ? relu?((x_i - mean_i) * rstd_i + beta_i) : x_i)
// One pass for conv_bn -> conv_bn -> add chains (cnn_wrapper/atvsnet.py U-Net skip adds) instead of a
// normalisation pass per tensor plus the add.  relu_mask bit i = ReLU after input i's batch norm.
__global__ __launch_bounds__(256) void bn_add_kernel(const float* __restrict__ x0, const float* __restrict__ p0,
                                                     const float* __restrict__ x1, const float* __restrict__ p1,
                                                     const float* __restrict__ x2, const float* __restrict__ p2,
                                                     float* __restrict__ y, long n, int C, int relu_mask,
                                                     long group_n, const float* __restrict__ base = nullptr,
                                                     float* __restrict__ y2 = nullptr) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  int c = (int)(i % C);
  const long pg = (i / group_n) * (3 * C);                    // parameter block of this element's sample
  auto term = [&](const float* x, const float* p, int relu) {
    float4 v = ld4(x + i);
    if (p) {
      p += pg;
      float4 m = ld4(p + c), s = ld4(p + C + c), b = ld4(p + 2 * C + c);
      v = atvs_bn4(v, s, atvs_bn_shift4(m, s, b));
      if (relu) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
    }
    return v;
  };
  float4 a = term(x0, p0, relu_mask & 1), b = term(x1, p1, relu_mask & 2);
  a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  if (x2) {
    float4 d = term(x2, p2, relu_mask & 4);
    a.x += d.x; a.y += d.y; a.z += d.z; a.w += d.w;
  }
  st4_stream(y + i, a);
  if (y2) {                                  // y2 = base + y, base one sample shared by all (atvs_add_n's arithmetic)
    float4 u = ld4(base + i % group_n);
    u.x += a.x; u.y += a.y; u.z += a.z; u.w += a.w;
    st4_stream(y2 + i, u);
  }
}

extern "C" int atvs_bn_add(const float* x0, const float* params0, const float* x1, const float* params1, const float* x2,
                           const float* params2, float* y, int groups, long rows, int C, int relu_mask,
                           atvs_stream_t stream) {
  if (!x0 || !x1 || !y) return ATVS_ERR_NULL;
  if (groups <= 0 || rows <= 0 || C <= 0 || (C % 4) != 0) return ATVS_ERR_SHAPE;
  long n = (long)groups * rows * C;          // rows per independent sample; params_i (groups, 3, C)
  hipLaunchKernelGGL(bn_add_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, as_stream(stream), x0, params0, x1, params1, x2,
                     params2, y, n, C, relu_mask, rows * C);
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}

// atvs_bn_add that ALSO writes y2 = base + y, base (rows, C) ONE sample shared by the `groups` samples: the refinement's last skip
// add global_refine_3dconv6_1 (the cost residual, read by the 8 -> 1 head) together with refined_cost = filtered_cost +
// cost_residual (model.py:438) of every source view -- one pass instead of the add and a tf.add_n per view.  y and y2: the bits of
// atvs_bn_add and atvs_add_n(base, y).
extern "C" int atvs_bn_add_plus(const float* x0, const float* params0, const float* x1, const float* params1, const float* x2,
                                const float* params2, float* y, const float* base, float* y2, int groups, long rows, int C,
                                int relu_mask, atvs_stream_t stream) {
  if (!x0 || !x1 || !y || !base || !y2) return ATVS_ERR_NULL;
  if (groups <= 0 || rows <= 0 || C <= 0 || (C % 4) != 0) return ATVS_ERR_SHAPE;
  long n = (long)groups * rows * C;
  hipLaunchKernelGGL(bn_add_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, as_stream(stream), x0, params0, x1, params1, x2,
                     params2, y, n, C, relu_mask, rows * C, base, y2);
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}

