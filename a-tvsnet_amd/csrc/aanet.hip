// Attention aggregation over source views (AANet) -- the cross-view softmax and
// weighted sum (gfx950).  The two 3x3x3 8->8 convolutions per view run in conv.hip as
// ONE 8->16 convolution with a ReLU epilogue (shared | unique weights side by side),
// giving SR_n = [S_n | R_n] (V,16) per view.
//
// Reference: Network.attention_activation / attention_aggregation,
// /root/reference/cnn_wrapper/network.py:282-351, 378-408 with second_weight=True,
// relu=True, biased=False (call sites cnn_wrapper/atvsnet.py:202,234):
//   S_sum = sum_n S_n ;  U_n = R_n - S_n + S_sum ;  score = softmax_n(U) ;
//   out   = sum_n score_n * X_n
// HBM-bound element-wise work over V*C values: every SR_n and X_n is read once.
//
// The *_partial kernels split the same arithmetic at its three reductions over views
// so that views sharded across GPUs combine with three all-reduces (SUM, MAX, SUM).
#include "common.h"

#define AANET_MAX_VIEWS 16

// e^(u - max) exactly as atvs_aanet_softmax_sum (common.h) forms it: 2^(u log2 e - ml), ml = max * log2 e
#define AA_L2E 1.44269504088896340736f
__device__ __forceinline__ float aa_exp(float u, float ml) { return __builtin_amdgcn_exp2f(__builtin_fmaf(u, AA_L2E, -ml)); }

struct ViewPtrs {
  const float* sr[AANET_MAX_VIEWS];
  const float* x[AANET_MAX_VIEWS];
};

// one lane per float4 of the (V, 8) output; every S_n, R_n, X_n is read once and kept in registers
template <int NV>
__global__ __launch_bounds__(256) void aanet_combine_kernel(ViewPtrs p, int nv_rt, float* __restrict__ out, long n4) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  long v = i >> 1;
  int half = (int)(i & 1) * 4;
  size_t so = (size_t)v * 16 + half, xo = (size_t)v * 8 + half;
  float4 u[NV], x[NV];
  float4 ssum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int n = 0; n < NV; ++n) {
    float4 s = ld4(p.sr[n] + so), r = ld4(p.sr[n] + so + 8);
    x[n] = ld4(p.x[n] + xo);
    ssum.x += s.x; ssum.y += s.y; ssum.z += s.z; ssum.w += s.w;
    u[n] = make_float4(r.x - s.x, r.y - s.y, r.z - s.z, r.w - s.w);
  }
  float ux[NV], uy[NV], uz[NV], uw[NV];
#pragma unroll
  for (int n = 0; n < NV; ++n) {
    ux[n] = u[n].x + ssum.x; uy[n] = u[n].y + ssum.y; uz[n] = u[n].z + ssum.z; uw[n] = u[n].w + ssum.w;      // (R - S) + S_sum
  }
  float4 o;
  atvs_aanet_softmax_sum<NV>(ux, [&](int n) { return x[n].x; }, &o.x);
  atvs_aanet_softmax_sum<NV>(uy, [&](int n) { return x[n].y; }, &o.y);
  atvs_aanet_softmax_sum<NV>(uz, [&](int n) { return x[n].z; }, &o.z);
  atvs_aanet_softmax_sum<NV>(uw, [&](int n) { return x[n].w; }, &o.w);
  st4(out + xo, o);
}

// any number of views: re-reads the operands per pass
__global__ __launch_bounds__(256) void aanet_combine_generic_kernel(ViewPtrs p, int nv, float* __restrict__ out, long n4) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  long v = i >> 1;
  int half = (int)(i & 1) * 4;
  size_t so = (size_t)v * 16 + half, xo = (size_t)v * 8 + half;
  float4 ssum = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int n = 0; n < nv; ++n) {
    float4 s = ld4(p.sr[n] + so);
    ssum.x += s.x; ssum.y += s.y; ssum.z += s.z; ssum.w += s.w;
  }
  float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  for (int n = 0; n < nv; ++n) {
    float4 s = ld4(p.sr[n] + so), r = ld4(p.sr[n] + so + 8);
    m.x = fmaxf(m.x, (r.x - s.x) + ssum.x);
    m.y = fmaxf(m.y, (r.y - s.y) + ssum.y);
    m.z = fmaxf(m.z, (r.z - s.z) + ssum.z);
    m.w = fmaxf(m.w, (r.w - s.w) + ssum.w);
  }
  // atvs_aanet_softmax_sum's arithmetic (common.h) with the operands re-read per pass
  const float4 ml = make_float4(m.x * AA_L2E, m.y * AA_L2E, m.z * AA_L2E, m.w * AA_L2E);
  float4 den = make_float4(0.f, 0.f, 0.f, 0.f), num = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int n = 0; n < nv; ++n) {
    float4 s = ld4(p.sr[n] + so), r = ld4(p.sr[n] + so + 8), x = ld4(p.x[n] + xo);
    const float ex = aa_exp((r.x - s.x) + ssum.x, ml.x), ey = aa_exp((r.y - s.y) + ssum.y, ml.y);
    const float ez = aa_exp((r.z - s.z) + ssum.z, ml.z), ew = aa_exp((r.w - s.w) + ssum.w, ml.w);
    den.x += ex; den.y += ey; den.z += ez; den.w += ew;
    num.x = __builtin_fmaf(ex, x.x, num.x); num.y = __builtin_fmaf(ey, x.y, num.y);
    num.z = __builtin_fmaf(ez, x.z, num.z); num.w = __builtin_fmaf(ew, x.w, num.w);
  }
  st4(out + xo, make_float4(num.x * __builtin_amdgcn_rcpf(den.x), num.y * __builtin_amdgcn_rcpf(den.y),
                            num.z * __builtin_amdgcn_rcpf(den.z), num.w * __builtin_amdgcn_rcpf(den.w)));
}

static int fill_ptrs(ViewPtrs* p, const float* const* sr, const float* const* x, int nv) {
  if (!sr || nv <= 0 || nv > AANET_MAX_VIEWS) return ATVS_ERR_SHAPE;
  for (int n = 0; n < AANET_MAX_VIEWS; ++n) {
    p->sr[n] = n < nv ? sr[n] : nullptr;
    p->x[n] = (x && n < nv) ? x[n] : nullptr;
    if (n < nv && (!p->sr[n] || (x && !p->x[n]))) return ATVS_ERR_NULL;
  }
  return ATVS_OK;
}

// sr_ptrs / x_ptrs: HOST arrays of nv device pointers: SR_n (V,16), X_n (V,8).  out (V,8).
extern "C" int atvs_aanet_combine(const float* const* sr_ptrs, const float* const* x_ptrs, int nv, float* out,
                                  long V, atvs_stream_t stream) {
  if (!out || !x_ptrs) return ATVS_ERR_NULL;
  if (V <= 0) return ATVS_ERR_SHAPE;
  ViewPtrs p;
  int rc = fill_ptrs(&p, sr_ptrs, x_ptrs, nv);
  if (rc) return rc;
  long n4 = V * 2;
  dim3 grid(cdiv(n4, 256)), block(256);
  hipStream_t s = as_stream(stream);
  switch (nv) {
    case 1: hipLaunchKernelGGL((aanet_combine_kernel<1>), grid, block, 0, s, p, nv, out, n4); break;
    case 2: hipLaunchKernelGGL((aanet_combine_kernel<2>), grid, block, 0, s, p, nv, out, n4); break;
    case 3: hipLaunchKernelGGL((aanet_combine_kernel<3>), grid, block, 0, s, p, nv, out, n4); break;
    case 4: hipLaunchKernelGGL((aanet_combine_kernel<4>), grid, block, 0, s, p, nv, out, n4); break;
    case 5: hipLaunchKernelGGL((aanet_combine_kernel<5>), grid, block, 0, s, p, nv, out, n4); break;
    case 6: hipLaunchKernelGGL((aanet_combine_kernel<6>), grid, block, 0, s, p, nv, out, n4); break;
    case 7: hipLaunchKernelGGL((aanet_combine_kernel<7>), grid, block, 0, s, p, nv, out, n4); break;
    case 8: hipLaunchKernelGGL((aanet_combine_kernel<8>), grid, block, 0, s, p, nv, out, n4); break;
    default: hipLaunchKernelGGL(aanet_combine_generic_kernel, grid, block, 0, s, p, nv, out, n4); break;
  }
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}

// ---- view-sharded form -----------------------------------------------------
// stage 0: ssum_local[v,c] = sum over local views of S_n[v,c]            -> all-reduce SUM
// stage 1: umax_local[v,c] = max over local views of U_n[v,c]            -> all-reduce MAX
// stage 2: acc_local[0][v,c] = sum e_n, acc_local[1][v,c] = sum e_n X_n   -> all-reduce SUM
// stage 3: out = acc[1] / acc[0]
__global__ __launch_bounds__(256) void aanet_partial_kernel(ViewPtrs p, int nv, int stage, const float* __restrict__ ssum,
                                                            const float* __restrict__ umax, float* __restrict__ out,
                                                            long n4, long V8) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  long v = i >> 1;
  int half = (int)(i & 1) * 4;
  size_t so = (size_t)v * 16 + half, xo = (size_t)v * 8 + half;
  if (stage == 0) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int n = 0; n < nv; ++n) {
      float4 s = ld4(p.sr[n] + so);
      a.x += s.x; a.y += s.y; a.z += s.z; a.w += s.w;
    }
    st4(out + xo, a);
    return;
  }
  float4 ss = ld4(ssum + xo);
  if (stage == 1) {
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int n = 0; n < nv; ++n) {
      float4 s = ld4(p.sr[n] + so), r = ld4(p.sr[n] + so + 8);
      m.x = fmaxf(m.x, (r.x - s.x) + ss.x);
      m.y = fmaxf(m.y, (r.y - s.y) + ss.y);
      m.z = fmaxf(m.z, (r.z - s.z) + ss.z);
      m.w = fmaxf(m.w, (r.w - s.w) + ss.w);
    }
    st4(out + xo, m);
    return;
  }
  float4 m = ld4(umax + xo);
  const float4 ml = make_float4(m.x * AA_L2E, m.y * AA_L2E, m.z * AA_L2E, m.w * AA_L2E);
  float4 den = make_float4(0.f, 0.f, 0.f, 0.f), num = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int n = 0; n < nv; ++n) {
    float4 s = ld4(p.sr[n] + so), r = ld4(p.sr[n] + so + 8), x = ld4(p.x[n] + xo);
    float ex = aa_exp((r.x - s.x) + ss.x, ml.x), ey = aa_exp((r.y - s.y) + ss.y, ml.y);
    float ez = aa_exp((r.z - s.z) + ss.z, ml.z), ew = aa_exp((r.w - s.w) + ss.w, ml.w);
    den.x += ex; den.y += ey; den.z += ez; den.w += ew;
    num.x = __builtin_fmaf(ex, x.x, num.x); num.y = __builtin_fmaf(ey, x.y, num.y);
    num.z = __builtin_fmaf(ez, x.z, num.z); num.w = __builtin_fmaf(ew, x.w, num.w);
  }
  st4(out + xo, den);
  st4(out + V8 + xo, num);
}

extern "C" int atvs_aanet_partial(const float* const* sr_ptrs, const float* const* x_ptrs, int nv, int stage,
                                  const float* ssum, const float* umax, float* out, long V, atvs_stream_t stream) {
  if (!out) return ATVS_ERR_NULL;
  if (V <= 0 || stage < 0 || stage > 2) return ATVS_ERR_SHAPE;
  if ((stage >= 1 && !ssum) || (stage == 2 && (!umax || !x_ptrs))) return ATVS_ERR_NULL;
  ViewPtrs p;
  int rc = fill_ptrs(&p, sr_ptrs, stage == 2 ? x_ptrs : nullptr, nv);
  if (rc) return rc;
  long n4 = V * 2;
  hipLaunchKernelGGL(aanet_partial_kernel, dim3(cdiv(n4, 256)), dim3(256), 0, as_stream(stream), p, nv, stage, ssum, umax,
                     out, n4, V * 8);
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}

__global__ __launch_bounds__(256) void divide_kernel(const float* __restrict__ num, const float* __restrict__ den,
                                                     float* __restrict__ out, long n) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  float4 a = ld4(num + i), b = ld4(den + i);
  st4(out + i, make_float4(a.x / b.x, a.y / b.y, a.z / b.z, a.w / b.w));
}

// out = num / den over n floats (n % 4 == 0)
extern "C" int atvs_divide(const float* num, const float* den, float* out, long n, atvs_stream_t stream) {
  if (!num || !den || !out) return ATVS_ERR_NULL;
  if (n <= 0 || (n % 4) != 0) return ATVS_ERR_SHAPE;
  hipLaunchKernelGGL(divide_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, as_stream(stream), num, den, out, n);
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}
