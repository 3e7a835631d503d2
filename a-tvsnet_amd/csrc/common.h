// Shared helpers for the gfx950 kernels behind include/atvsnet_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/atvsnet_hip.h"

#define ATVS_LAUNCH_CHECK()                                   \
  do {                                                        \
    hipError_t e_ = hipGetLastError();                        \
    if (e_ != hipSuccess) return ATVS_ERR_LAUNCH;             \
  } while (0)

static inline hipStream_t as_stream(atvs_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) ONCE per (kernel instantiation, device ordinal of this process).  The entry points
// are re-entrant per stream and may be called from several host threads (include/atvsnet_hip.h): the flags are atomics -- a thread
// reads `true` (acquire) only after the call it stands for has returned (release); two threads that race on a clear flag both make
// the (idempotent) call.  No other process-global state exists behind the ABI.
struct AtvsAttrOnce {
  std::atomic<bool> done[64];
};
static inline int atvs_set_max_lds_once(AtvsAttrOnce& once, const void* kernel, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return ATVS_ERR_LAUNCH;
  if (!once.done[dev].load(std::memory_order_acquire)) {
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return ATVS_ERR_LAUNCH;
    once.done[dev].store(true, std::memory_order_release);
  }
  return ATVS_OK;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------
// Sampling coordinates, exactly the reference's order of operations
// (homography_warping.py:31-104, 230-257).  Built with -ffp-contract=off so
// that the oracle (oracle/homography_warping.py) matches bit for bit.
// ---------------------------------------------------------------------------
struct Tap4 {
  int i00, i01, i10, i11;   // pixel indices y*W+x of the four taps
  float wa, wb, wc, wd;     // area weights
  float valid;              // 1.f / 0.f
};

__device__ __forceinline__ Tap4 bilinear_taps(float xw, float yw, int H, int W) {
  Tap4 t;
  float x = xw - 0.5f, y = yw - 0.5f;
  bool v = (x >= 0.f) && (y >= 0.f) && (x < (float)(W - 1)) && (y < (float)(H - 1));
  int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
  if (v) {
    x0 = (int)floorf(x);
    y0 = (int)floorf(y);
    x1 = x0 + 1;
    y1 = y0 + 1;
  }
  float m = v ? 1.f : 0.f;
  x = x * m;  // a non-finite coordinate stays NaN, as tf.multiply does
  y = y * m;
  x0 = min(max(x0, 0), W - 1);
  x1 = min(max(x1, 0), W - 1);
  y0 = min(max(y0, 0), H - 1);
  y1 = min(max(y1, 0), H - 1);
  float x0f = (float)x0, x1f = (float)x1, y0f = (float)y0, y1f = (float)y1;
  t.wa = (y1f - y) * (x1f - x);
  t.wb = (y1f - y) * (x - x0f);
  t.wc = (y - y0f) * (x1f - x);
  t.wd = (y - y0f) * (x - x0f);
  t.i00 = y0 * W + x0;
  t.i01 = y0 * W + x1;
  t.i10 = y1 * W + x0;
  t.i11 = y1 * W + x1;
  t.valid = m;
  return t;
}

// nearest: tf.round (half to even), invalid -> pixel (0,0), value not masked (quirk C4)
__device__ __forceinline__ int nearest_tap(float xw, float yw, int H, int W, float* valid) {
  float x = xw - 0.5f, y = yw - 0.5f;
  bool v = (x >= 0.f) && (y >= 0.f) && (x < (float)(W - 1)) && (y < (float)(H - 1));
  int x0 = 0, y0 = 0;
  if (v) {
    x0 = (int)rintf(x);
    y0 = (int)rintf(y);
  }
  *valid = v ? 1.f : 0.f;
  return y0 * W + x0;
}

// p' = H (x+.5, y+.5, 1), projective divide with the reference's /0 guard (:251-254)
__device__ __forceinline__ void homography_apply(const float* __restrict__ Hm, int x, int y, float* xw, float* yw) {
  float px = (float)x + 0.5f, py = (float)y + 0.5f;
  float xa = (Hm[0] * px + Hm[1] * py) + Hm[2];
  float ya = (Hm[3] * px + Hm[4] * py) + Hm[5];
  float dv = (Hm[6] * px + Hm[7] * py) + Hm[8];
  dv = dv + ((dv == 0.0f) ? 1.f : 0.f) * 1e-7f;
  *xw = xa / dv;
  *yw = ya / dv;
}

// Training-mode batch norm of one value, y = (x - mean) * scale + beta, as ONE fused multiply-add: x * scale + shift with
// shift = beta - mean * scale formed once per channel -- the form tf.nn.batch_normalization itself computes (inv = rsqrt(var + eps);
// x * inv + (offset - mean * inv); the reference's 3-D batch norms take that path: tf.layers.batch_normalization on 5-D inputs,
// /root/reference/cnn_wrapper/network.py:206-212).  ONE definition for every kernel that normalises: the element-wise passes
// (norm.hip) and every convolution that normalises its input on load must agree bit for bit.  Two instructions per value with the
// ReLU instead of four (sub, mul, add, max): in the two-role kernels every staging instruction is stage time.
__device__ __forceinline__ float atvs_bn_shift(float mean, float scale, float beta) { return __builtin_fmaf(-mean, scale, beta); }
__device__ __forceinline__ float atvs_bn1(float x, float scale, float shift) { return __builtin_fmaf(x, scale, shift); }
__device__ __forceinline__ float4 atvs_bn_shift4(const float4& m, const float4& s, const float4& b) {
  return make_float4(atvs_bn_shift(m.x, s.x, b.x), atvs_bn_shift(m.y, s.y, b.y), atvs_bn_shift(m.z, s.z, b.z), atvs_bn_shift(m.w, s.w, b.w));
}
__device__ __forceinline__ float4 atvs_bn4(const float4& v, const float4& s, const float4& sh) {
  return make_float4(atvs_bn1(v.x, s.x, sh.x), atvs_bn1(v.y, s.y, sh.y), atvs_bn1(v.z, s.z, sh.z), atvs_bn1(v.w, s.w, sh.w));
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// 16-byte store of data its producer does not read again (large volumes consumed by a LATER kernel): the non-temporal hint
// keeps the line out of the way of what the L2 is holding for reuse.  -DATVS_NO_NT_STORES: plain stores (A/B).
// cache-policy operand of raw buffer stores for the same purpose (gfx940+: bit 1 = nt)
#ifdef ATVS_NO_NT_STORES
#define ATVS_BUF_NT 0
#else
#define ATVS_BUF_NT 2
#endif
typedef float atvs_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned atvs_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4_stream(float* p, float4 v) {
#ifdef ATVS_NO_NT_STORES
  *reinterpret_cast<float4*>(p) = v;
#else
  __builtin_nontemporal_store((atvs_f32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<atvs_f32x4*>(p));
#endif
}
__device__ __forceinline__ void st4u_stream(void* p, uint4 v) {
#ifdef ATVS_NO_NT_STORES
  *reinterpret_cast<uint4*>(p) = v;
#else
  __builtin_nontemporal_store((atvs_u32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<atvs_u32x4*>(p));
#endif
}

// The two fp16 pieces of two fp32 values, packed: h0 = f16(x), h1 = f16((x - h0) * 2^11)  (x - h0 is exact in fp32), in FIVE
// vector instructions: one packed conversion, two mixed-precision fused multiply-adds h0 * -1 + x that read the fp16 halves
// directly, two that scale by 2^11 (rs = 2048.f in a scalar register) and round into the halves of the second piece (plain C
// costs 8-9: the compiler converts h0 back to fp32 first and, with -ffp-contract=off, rewrites fma(h0, -1, x) as a subtraction).
// Same values as the C form: every fma is exact before its one rounding; r * 2048 + 0 keeps r's zero (+0, as x - h0 gives it).
// ONE definition for every producer of pieces: conv_xb.hip's staging wavefronts and the plane-sweep warp (geometry.hip) must
// split bit-identically.
__device__ __forceinline__ void atvs_split2_f16(float x0, float x1, float rs, unsigned* h0, unsigned* h1) {
  float r0, r1;
  unsigned a, b;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(a) : "v"(x0), "v"(x1));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(a), "v"(x0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(a), "v"(x1));
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(b) : "v"(r0), "s"(rs));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(b) : "v"(r1), "s"(rs));
  *h0 = a;
  *h1 = b;
}

// ... of a float4: two 8-byte records (h0 of the four values | h1 of the four values), residual scale 2^11
__device__ __forceinline__ void atvs_split4_f16(const float4& v, uint2* h0, uint2* h1) {
  atvs_split2_f16(v.x, v.y, 2048.f, &h0->x, &h1->x);
  atvs_split2_f16(v.z, v.w, 2048.f, &h0->y, &h1->y);
}

// The AANet's softmax over views and weighted sum for ONE value (reference cnn_wrapper/network.py:326-406: tf.nn.softmax over the
// view axis, then sum_n score_n * X_n):  *out = sum_n softmax_n(u)[n] * x(n) = (sum_n e_n x_n) / (sum_n e_n), e_n = e^(u_n - max).
// ONE definition for aanet_combine_kernel (aanet.hip) and the fused module (aanet_b.hip), which must agree bit for bit.  u[] is
// overwritten.  Written for FEW vector instructions -- in aanet_b it runs on wavefronts that share their SIMD's issue slots with MFMA
// wavefronts: e_n = 2^(u_n log2 e - max log2 e) as one fused multiply-add + v_exp_f32 (the rounding of max log2 e is a factor
// common to every e_n: it cancels in the ratio), numerator and denominator accumulated side by side, ONE v_rcp_f32 (1 ulp) and one
// multiplication at the end: 5 NV + 3 instructions instead of ~25 NV for expf and NV IEEE divisions; within a few ulp of the exact
// softmax-weighted sum wherever a term matters (tests: 2e-5 of the output maximum against the oracle).
template <int NV, class X>
__device__ __forceinline__ void atvs_aanet_softmax_sum(float* u, X&& x, float* out) {
  constexpr float L2E = 1.44269504088896340736f;
  float m = -INFINITY;
#pragma unroll
  for (int n = 0; n < NV; ++n) m = fmaxf(m, u[n]);
  const float ml = m * L2E;
  float den = 0.f, num = 0.f;
#pragma unroll
  for (int n = 0; n < NV; ++n) {
    const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(u[n], L2E, -ml));
    den += e;
    num = __builtin_fmaf(e, x(n), num);
  }
  *out = num * __builtin_amdgcn_rcpf(den);
}

__device__ __forceinline__ float4 blend4(const Tap4& t, float4 a, float4 b, float4 c, float4 d) {
  float4 o;
  o.x = ((t.wa * a.x + t.wb * b.x) + t.wc * c.x) + t.wd * d.x;
  o.y = ((t.wa * a.y + t.wb * b.y) + t.wc * c.y) + t.wd * d.y;
  o.z = ((t.wa * a.z + t.wb * b.z) + t.wc * c.z) + t.wd * d.z;
  o.w = ((t.wa * a.w + t.wb * b.w) + t.wc * c.w) + t.wd * d.w;
  return o;
}
