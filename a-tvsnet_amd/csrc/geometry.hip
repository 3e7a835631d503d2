// Plane-sweep geometry kernels (gfx950): homographies, homography warps feeding
// the cost / photo / geo volumes, per-pixel-depth warp, depth transform, visual hull.
//
// Reference semantics: /root/reference/atvsnet/homography_warping.py (whole
// file) and the volume construction of /root/reference/atvsnet/model.py:157-200,
// 270-336.  All HBM-bound: the D x h x w x C output volume is written once,
// coalesced along the channel-last rows (a wavefront covers 64*16 B = 1 KiB of
// consecutive output); the small source map (h x w x C) is gathered through L2.
#include "common.h"

// buffer_load_dwordx4 (offen).  hipcc 7.2's __builtin_amdgcn_raw_buffer_load_b128 compiles to a ONE-dword load whose value is
// splat over the four components (checked in the ISA), so the LLVM intrinsic is bound by name instead.
typedef float geo_f32x4 __attribute__((ext_vector_type(4)));
__device__ geo_f32x4 atvs_buffer_load_x4(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.ptr.buffer.load.v4f32");

// ---------------------------------------------------------------------------
// 3x3 helpers, fixed operation order (matches oracle mm3 / inv3)
// ---------------------------------------------------------------------------
struct M3 { float m[9]; };

__device__ __forceinline__ M3 mm3(const M3& a, const M3& b) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      r.m[i * 3 + j] = (a.m[i * 3 + 0] * b.m[0 * 3 + j] + a.m[i * 3 + 1] * b.m[1 * 3 + j]) + a.m[i * 3 + 2] * b.m[2 * 3 + j];
  return r;
}
__device__ __forceinline__ void mv3(const M3& a, const float* v, float* o) {
  for (int i = 0; i < 3; ++i) o[i] = (a.m[i * 3 + 0] * v[0] + a.m[i * 3 + 1] * v[1]) + a.m[i * 3 + 2] * v[2];
}
__device__ __forceinline__ M3 transpose3(const M3& a) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i * 3 + j] = a.m[j * 3 + i];
  return r;
}
__device__ __forceinline__ M3 inv3(const M3& k) {
  float a = k.m[0], b = k.m[1], c = k.m[2], d = k.m[3], e = k.m[4], f = k.m[5], g = k.m[6], h = k.m[7], i = k.m[8];
  float c00 = e * i - f * h, c01 = c * h - b * i, c02 = b * f - c * e;
  float c10 = f * g - d * i, c11 = a * i - c * g, c12 = c * d - a * f;
  float c20 = d * h - e * g, c21 = b * g - a * h, c22 = a * e - b * d;
  float det = (a * c00 + b * c10) + c * c20;
  M3 r;
  r.m[0] = c00 / det; r.m[1] = c01 / det; r.m[2] = c02 / det;
  r.m[3] = c10 / det; r.m[4] = c11 / det; r.m[5] = c12 / det;
  r.m[6] = c20 / det; r.m[7] = c21 / det; r.m[8] = c22 / det;
  return r;
}
// cam (2,4,4): [0] = extrinsic [R|t], [1][:3,:3] = K
__device__ __forceinline__ void split_cam(const float* cam, M3* R, float* t, M3* K) {
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) {
      R->m[i * 3 + j] = cam[i * 4 + j];
      K->m[i * 3 + j] = cam[16 + i * 4 + j];
    }
    t[i] = cam[i * 4 + 3];
  }
}

// H_d = K_r R_r (I - (c_r - c_l) n_l^T delta_d) R_l^T K_l^-1   (homography_warping.py:179-227)
__global__ void homographies_kernel(const float* __restrict__ left_cam, const float* __restrict__ right_cam,
                                    const float* __restrict__ depth_start, const float* __restrict__ depth_interval,
                                    float* __restrict__ Hout, int depth_num, int inverse_depth) {
  int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= depth_num) return;
  M3 Rl, Kl, Rr, Kr;
  float tl[3], tr[3];
  split_cam(left_cam, &Rl, tl, &Kl);
  split_cam(right_cam, &Rr, tr, &Kr);
  float depth = depth_start[0] + (float)d * depth_interval[0];
  M3 Kli = inv3(Kl);
  M3 RlT = transpose3(Rl), RrT = transpose3(Rr);
  float cl[3], cr[3], crel[3];
  mv3(RlT, tl, cl);
  mv3(RrT, tr, cr);
  for (int i = 0; i < 3; ++i) crel[i] = (-cr[i]) - (-cl[i]);
  M3 mid0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float tv = crel[i] * Rl.m[2 * 3 + j];
      float e = (i == j) ? 1.f : 0.f;
      mid0.m[i * 3 + j] = inverse_depth ? (e - tv * depth) : (e - tv / depth);
    }
  M3 mid1 = mm3(RlT, Kli);
  M3 mid2 = mm3(mid0, mid1);
  M3 Hm = mm3(Kr, mm3(Rr, mid2));
  for (int i = 0; i < 9; ++i) Hout[d * 9 + i] = Hm.m[i];
}

extern "C" int atvs_get_homographies(const float* left_cam, const float* right_cam, const float* depth_start,
                                     const float* depth_interval, float* homographies, int depth_num,
                                     int inverse_depth, atvs_stream_t stream) {
  if (!left_cam || !right_cam || !depth_start || !depth_interval || !homographies) return ATVS_ERR_NULL;
  if (depth_num <= 0) return ATVS_ERR_SHAPE;
  hipLaunchKernelGGL(homographies_kernel, dim3(cdiv(depth_num, 64)), dim3(64), 0, as_stream(stream), left_cam,
                     right_cam, depth_start, depth_interval, homographies, depth_num, inverse_depth);
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}

// ---------------------------------------------------------------------------
// Homography warp of an (h, w, C) map onto D planes -> out (D, h, w, ld) at
// channel offset c_off.  MODE selects the fused epilogue:
//   0 plain warp                              (model.py:190-194, cost volume)
//   1 |warp - ref| * mask                     (model.py:272-279, photo volume)
//   2 (|warp - delta_d| / interval / D) * mask replicated to `rep` channels,
//     source has one channel                  (model.py:292-297, geo view volume)
//   3 nearest-neighbour warp (homography_warping.py:45-56: tf.round, invalid -> pixel (0,0), value NOT masked)
// VEC = 4: C % 4 == 0, float4 per lane.  VEC = 1: scalar per lane.
// ---------------------------------------------------------------------------
template <int MODE, int VEC>
__global__ __launch_bounds__(256) void warp_planes_kernel(
    const float* __restrict__ src, const float* __restrict__ Hmats, const float* __restrict__ ref,
    const float* __restrict__ depth_start, const float* __restrict__ depth_interval, float* __restrict__ out,
    float* __restrict__ mask_out, int D, int h, int w, int C, int ld, int c_off, int rep) {
  const int d = blockIdx.y;
  const int cg = (MODE == 2) ? 1 : C / VEC;   // lanes per pixel
  long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long npix = (long)h * w;
  if (gid >= npix * cg) return;
  int pix = (int)(gid / cg);
  int c = (int)(gid % cg) * VEC;
  int y = pix / w, x = pix % w;
  float Hm[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) Hm[i] = Hmats[d * 9 + i];
  float xw, yw;
  homography_apply(Hm, x, y, &xw, &yw);
  size_t obase = ((size_t)d * npix + pix) * (size_t)ld + c_off;
  if (MODE == 3) {
    float valid;
    const int idx = nearest_tap(xw, yw, h, w, &valid);
    if (VEC == 4) st4(out + obase + c, ld4(src + (size_t)idx * C + c));
    else out[obase + c] = src[(size_t)idx * C + c];
    if (mask_out && c == 0) mask_out[(size_t)d * npix + pix] = valid;
    return;
  }
  Tap4 t = bilinear_taps(xw, yw, h, w);
  if (MODE == 2) {
    float v = ((t.wa * src[t.i00] + t.wb * src[t.i01]) + t.wc * src[t.i10]) + t.wd * src[t.i11];
    float val = depth_start[0] + (float)d * depth_interval[0];
    float g = (fabsf(v - val) / depth_interval[0] / (float)D) * t.valid;
    for (int r = 0; r < rep; ++r) out[obase + r] = g;
    if (mask_out) mask_out[(size_t)d * npix + pix] = t.valid;
    return;
  }
  if (VEC == 4) {
    float4 a = ld4(src + (size_t)t.i00 * C + c), b = ld4(src + (size_t)t.i01 * C + c);
    float4 cc = ld4(src + (size_t)t.i10 * C + c), dd = ld4(src + (size_t)t.i11 * C + c);
    float4 o = blend4(t, a, b, cc, dd);
    if (MODE == 1) {
      float4 r = ld4(ref + (size_t)pix * C + c);
      o.x = fabsf(o.x - r.x) * t.valid;
      o.y = fabsf(o.y - r.y) * t.valid;
      o.z = fabsf(o.z - r.z) * t.valid;
      o.w = fabsf(o.w - r.w) * t.valid;
    }
    st4(out + obase + c, o);
  } else {
    float o = ((t.wa * src[(size_t)t.i00 * C + c] + t.wb * src[(size_t)t.i01 * C + c]) +
               t.wc * src[(size_t)t.i10 * C + c]) + t.wd * src[(size_t)t.i11 * C + c];
    if (MODE == 1) o = fabsf(o - ref[(size_t)pix * C + c]) * t.valid;
    out[obase + c] = o;
  }
  if (mask_out && c == 0) mask_out[(size_t)d * npix + pix] = t.valid;
}

// The same warp with the per-pixel geometry computed ONCE and shared by the pixel's channel-group lanes.  In the kernel
// above every one of the C/4 lanes of a pixel repeats the homography, the floor / clip and the four area weights (~100
// VALU instructions for one 16-byte store: the kernel was instruction-bound at 46 % of the HBM peak).  Here a workgroup
// owns 256 pixels of one plane: phase 1, one pixel per thread -> taps and weights into LDS (48 bytes per pixel); phase 2,
// C/4 passes in which thread t handles pixel pass * (256 / cg) + t / cg, channel group t % cg: two broadcast LDS reads,
// four gathers, the blend (the same IEEE operations in the same order as blend4, so the result is bit-identical), one
// 16-byte store -- every pass writes 4 KB contiguous.
// (Round 3: an unrolled form of this kernel, and homographies_kernel, produced wrong lane quarters beside wavefronts of a
// bf16-MFMA kernel on the same SIMD -- two depth maps in flight.  Common factor: compiler-formed packed fp32 arithmetic.  This
// file is built with -fno-slp-vectorize (_lib.flags_for) and the blend below is scalar.  DESIGN.md appendix B;
// tests/test_gpu_pipeline.py::test_small_kernels_beside_other_wavefronts, ::test_two_depth_maps_in_flight_fullsize.)
// PIECES (chunk-planar output only): every value leaves as its two fp16 pieces (atvs_split2_f16: the split conv_xb.hip's staging
// wavefronts would otherwise perform, once per value instead of once per halo copy) -- a chunk plane is then
// [2 pieces][D][h][w][8 fp16], the same bytes as [D][h][w][8 fp32]; piece_bytes = D * h * w * 16.
template <int MODE, bool PIECES>
__global__ __launch_bounds__(256) void warp_planes_shared_kernel(
    const float* __restrict__ src, const float* __restrict__ Hmats, const float* __restrict__ ref,
    float* __restrict__ out, float* __restrict__ mask_out, int h, int w, int C, int ld, int c_off, long plane_stride,
    long piece_bytes) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  __shared__ __attribute__((aligned(16))) float s_geo[256 * 12];
  const int d = blockIdx.y;
  const int tid = threadIdx.x;
  const long npix = (long)h * w;
  // XCD-aware: workgroups are dealt round-robin to the 8 XCDs by their linear id, and gridDim.x is a multiple of 8 (host), so
  // blockIdx.x & 7 is the XCD for every depth plane -- XCD k takes the k-th eighth of the plane's pixel blocks at every depth, and
  // its L2 only ever holds that eighth of the source map (+ the sweep's shift).  At configs[4] the map is 15 MB against 4 MB of L2 per
  // XCD: the cost-volume warp 2.8 -> 3.9 TB/s (1.39 -> 0.99 ms), the photo-volume warp 0.91 -> 0.45 ms; at configs[2] (2.6 MB) nothing changes.  Which workgroup writes which pixels does not change a bit.
  const long pix0 = (long)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) * 256;
  {
    const long pix = pix0 + tid;
    Tap4 t;
    t.i00 = t.i01 = t.i10 = t.i11 = 0;
    t.wa = t.wb = t.wc = t.wd = t.valid = 0.f;
    if (pix < npix) {
      float Hm[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) Hm[i] = Hmats[d * 9 + i];
      const int y = (int)(pix / w), x = (int)(pix % w);
      float xw, yw;
      homography_apply(Hm, x, y, &xw, &yw);
      t = bilinear_taps(xw, yw, h, w);
      if (mask_out) mask_out[(size_t)d * npix + pix] = t.valid;
    }
    float4* g = reinterpret_cast<float4*>(s_geo + tid * 12);
    // the taps as BYTE offsets of the pixels in src (32 bits: checked by the host): the channel-group lanes of phase 2 then add
    // their own constant and gather through a buffer descriptor -- one vector instruction per gather address instead of seven
    const int pixb = C * 4;
    g[0] = make_float4(__int_as_float(t.i00 * pixb), __int_as_float(t.i01 * pixb), __int_as_float(t.i10 * pixb), __int_as_float(t.i11 * pixb));
    g[1] = make_float4(t.wa, t.wb, t.wc, t.wd);
    g[2] = make_float4(t.valid, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  const int cg = C >> 2;                 // lanes per pixel: 4, 8 or 16 (a divisor of 256)
  const int ppp = 256 / cg;              // pixels per pass
  // channel-last: the cg lanes of a pixel are neighbours (one pixel = C*4 contiguous bytes; cg is a power of two).
  // Chunk-planar: the same pixels per wavefront (all their channel groups: the gathers stay whole 128-byte rows), but lane
  // order (pixel / 4, chunk, pixel % 4, half) so that eight neighbouring lanes write one whole 128-byte line of a chunk plane
  const int cgs = __builtin_ctz(cg);
  int lp = tid >> cgs, c = (tid & (cg - 1)) * 4;
  if (plane_stride > 0) {
    const int i = tid & (4 * cg - 1);               // index inside a group of four pixels
    lp = ((tid >> (cgs + 2)) << 2) | ((i >> 1) & 3);
    c = (((i >> 3) << 1) | (i & 1)) * 4;
  }
  const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)(npix * C * 4), 0x00020000);
  const int cb = c * 4;                  // this lane's channel group, in bytes
  for (int pass = 0; pass < cg; ++pass) {
    const int pl = pass * ppp + lp;
    const long pix = pix0 + pl;
    if (pix >= npix) break;              // pixels ascend with the pass
    const float4* g = reinterpret_cast<const float4*>(s_geo + pl * 12);
    const float4 gi = g[0], gw = g[1];
    const geo_f32x4 ga = atvs_buffer_load_x4(srs, __float_as_int(gi.x) + cb, 0, 0), gb = atvs_buffer_load_x4(srs, __float_as_int(gi.y) + cb, 0, 0);
    const geo_f32x4 gc = atvs_buffer_load_x4(srs, __float_as_int(gi.z) + cb, 0, 0), gd = atvs_buffer_load_x4(srs, __float_as_int(gi.w) + cb, 0, 0);
    const float4 a = make_float4(ga[0], ga[1], ga[2], ga[3]), b = make_float4(gb[0], gb[1], gb[2], gb[3]);
    const float4 cc = make_float4(gc[0], gc[1], gc[2], gc[3]), dd = make_float4(gd[0], gd[1], gd[2], gd[3]);
    // ((wa a + wb b) + wc c) + wd d per component, two components per instruction
    // scalar arithmetic on purpose (the same IEEE operations in the same order as blend4): packed fp32 instructions gave wrong
    // lane quarters beside another kernel's wavefronts on the SIMD
    float4 o;
    o.x = ((gw.x * a.x + gw.y * b.x) + gw.z * cc.x) + gw.w * dd.x;
    o.y = ((gw.x * a.y + gw.y * b.y) + gw.z * cc.y) + gw.w * dd.y;
    o.z = ((gw.x * a.z + gw.y * b.z) + gw.z * cc.z) + gw.w * dd.z;
    o.w = ((gw.x * a.w + gw.y * b.w) + gw.z * cc.w) + gw.w * dd.w;
    if (MODE == 1) {
      const float valid = g[2].x;
      const float4 r = ld4(ref + (size_t)pix * C + c);
      o.x = fabsf(o.x - r.x) * valid;
      o.y = fabsf(o.y - r.y) * valid;
      o.z = fabsf(o.z - r.z) * valid;
      o.w = fabsf(o.w - r.w) * valid;
    }
    if (PIECES) {
      // The lane pair (tid, tid ^ 1) holds channels 0..3 | 4..7 of one voxel's chunk.  After the split the even lane keeps its h0
      // and takes the partner's, the odd lane keeps its h1 and takes the partner's (two DPP moves): each lane stores ONE 16-byte
      // record -- the voxel's eight h0 into the chunk's first piece plane, its eight h1 into the second -- so a store
      // instruction still writes whole lines (32 consecutive voxels per piece plane).
      unsigned h0a, h1a, h0b, h1b;
      atvs_split2_f16(o.x, o.y, 2048.f, &h0a, &h1a);
      atvs_split2_f16(o.z, o.w, 2048.f, &h0b, &h1b);
      const bool hi = (c & 4) != 0;                   // this lane holds channels 4..7 (it is the odd lane of its pair)
      const unsigned s0 = hi ? h0a : h1a, s1 = hi ? h0b : h1b;
      const unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s0, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
      const unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s1, 0xB1, 0xF, 0xF, true);
      uint4 rec;
      if (hi) rec = make_uint4(r0, r1, h1a, h1b);     // h1 of channels 0..3 (partner's) | 4..7 (own)
      else rec = make_uint4(h0a, h0b, r0, r1);        // h0 of channels 0..3 (own) | 4..7 (partner's)
      unsigned char* dst = reinterpret_cast<unsigned char*>(out + (size_t)(c >> 3) * (size_t)plane_stride) +
                           (hi ? (size_t)piece_bytes : (size_t)0) + ((size_t)d * npix + pix) * 16;
      st4u_stream(dst, rec);
    } else if (plane_stride > 0) {
      // chunk-planar output [C/8][D][h][w][8] (the layout atvs_conv_xw_f32 reads with x_planar)
      st4(out + (size_t)(c >> 3) * (size_t)plane_stride + ((size_t)d * npix + pix) * 8 + (c & 7), o);
    } else {
      st4(out + ((size_t)d * npix + pix) * (size_t)ld + c_off + c, o);
    }
  }
}

extern "C" int atvs_warp_planes(const float* src, const float* homographies, const float* ref,
                                const float* depth_start, const float* depth_interval, float* out, float* mask_out,
                                int D, int h, int w, int C, int ld_out, int c_off, int mode, int rep, long planar,
                                int pieces, atvs_stream_t stream) {
  if (!src || !homographies || !out) return ATVS_ERR_NULL;
  if (pieces && !planar) return ATVS_ERR_ARG;
  if (planar && ((mode != 0 && mode != 1) || (C != 16 && C != 32 && C != 64) || ld_out != C || c_off != 0)) return ATVS_ERR_ARG;
  if (planar && planar < (long)D * h * w * 8) return ATVS_ERR_ARG;
  const long plane_stride = planar;       // floats between the 8-channel chunk planes (>= D*h*w*8; callers pad it)
  if (D <= 0 || h <= 0 || w <= 0 || C <= 0 || ld_out < C || c_off < 0) return ATVS_ERR_SHAPE;
  if (mode == 1 && !ref) return ATVS_ERR_NULL;
  if (mode == 2 && (C != 1 || !depth_start || !depth_interval || rep < 1 || c_off + rep > ld_out)) return ATVS_ERR_SHAPE;
  if (mode != 2 && c_off + C > ld_out) return ATVS_ERR_SHAPE;
  if (mode < 0 || mode > 3) return ATVS_ERR_ARG;
  bool vec = (C % 4 == 0) && (ld_out % 4 == 0) && (c_off % 4 == 0) && mode != 2;
  long lanes = (long)h * w * (mode == 2 ? 1 : (vec ? C / 4 : C));
  dim3 grid(cdiv(lanes, 256), D), block(256);
  hipStream_t s = as_stream(stream);
  if (vec && mode < 2 && (C == 16 || C == 32 || C == 64) && (double)h * w * C * 4.0 < 2147483648.0) {
    // geometry once per pixel, shared by its channel-group lanes (31-bit byte offsets into src)
    dim3 g2((cdiv((long)h * w, 256) + 7) / 8 * 8, D);      // a multiple of 8 workgroups per plane (XCD-aware dealing)
    const long piece_bytes = (long)D * h * w * 16;
#define SHARED(M, P)                                                                                                       \
  hipLaunchKernelGGL((warp_planes_shared_kernel<M, P>), g2, block, 0, s, src, homographies, ref, out, mask_out, h, w, C, ld_out, \
                     c_off, plane_stride, (P) ? piece_bytes : 0L)
    if (mode == 0) { if (pieces) SHARED(0, true); else SHARED(0, false); }
    else { if (pieces) SHARED(1, true); else SHARED(1, false); }
#undef SHARED
    ATVS_LAUNCH_CHECK();
    return ATVS_OK;
  }
  // the gather kernel below writes channel-last fp32 only: a chunk-planar / pieces request the shared kernel cannot take (a
  // source map of 2 GiB or more) is refused, never silently written in another layout
  if (planar || pieces) return ATVS_ERR_SHAPE;
#define LAUNCH(M, V)                                                                                            \
  hipLaunchKernelGGL((warp_planes_kernel<M, V>), grid, block, 0, s, src, homographies, ref, depth_start,        \
                     depth_interval, out, mask_out, D, h, w, C, ld_out, c_off, rep)
  if (mode == 0) { if (vec) LAUNCH(0, 4); else LAUNCH(0, 1); }
  else if (mode == 1) { if (vec) LAUNCH(1, 4); else LAUNCH(1, 1); }
  else if (mode == 2) LAUNCH(2, 1);
  else { if (vec) LAUNCH(3, 4); else LAUNCH(3, 1); }
#undef LAUNCH
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}

// ---------------------------------------------------------------------------
// Cost volume (model.py:157-200): out[d, y, x, :] = concat(ref[y, x, :], warp_d(view)[y, x, :]).
// One lane per float4 of the 2C-wide row; a wavefront writes 1 KiB contiguous.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cost_volume_kernel(const float* __restrict__ ref_f, const float* __restrict__ view_f,
                                                          const float* __restrict__ Hmats, float* __restrict__ out,
                                                          int D, int h, int w, int C) {
  const int d = blockIdx.y;
  const int cg = (2 * C) / 4;
  long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long npix = (long)h * w;
  if (gid >= npix * cg) return;
  int pix = (int)(gid / cg);
  int c = (int)(gid % cg) * 4;
  float4 o;
  if (c < C) {
    o = ld4(ref_f + (size_t)pix * C + c);
  } else {
    int cv = c - C;
    int y = pix / w, x = pix % w;
    float Hm[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) Hm[i] = Hmats[d * 9 + i];
    float xw, yw;
    homography_apply(Hm, x, y, &xw, &yw);
    Tap4 t = bilinear_taps(xw, yw, h, w);
    o = blend4(t, ld4(view_f + (size_t)t.i00 * C + cv), ld4(view_f + (size_t)t.i01 * C + cv),
               ld4(view_f + (size_t)t.i10 * C + cv), ld4(view_f + (size_t)t.i11 * C + cv));
  }
  st4(out + ((size_t)d * npix + pix) * (size_t)(2 * C) + c, o);
}

extern "C" int atvs_build_cost_volume(const float* ref_feature, const float* view_feature, const float* homographies,
                                      float* cost_volume, int D, int h, int w, int C, atvs_stream_t stream) {
  if (!ref_feature || !view_feature || !homographies || !cost_volume) return ATVS_ERR_NULL;
  if (D <= 0 || h <= 0 || w <= 0 || C <= 0 || (C % 4) != 0) return ATVS_ERR_SHAPE;
  long lanes = (long)h * w * (2 * C / 4);
  hipLaunchKernelGGL(cost_volume_kernel, dim3(cdiv(lanes, 256), D), dim3(256), 0, as_stream(stream), ref_feature,
                     view_feature, homographies, cost_volume, D, h, w, C);
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}

// ---------------------------------------------------------------------------
// Broadcast an (h, w, C) map along D into out (D, h, w, ld) at c_off
// (tf.tile call sites model.py:311,316,329-330).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tile_planes_kernel(const float* __restrict__ src, float* __restrict__ out, int D,
                                                          long npix, int C, int ld, int c_off) {
  const int d = blockIdx.y;
  long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= npix * C) return;
  long pix = gid / C;
  int c = (int)(gid % C);
  out[((size_t)d * npix + pix) * (size_t)ld + c_off + c] = src[gid];
}

extern "C" int atvs_tile_planes(const float* src, float* out, int D, int h, int w, int C, int ld_out, int c_off,
                                atvs_stream_t stream) {
  if (!src || !out) return ATVS_ERR_NULL;
  if (D <= 0 || h <= 0 || w <= 0 || C <= 0 || c_off < 0 || c_off + C > ld_out) return ATVS_ERR_SHAPE;
  long n = (long)h * w * C;
  hipLaunchKernelGGL(tile_planes_kernel, dim3(cdiv(n, 256), D), dim3(256), 0, as_stream(stream), src, out, D,
                     (long)h * w, C, ld_out, c_off);
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}

// ---------------------------------------------------------------------------
// geo reference volume (model.py:289-290): out[d,pix] = |depth_ref[pix] - delta_d| / interval / D
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void geo_ref_kernel(const float* __restrict__ depth_ref, const float* __restrict__ depth_start,
                                                      const float* __restrict__ depth_interval, float* __restrict__ out,
                                                      int D, long npix, int ld, int c_off) {
  const int d = blockIdx.y;
  long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= npix) return;
  float val = depth_start[0] + (float)d * depth_interval[0];
  out[((size_t)d * npix + pix) * (size_t)ld + c_off] = fabsf(depth_ref[pix] - val) / depth_interval[0] / (float)D;
}

extern "C" int atvs_geo_ref_planes(const float* depth_ref, const float* depth_start, const float* depth_interval,
                                   float* out, int D, int h, int w, int ld_out, int c_off, atvs_stream_t stream) {
  if (!depth_ref || !depth_start || !depth_interval || !out) return ATVS_ERR_NULL;
  if (D <= 0 || h <= 0 || w <= 0 || c_off < 0 || c_off >= ld_out) return ATVS_ERR_SHAPE;
  long npix = (long)h * w;
  hipLaunchKernelGGL(geo_ref_kernel, dim3(cdiv(npix, 256), D), dim3(256), 0, as_stream(stream), depth_ref,
                     depth_start, depth_interval, out, D, npix, ld_out, c_off);
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}

// The refinement's geo volume in one launch (model.py:285-300): out[d, pix, c_off] = geo_ref (geo_ref_kernel),
// out[d, pix, c_off + 1 .. + rep] = (|warp_d(view_depth) - delta_d| / interval / D) * mask (warp_planes_kernel<2, 1>) -- the same
// operations; with rep = 1 and an even c_off the two channels leave as one 8-byte store.
__global__ __launch_bounds__(256) void geo_volume_kernel(const float* __restrict__ depth_ref, const float* __restrict__ src,
                                                         const float* __restrict__ Hmats, const float* __restrict__ depth_start,
                                                         const float* __restrict__ depth_interval, float* __restrict__ out, int D,
                                                         int h, int w, int ld, int c_off, int rep) {
  const int d = blockIdx.y;
  const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long npix = (long)h * w;
  if (pix >= npix) return;
  const int y = (int)(pix / w), x = (int)(pix % w);
  float Hm[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) Hm[i] = Hmats[d * 9 + i];
  float xw, yw;
  homography_apply(Hm, x, y, &xw, &yw);
  const Tap4 t = bilinear_taps(xw, yw, h, w);
  const float v = ((t.wa * src[t.i00] + t.wb * src[t.i01]) + t.wc * src[t.i10]) + t.wd * src[t.i11];
  const float val = depth_start[0] + (float)d * depth_interval[0];
  const float g = (fabsf(v - val) / depth_interval[0] / (float)D) * t.valid;
  const float gr = fabsf(depth_ref[pix] - val) / depth_interval[0] / (float)D;
  float* o = out + ((size_t)d * npix + pix) * (size_t)ld + c_off;
  if (rep == 1 && ((ld | c_off) & 1) == 0) {
    *reinterpret_cast<float2*>(o) = make_float2(gr, g);
  } else {
    o[0] = gr;
    for (int r = 0; r < rep; ++r) o[1 + r] = g;
  }
}

extern "C" int atvs_geo_volume(const float* depth_ref, const float* view_depth, const float* homographies,
                               const float* depth_start, const float* depth_interval, float* out, int D, int h, int w,
                               int ld_out, int c_off, int rep, atvs_stream_t stream) {
  if (!depth_ref || !view_depth || !homographies || !depth_start || !depth_interval || !out) return ATVS_ERR_NULL;
  if (D <= 0 || h <= 0 || w <= 0 || rep < 1 || c_off < 0 || c_off + 1 + rep > ld_out) return ATVS_ERR_SHAPE;
  const long npix = (long)h * w;
  hipLaunchKernelGGL(geo_volume_kernel, dim3(cdiv(npix, 256), D), dim3(256), 0, as_stream(stream), depth_ref, view_depth,
                     homographies, depth_start, depth_interval, out, D, h, w, ld_out, c_off, rep);
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}

// ---------------------------------------------------------------------------
// Visual hull for two depth maps (homography_warping.py:329-387 with view_num = 2,
// the only configuration the path uses: model.py:323-324 via :436 / :373):
//   hull[d] = ([ref>0][ref>delta_d] + [wd>0][wd>delta_d]) / 2,  wd = nearest-warp_d(view_depth_in_ref)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void visual_hull_kernel(const float* __restrict__ ref_depth, const float* __restrict__ view_depth_trans,
                                                          const float* __restrict__ Hmats, const float* __restrict__ depth_start,
                                                          const float* __restrict__ depth_interval, float* __restrict__ out,
                                                          int D, int h, int w, int inverse_depth, float view_num) {
  const int d = blockIdx.y;
  long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long npix = (long)h * w;
  if (pix >= npix) return;
  int y = (int)(pix / w), x = (int)(pix % w);
  float cur = depth_start[0] + depth_interval[0] * (float)d;
  float rd = ref_depth[pix];
  float s = ((rd > 0.f) ? 1.f : 0.f) * (inverse_depth ? ((rd > cur) ? 1.f : 0.f) : ((cur > rd) ? 1.f : 0.f));
  float Hm[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) Hm[i] = Hmats[d * 9 + i];
  float xw, yw, valid;
  homography_apply(Hm, x, y, &xw, &yw);
  int idx = nearest_tap(xw, yw, h, w, &valid);
  float wd = view_depth_trans[idx];
  s = s + ((wd > 0.f) ? 1.f : 0.f) * (inverse_depth ? ((wd > cur) ? 1.f : 0.f) : ((cur > wd) ? 1.f : 0.f));
  out[(size_t)d * npix + pix] = s / view_num;
}

extern "C" int atvs_visual_hull(const float* ref_depth, const float* view_depth_in_ref, const float* homographies,
                                const float* depth_start, const float* depth_interval, float* out, int D, int h,
                                int w, int inverse_depth, atvs_stream_t stream) {
  if (!ref_depth || !view_depth_in_ref || !homographies || !depth_start || !depth_interval || !out) return ATVS_ERR_NULL;
  if (D <= 0 || h <= 0 || w <= 0) return ATVS_ERR_SHAPE;
  long npix = (long)h * w;
  hipLaunchKernelGGL(visual_hull_kernel, dim3(cdiv(npix, 256), D), dim3(256), 0, as_stream(stream), ref_depth,
                     view_depth_in_ref, homographies, depth_start, depth_interval, out, D, h, w, inverse_depth, 2.0f);
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}

// ---------------------------------------------------------------------------
// Relative pose  mat = K_r R_r R_l^T K_l^-1,  vec = K_r R_r c_l + K_r t_r
// (homography_warping.py:123-146, 290-313), recomputed per thread: 12 floats.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void relative_pose(const float* left_cam, const float* right_cam, M3* mat, float* vec) {
  M3 Rl, Kl, Rr, Kr;
  float tl[3], tr[3];
  split_cam(left_cam, &Rl, tl, &Kl);
  split_cam(right_cam, &Rr, tr, &Kr);
  M3 Kli = inv3(Kl);
  M3 RlT = transpose3(Rl);
  float cl[3];
  mv3(RlT, tl, cl);
  for (int i = 0; i < 3; ++i) cl[i] = -cl[i];
  *mat = mm3(Kr, mm3(Rr, mm3(RlT, Kli)));
  float v0[3], v1[3], v2[3];
  mv3(Rr, cl, v0);
  mv3(Kr, v0, v1);
  mv3(Kr, tr, v2);
  for (int i = 0; i < 3; ++i) vec[i] = v1[i] + v2[i];
}

// pose (12 floats: mat row-major, vec) computed by thread 0 of every workgroup into LDS (round 5: it was a 1-thread launch of its
// own in front of every consumer, 15 launches of 4.6 us per depth map; the same function, the same values)
__device__ __forceinline__ void workgroup_pose(const float* left_cam, const float* right_cam, float* pose_s) {
  if (threadIdx.x == 0) {
    M3 mat;
    float vec[3];
    relative_pose(left_cam, right_cam, &mat, vec);
    for (int i = 0; i < 9; ++i) pose_s[i] = mat.m[i];
    for (int i = 0; i < 3; ++i) pose_s[9 + i] = vec[i];
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------
// Per-pixel-depth warp (homography_warping.py:108-176): p' ~ M p + v * delta(p).
// method 0 bilinear / 1 nearest; mask written when mask_out != nullptr.
// ---------------------------------------------------------------------------
// ERR: out[pix, c_off + c] (rows of ld floats) = |warp - ref[pix, c]| * valid -- the warp, tf.abs(. - ref) * mask and the copy into
// the tiled-channel buffer of the refinement's photo_err / geo_err (model.py:309-316) as one launch
template <int NEAREST, bool ERR = false>
__global__ __launch_bounds__(256) void warp_by_depth_kernel(const float* __restrict__ src, const float* __restrict__ left_cam,
                                                            const float* __restrict__ right_cam,
                                                            const float* __restrict__ depth, float* __restrict__ out,
                                                            float* __restrict__ mask_out, int h, int w, int C,
                                                            int inverse_depth, const float* __restrict__ ref = nullptr,
                                                            int ld = 0, int c_off = 0, int copy_ref = 0) {
  __shared__ float pose[12];
  workgroup_pose(left_cam, right_cam, pose);
  long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long npix = (long)h * w;
  if (gid >= npix * C) return;
  int pix = (int)(gid / C), c = (int)(gid % C);
  int y = pix / w, x = pix % w;
  float px = (float)x + 0.5f, py = (float)y + 0.5f;
  float dd = depth[pix];
  float r[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float v = inverse_depth ? (pose[9 + i] * dd) : (pose[9 + i] / dd);
    r[i] = ((pose[i * 3 + 0] * px + pose[i * 3 + 1] * py) + pose[i * 3 + 2]) + v;
  }
  float xw = r[0] / r[2], yw = r[1] / r[2];
  float o, valid;
  if (NEAREST) {
    int idx = nearest_tap(xw, yw, h, w, &valid);
    o = src[(size_t)idx * C + c];
  } else {
    Tap4 t = bilinear_taps(xw, yw, h, w);
    valid = t.valid;
    o = ((t.wa * src[(size_t)t.i00 * C + c] + t.wb * src[(size_t)t.i01 * C + c]) + t.wc * src[(size_t)t.i10 * C + c]) +
        t.wd * src[(size_t)t.i11 * C + c];
  }
  if (ERR) {
    const float rv = ref[gid];
    out[(size_t)pix * ld + c_off + c] = fabsf(o - rv) * valid;
    if (copy_ref) out[(size_t)pix * ld + c_off + C + c] = rv;      // the tiled reference map that follows the error map in the buffer
  } else {
    out[gid] = o;
  }
  if (mask_out && c == 0) mask_out[pix] = valid;
}

extern "C" int atvs_warp_by_depth(const float* src, const float* left_cam, const float* right_cam, const float* depth,
                                  float* out, float* mask_out, float* pose_ws, int h, int w, int C, int method,
                                  int inverse_depth, atvs_stream_t stream) {
  if (!src || !left_cam || !right_cam || !depth || !out || !pose_ws) return ATVS_ERR_NULL;
  if (h <= 0 || w <= 0 || C <= 0) return ATVS_ERR_SHAPE;
  hipStream_t s = as_stream(stream);
  long n = (long)h * w * C;
  if (method == 0)
    hipLaunchKernelGGL((warp_by_depth_kernel<0>), dim3(cdiv(n, 256)), dim3(256), 0, s, src, left_cam, right_cam, depth, out, mask_out, h, w, C, inverse_depth);
  else if (method == 1)
    hipLaunchKernelGGL((warp_by_depth_kernel<1>), dim3(cdiv(n, 256)), dim3(256), 0, s, src, left_cam, right_cam, depth, out, mask_out, h, w, C, inverse_depth);
  else
    return ATVS_ERR_ARG;
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}

// |homography_warping_by_depth(src) - ref| * mask written into channels [c_off, c_off + C) of out (h, w, ld): photo_err / geo_err
// of the refinement (model.py:309-316: the warp, tf.abs(warped - ref) * mask) straight into the tiled-channel buffer; copy_ref: ref
// itself goes to the C channels behind (the tiled reference features / depth that follow the error map, model.py:329-334).  The same
// operations in the same order as atvs_warp_by_depth followed by atvs_absdiff_mask.
extern "C" int atvs_warp_by_depth_err(const float* src, const float* ref, const float* left_cam, const float* right_cam,
                                      const float* depth, float* out, int ld_out, int c_off, int h, int w, int C, int method,
                                      int inverse_depth, int copy_ref, atvs_stream_t stream) {
  if (!src || !ref || !left_cam || !right_cam || !depth || !out) return ATVS_ERR_NULL;
  if (h <= 0 || w <= 0 || C <= 0 || c_off < 0 || c_off + (copy_ref ? 2 : 1) * C > ld_out) return ATVS_ERR_SHAPE;
  hipStream_t s = as_stream(stream);
  long n = (long)h * w * C;
  if (method == 0)
    hipLaunchKernelGGL((warp_by_depth_kernel<0, true>), dim3(cdiv(n, 256)), dim3(256), 0, s, src, left_cam, right_cam, depth, out,
                       (float*)nullptr, h, w, C, inverse_depth, ref, ld_out, c_off, copy_ref);
  else if (method == 1)
    hipLaunchKernelGGL((warp_by_depth_kernel<1, true>), dim3(cdiv(n, 256)), dim3(256), 0, s, src, left_cam, right_cam, depth, out,
                       (float*)nullptr, h, w, C, inverse_depth, ref, ld_out, c_off, copy_ref);
  else
    return ATVS_ERR_ARG;
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}

// ---------------------------------------------------------------------------
// interpolate (homography_warping.py:31-104) with CALLER-supplied sampling coordinates: the function the warps above
// have folded into them, exported for callers that bring their own x / y (B = 1: n = h*w points in the reference,
// any n here).  get_pixel_grids (homography_warping.py:8-17): [x + 0.5 | y + 0.5 | 1], each h*w long, concatenated;
// tf.linspace = start + i * step with step = (stop - start) / (num - 1) in fp32.
// ---------------------------------------------------------------------------
template <int NEAREST>
__global__ __launch_bounds__(256) void interpolate_kernel(const float* __restrict__ src, const float* __restrict__ xs,
                                                          const float* __restrict__ ys, float* __restrict__ out,
                                                          float* __restrict__ mask_out, long n, int h, int w, int C) {
  long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n * C) return;
  long pt = gid / C;
  int c = (int)(gid % C);
  float xw = xs[pt], yw = ys[pt];
  float o, valid;
  if (NEAREST) {
    int idx = nearest_tap(xw, yw, h, w, &valid);
    if (xw != xw || yw != yw) { idx = 0; valid = 0.f; }   // comparisons with NaN are false already; kept explicit (:42-43)
    o = src[(size_t)idx * C + c];
  } else {
    Tap4 t = bilinear_taps(xw, yw, h, w);
    valid = t.valid;
    o = ((t.wa * src[(size_t)t.i00 * C + c] + t.wb * src[(size_t)t.i01 * C + c]) + t.wc * src[(size_t)t.i10 * C + c]) +
        t.wd * src[(size_t)t.i11 * C + c];
  }
  out[gid] = o;
  if (mask_out && c == 0) mask_out[pt] = valid;
}

extern "C" int atvs_interpolate(const float* src, const float* x, const float* y, float* out, float* mask_out, long n,
                                int h, int w, int C, int method, atvs_stream_t stream) {
  if (!src || !x || !y || !out) return ATVS_ERR_NULL;
  if (h <= 0 || w <= 0 || C <= 0 || n < 0) return ATVS_ERR_SHAPE;
  if (method != 0 && method != 1) return ATVS_ERR_ARG;
  if (n == 0) return ATVS_OK;
  hipStream_t s = as_stream(stream);
  long tot = n * C;
  if (method == 0)
    hipLaunchKernelGGL((interpolate_kernel<0>), dim3(cdiv(tot, 256)), dim3(256), 0, s, src, x, y, out, mask_out, n, h, w, C);
  else
    hipLaunchKernelGGL((interpolate_kernel<1>), dim3(cdiv(tot, 256)), dim3(256), 0, s, src, x, y, out, mask_out, n, h, w, C);
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}

__global__ __launch_bounds__(256) void pixel_grids_kernel(float* __restrict__ out, int h, int w) {
  long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long npix = (long)h * w;
  if (gid >= npix) return;
  int y = (int)(gid / w), x = (int)(gid % w);
  // tf.linspace(0.5, n - 0.5, n): step = ((n - 0.5) - 0.5) / (n - 1)
  float sx = (w > 1) ? (((float)w - 0.5f) - 0.5f) / (float)(w - 1) : 0.f;
  float sy = (h > 1) ? (((float)h - 0.5f) - 0.5f) / (float)(h - 1) : 0.f;
  out[gid] = 0.5f + sx * (float)x;
  out[npix + gid] = 0.5f + sy * (float)y;
  out[2 * npix + gid] = 1.f;
}

extern "C" int atvs_pixel_grids(float* out, int h, int w, atvs_stream_t stream) {
  if (!out) return ATVS_ERR_NULL;
  if (h <= 0 || w <= 0) return ATVS_ERR_SHAPE;
  hipLaunchKernelGGL(pixel_grids_kernel, dim3(cdiv((long)h * w, 256)), dim3(256), 0, as_stream(stream), out, h, w);
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}

// ---------------------------------------------------------------------------
// transform_depth (homography_warping.py:275-326): two global maxima (quirk C15)
// -> three tiny passes.  ws: 2 floats (max of input, max of transformed z).
// ---------------------------------------------------------------------------
__device__ __forceinline__ float block_max_256(float v) {
  __shared__ float sm[4];
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  v = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
  __syncthreads();
  return v;
}

// fmaxf drops NaNs, like tf.reduce_max on finite data; order-independent, so atomics are exact.
__device__ __forceinline__ void atomic_max_float(float* addr, float v) {
  // monotone int mapping of finite floats
  if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

__global__ void fill_neg_inf_kernel(float* ws) {
  ws[0] = -INFINITY;
  ws[1] = -INFINITY;
}

__global__ __launch_bounds__(256) void max_kernel(const float* __restrict__ x, long n, float* __restrict__ ws) {
  float v = -INFINITY;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) v = fmaxf(v, x[i]);
  v = block_max_256(v);
  if (threadIdx.x == 0) atomic_max_float(ws, v);
}

template <int PASS>
__global__ __launch_bounds__(256) void transform_depth_kernel(const float* __restrict__ depth, const float* __restrict__ left_cam,
                                                              const float* __restrict__ right_cam,
                                                              float* __restrict__ out, float* __restrict__ ws, int h, int w,
                                                              int inverse_depth) {
  __shared__ float pose[12];
  workgroup_pose(left_cam, right_cam, pose);
  long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long npix = (long)h * w;
  float z = -INFINITY;
  bool in = pix < npix;
  float valid = 0.f;
  if (in) {
    float d = depth[pix];
    if (inverse_depth) {
      valid = (d > 1e-10f) ? 1.f : 0.f;
      d = fminf(fmaxf(d, 1e-10f), ws[0]);
      d = 1.0f / d;
      d = d * valid;
    }
    int y = (int)(pix / w), x = (int)(pix % w);
    float gx = ((float)x + 0.5f) * d, gy = ((float)y + 0.5f) * d;
    z = ((pose[6] * gx + pose[7] * gy) + pose[8] * d) + pose[11];
  }
  if (PASS == 0) {
    float m = block_max_256(z);
    if (threadIdx.x == 0) atomic_max_float(ws + 1, m);
  } else if (in) {
    if (inverse_depth) {
      z = fminf(fmaxf(z, 1e-10f), ws[1]);
      z = 1.0f / z;
      z = z * valid;
    }
    out[pix] = z;
  }
}

// The whole of transform_depth for a small map (<= 32 pixels per thread of ONE workgroup of 1,024): max of the input, the
// transformed z and its max, the clip -- the four launches of the general form as one (round 5: a dependent launch costs ~4.7 us
// of dispatch latency in a replayed graph, the map of configs[2] has 20,480 pixels).  The same operations per pixel; a maximum
// is order-independent: identical bits.
__device__ __forceinline__ float block_max_1024(float v, float* sm) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  float m = sm[0];
  for (int i = 1; i < 16; ++i) m = fmaxf(m, sm[i]);
  return m;
}

// A thread keeps its (up to 32) pixels in registers: the loads are issued back to back (ONE memory round trip instead of one per
// pixel and pass: 26 -> .. us at 20,480 pixels), the transformed depths are computed once.
// Several maps per launch (workgroup = map): the refinement transforms the depth map of every source view into the reference camera
// and into the hull camera (model.py:289,321-324: 7 maps of configs[2]) -- one launch instead of one per map.
struct TdJobs {
  const float* depth[16];
  const float* left_cam[16];
  const float* right_cam[16];
  float* out[16];
};
__global__ __launch_bounds__(1024) void transform_depth_small_kernel(TdJobs jobs, int h, int w, int inverse_depth) {
  const float* __restrict__ depth = jobs.depth[blockIdx.x];
  const float* __restrict__ left_cam = jobs.left_cam[blockIdx.x];
  const float* __restrict__ right_cam = jobs.right_cam[blockIdx.x];
  float* __restrict__ out = jobs.out[blockIdx.x];
  constexpr int ITEMS = 32;                  // pixels per thread: maps up to 32,768 pixels
  __shared__ float pose[12];
  __shared__ float sm[16];
  const int npix = h * w;
  float d[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int i = threadIdx.x + k * 1024;
    d[k] = (i < npix) ? depth[i] : -INFINITY;
  }
  workgroup_pose(left_cam, right_cam, pose);      // (thread 0's serial arithmetic: under the loads' round trip)
  float v = -INFINITY;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) v = fmaxf(v, d[k]);
  const float dmax = block_max_1024(v, sm);
  unsigned validm = 0;                       // bit k: pixel k of this thread has a positive depth
  // pixel threadIdx.x + 1024 k = (y, x): by carries, no division per pixel
  const int sy = 1024 / w, sx = 1024 % w;
  int y = (int)threadIdx.x / w, x = (int)threadIdx.x % w;
  float zm = -INFINITY;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int pix = threadIdx.x + k * 1024;
    if (pix < npix) {                        // d[k] <- the depth of the pixel's point in the right camera
      float dd = d[k];
      if (inverse_depth) {
        const bool ok = dd > 1e-10f;
        validm |= ok ? (1u << k) : 0u;
        dd = fminf(fmaxf(dd, 1e-10f), dmax);
        dd = 1.0f / dd;
        dd = dd * (ok ? 1.f : 0.f);
      }
      float gx = ((float)x + 0.5f) * dd;
      asm volatile("" : "+v"(gx));           // keeps the two products apart: no v_pk_mul_f32 here (Appendix B; tests/test_packed_fp32_census.py)
      const float gy = ((float)y + 0.5f) * dd;
      const float z = ((pose[6] * gx + pose[7] * gy) + pose[8] * dd) + pose[11];
      d[k] = z;
      zm = fmaxf(zm, z);
    }
    x += sx; y += sy;
    if (x >= w) { x -= w; ++y; }
  }
  const float zmax = block_max_1024(zm, sm);
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int pix = threadIdx.x + k * 1024;
    if (pix < npix) {
      float z = d[k];
      if (inverse_depth) {
        z = fminf(fmaxf(z, 1e-10f), zmax);
        z = 1.0f / z;
        z = z * (((validm >> k) & 1u) ? 1.f : 0.f);
      }
      out[pix] = z;
    }
  }
}

extern "C" int atvs_transform_depth(const float* depth, const float* left_cam, const float* right_cam, float* out,
                                    float* ws14, int h, int w, int inverse_depth, atvs_stream_t stream) {
  if (!depth || !left_cam || !right_cam || !out || !ws14) return ATVS_ERR_NULL;
  if (h <= 0 || w <= 0) return ATVS_ERR_SHAPE;
  hipStream_t s = as_stream(stream);
  long npix = (long)h * w;
  if (npix <= 32 * 1024) {
    TdJobs jobs;
    for (int i = 0; i < 16; ++i) { jobs.depth[i] = depth; jobs.left_cam[i] = left_cam; jobs.right_cam[i] = right_cam; jobs.out[i] = out; }
    hipLaunchKernelGGL(transform_depth_small_kernel, dim3(1), dim3(1024), 0, s, jobs, h, w, inverse_depth);
    ATVS_LAUNCH_CHECK();
    return ATVS_OK;
  }
  hipLaunchKernelGGL(fill_neg_inf_kernel, dim3(1), dim3(1), 0, s, ws14);
  hipLaunchKernelGGL(max_kernel, dim3(min(cdiv(npix, 256), 1024)), dim3(256), 0, s, depth, npix, ws14);
  hipLaunchKernelGGL((transform_depth_kernel<0>), dim3(cdiv(npix, 256)), dim3(256), 0, s, depth, left_cam, right_cam, out, ws14, h, w, inverse_depth);
  hipLaunchKernelGGL((transform_depth_kernel<1>), dim3(cdiv(npix, 256)), dim3(256), 0, s, depth, left_cam, right_cam, out, ws14, h, w, inverse_depth);
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}

// n <= 16 maps of one size in ONE launch where a map fits a workgroup (h * w <= 32,768: atvs_transform_depth_batch_supported), the
// values of n calls of atvs_transform_depth.  depth / left_cam / right_cam / out: HOST arrays of n device pointers.
extern "C" int atvs_transform_depth_batch_supported(int h, int w) { return (h > 0 && w > 0 && (long)h * w <= 32 * 1024) ? 1 : 0; }

extern "C" int atvs_transform_depth_batch(const float* const* depth, const float* const* left_cam, const float* const* right_cam,
                                          float* const* out, int n, int h, int w, int inverse_depth, atvs_stream_t stream) {
  if (!depth || !left_cam || !right_cam || !out) return ATVS_ERR_NULL;
  if (n <= 0 || n > 16 || !atvs_transform_depth_batch_supported(h, w)) return ATVS_ERR_SHAPE;
  TdJobs jobs;
  for (int i = 0; i < 16; ++i) {
    const int k = i < n ? i : 0;
    if (!depth[k] || !left_cam[k] || !right_cam[k] || !out[k]) return ATVS_ERR_NULL;
    jobs.depth[i] = depth[k]; jobs.left_cam[i] = left_cam[k]; jobs.right_cam[i] = right_cam[k]; jobs.out[i] = out[k];
  }
  hipLaunchKernelGGL(transform_depth_small_kernel, dim3(n), dim3(1024), 0, as_stream(stream), jobs, h, w, inverse_depth);
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}

// ---------------------------------------------------------------------------
// |a - b| * mask[pixel]  (photo_err / geo_err, model.py:310,315)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void absdiff_mask_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           const float* __restrict__ mask, float* __restrict__ out,
                                                           long n, int C) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = fabsf(a[i] - b[i]) * mask[i / C];
}

extern "C" int atvs_absdiff_mask(const float* a, const float* b, const float* mask, float* out, int npix, int C,
                                 atvs_stream_t stream) {
  if (!a || !b || !mask || !out) return ATVS_ERR_NULL;
  if (npix <= 0 || C <= 0) return ATVS_ERR_SHAPE;
  long n = (long)npix * C;
  hipLaunchKernelGGL(absdiff_mask_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), a, b, mask, out, n, C);
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}
