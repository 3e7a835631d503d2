// Depth-map fusion by cross-view consistency voting (gfx950): the `fusibile` kernel of the reference's point-cloud
// stage, /root/reference/fusibile/fusibile.cu:138-277 (called per reference camera from :422-427; cameras from
// cameraGeometryUtils.h:194-499, textures from main.cpp:459-498).
//
// One thread per pixel of the reference camera: back-project the pixel with its depth, project the 3-D point into
// every other view, sample that view's (normal, depth) map and colour bilinearly, accept the view when the relative
// disparity difference and the normal angle are under their thresholds, average normals and colours over the
// accepted views, and create the point when enough views agree.  HBM / L2-bound gathers (N - 1 bilinear float4
// fetches x 2 per pixel); no matrix work.
//
// Arithmetic contract (oracle/fusibile.py, bit for bit; this file is built with -ffp-contract=off): float32, left to
// right; the CUDA texture fetch tex2D<float4>(t, x + 0.5, y + 0.5) of an un-normalised linear texture is restated as
// the bilinear blend of texels floor(x), floor(x) + 1 (clamped to the image) with weights rounded to 8 fractional
// bits -- plain loads, no texture unit.
#include "common.h"

namespace {

struct Cam {      // 28 floats per camera: P[12] | M_inv[9] | C[3] | P_col34[3] | f
  const float* p;
  __device__ __forceinline__ float P(int i) const { return p[i]; }
  __device__ __forceinline__ float Mi(int i) const { return p[12 + i]; }
  __device__ __forceinline__ float C(int i) const { return p[21 + i]; }
  __device__ __forceinline__ float pc(int i) const { return p[24 + i]; }
  __device__ __forceinline__ float f() const { return p[27]; }
};

__device__ __forceinline__ float4 tex_fetch(const float4* __restrict__ tex, float x, float y, int rows, int cols) {
  const float xf = floorf(x), yf = floorf(y);
  const float ax = floorf((x - xf) * 256.0f + 0.5f) / 256.0f;
  const float ay = floorf((y - yf) * 256.0f + 0.5f) / 256.0f;
  const int xi = (int)xf, yi = (int)yf;
  const int x0 = min(max(xi, 0), cols - 1), x1 = min(max(xi + 1, 0), cols - 1);
  const int y0 = min(max(yi, 0), rows - 1), y1 = min(max(yi + 1, 0), rows - 1);
  const float4 a = tex[(size_t)y0 * cols + x0], b = tex[(size_t)y0 * cols + x1];
  const float4 c = tex[(size_t)y1 * cols + x0], d = tex[(size_t)y1 * cols + x1];
  const float bx = 1.0f - ax, by = 1.0f - ay;
  float4 top, bot, o;
  top.x = bx * a.x + ax * b.x; top.y = bx * a.y + ax * b.y; top.z = bx * a.z + ax * b.z; top.w = bx * a.w + ax * b.w;
  bot.x = bx * c.x + ax * d.x; bot.y = bx * c.y + ax * d.y; bot.z = bx * c.z + ax * d.z; bot.w = bx * c.w + ax * d.w;
  o.x = by * top.x + ay * bot.x; o.y = by * top.y + ay * bot.y; o.z = by * top.z + ay * bot.z; o.w = by * top.w + ay * bot.w;
  return o;
}

__global__ __launch_bounds__(256) void fusibile_kernel(const float* __restrict__ cams, const float4* __restrict__ nd,
                                                       const float4* __restrict__ img, int nviews, int ref, int rows,
                                                       int cols, float disp_thresh, float normal_thresh, int num_consistent,
                                                       float4* __restrict__ coord, float4* __restrict__ normal_out,
                                                       float4* __restrict__ tex_out, float* __restrict__ created) {
  const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
  if (x >= cols || y >= rows) return;
  const size_t center = (size_t)y * cols + x, plane = (size_t)rows * cols;
  const Cam cr = {cams + (size_t)ref * 28};
  const float4 nrm = nd[(size_t)ref * plane + center];
  const float depth = nrm.w;
  // get3Dpoint_cu (:53-62)
  const float ptx = depth * (float)x - cr.pc(0), pty = depth * (float)y - cr.pc(1), ptz = depth - cr.pc(2);
  const float Xx = cr.Mi(0) * ptx + cr.Mi(1) * pty + cr.Mi(2) * ptz;
  const float Xy = cr.Mi(3) * ptx + cr.Mi(4) * pty + cr.Mi(5) * ptz;
  const float Xz = cr.Mi(6) * ptx + cr.Mi(7) * pty + cr.Mi(8) * ptz;
  float4 cn = nrm;
  float4 ct = img[(size_t)ref * plane + center];
  int count = 0;
  for (int i = 0; i < nviews; ++i) {
    if (i == ref) continue;
    const Cam c = {cams + (size_t)i * 28};
    // project_on_camera (:127-133)
    const float tx = c.P(0) * Xx + c.P(1) * Xy + c.P(2) * Xz + c.P(3);
    const float ty = c.P(4) * Xx + c.P(5) * Xy + c.P(6) * Xz + c.P(7);
    const float tz = c.P(8) * Xx + c.P(9) * Xy + c.P(10) * Xz + c.P(11);
    const float px = tx / tz, py = ty / tz, d = tz;
    if (!(px >= 0.f && px < (float)cols && py >= 0.f && py < (float)rows)) continue;
    const float4 other = tex_fetch(nd + (size_t)i * plane, px, py, rows, cols);
    const float dx = cr.C(0) - c.C(0), dy = cr.C(1) - c.C(1), dz = cr.C(2) - c.C(2);
    const float base = sqrtf(dx * dx + dy * dy + dz * dz);
    const float fb = cr.f() * base;
    const float d_disp = fb / d, o_disp = fb / other.w;
    if (!((fabsf(d_disp - o_disp) / d_disp) < disp_thresh)) continue;
    const float dot = other.x * nrm.x + other.y * nrm.y + other.z * nrm.z;
    float ang = acosf(dot);
    if (ang != ang) ang = 0.f;
    if (!(ang < normal_thresh)) continue;
    const float4 t = tex_fetch(img + (size_t)i * plane, px, py, rows, cols);
    cn = make_float4(cn.x + other.x, cn.y + other.y, cn.z + other.z, 0.f);
    ct = make_float4(ct.x + t.x, ct.y + t.y, ct.z + t.z, 0.f);
    ++count;
  }
  const float k = (float)count + 1.0f;
  coord[center] = make_float4(Xx, Xy, Xz, 0.f);
  normal_out[center] = make_float4(cn.x / k, cn.y / k, cn.z / k, 0.f);
  tex_out[center] = make_float4(ct.x / k, ct.y / k, ct.z / k, 0.f);
  created[center] = (count >= num_consistent) ? 1.f : 0.f;
}

}  // namespace

// cams (nviews, 28) floats per camera: P[12] | M_inv[9] | C[3] | P_col34[3] | f_ref-candidate (K[0,0]); normals_depths and
// images (nviews, rows, cols, 4) float: (nx, ny, nz, depth) and (b, g, r, unused).  Outputs for reference camera `ref`,
// each (rows, cols, 4) resp. (rows, cols): the 3-D point of every pixel, the averaged normal and colour, and 1 / 0 for
// "at least num_consistent other views agree" (the reference stores the point only then, fusibile.cu:252-262).
extern "C" int atvs_fusibile(const float* cams, const float* normals_depths, const float* images, int nviews, int ref,
                             int rows, int cols, float disp_thresh, float normal_thresh, int num_consistent, float* coord,
                             float* normal, float* texture, float* created, atvs_stream_t stream) {
  if (!cams || !normals_depths || !images || !coord || !normal || !texture || !created) return ATVS_ERR_NULL;
  if (nviews <= 0 || ref < 0 || ref >= nviews || rows <= 0 || cols <= 0) return ATVS_ERR_SHAPE;
  dim3 grid(cdiv(cols, 32), cdiv(rows, 8)), block(256);
  hipLaunchKernelGGL(fusibile_kernel, grid, block, 0, as_stream(stream), cams, reinterpret_cast<const float4*>(normals_depths),
                     reinterpret_cast<const float4*>(images), nviews, ref, rows, cols, disp_thresh, normal_thresh,
                     num_consistent, reinterpret_cast<float4*>(coord), reinterpret_cast<float4*>(normal),
                     reinterpret_cast<float4*>(texture), created);
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}
