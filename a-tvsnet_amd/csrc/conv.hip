// Convolution as an implicit GEMM on the fp32 matrix cores of gfx950
// (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate).
//
// Replaces tf.layers.conv2d / conv3d / conv3d_transpose, slim.conv2d and
// tf.nn.conv3d (+ bias_add, relu, the residual add of the bottleneck) at
// /root/reference/cnn_wrapper/network.py:141-215, 282-351, 510-602, and feeds the
// training-mode batch-norm statistics of :206-212, 541-547, 570-571 from its epilogue.
//
// GEMM view (per launch):  Y^T[co, v] = sum_k  Wp[co, k] * X[v, k]
//   M = output channels (16 per MFMA tile, NT tiles), N = 16 output voxels per tile
//   (TM tiles per wavefront), K = (tap, input channel) pairs listed by a host-built
//   "group table" (so SAME / explicit padding, stride, dilation and the parity classes
//   of a stride-2 transposed convolution are all the same kernel).
// Orientation: weights are the A operand (lane: co = l&15, k = l>>4), inputs the B
// operand (lane: voxel = l&15, k = l>>4); the accumulator then holds, per lane, 4
// CONSECUTIVE output channels of one voxel -> 16-byte channel-last stores.
// K ordering trick: with Cin % 4 == 0 a lane loads one float4 = 4 consecutive input
// channels of its tap and feeds element s to MFMA s; the packed weight for MFMA s is
// arranged to match, so one 16-byte load per lane feeds four MFMAs.
#include "conv_common.h"

struct ConvArgs {
  const float* x;
  const float* wp;
  const int4* tab;
  const float* bias;
  const float* res;
  float* y;
  double* stats;
  int Di, Hi, Wi, Cin;
  int Do, Ho, Wo;          // logical output grid of this launch
  long M;                  // Do*Ho*Wo
  int sI;                  // input step per output step
  int Hy, Wy;              // full output tensor dims (rows)
  int oS, offz, offy, offx;  // output voxel = o*oS + off
  int ldy, ycoff, Cout;
  int J;                   // K steps (4 groups each)
  int relu;
  int vec_out;
  const float* pbias;      // (Ho, Wo, 3*Cout) depth-plane bias variants or nullptr (see plane_variant)
  int pb_pz;               // z padding before, to pick the variant
  // groups: G independent samples stacked on the leading axis of x / y / residual / plane bias; a workgroup
  // belongs to one sample (bpg workgroups per sample), the statistics rows are (sample, workgroup)
  int bpg;
  long gx, gy, gpb;        // elements per sample of x, of y / residual, of the plane bias
};

template <int V>
struct AVal;
template <>
struct AVal<4> { typedef float4 T; };
template <>
struct AVal<1> { typedef float T; };

template <int V>
__device__ __forceinline__ float aget(const typename AVal<V>::T& a, int s);
template <>
__device__ __forceinline__ float aget<4>(const float4& a, int s) {
  return s == 0 ? a.x : (s == 1 ? a.y : (s == 2 ? a.z : a.w));
}
template <>
__device__ __forceinline__ float aget<1>(const float& a, int) { return a; }

template <int V>
__device__ __forceinline__ typename AVal<V>::T azero();
template <>
__device__ __forceinline__ float4 azero<4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <>
__device__ __forceinline__ float azero<1>() { return 0.f; }

template <int NT, int TM, int V>
__global__ __launch_bounds__(256) void conv_mfma_f32_kernel(ConvArgs p) {
  extern __shared__ int4 s_tab[];
  typedef typename AVal<V>::T AT;
  const int tid = threadIdx.x;
  for (int i = tid; i < p.J * 4; i += 256) s_tab[i] = p.tab[i];
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;

  // voxel of each of this wave's TM tiles (lane r <-> voxel r of the tile)
  int iz[TM], iy[TM], ix[TM];
  long mvox[TM];
  const int grp = blockIdx.x / p.bpg;
  const long m0 = ((long)(blockIdx.x - grp * p.bpg) * 4 + wave) * (TM * 16);
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    long m = m0 + t * 16 + r;
    mvox[t] = m;
    if (m < p.M) {
      int xo = (int)(m % p.Wo);
      long rest = m / p.Wo;
      int yo = (int)(rest % p.Ho);
      int zo = (int)(rest / p.Ho);
      iz[t] = zo * p.sI;
      iy[t] = yo * p.sI;
      ix[t] = xo * p.sI;
    } else {
      iz[t] = -(1 << 29);   // every tap fails the bounds test
      iy[t] = 0;
      ix[t] = 0;
    }
  }

  f32x4 acc[TM][NT];
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const float* __restrict__ x = p.x + (size_t)grp * p.gx;
  const AT* __restrict__ wp = reinterpret_cast<const AT*>(p.wp);

  AT a_cur[TM], w_cur[NT];
  auto load_step = [&](int j, AT* a, AT* w) {
    int4 e = s_tab[4 * j + q];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      int zz = iz[t] + e.x, yy = iy[t] + e.y, xx = ix[t] + e.z;
      bool ok = ((unsigned)zz < (unsigned)p.Di) && ((unsigned)yy < (unsigned)p.Hi) && ((unsigned)xx < (unsigned)p.Wi);
      size_t off = (((size_t)zz * p.Hi + yy) * p.Wi + xx) * (size_t)p.Cin + e.w;
      a[t] = ok ? *reinterpret_cast<const AT*>(x + off) : azero<V>();
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) w[n] = wp[((size_t)j * NT + n) * 64 + lane];
  };

  load_step(0, a_cur, w_cur);
  for (int j = 0; j < p.J; ++j) {
    AT a_nxt[TM], w_nxt[NT];
    if (j + 1 < p.J) load_step(j + 1, a_nxt, w_nxt);
#pragma unroll
    for (int s = 0; s < V; ++s)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int t = 0; t < TM; ++t)
          acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(aget<V>(w_cur[n], s), aget<V>(a_cur[t], s), acc[t][n], 0, 0, 0);
    if (j + 1 < p.J) {
#pragma unroll
      for (int t = 0; t < TM; ++t) a_cur[t] = a_nxt[t];
#pragma unroll
      for (int n = 0; n < NT; ++n) w_cur[n] = w_nxt[n];
    }
  }

  // ---- epilogue: lane holds channels n*16 + 4q .. +3 of voxel r of each tile
  float ssum[NT][4], ssq[NT][4];
#pragma unroll
  for (int n = 0; n < NT; ++n)
#pragma unroll
    for (int k = 0; k < 4; ++k) ssum[n][k] = ssq[n][k] = 0.f;

#pragma unroll
  for (int t = 0; t < TM; ++t) {
    long m = mvox[t];
    if (m >= p.M) continue;
    int xo = (int)(m % p.Wo);
    long rest = m / p.Wo;
    int yo = (int)(rest % p.Ho);
    int zo = (int)(rest / p.Ho);
    size_t vox = ((size_t)(zo * p.oS + p.offz) * p.Hy + (yo * p.oS + p.offy)) * p.Wy + (xo * p.oS + p.offx);
    size_t base = (size_t)grp * p.gy + vox * (size_t)p.ldy + p.ycoff;
    const float* pb = nullptr;
    if (p.pbias)
      pb = p.pbias + (size_t)grp * p.gpb + ((size_t)yo * p.Wo + xo) * (size_t)(3 * p.Cout) +
           plane_variant(zo * p.sI - p.pb_pz, p.Di) * p.Cout;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      int co = n * 16 + 4 * q;
      if (co >= p.Cout) continue;
      float v[4] = {acc[t][n][0], acc[t][n][1], acc[t][n][2], acc[t][n][3]};
      if (pb) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (co + k < p.Cout) v[k] += pb[co + k];
      }
      if (p.vec_out) {
        if (p.bias) {
          float4 b = ld4(p.bias + co);
          v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
        if (p.res) {
          float4 rr = ld4(p.res + base + co);
          v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
        }
        if (p.relu) {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
        }
        st4(p.y + base + co, make_float4(v[0], v[1], v[2], v[3]));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          ssum[n][k] += v[k];
          ssq[n][k] += v[k] * v[k];
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (co + k < p.Cout) {
            float u = v[k];
            if (p.bias) u += p.bias[co + k];
            if (p.res) u += p.res[base + co + k];
            if (p.relu) u = fmaxf(u, 0.f);
            p.y[base + co + k] = u;
            ssum[n][k] += u;
            ssq[n][k] += u * u;
          }
        }
      }
    }
  }

  if (p.stats) {
    // per-channel partial sums of this workgroup: lanes r -> shuffle, waves -> LDS
    __shared__ double s_red[4][2][NT * 16];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        double a = (double)ssum[n][k], b = (double)ssq[n][k];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          a += __shfl_xor(a, o);
          b += __shfl_xor(b, o);
        }
        if (r == 0) {
          s_red[wave][0][n * 16 + 4 * q + k] = a;
          s_red[wave][1][n * 16 + 4 * q + k] = b;
        }
      }
    __syncthreads();
    if (tid < 2 * NT * 16) {
      int which = tid / (NT * 16), c = tid % (NT * 16);
      double v = (s_red[0][which][c] + s_red[1][which][c]) + (s_red[2][which][c] + s_red[3][which][c]);
      p.stats[((size_t)blockIdx.x * 2 + which) * (NT * 16) + c] = v;
    }
  }
}

extern "C" int atvs_conv_pack_size(int ntaps, int Cin, int Cout, int* vec, int* ksteps, int* ntiles,
                                   long* packed_floats, long* table_ints) {
  if (ntaps <= 0 || Cin <= 0 || Cout <= 0 || Cout > 128) return ATVS_ERR_SHAPE;
  int V = (Cin % 4 == 0) ? 4 : 1;
  long G = (long)ntaps * (V == 4 ? Cin / 4 : Cin);
  int J = (int)((G + 3) / 4);
  int NT = pow2_tiles(Cout);
  if (vec) *vec = V;
  if (ksteps) *ksteps = J;
  if (ntiles) *ntiles = NT;
  if (packed_floats) *packed_floats = (long)J * NT * 64 * V;
  if (table_ints) *table_ints = (long)J * 16;
  return ATVS_OK;
}

// HOST function: arrange TF-layout weights for the kernel.
//   w: host, [n_w_taps][Cin][Cout] (w_transposed = 0: conv kernels [k.., Cin, Cout]) or
//      [n_w_taps][Cout][Cin] (w_transposed = 1: conv3d_transpose kernels [k.., Cout, Cin]).
//   taps: host, ntaps x 4 ints: (index of the tap in w, dz, dy, dx) with (dz,dy,dx) the
//      input offset relative to output_index * in_stride (padding / dilation folded in).
extern "C" int atvs_conv_pack(const float* w, int w_transposed, const int32_t* taps, int ntaps, int Cin, int Cout,
                              float* packed, int32_t* table) {
  if (!w || !taps || !packed || !table) return ATVS_ERR_NULL;
  int V, J, NT;
  long pf, ti;
  int rc = atvs_conv_pack_size(ntaps, Cin, Cout, &V, &J, &NT, &pf, &ti);
  if (rc) return rc;
  const int cg = (V == 4) ? Cin / 4 : Cin;   // groups per tap
  const long G = (long)ntaps * cg;
  for (long i = 0; i < pf; ++i) packed[i] = 0.f;
  for (int j = 0; j < J; ++j)
    for (int q = 0; q < 4; ++q) {
      long g = (long)j * 4 + q;
      int32_t* e = table + (j * 4 + q) * 4;
      if (g >= G) {
        e[0] = 1 << 28; e[1] = 0; e[2] = 0; e[3] = 0;
        continue;
      }
      int t = (int)(g / cg), c0 = (int)(g % cg) * V;
      e[0] = taps[t * 4 + 1]; e[1] = taps[t * 4 + 2]; e[2] = taps[t * 4 + 3]; e[3] = c0;
      int wt = taps[t * 4 + 0];
      for (int n = 0; n < NT; ++n)
        for (int co16 = 0; co16 < 16; ++co16) {
          int co = n * 16 + co16;
          if (co >= Cout) continue;
          for (int s = 0; s < V; ++s) {
            int ci = c0 + s;
            float val = w_transposed ? w[((size_t)wt * Cout + co) * Cin + ci] : w[((size_t)wt * Cin + ci) * Cout + co];
            packed[((((size_t)j * NT + n) * 64) + q * 16 + co16) * V + s] = val;
          }
        }
    }
  return ATVS_OK;
}

extern "C" long atvs_conv_num_blocks(long M, int tile_m) { return (M + 64L * tile_m - 1) / (64L * tile_m); }

template <int NT, int V>
static int launch_tm(ConvArgs& a, int TM, int groups, hipStream_t s) {
  long bpg = atvs_conv_num_blocks(a.M, TM);
  long blocks = bpg * groups;
  if (bpg <= 0 || blocks > 0x7fffffffL) return ATVS_ERR_SHAPE;
  a.bpg = (int)bpg;
  size_t lds = (size_t)a.J * 4 * sizeof(int4);
  dim3 grid((unsigned)blocks), block(256);
  switch (TM) {
    case 1: hipLaunchKernelGGL((conv_mfma_f32_kernel<NT, 1, V>), grid, block, lds, s, a); break;
    case 2: hipLaunchKernelGGL((conv_mfma_f32_kernel<NT, 2, V>), grid, block, lds, s, a); break;
    case 4:
      if (NT > 4) return ATVS_ERR_ARG;
      hipLaunchKernelGGL((conv_mfma_f32_kernel<(NT > 4 ? 4 : NT), 4, V>), grid, block, lds, s, a);
      break;
    case 8:
      if (NT > 2) return ATVS_ERR_ARG;
      hipLaunchKernelGGL((conv_mfma_f32_kernel<(NT > 2 ? 2 : NT), 8, V>), grid, block, lds, s, a);
      break;
    default: return ATVS_ERR_ARG;
  }
  return ATVS_OK;
}

template <int V>
static int launch_nt(ConvArgs& a, int NT, int TM, int groups, hipStream_t s) {
  switch (NT) {
    case 1: return launch_tm<1, V>(a, TM, groups, s);
    case 2: return launch_tm<2, V>(a, TM, groups, s);
    case 4: return launch_tm<4, V>(a, TM, groups, s);
    case 8: return launch_tm<8, V>(a, TM, groups, s);
  }
  return ATVS_ERR_ARG;
}

extern "C" int atvs_conv_mfma_f32(const float* x, const float* packed_w, const int32_t* group_table, const float* bias,
                                  const float* residual, const float* plane_bias, int pad_z, float* y,
                                  double* stats_partial, int groups, int Di, int Hi, int Wi,
                                  int Cin, int Do, int Ho, int Wo, int in_stride, int Dy, int Hy, int Wy,
                                  int out_stride, int off_z, int off_y, int off_x, int ldy, int y_coff, int Cout,
                                  int ntaps, int tile_m, int relu, atvs_stream_t stream) {
  if (!x || !packed_w || !group_table || !y) return ATVS_ERR_NULL;
  if (groups <= 0 || Di <= 0 || Hi <= 0 || Wi <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0 || in_stride <= 0 || out_stride <= 0)
    return ATVS_ERR_SHAPE;
  if (y_coff < 0 || y_coff + Cout > ldy) return ATVS_ERR_SHAPE;
  if ((double)Di * Hi * Wi * Cin >= 9.0e18) return ATVS_ERR_SHAPE;
  if ((Do - 1) * out_stride + off_z >= Dy || (Ho - 1) * out_stride + off_y >= Hy || (Wo - 1) * out_stride + off_x >= Wy)
    return ATVS_ERR_SHAPE;
  int V, J, NT;
  int rc = atvs_conv_pack_size(ntaps, Cin, Cout, &V, &J, &NT, nullptr, nullptr);
  if (rc) return rc;
  ConvArgs a;
  a.x = x; a.wp = packed_w; a.tab = reinterpret_cast<const int4*>(group_table); a.bias = bias; a.res = residual;
  a.y = y; a.stats = stats_partial;
  a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Cin = Cin;
  a.Do = Do; a.Ho = Ho; a.Wo = Wo; a.M = (long)Do * Ho * Wo;
  a.sI = in_stride; a.Hy = Hy; a.Wy = Wy; a.oS = out_stride; a.offz = off_z; a.offy = off_y; a.offx = off_x;
  a.ldy = ldy; a.ycoff = y_coff; a.Cout = Cout; a.J = J; a.relu = relu;
  a.vec_out = (Cout % 4 == 0) && (ldy % 4 == 0) && (y_coff % 4 == 0);
  a.pbias = plane_bias; a.pb_pz = pad_z;
  a.gx = (long)Di * Hi * Wi * Cin; a.gy = (long)Dy * Hy * Wy * ldy; a.gpb = (long)Ho * Wo * 3 * Cout; a.bpg = 0;
  if (plane_bias && (out_stride != 1 || Di < 2)) return ATVS_ERR_ARG;
  if (residual && y_coff != 0) return ATVS_ERR_ARG;   // residual shares y's addressing
  hipStream_t s = as_stream(stream);
  rc = (V == 4) ? launch_nt<4>(a, NT, tile_m, groups, s) : launch_nt<1>(a, NT, tile_m, groups, s);
  if (rc) return rc;
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}
