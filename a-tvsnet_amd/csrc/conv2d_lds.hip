// 3x3 (dilated) stride-1 SAME 2-D convolution of wide feature maps on the fp32 matrix cores, LDS-tiled (gfx950).
//
// The heavy layers of the 2-D feature towers (ResNetDS2SPP, /root/reference/cnn_wrapper/atvsnet.py:254-292): the
// bottlenecks' conv2 (slim.conv2d 3x3, rate 1 / 2 / 4, network.py:585-587), conv0_1 / conv0_2 and fusion0
// (tf.layers.conv2d, network.py:198-200), Cin in {32, 64, 128, 320}, Cout in {32, 64, 128}, on 128x160 or 256x320
// maps -- a GEMM with few rows (20 480 pixels) and wide K (9 * Cin) and N (Cout).
//
// x is (G, H, W, Cin): G independent images (the reference calls the tower once per view with batch 1 and
// per-call batch statistics, quirk C1); tiles never span images and the batch-norm moments are per image.
//
// Structure: one workgroup (4 wavefronts) per tile of R x 16 output pixels of one image.
//   * the WAVES SPLIT THE OUTPUT CHANNELS (wave -> 16*NTW channels) and share the pixels: the packed weights are
//     streamed from L2 exactly once per workgroup (operand A, a 3-slot register ring filled two K steps ahead),
//     the pixels come from LDS (operand B) and are read by all four waves;
//   * K loop = chunks of 16 input channels x 9 taps, fully unrolled per chunk; the (R + 2d) x (16 + 2d) halo of
//     the NEXT chunk is fetched into registers one 16-byte slot per K step between the MFMAs and written to the
//     other LDS buffer after the loop: one barrier per chunk;
//   * LDS image [row][24 pixels][16 channels]: 64 B per pixel, row pitch 1536 B = 3 * 512 B, bit 5 of the byte
//     address XOR-ed with bit 8 (conflict-free ds_read_b128 for every tap alignment, as in conv_tiled.hip); the
//     three x displacements are three swizzled base registers, every (ky, row) displacement an immediate offset;
//   * optional prologue: the producer's batch norm (+ ReLU) applied while staging (normalise-on-load) -- padding
//     stays zero, as the reference pads the normalised tensor;
//   * epilogue: bias, residual, ReLU, 16-byte channel-last stores, per-(image, workgroup) partial moments.
#include <type_traits>

#include "conv_common.h"

namespace {

constexpr int C2_PITCH = 24;              // pixels per LDS row (16 + 2 * 4)
constexpr int C2_ROWB = C2_PITCH * 64;    // bytes per LDS row

struct C2Args {
  const float* x;
  const float* wp;
  const float* bias;
  const float* res;
  const float* in_params;   // (G, 3, Cin) = mean, rstd, beta of the producer's batch norm, or nullptr
  float* y;
  double* stats;
  int G, H, W, Cin, Cout;
  int ldy, ycoff;
  int nchunk;
  int tiles_x, tiles;       // per image
  int relu, in_relu;
  long gx, gy;              // elements per image of x and of y / res
  long total;               // G * tiles
};

__device__ __forceinline__ int c2_swz(int a) { return a ^ (((a >> 8) & 1) << 5); }

// NTW = 16-channel output tiles per wave, WR = row groups across the waves (4 / WR waves split the channels),
// TY = rows per wave, DIL = dilation.  Cout = 16 * NTW * (4 / WR); tile = (TY * WR) rows x 16 columns.
template <int NTW, int WR, int TY, int DIL>
__global__ __launch_bounds__(256, 2) void conv2d_lds_kernel(C2Args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int R = TY * WR;
  constexpr int HR = R + 2 * DIL, HC = 16 + 2 * DIL;
  constexpr int SLOTS = HR * HC * 4;
  constexpr int MAXS = (SLOTS + 255) / 256;
  constexpr int BUFB = HR * C2_ROWB;
  constexpr int WN = 4 / WR;                 // waves across the output channels
  constexpr int NT = NTW * WN;               // 16-channel tiles of the whole output
  static_assert(MAXS <= 9, "one halo slot per K step");
  static_assert(HC <= C2_PITCH, "row pitch");

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int wn = wave % WN, wr = wave / WN;

  // tile of this workgroup; blocks that share an XCD (blockIdx % 8) take a contiguous eighth of the tiles
  const long per = (p.total + 7) >> 3;
  const long lin = (long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (lin >= p.total) return;
  const int g = (int)(lin / p.tiles), tile = (int)(lin % p.tiles);
  const int y0 = (tile / p.tiles_x) * R, x0 = (tile % p.tiles_x) * 16;

  // ---- halo slots of this thread
  const float* xg = p.x + (size_t)g * p.gx;
  int goff[MAXS], laddr[MAXS];
  unsigned valid = 0;
#pragma unroll
  for (int i = 0; i < MAXS; ++i) {
    int s = tid + i * 256;
    const bool live = s < SLOTS;
    s = min(s, SLOTS - 1);
    const int c4 = s & 3, v = s >> 2;
    const int xx = v % HC, yy = v / HC;
    const int gy = y0 - DIL + yy, gxx = x0 - DIL + xx;
    const bool ok = live && (unsigned)gy < (unsigned)p.H && (unsigned)gxx < (unsigned)p.W;
    goff[i] = ok ? ((gy * p.W + gxx) * p.Cin + c4 * 4) : 0;
    laddr[i] = live ? c2_swz((yy * C2_PITCH + xx) * 64 + c4 * 16) : -1;
    valid |= (ok ? 1u : 0u) << i;
  }
  float4 pf[MAXS];
  auto pf_slot = [&](int i, int ch) __attribute__((always_inline)) {
    pf[i] = ((valid >> i) & 1u) ? ld4(xg + goff[i] + ch * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto write_image = [&](int buf, int ch) __attribute__((always_inline)) {
    const float* ip = p.in_params ? p.in_params + (size_t)g * 3 * p.Cin + ch * 16 : nullptr;
#pragma unroll
    for (int i = 0; i < MAXS; ++i) {
      if (laddr[i] < 0) continue;
      float4 v = pf[i];
      if (ip && ((valid >> i) & 1u)) {
        const int c = ((tid + i * 256) & 3) * 4;
        const float4 m = ld4(ip + c), s = ld4(ip + p.Cin + c), b = ld4(ip + 2 * p.Cin + c);
        v = atvs_bn4(v, s, atvs_bn_shift4(m, s, b));
        if (p.in_relu) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
      }
      *reinterpret_cast<float4*>(smem + buf * BUFB + laddr[i]) = v;
    }
  };

  // ---- LDS read bases: x displacement kx -> this lane's fragment for (ky, row) = (0, 0) of its row group
  int base[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) base[kx] = c2_swz(((wr * TY) * C2_PITCH + r + kx * DIL) * 64 + q * 16);

  // ---- packed weights: [K step = chunk * 9 + tap][NT tiles][64 lanes] float4, 2 padding steps at the end
  const float4* __restrict__ wl = reinterpret_cast<const float4*>(p.wp) + (size_t)(wn * NTW) * 64 + lane;
  constexpr int WSTEP = NT * 64;
  float4 w[3][NTW];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int n = 0; n < NTW; ++n) w[j][n] = wl[(size_t)j * WSTEP + n * 64];

  f32x4 acc[TY][NTW];
#pragma unroll
  for (int t = 0; t < TY; ++t)
#pragma unroll
    for (int n = 0; n < NTW; ++n) acc[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int i = 0; i < MAXS; ++i) pf_slot(i, 0);
  write_image(0, 0);
  __syncthreads();

  float4 b[2][TY];
  for (int ch = 0; ch < p.nchunk; ++ch) {
    const unsigned char* lb = smem + (ch & 1) * BUFB;
    const bool more = ch + 1 < p.nchunk;
    const float4* wc = wl + (size_t)ch * 9 * WSTEP;
    auto request_b = [&](int tap) __attribute__((always_inline)) {
      const int ky = tap / 3, kx = tap % 3;
#pragma unroll
      for (int t = 0; t < TY; ++t)
        b[tap & 1][t] = *reinterpret_cast<const float4*>(lb + base[kx] + (ky * DIL + t) * C2_ROWB);
    };
    request_b(0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      // weights of K step tap + 2 into the ring slot that step tap - 1 has just consumed
#pragma unroll
      for (int n = 0; n < NTW; ++n) w[(tap + 2) % 3][n] = wc[(size_t)(tap + 2) * WSTEP + n * 64];
      if (tap + 1 < 9) request_b(tap + 1);
      if (tap < MAXS && more) pf_slot(tap, ch + 1);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int n = 0; n < NTW; ++n)
#pragma unroll
          for (int t = 0; t < TY; ++t)
            acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(w[tap % 3][n], s), f4get(b[tap & 1][t], s), acc[t][n], 0, 0, 0);
    }
    if (more) {
      write_image((ch + 1) & 1, ch + 1);     // the other buffer: last read in chunk ch - 1, behind the barrier below
      __syncthreads();
    }
  }

  // ---- epilogue: lane holds channels co..co+3 of pixel (y0 + wr*TY + t, x0 + r)
  const int xo = x0 + r;
  float ssum[NTW][4], ssq[NTW][4];
#pragma unroll
  for (int n = 0; n < NTW; ++n)
#pragma unroll
    for (int k = 0; k < 4; ++k) ssum[n][k] = ssq[n][k] = 0.f;
  float* yg = p.y + (size_t)g * p.gy;
  const float* rg = p.res ? p.res + (size_t)g * p.gy : nullptr;
#pragma unroll
  for (int t = 0; t < TY; ++t) {
    const int yo = y0 + wr * TY + t;
    if (yo >= p.H || xo >= p.W) continue;
    const size_t rowb = ((size_t)yo * p.W + xo) * p.ldy + p.ycoff;
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
      const int co = (wn * NTW + n) * 16 + 4 * q;
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
    // row (image, tile): [2][Cout] doubles.  Channels are private to a wave (WR == 1) or shared by the WR waves of
    // a column of row groups (combined through LDS).
    double* row = p.stats + (size_t)lin * 2 * p.Cout;
    double* s_red = reinterpret_cast<double*>(smem);           // [wr][2][Cout], the images are dead
    if (WR > 1) __syncthreads();
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        double a = (double)ssum[n][k], bq = (double)ssq[n][k];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          a += __shfl_xor(a, o);
          bq += __shfl_xor(bq, o);
        }
        if (r == 0) {
          const int c = (wn * NTW + n) * 16 + 4 * q + k;
          if (WR == 1) {
            row[c] = a;
            row[p.Cout + c] = bq;
          } else {
            s_red[(wr * 2 + 0) * p.Cout + c] = a;
            s_red[(wr * 2 + 1) * p.Cout + c] = bq;
          }
        }
      }
    if (WR > 1) {
      __syncthreads();
      for (int i = tid; i < 2 * p.Cout; i += 256) {
        double v = 0.0;
#pragma unroll
        for (int a = 0; a < WR; ++a) v += s_red[a * 2 * p.Cout + i];
        row[i] = v;
      }
    }
  }
}

template <int NTW, int WR, int TY, int DIL>
int launch_c2(const C2Args& a, hipStream_t s) {
  constexpr int R = TY * WR, HR = R + 2 * DIL;
  size_t lds = (size_t)2 * HR * C2_ROWB;
  const long blocks = ((a.total + 7) / 8) * 8;
  if (blocks > 0x7fffffffL) return ATVS_ERR_SHAPE;
  hipLaunchKernelGGL((conv2d_lds_kernel<NTW, WR, TY, DIL>), dim3((unsigned)blocks), dim3(256), lds, s, a);
  return ATVS_OK;
}

// rows of an output tile for a channel count
int c2_tile_rows(int Cout) { return Cout == 32 ? 8 : 4; }

}  // namespace

extern "C" int atvs_conv2d_lds_supported(int Cin, int Cout, int dilation) {
  const bool cin_ok = Cin > 0 && Cin % 16 == 0;
  const bool cout_ok = Cout == 32 || Cout == 64 || Cout == 128;
  const bool dil_ok = dilation == 1 || (Cout == 128 && (dilation == 2 || dilation == 4));
  return (cin_ok && cout_ok && dil_ok) ? 1 : 0;
}

// Workgroups per image of a launch = rows per image of stats_partial ([2][Cout] doubles each).
extern "C" long atvs_conv2d_lds_rows(int H, int W, int Cout) {
  const int R = c2_tile_rows(Cout);
  return (long)((H + R - 1) / R) * ((W + 15) / 16);
}

extern "C" int atvs_conv2d_lds_pack_size(int Cin, int Cout, long* packed_floats) {
  if (!packed_floats) return ATVS_ERR_NULL;
  if (!atvs_conv2d_lds_supported(Cin, Cout, 1)) return ATVS_ERR_SHAPE;
  *packed_floats = (long)((Cin / 16) * 9 + 2) * (Cout / 16) * 64 * 4;
  return ATVS_OK;
}

// HOST function.  w: TF kernel [3][3][Cin][Cout].  packed[K step = chunk * 9 + (ky * 3 + kx)][tile n][lane = q * 16 + co16][s]
// = w[ky][kx][chunk * 16 + 4 q + s][n * 16 + co16]; two zero K steps of padding at the end (the weight ring reads ahead).
extern "C" int atvs_conv2d_lds_pack(const float* w, int Cin, int Cout, float* packed) {
  if (!w || !packed) return ATVS_ERR_NULL;
  long pf;
  int rc = atvs_conv2d_lds_pack_size(Cin, Cout, &pf);
  if (rc) return rc;
  const int NT = Cout / 16, nch = Cin / 16;
  for (long i = 0; i < pf; ++i) packed[i] = 0.f;
  for (int ch = 0; ch < nch; ++ch)
    for (int tap = 0; tap < 9; ++tap)
      for (int n = 0; n < NT; ++n)
        for (int q = 0; q < 4; ++q)
          for (int co16 = 0; co16 < 16; ++co16)
            for (int s = 0; s < 4; ++s) {
              const int ci = ch * 16 + 4 * q + s, co = n * 16 + co16;
              packed[((((size_t)(ch * 9 + tap) * NT + n) * 64) + q * 16 + co16) * 4 + s] = w[((size_t)tap * Cin + ci) * Cout + co];
            }
  return ATVS_OK;
}

// y (G,H,W,ldy)[..., y_coff + co] = conv3x3(x (G,H,W,Cin), dilation, SAME) (+ bias, + residual, ReLU).
// residual: same addressing as y (y_coff must be 0).  in_params (G,3,Cin) != NULL: x is a raw convolution output whose
// training-mode batch norm (mean, rstd, beta per image) [+ ReLU, in_relu] is applied on load.
// stats_partial: G * atvs_conv2d_lds_rows rows of [2][Cout] doubles (image-major) or NULL.
extern "C" int atvs_conv2d_lds_f32(const float* x, const float* packed_w, const float* bias, const float* residual,
                                   const float* in_params, int in_relu, float* y, double* stats_partial, int G, int H,
                                   int W, int Cin, int Cout, int dilation, int ldy, int y_coff, int relu,
                                   atvs_stream_t stream) {
  if (!x || !packed_w || !y) return ATVS_ERR_NULL;
  if (G <= 0 || H <= 0 || W <= 0 || !atvs_conv2d_lds_supported(Cin, Cout, dilation)) return ATVS_ERR_SHAPE;
  if (y_coff < 0 || y_coff + Cout > ldy || (ldy % 4) || (y_coff % 4)) return ATVS_ERR_SHAPE;
  if (residual && y_coff != 0) return ATVS_ERR_ARG;
  if ((double)H * W * Cin >= 2147483648.0) return ATVS_ERR_SHAPE;      // 31-bit element offsets inside an image
  C2Args a;
  a.x = x; a.wp = packed_w; a.bias = bias; a.res = residual; a.in_params = in_params; a.y = y; a.stats = stats_partial;
  a.G = G; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.ldy = ldy; a.ycoff = y_coff; a.nchunk = Cin / 16;
  const int R = c2_tile_rows(Cout);
  a.tiles_x = (W + 15) / 16;
  a.tiles = ((H + R - 1) / R) * a.tiles_x;
  a.relu = relu; a.in_relu = in_relu;
  a.gx = (long)H * W * Cin; a.gy = (long)H * W * ldy;
  a.total = (long)G * a.tiles;
  hipStream_t s = as_stream(stream);
  int rc = ATVS_ERR_ARG;
  if (Cout == 128) {
    if (dilation == 1) rc = launch_c2<2, 1, 4, 1>(a, s);
    else if (dilation == 2) rc = launch_c2<2, 1, 4, 2>(a, s);
    else rc = launch_c2<2, 1, 4, 4>(a, s);
  } else if (Cout == 64) {
    rc = launch_c2<1, 1, 4, 1>(a, s);
  } else {
    rc = launch_c2<1, 2, 4, 1>(a, s);
  }
  if (rc) return rc;
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}
