// 1x1 convolution of feature maps (a tall GEMM Y[p, co] = sum_ci X[p, ci] W[ci, co]) on the 16-bit matrix cores with SPLIT operands
// (gfx950): conv1x1.hip's layers (the bottlenecks' conv1 / conv3 / shortcut and fusion1 of the 2-D towers,
// /root/reference/cnn_wrapper/network.py:552-602, cnn_wrapper/atvsnet.py:254-292) with every fp32 operand split into TWO fp16
// pieces, three products, fp32 accumulation (conv_c16b.hip / conv_xb.hip have the arithmetic; round 3: three bf16 pieces, six products).  At 128 -> 128 the fp32 MFMA form is as much
// matrix-core- as HBM-bound (32 FLOP per byte); with the products 5x cheaper the layer is HBM-bound.
//
// Structure = conv2d_b.hip without a halo: one workgroup per 128 consecutive pixels of one image, the WAVES SPLIT THE OUTPUT
// CHANNELS and share the pixels, K loop in chunks of 32 input channels = ONE K = 32 step (lane group q = channels 8 q ..), two
// phases per step (h0 with both weight pieces, h1 with g0); two LDS buffers of two piece images [128 pixels][32 channels] (64-byte pixels, bit 5 of the byte address
// XOR-ed with bit 9: conflict-free ds_read_b128); the next chunk's pixels are fetched during the phases and split + written after
// them; weight pieces streamed from L2 one step ahead (two register slots, the chunk loop unrolled by two); optional
// normalise-on-load; epilogue as conv1x1.hip (bias, residual, ReLU, per-(image, workgroup) moments).
#include <cstring>
#include <type_traits>

#include "conv_common.h"

namespace {

constexpr int C1B_PX = 128;                   // pixels per workgroup
constexpr int C1B_PIMG = C1B_PX * 64;         // bytes of one piece image (32 channels x 2 B per pixel)
constexpr int C1B_NP = 2;                     // operand pieces: x = h0 + h1 / 2048 (conv_c16b.hip, round 4)
constexpr float C1B_RS = 2048.f, C1B_IRS = 1.f / 2048.f;
constexpr int C1B_BUFB = C1B_NP * C1B_PIMG;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

struct C1bArgs {
  const float* x;
  const f16x8* wp;
  const float* bias;
  const float* res;
  const float* in_params;
  float* y;
  double* stats;
  int Cin, Cout, ldy, ycoff;
  int relu, in_relu;
  long rows;                // pixels per image
  int wgs;                  // workgroups per image
  int nchunk;
};

__device__ __forceinline__ int c1b_swz(int a) { return a ^ (((a >> 9) & 1) << 5); }

__device__ __forceinline__ void c1b_split(const float4& v, f16x4* p0, f16x4* p1) {
  // atvs_split2_f16 (common.h): five vector instructions per two values instead of the 8-9 of the C form, the same values
  uint2 a, b;
  atvs_split4_f16(v, &a, &b);
  *p0 = __builtin_bit_cast(f16x4, a);
  *p1 = __builtin_bit_cast(f16x4, b);
}

// NTW = 16-channel output tiles per wave, WR = pixel groups across the waves (4 / WR waves split the channels).
// Cout = 16 * NTW * (4 / WR); a wave owns TYW = 8 / WR tiles of 16 pixels.
template <int NTW, int WR>
__global__ __launch_bounds__(256, 2) void conv1x1_b_kernel(C1bArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int WN = 4 / WR, NT = NTW * WN, TYW = 8 / WR;
  constexpr int MAXS = C1B_PX * 8 / 256;       // float4 slots per thread and chunk: 4
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int wn = wave % WN, wr = wave / WN;
  const int grp = blockIdx.x / p.wgs, wb = blockIdx.x - grp * p.wgs;
  const long pix0 = (long)wb * C1B_PX;
  const float* __restrict__ xg = p.x + (size_t)grp * p.rows * p.Cin;

  // ---- staging slots: float4 = channels 4 c4 .. of pixel px of the fp32 chunk -> 8 bytes of each piece image
  int goff[MAXS], laddr[MAXS];
  unsigned valid = 0;
#pragma unroll
  for (int i = 0; i < MAXS; ++i) {
    const int s = tid + i * 256;
    const int c4 = s & 7, px = s >> 3;
    const bool ok = pix0 + px < p.rows;
    goff[i] = ok ? (px * p.Cin + c4 * 4) : 0;
    laddr[i] = c1b_swz(px * 64 + c4 * 8);
    valid |= (ok ? 1u : 0u) << i;
  }
  const float* xt = xg + (size_t)pix0 * p.Cin;
  float4 pf[MAXS];
  auto pf_slot = [&](int i, int ch) __attribute__((always_inline)) {
    pf[i] = ((valid >> i) & 1u) ? ld4(xt + goff[i] + ch * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto write_image = [&](int buf, int ch) __attribute__((always_inline)) {
    const float* ip = p.in_params ? p.in_params + (size_t)grp * 3 * p.Cin + ch * 32 : nullptr;
#pragma unroll
    for (int i = 0; i < MAXS; ++i) {
      float4 v = pf[i];
      if (ip && ((valid >> i) & 1u)) {
        const int c = ((tid + i * 256) & 7) * 4;
        const float4 m = ld4(ip + c), s = ld4(ip + p.Cin + c), b = ld4(ip + 2 * p.Cin + c);
        v = atvs_bn4(v, s, atvs_bn_shift4(m, s, b));
        if (p.in_relu) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
      }
      f16x4 p0, p1;
      c1b_split(v, &p0, &p1);
      unsigned char* d = smem + buf * C1B_BUFB + laddr[i];
      *reinterpret_cast<f16x4*>(d) = p0;
      *reinterpret_cast<f16x4*>(d + C1B_PIMG) = p1;
    }
  };

  // this lane's fragment (channels 8 q .. of pixel r of a 16-pixel tile); tiles are 1024 bytes apart
  const int fb = c1b_swz(r * 64 + q * 16) + wr * TYW * 1024;

  // packed weight pieces: [chunk][NT tiles][2 pieces][64 lanes] f16x8, one zero chunk at the end
  const f16x8* __restrict__ wl = p.wp + (size_t)(wn * NTW) * C1B_NP * 64 + lane;
  constexpr int WSTEP = NT * C1B_NP * 64;
  // g0 fragments double-buffered (both phases of a chunk read them; the next chunk's are requested in the first phase), g1
  // fragments single (read in the first phase only; the next chunk's are requested behind it into the same registers)
  f16x8 Aw0[2][NTW], Aw1[NTW];
#pragma unroll
  for (int n = 0; n < NTW; ++n) {
    Aw0[0][n] = wl[(n * C1B_NP + 0) * 64];
    Aw1[n] = wl[(n * C1B_NP + 1) * 64];
  }

  f32x4 acc[TYW][NTW], accx[TYW][NTW];  // h0 g0 | (h0 g1 + h1 g0) * 2^11
#pragma unroll
  for (int t = 0; t < TYW; ++t)
#pragma unroll
    for (int n = 0; n < NTW; ++n) acc[t][n] = accx[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int i = 0; i < MAXS; ++i) pf_slot(i, 0);
  write_image(0, 0);
  __syncthreads();

  f16x8 Bq[2][TYW];
  auto chunk = [&](auto PAR, int ch) __attribute__((always_inline)) {
    constexpr int par = decltype(PAR)::value;                 // ch & 1: LDS buffer and weight slot of the chunk
    const unsigned char* lb = smem + par * C1B_BUFB;
    const bool more = ch + 1 < p.nchunk;
    auto request_b = [&](int pc) __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < TYW; ++t) Bq[pc & 1][t] = *reinterpret_cast<const f16x8*>(lb + pc * C1B_PIMG + fb + t * 1024);
    };
    request_b(0);
#pragma unroll
    for (int pc = 0; pc < C1B_NP; ++pc) {
      if (pc == 0) {
#pragma unroll
        for (int n = 0; n < NTW; ++n) Aw0[par ^ 1][n] = wl[(size_t)(ch + 1) * WSTEP + (n * C1B_NP + 0) * 64];
      }
      if (pc + 1 < C1B_NP) request_b(pc + 1);
      if (more) {
        if (pc == 0) { pf_slot(0, ch + 1); pf_slot(1, ch + 1); }
        if (pc == 1) { pf_slot(2, ch + 1); pf_slot(3, ch + 1); }
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < NTW; ++n) {
        if (pc == 0) {
#pragma unroll
          for (int t = 0; t < TYW; ++t) acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aw0[par][n], Bq[pc & 1][t], acc[t][n], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < TYW; ++t) accx[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aw1[n], Bq[pc & 1][t], accx[t][n], 0, 0, 0);
        } else {
#pragma unroll
          for (int t = 0; t < TYW; ++t) accx[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aw0[par][n], Bq[pc & 1][t], accx[t][n], 0, 0, 0);
        }
      }
      if (pc == 0) {                          // behind the MFMAs that read Aw1: the next chunk's g1 fragments
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NTW; ++n) Aw1[n] = wl[(size_t)(ch + 1) * WSTEP + (n * C1B_NP + 1) * 64];
      }
    }
    if (more) {
      write_image(par ^ 1, ch + 1);          // the other buffer: last read in chunk ch - 1, behind the barrier below
      __syncthreads();
    }
  };
  for (int ch = 0; ch < p.nchunk; ch += 2) {
    chunk(std::integral_constant<int, 0>{}, ch);
    if (ch + 1 < p.nchunk) chunk(std::integral_constant<int, 1>{}, ch + 1);
  }

  // ---- epilogue: lane holds channels (wn*NTW + n)*16 + 4q .. +3 of pixel pix0 + (wr*TYW + t)*16 + r
  float ssum[NTW][4], ssq[NTW][4];
#pragma unroll
  for (int n = 0; n < NTW; ++n)
#pragma unroll
    for (int k = 0; k < 4; ++k) ssum[n][k] = ssq[n][k] = 0.f;
  float* yg = p.y + (size_t)grp * p.rows * p.ldy;
  const float* rg = p.res ? p.res + (size_t)grp * p.rows * p.ldy : nullptr;
#pragma unroll
  for (int t = 0; t < TYW; ++t) {
    const long px = pix0 + (wr * TYW + t) * 16 + r;
    if (px >= p.rows) continue;
    const size_t rowb = (size_t)px * p.ldy + p.ycoff;
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
      const int co = (wn * NTW + n) * 16 + 4 * q;
      float4 v = make_float4(acc[t][n][0] + accx[t][n][0] * C1B_IRS, acc[t][n][1] + accx[t][n][1] * C1B_IRS,
                             acc[t][n][2] + accx[t][n][2] * C1B_IRS, acc[t][n][3] + accx[t][n][3] * C1B_IRS);
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
    // row (image, workgroup): [2][Cout] doubles; channels are private to a wave (WR == 1) or shared by WR waves
    double* row = p.stats + (size_t)blockIdx.x * 2 * p.Cout;
    double* s_red = reinterpret_cast<double*>(smem);
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

template <int NTW, int WR>
int launch_c1b(const C1bArgs& a, int groups, hipStream_t s) {
  const long blocks = (long)groups * a.wgs;
  if (blocks > 0x7fffffffL) return ATVS_ERR_SHAPE;
  hipLaunchKernelGGL((conv1x1_b_kernel<NTW, WR>), dim3((unsigned)blocks), dim3(256), 2 * C1B_BUFB, s, a);
  return ATVS_OK;
}


}  // namespace

extern "C" int atvs_conv1x1_b_supported(int Cin, int Cout) {
  return (Cin > 0 && Cin % 32 == 0 && Cin <= 1024 && (Cout == 32 || Cout == 64 || Cout == 128)) ? 1 : 0;
}

// workgroups per image = rows per image of stats_partial ([2][Cout] doubles each): 128 pixels per workgroup
extern "C" long atvs_conv1x1_b_rows(long pixels) { return (pixels + C1B_PX - 1) / C1B_PX; }

extern "C" int atvs_conv1x1_b_pack_size(int Cin, int Cout, long* packed_bytes) {
  if (!packed_bytes) return ATVS_ERR_NULL;
  if (!atvs_conv1x1_b_supported(Cin, Cout)) return ATVS_ERR_SHAPE;
  *packed_bytes = (long)(Cin / 32 + 1) * (Cout / 16) * C1B_NP * 1024;
  return ATVS_OK;
}

// HOST function.  w: [Cin][Cout].  packed[chunk][tile n][piece][lane = q * 16 + co16][e] = piece of
// w[chunk * 32 + q * 8 + e][n * 16 + co16]; one zero chunk of padding at the end (the weight slots read ahead).
extern "C" int atvs_conv1x1_b_pack(const float* w, int Cin, int Cout, unsigned char* packed) {
  if (!w || !packed) return ATVS_ERR_NULL;
  long pb;
  int rc = atvs_conv1x1_b_pack_size(Cin, Cout, &pb);
  if (rc) return rc;
  std::memset(packed, 0, (size_t)pb);
  uint16_t* out = reinterpret_cast<uint16_t*>(packed);
  const int NT = Cout / 16;
  bool fits = true;
  for (int ch = 0; ch < Cin / 32; ++ch)
    for (int n = 0; n < NT; ++n)
      for (int q = 0; q < 4; ++q)
        for (int co16 = 0; co16 < 16; ++co16)
          for (int e = 0; e < 8; ++e) {
            const float v = w[(size_t)(ch * 32 + q * 8 + e) * Cout + n * 16 + co16];
            const _Float16 g0 = (_Float16)v, g1 = (_Float16)((v - (float)g0) * C1B_RS);
            std::memcpy(&out[((((size_t)ch * NT + n) * C1B_NP + 0) * 64 + q * 16 + co16) * 8 + e], &g0, 2);
            std::memcpy(&out[((((size_t)ch * NT + n) * C1B_NP + 1) * 64 + q * 16 + co16) * 8 + e], &g1, 2);
            const float back = (float)g0;
            fits &= (back - back == 0.f);
          }
  return fits ? ATVS_OK : ATVS_ERR_ARG;
}

// Same contract as atvs_conv1x1_f32 except the statistics rows (atvs_conv1x1_b_rows: 128 pixels per workgroup) and the weights
// (atvs_conv1x1_b_pack); split-fp16 operands, fp32-class results.
extern "C" int atvs_conv1x1_b_f32(const float* x, const unsigned char* packed_w, const float* bias, const float* residual,
                                  const float* in_params, int in_relu, float* y, double* stats_partial, int groups, long pixels,
                                  int Cin, int Cout, int ldy, int y_coff, int relu, atvs_stream_t stream) {
  if (!x || !packed_w || !y) return ATVS_ERR_NULL;
  if (groups <= 0 || pixels <= 0 || !atvs_conv1x1_b_supported(Cin, Cout)) return ATVS_ERR_SHAPE;
  if (y_coff < 0 || y_coff + Cout > ldy || (ldy % 4) || (y_coff % 4)) return ATVS_ERR_SHAPE;
  if (residual && y_coff != 0) return ATVS_ERR_ARG;
  if ((double)pixels * Cin >= 2147483648.0) return ATVS_ERR_SHAPE;
  C1bArgs a;
  a.x = x; a.wp = reinterpret_cast<const f16x8*>(packed_w); a.bias = bias; a.res = residual; a.in_params = in_params;
  a.y = y; a.stats = stats_partial;
  a.Cin = Cin; a.Cout = Cout; a.ldy = ldy; a.ycoff = y_coff; a.relu = relu; a.in_relu = in_relu;
  a.rows = pixels; a.wgs = (int)atvs_conv1x1_b_rows(pixels); a.nchunk = Cin / 32;
  hipStream_t s = as_stream(stream);
  int rc;
  if (Cout == 128) rc = launch_c1b<2, 1>(a, groups, s);
  else if (Cout == 64) rc = launch_c1b<1, 1>(a, groups, s);
  else rc = launch_c1b<1, 2>(a, groups, s);
  if (rc) return rc;
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}
