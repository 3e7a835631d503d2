// 3x3 (dilated) stride-1 SAME 2-D convolution of wide feature maps on the 16-bit matrix cores with SPLIT operands (gfx950):
// conv2d_lds.hip's layers (the bottlenecks' conv2, conv0_1 / conv0_2, fusion0 of the 2-D feature towers,
// /root/reference/cnn_wrapper/atvsnet.py:254-292, network.py:198-200,585-587) with every fp32 operand split into TWO fp16
// pieces (x = h0 + h1 / 2048), the three products h0 g0 + (h0 g1 + h1 g0) / 2048 accumulated in fp32 by v_mfma_f32_16x16x32_f16, the
// cross terms in an accumulator of their own (conv_c16b.hip / conv_xb.hip have the arithmetic and its error measurements:
// fp32-class; round 3: three bf16 pieces, six products).
//
// Structure = conv2d_lds.hip (one workgroup per tile of R x 16 pixels of one image, the WAVES SPLIT THE OUTPUT CHANNELS and share
// the pixels, two LDS buffers, the next 16-channel chunk's halo fetched one slot per phase and written after the loop, optional
// normalise-on-load, same epilogue / statistics rows) with: a K = 32 step = two taps x 16 channels (lane half q >> 1 picks the
// tap: 5 steps per chunk, the 10th tap has zero weights); two piece images per buffer (32-byte pixels, row pitch 768 B:
// conflict-free ds_read_b128 without a swizzle); the split done once per staged element on its way into LDS; two phases per
// step (h0 with both weight pieces, h1 with g0), the weight pieces (two per step and output tile, split by the
// host packer) streamed from L2 one step ahead.
#include <cstring>
#include <type_traits>

#include "conv_common.h"

namespace {

constexpr int C2B_PITCH = 24;                 // pixels per LDS row (16 + 2 * 4)
constexpr int C2B_ROWB = C2B_PITCH * 32;      // bytes per row of one piece image (16 channels x 2 B per pixel)
constexpr int C2B_JS = 5;                     // K steps per 16-channel chunk: taps 2 j + (q >> 1) of the 9 (10th = zero)

constexpr int C2B_NP = 2;                                      // operand pieces: x = h0 + h1 / 2048 (conv_c16b.hip, round 4)
constexpr float C2B_RS = 2048.f, C2B_IRS = 1.f / 2048.f;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

struct C2bArgs {
  const float* x;
  const f16x8* wp;
  const float* bias;
  const float* res;
  const float* in_params;
  // TAIL form (a residual unit's conv2 AND conv3 in one launch): y = conv3_1x1(relu(conv2(x) + bias)) + b3 + res
  const f16x8* w3;            // atvs_conv1x1_b_pack of conv3 [Cout][Cout]
  const float* b3;
  float* y;
  double* stats;
  int G, H, W, Cin, Cout;
  int Ho, Wo;                 // output grid (= H, W at stride 1)
  int ldy, ycoff;
  int nchunk;
  int tiles_x, tiles;
  int relu, in_relu;
  long gx, gy;
  long total;
};

__device__ __forceinline__ void c2b_split(const float4& v, f16x4* p0, f16x4* p1) {
  // atvs_split2_f16 (common.h): five vector instructions per two values instead of the 8-9 of the C form, the same values
  uint2 a, b;
  atvs_split4_f16(v, &a, &b);
  *p0 = __builtin_bit_cast(f16x4, a);
  *p1 = __builtin_bit_cast(f16x4, b);
}

// NTW = 16-channel output tiles per wave, WR = row groups across the waves (4 / WR waves split the channels),
// TY = rows per wave, DIL = dilation.  Cout = 16 * NTW * (4 / WR); tile = (TY * WR) rows x 16 columns.
// Workgroup barrier for LDS hand-offs that leaves global loads in flight (bottleneck_b.hip)
__device__ __forceinline__ void c2b_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// TAIL: the 1x1 convolution that follows in a residual unit (conv3, reference cnn_wrapper/network.py:598-601) runs in the same
// launch: r2 = relu(conv2 + bias) leaves the accumulators as fp16 pieces into the (dead) image buffers -- conv3 needs all
// channels of a pixel as K, and they are spread over the four waves -- and each wave multiplies its own output channels; bias b3,
// the shortcut `res` and the moments of the unit's output in the common epilogue.  K order and packed weights of conv1x1_b.hip:
// bit for bit conv2d_b followed by conv1x1_b.
// STRIDE 2 (dilation 1): the strided conv2 of a residual unit's first block (explicit symmetric padding 1 + VALID, taps centred on
// pixel 2 i: reference cnn_wrapper/network.py:588-595, quirk C17).  Halo (2 R + 1) x 33 pixels; even and odd columns of a row
// are stored apart (odd | even halves of the row), so that the 16 lanes of a fragment read -- input columns 2 r + kx -- still
// touch 16 consecutive 32-byte pixels.
template <int NTW, int WR, int TY, int DIL, bool TAIL = false, int STRIDE = 1>
__global__ __launch_bounds__(256, 2) void conv2d_b_kernel(C2bArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  static_assert(STRIDE == 1 || (STRIDE == 2 && DIL == 1 && !TAIL), "stride 2: dilation 1, no tail");
  constexpr int R = TY * WR;
  constexpr int HR = (R - 1) * STRIDE + 1 + 2 * DIL, HC = 15 * STRIDE + 1 + 2 * DIL;
  constexpr int SLOTS = HR * HC * 4;
  constexpr int MAXS = (SLOTS + 255) / 256;
  constexpr int PITCH = (STRIDE == 1) ? C2B_PITCH : 34;          // pixels per LDS row (stride 2: 17 odd + 17 even columns)
  constexpr int ROWB = PITCH * 32;
  constexpr int PIMG = HR * ROWB;              // bytes of one piece image
  constexpr int BUFB = C2B_NP * PIMG;
  constexpr int WN = 4 / WR;
  constexpr int NT = NTW * WN;
  constexpr int JS = C2B_JS;
  static_assert(MAXS <= C2B_NP * JS, "one halo slot per phase");
  static_assert(HC <= PITCH, "row pitch");

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int wn = wave % WN, wr = wave / WN;

  const long per = (p.total + 7) >> 3;
  const long lin = (long)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (lin >= p.total) return;
  const int g = (int)(lin / p.tiles), tile = (int)(lin % p.tiles);
  const int y0 = (tile / p.tiles_x) * R, x0 = (tile % p.tiles_x) * 16;

  // ---- halo slots of this thread: float4 = channels 4 c4 .. of a pixel of the fp32 chunk -> 8 bytes of each piece image
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
    const int gy = y0 * STRIDE - DIL + yy, gxx = x0 * STRIDE - DIL + xx;
    const bool ok = live && (unsigned)gy < (unsigned)p.H && (unsigned)gxx < (unsigned)p.W;
    goff[i] = ok ? ((gy * p.W + gxx) * p.Cin + c4 * 4) : 0;
    // stride 2: halo column xx = 2 x0 - 1 + ... : odd xx are EVEN input offsets; columns of one parity sit together
    const int lx = (STRIDE == 1) ? xx : ((xx & 1) * 17 + (xx >> 1));
    laddr[i] = live ? ((yy * PITCH + lx) * 32 + c4 * 8) : -1;
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
      f16x4 p0, p1;
      c2b_split(v, &p0, &p1);
      unsigned char* d = smem + buf * BUFB + laddr[i];
      *reinterpret_cast<f16x4*>(d) = p0;
      *reinterpret_cast<f16x4*>(d + PIMG) = p1;
    }
  };

  // ---- LDS read offsets: this lane's fragment (8 channels (q & 1) * 8 .. of a pixel) for its tap of step j, row 0 of its row
  // group; taps past 8: zero weights, the 9th tap's fragment
  int bd[JS];
#pragma unroll
  for (int j = 0; j < JS; ++j) {
    const int tap = min(2 * j + (q >> 1), 8);
    const int ky = tap / 3, kx = tap % 3;
    // stride 2: output pixel (t, r) reads halo pixel (2 t + ky, 2 r + kx), stored at column (kx & 1) * 17 + r + (kx >> 1)
    bd[j] = (STRIDE == 1) ? ((wr * TY + ky * DIL) * PITCH + r + kx * DIL) * 32 + (q & 1) * 16
                          : ((wr * TY * 2 + ky) * PITCH + (kx & 1) * 17 + r + (kx >> 1)) * 32 + (q & 1) * 16;
  }

  // ---- packed weight pieces: [K step = chunk * 5 + j][NT tiles][3 pieces][64 lanes] bf16x8, zero steps at the end.  Two
  // register slots, the pieces of step gs + 1 requested at the first phase of step gs (with two workgroups per CU the
  // other wavefront of the SIMD covers what is left of the L2 latency); a chunk has 5 steps, so the slot parity flips from
  // chunk to chunk: the chunk loop is unrolled by two (Cin / 16 is even for every shape the towers have).
  const f16x8* __restrict__ wl = p.wp + (size_t)(wn * NTW) * C2B_NP * 64 + lane;
  constexpr int WSTEP = NT * C2B_NP * 64;
  f16x8 Aw[2][NTW][C2B_NP];
#pragma unroll
  for (int n = 0; n < NTW; ++n)
#pragma unroll
    for (int w3 = 0; w3 < C2B_NP; ++w3) Aw[0][n][w3] = wl[(n * C2B_NP + w3) * 64];

  f32x4 acc[TY][NTW], accx[TY][NTW];   // h0 g0 | (h0 g1 + h1 g0) * 2^11
#pragma unroll
  for (int t = 0; t < TY; ++t)
#pragma unroll
    for (int n = 0; n < NTW; ++n) acc[t][n] = accx[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int i = 0; i < MAXS; ++i) pf_slot(i, 0);
  write_image(0, 0);
  __syncthreads();

  f16x8 Bq[2][TY];
  auto chunk = [&](auto PAR, int ch) __attribute__((always_inline)) {
    constexpr int par = decltype(PAR)::value;                 // ch & 1: LDS buffer and weight-slot parity of the chunk
    const unsigned char* lb = smem + par * BUFB;
    const bool more = ch + 1 < p.nchunk;
    const f16x8* wc = wl + (size_t)ch * JS * WSTEP;
    auto request_b = [&](int ph) __attribute__((always_inline)) {
      const int j = ph / C2B_NP, pc = ph % C2B_NP;
#pragma unroll
      for (int t = 0; t < TY; ++t) Bq[ph & 1][t] = *reinterpret_cast<const f16x8*>(lb + pc * PIMG + bd[j] + t * (STRIDE * ROWB));
    };
    request_b(0);
#pragma unroll
    for (int ph = 0; ph < C2B_NP * JS; ++ph) {
      const int j = ph / C2B_NP, pc = ph % C2B_NP;
      const int slot = (par * JS + j) & 1;
      if (pc == 0) {
#pragma unroll
        for (int n = 0; n < NTW; ++n)
#pragma unroll
          for (int w3 = 0; w3 < C2B_NP; ++w3) Aw[slot ^ 1][n][w3] = wc[(size_t)(j + 1) * WSTEP + (n * C2B_NP + w3) * 64];
      }
      if (ph + 1 < C2B_NP * JS) request_b(ph + 1);
      if (ph < MAXS && more) pf_slot(ph, ch + 1);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < NTW; ++n) {
        if (pc == 0) {
#pragma unroll
          for (int t = 0; t < TY; ++t) acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aw[slot][n][0], Bq[ph & 1][t], acc[t][n], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < TY; ++t) accx[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aw[slot][n][1], Bq[ph & 1][t], accx[t][n], 0, 0, 0);
        } else {
#pragma unroll
          for (int t = 0; t < TY; ++t) accx[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aw[slot][n][0], Bq[ph & 1][t], accx[t][n], 0, 0, 0);
        }
      }
    }
    if (more) {
      write_image(par ^ 1, ch + 1);          // the other buffer: last read in chunk ch - 1, behind the barrier below
      __syncthreads();
    }
  };
  for (int ch = 0; ch < p.nchunk; ch += 2) {
    chunk(std::integral_constant<int, 0>{}, ch);
    chunk(std::integral_constant<int, 1>{}, ch + 1);
  }

  if constexpr (TAIL) {
    constexpr int XP = NT * 32 + 16;           // bytes per pixel of an exchange piece image (16 pixels x 16 B cover all banks)
    constexpr int XIMG = R * 16 * XP;
    // (the launch sizes LDS for the larger of the two image buffers and the two exchange images)
    static_assert(NT % 2 == 0, "conv3 runs in 32-channel chunks");
    // conv3's weight pieces of this wave's output tiles, b3 and the shortcut: requested in front of the exchange
    f16x8 A3[NT / 2][NTW][C2B_NP];
#pragma unroll
    for (int ch = 0; ch < NT / 2; ++ch)
#pragma unroll
      for (int n = 0; n < NTW; ++n)
#pragma unroll
        for (int w3 = 0; w3 < C2B_NP; ++w3) A3[ch][n][w3] = p.w3[((size_t)(ch * NT + wn * NTW + n) * C2B_NP + w3) * 64 + lane];
    float4 b2v[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) b2v[n] = p.bias ? ld4(p.bias + (wn * NTW + n) * 16 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    c2b_lds_barrier();                         // every wave has read its last fragment of the images
#pragma unroll
    for (int t = 0; t < TY; ++t) {
      unsigned char* d = smem + ((wr * TY + t) * 16 + r) * XP + q * 8;
#pragma unroll
      for (int n = 0; n < NTW; ++n) {
        float4 v = make_float4(acc[t][n][0] + accx[t][n][0] * C2B_IRS, acc[t][n][1] + accx[t][n][1] * C2B_IRS,
                               acc[t][n][2] + accx[t][n][2] * C2B_IRS, acc[t][n][3] + accx[t][n][3] * C2B_IRS);
        if (p.bias) { v.x += b2v[n].x; v.y += b2v[n].y; v.z += b2v[n].z; v.w += b2v[n].w; }
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        f16x4 p0, p1;
        c2b_split(v, &p0, &p1);
        *reinterpret_cast<f16x4*>(d + (wn * NTW + n) * 32) = p0;
        *reinterpret_cast<f16x4*>(d + (wn * NTW + n) * 32 + XIMG) = p1;
      }
    }
    c2b_lds_barrier();
#pragma unroll
    for (int t = 0; t < TY; ++t)
#pragma unroll
      for (int n = 0; n < NTW; ++n) acc[t][n] = accx[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ch = 0; ch < NT / 2; ++ch) {
      f16x8 B0[TY], B1[TY];
#pragma unroll
      for (int t = 0; t < TY; ++t) {
        const unsigned char* a = smem + ((wr * TY + t) * 16 + r) * XP + ch * 64 + q * 16;
        B0[t] = *reinterpret_cast<const f16x8*>(a);
        B1[t] = *reinterpret_cast<const f16x8*>(a + XIMG);
      }
#pragma unroll
      for (int n = 0; n < NTW; ++n) {
#pragma unroll
        for (int t = 0; t < TY; ++t) acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A3[ch][n][0], B0[t], acc[t][n], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TY; ++t) accx[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A3[ch][n][1], B0[t], accx[t][n], 0, 0, 0);
      }
#pragma unroll
      for (int n = 0; n < NTW; ++n)
#pragma unroll
        for (int t = 0; t < TY; ++t) accx[t][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A3[ch][n][0], B1[t], accx[t][n], 0, 0, 0);
    }
  }
  const float* ebias = TAIL ? p.b3 : p.bias;
  const bool erelu = TAIL ? false : (p.relu != 0);

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
    if (yo >= p.Ho || xo >= p.Wo) continue;
    const size_t rowb = ((size_t)yo * p.Wo + xo) * p.ldy + p.ycoff;
#pragma unroll
    for (int n = 0; n < NTW; ++n) {
      const int co = (wn * NTW + n) * 16 + 4 * q;
      float4 v = make_float4(acc[t][n][0] + accx[t][n][0] * C2B_IRS, acc[t][n][1] + accx[t][n][1] * C2B_IRS,
                             acc[t][n][2] + accx[t][n][2] * C2B_IRS, acc[t][n][3] + accx[t][n][3] * C2B_IRS);
      if (ebias) {
        const float4 bb = ld4(ebias + co);
        v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
      }
      if (rg) {
        const float4 rr = ld4(rg + rowb + co);
        v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
      }
      if (erelu) {
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
    if (WR > 1 || TAIL) __syncthreads();       // s_red lies over the images / the exchange images
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

template <int NTW, int WR, int TY, int DIL, bool TAIL = false, int STRIDE = 1>
int launch_c2b(const C2bArgs& a, hipStream_t s) {
  constexpr int R = TY * WR, HR = (R - 1) * STRIDE + 1 + 2 * DIL;
  size_t lds = (size_t)2 * C2B_NP * HR * (STRIDE == 1 ? C2B_ROWB : 34 * 32);
  if (TAIL) {
    constexpr size_t xch = (size_t)2 * R * 16 * ((NTW * (4 / WR)) * 32 + 16);
    if (xch > lds) lds = xch;
  }
  const long blocks = ((a.total + 7) / 8) * 8;
  if (blocks > 0x7fffffffL) return ATVS_ERR_SHAPE;
  hipLaunchKernelGGL((conv2d_b_kernel<NTW, WR, TY, DIL, TAIL, STRIDE>), dim3((unsigned)blocks), dim3(256), lds, s, a);
  return ATVS_OK;
}

}  // namespace

// Bytes of the packed form of a TF kernel [3][3][Cin][Cout] for atvs_conv2d_b_f32 (the shapes atvs_conv2d_lds_supported takes).
extern "C" int atvs_conv2d_b_pack_size(int Cin, int Cout, long* packed_bytes) {
  if (!packed_bytes) return ATVS_ERR_NULL;
  if (!atvs_conv2d_lds_supported(Cin, Cout, 1)) return ATVS_ERR_SHAPE;
  *packed_bytes = (long)((Cin / 16) * C2B_JS + 1) * (Cout / 16) * C2B_NP * 1024;
  return ATVS_OK;
}

// HOST function.  packed[K step = chunk * 5 + j][tile n][piece][lane = q * 16 + co16][e] = piece of
// w[tap = 2 j + (q >> 1)][chunk * 16 + (q & 1) * 8 + e][n * 16 + co16] (zero for tap 9); one zero K step of padding at the end (the weight slots read ahead).
extern "C" int atvs_conv2d_b_pack(const float* w, int Cin, int Cout, unsigned char* packed) {
  if (!w || !packed) return ATVS_ERR_NULL;
  long pb;
  int rc = atvs_conv2d_b_pack_size(Cin, Cout, &pb);
  if (rc) return rc;
  std::memset(packed, 0, (size_t)pb);
  uint16_t* out = reinterpret_cast<uint16_t*>(packed);
  const int NT = Cout / 16, nch = Cin / 16;
  bool fits = true;
  for (int ch = 0; ch < nch; ++ch)
    for (int j = 0; j < C2B_JS; ++j)
      for (int n = 0; n < NT; ++n)
        for (int q = 0; q < 4; ++q) {
          const int tap = 2 * j + (q >> 1);
          if (tap > 8) continue;
          for (int co16 = 0; co16 < 16; ++co16)
            for (int e = 0; e < 8; ++e) {
              const int ci = ch * 16 + (q & 1) * 8 + e, co = n * 16 + co16;
              const float v = w[((size_t)tap * Cin + ci) * Cout + co];
              const _Float16 g0 = (_Float16)v, g1 = (_Float16)((v - (float)g0) * C2B_RS);
              std::memcpy(&out[(((((size_t)(ch * C2B_JS + j) * NT + n) * C2B_NP + 0) * 64) + q * 16 + co16) * 8 + e], &g0, 2);
              std::memcpy(&out[(((((size_t)(ch * C2B_JS + j) * NT + n) * C2B_NP + 1) * 64) + q * 16 + co16) * 8 + e], &g1, 2);
              const float back = (float)g0;
              fits &= (back - back == 0.f);
            }
        }
  return fits ? ATVS_OK : ATVS_ERR_ARG;
}

// Same contract as atvs_conv2d_lds_f32 (shapes, statistics rows = atvs_conv2d_lds_rows) with split-fp16 operands; weights from
// atvs_conv2d_b_pack.  fp32-class results; rounding differs from the fp32 MFMA form.
namespace {
int c2b_run(const float* x, const unsigned char* packed_w, const float* bias, const float* residual, const float* in_params,
            int in_relu, const unsigned char* packed_w3, const float* b3, float* y, double* stats_partial, int G, int H, int W,
            int Cin, int Cout, int dilation, int ldy, int y_coff, int relu, atvs_stream_t stream, int stride = 1) {
  if (!x || !packed_w || !y) return ATVS_ERR_NULL;
  if (stride != 1 && (stride != 2 || dilation != 1 || Cout != 64 || packed_w3 || (H % 2) || (W % 2))) return ATVS_ERR_SHAPE;
  if (G <= 0 || H <= 0 || W <= 0 || !atvs_conv2d_lds_supported(Cin, Cout, dilation) || (Cin % 32)) return ATVS_ERR_SHAPE;
  if (y_coff < 0 || y_coff + Cout > ldy || (ldy % 4) || (y_coff % 4)) return ATVS_ERR_SHAPE;
  if (residual && y_coff != 0) return ATVS_ERR_ARG;
  if ((double)H * W * Cin >= 2147483648.0) return ATVS_ERR_SHAPE;
  C2bArgs a;
  a.x = x; a.wp = reinterpret_cast<const f16x8*>(packed_w); a.bias = bias; a.res = residual; a.in_params = in_params;
  a.w3 = reinterpret_cast<const f16x8*>(packed_w3); a.b3 = b3;
  a.y = y; a.stats = stats_partial;
  a.G = G; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.ldy = ldy; a.ycoff = y_coff; a.nchunk = Cin / 16;
  a.Ho = H / stride; a.Wo = W / stride;
  const int R = (Cout == 32) ? 8 : 4;
  a.tiles_x = (a.Wo + 15) / 16;
  a.tiles = ((a.Ho + R - 1) / R) * a.tiles_x;
  a.relu = relu; a.in_relu = in_relu;
  a.gx = (long)H * W * Cin; a.gy = (long)a.Ho * a.Wo * ldy;
  a.total = (long)G * a.tiles;
  hipStream_t s = as_stream(stream);
  int rc = ATVS_ERR_ARG;
  if (stride == 2) {
    rc = launch_c2b<1, 1, 4, 1, false, 2>(a, s);
  } else if (packed_w3) {
    if (dilation == 2) rc = launch_c2b<2, 1, 4, 2, true>(a, s);
    else rc = launch_c2b<2, 1, 4, 4, true>(a, s);
  } else if (Cout == 128) {
    if (dilation == 1) rc = launch_c2b<2, 1, 4, 1>(a, s);
    else if (dilation == 2) rc = launch_c2b<2, 1, 4, 2>(a, s);
    else rc = launch_c2b<2, 1, 4, 4>(a, s);
  } else if (Cout == 64) {
    rc = launch_c2b<1, 1, 4, 1>(a, s);
  } else {
    rc = launch_c2b<1, 2, 4, 1>(a, s);
  }
  if (rc) return rc;
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}
}  // namespace

extern "C" int atvs_conv2d_b_f32(const float* x, const unsigned char* packed_w, const float* bias, const float* residual,
                                 const float* in_params, int in_relu, float* y, double* stats_partial, int G, int H,
                                 int W, int Cin, int Cout, int dilation, int ldy, int y_coff, int relu,
                                 atvs_stream_t stream) {
  return c2b_run(x, packed_w, bias, residual, in_params, in_relu, nullptr, nullptr, y, stats_partial, G, H, W, Cin, Cout, dilation,
                 ldy, y_coff, relu, stream);
}

// A residual unit's conv2 and conv3 in ONE launch (reference cnn_wrapper/network.py:585-601):
//   y = conv3_1x1(relu(conv2_3x3_dil(x) + b2)) + b3 + residual
// x (G,H,W,C) = r1, the unit's conv1 output; packed_w2 = atvs_conv2d_b_pack of [3][3][C][C], packed_w3 = atvs_conv1x1_b_pack of
// [C][C]; residual (G,H,W,C) = the shortcut (identity or projection) or NULL; stats_partial: the moments of y, rows as
// atvs_conv2d_lds_rows.  Built for the 128-channel dilated units (atvs_conv2d_b_tail_supported: C = 128, dilation 2 / 4) -- the
// ones whose conv1 halo does not fit a fully fused unit (atvs_bottleneck_b_f32).  Bit for bit atvs_conv2d_b_f32 followed by
// atvs_conv1x1_b_f32.
extern "C" int atvs_conv2d_b_tail_supported(int C, int dilation) { return (C == 128 && (dilation == 2 || dilation == 4)) ? 1 : 0; }

extern "C" int atvs_conv2d_b_tail_f32(const float* x, const unsigned char* packed_w2, const float* b2,
                                      const unsigned char* packed_w3, const float* b3, const float* residual, float* y,
                                      double* stats_partial, int G, int H, int W, int C, int dilation, atvs_stream_t stream) {
  if (!packed_w3 || !b2 || !b3) return ATVS_ERR_NULL;
  if (!atvs_conv2d_b_tail_supported(C, dilation)) return ATVS_ERR_SHAPE;
  return c2b_run(x, packed_w2, b2, residual, nullptr, 0, packed_w3, b3, y, stats_partial, G, H, W, C, C, dilation, C, 0, 1, stream);
}

// The strided conv2 of a residual unit's first block (reference cnn_wrapper/network.py:588-595: explicit symmetric padding 1,
// VALID, stride 2 -- taps centred on input pixel 2 i, quirk C17) on the split-operand kernel: x (G,H,W,Cin) with H, W even ->
// y (G,H/2,W/2,Cout).  Built for 64 output channels (conv1_x_0/conv2 of ResNetDS2SPP; atvs_conv2d_b_s2_supported); weights
// atvs_conv2d_b_pack; stats_partial rows = atvs_conv2d_lds_rows(H/2, W/2, Cout).  Same arithmetic as atvs_conv2d_b_f32.
extern "C" int atvs_conv2d_b_s2_supported(int Cin, int Cout) { return (Cout == 64 && Cin > 0 && Cin % 32 == 0 && Cin <= 512) ? 1 : 0; }

extern "C" int atvs_conv2d_b_s2_f32(const float* x, const unsigned char* packed_w, const float* bias, float* y,
                                    double* stats_partial, int G, int H, int W, int Cin, int Cout, int relu, atvs_stream_t stream) {
  if (!atvs_conv2d_b_s2_supported(Cin, Cout)) return ATVS_ERR_SHAPE;
  return c2b_run(x, packed_w, bias, nullptr, nullptr, 0, nullptr, nullptr, y, stats_partial, G, H, W, Cin, Cout, 1, Cout, 0, relu,
                 stream, 2);
}
