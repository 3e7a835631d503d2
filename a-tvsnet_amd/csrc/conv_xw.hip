// 3x3x3 SAME stride-1 convolution to EIGHT output channels: x-pair rows x Winograd F(2,3) along y, one wavefront per
// (conv_xp.hip, referred to below, was this kernel's direct fp32 predecessor -- round 2, removed from the tree in round 4: git history.)
// SIMD (gfx950).  Successor of conv_xp.hip for the widest layers of the stacked U-Nets / the refinement net
// (conv_b*_0_1, global_refine_3dconv0_1, the photo stem: cnn_wrapper/atvsnet.py StackedUNet / CostVolRefineNet, layer
// code /root/reference/cnn_wrapper/network.py:165-215).
//
// Why.  These layers are MFMA-bound (PMC, profiles/round3_pmc_baseline.json: fp32 MFMA pipe 80 % busy in conv_xp) and 8
// output channels fill half of a 16-row MFMA tile.  conv_xp pairs two x-adjacent voxels on the rows, which makes 3/4 of
// the issued MFMAs useful (12 tap-steps per voxel pair and (kd, kh) instead of the 6 a full tile would need... 36 K
// steps per 16-channel chunk for a 32-voxel row).  The y axis is still free: with the minimal-filtering form F(2,3)
//     t = [d0 - d2, d1 + d2, d2 - d1, d1 - d3],   U = [g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2],
//     m_p = sum U_p * t_p,   y0 = m0 + m1 + m2,   y1 = m1 - m2 - m3
// two output rows cost 4 products per (kd, x offset, channel) instead of 6: 2/3 of the MFMAs of conv_xp for the same
// outputs.  U is formed on the host in double and rounded once; t is formed in registers (v_pk_add_f32) from the raw
// image rows, so the LDS image stays the raw input and the stride-2 sibling convolution can still be computed from it.
// Numerics (tools_dev/winograd_emulation.py, CPU emulation patched into the oracle): one 32 -> 8 layer 7.7e-7 of the
// output maximum against 4.2e-7 for the direct fp32 sum; final depth map of the two-view 640x512x192 pipeline 4.9e-4
// from the float64-network evaluation against 4.3e-4 for the direct fp32 oracle.
//
// Structure (differences from conv_xp.hip): the input is staged in 8-channel chunks (32-byte voxels, no bank swizzle
// needed) into a DOUBLE-BUFFERED image of 2 x 76.8 KB, so the next (tile, chunk) stage's halo goes from registers to
// the other buffer a few K steps after it was requested -- no register-resident halo (conv_xp keeps 32 float4 per
// thread), one barrier per stage, and the registers pay for the 16 accumulator tiles (4 products x 4 row pairs) a
// wavefront needs.  Tile 4(z) x 8(y) x 32(x); wavefront w owns plane z0 + w.  K step j = (kd, xh, p): x offset
// xl = 2 xh + (q >> 1) of the pair window, channel group q & 1, product p; per (kd, xh) the wavefront reads the 10 halo
// rows once (10 ds_read_b128) and feeds 4 products x 4 row pairs x 4 = 64 MFMAs from them.
#include <type_traits>

#include "conv_common.h"

// Development build (-DATVS_XW_DEBUG): per-wavefront cycle counts of the phases, read back with atvs_debug_read_xw
// (tools_dev/phase_xw.py).
#ifdef ATVS_XW_DEBUG
__device__ unsigned long long atvs_dbg_xw[4096 * 8];
extern "C" int atvs_debug_read_xw(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(atvs_dbg_xw), sizeof(atvs_dbg_xw));
}
#define XDBG(i) { unsigned long long t_ = clock64(); dbg_acc[i] += t_ - dbg_t; dbg_t = t_; }
#else
#define XDBG(i)
#endif

namespace {

constexpr int XW_TZ = 4, XW_TY = 8, XW_TXV = 32;
constexpr int XW_HZ = XW_TZ + 2, XW_HY = XW_TY + 2, XW_HX = XW_TXV + 2;
constexpr int XW_HXP = 40;       // voxels per image row: even x in columns 0..16, odd x in columns XW_ODD..XW_ODD+16
constexpr int XW_ODD = 18;
constexpr int XW_VB = 32;        // bytes per voxel of an 8-channel chunk
constexpr int XW_ROWB = XW_HXP * XW_VB;                     // 1280
constexpr int XW_IMG = XW_HZ * XW_HY * XW_ROWB;             // 76,800 bytes per buffer
constexpr int XW_SLOTS = XW_HZ * XW_HY * XW_HX * 2;         // 16-byte halo slots per stage
constexpr int XW_MAXS = (XW_SLOTS + 255) / 256;             // 16 per thread
constexpr int XW_NP = XW_TY / 2;                            // row pairs
constexpr int XW_JC = 3 * 2 * 4;                            // main K steps per chunk: (kd, xh, product)
constexpr int XW_J2 = 14;                                   // sibling K steps per chunk: taps 2 i + (q >> 1) of the 27 (28th = zero)
#ifndef XW_LOOK_V
#define XW_LOOK_V 4
#endif
#ifndef XW_LAG_V
#define XW_LAG_V 6
#endif
constexpr int XW_LOOK = XW_LOOK_V;                          // weight look-ahead in K steps
constexpr int XW_LAG = XW_LAG_V;                            // K steps between a halo slot's request and its LDS write
static_assert(XW_LOOK <= XW_J2 && XW_MAXS - 1 + XW_LAG < XW_JC, "every halo slot is written inside the main K loop");
static_assert((2 * XW_HY + XW_HY) * XW_ROWB < 65536, "ds_read immediate offset");

struct XwArgs {
  const float* x;
  const float* wp;       // packed weights, see atvs_conv_xw_pack
  const float* zeros;    // 16 bytes of zeros (tail of the packed weights): source of the zero padding
  const float* bias;
  const float* pbias;    // (H, W, 24) or nullptr
  float* y;
  double* stats;
  int Di, Hi, Wi, Cin;
  int ldy, ycoff;
  int nchunk;
  int tiles_y, tiles_x, ntiles;
  int relu;
  const float* wp2;      // packed sibling weights (atvs_conv_xw_pack_sibling) or nullptr
  const float* pbias2;   // (Ho2, Wo2, 48) or nullptr
  float* y2;
  double* stats2;
  int Do2, Ho2, Wo2, ldy2, ycoff2;
  int pbz, pby, pbx;     // SAME padding in front of each axis of the stride-2 sibling (0 or 1)
  int wg;
  long gx, gy, gpb, gy2, gpb2;
  const float* x2;       // prologue, as in conv_xp.hip
  const float* in_pa;
  const float* in_pb;
  int relu_a, relu_b;
  int sample_major;
  // layout of x (and x2): floats between voxels / between 8-channel chunks.  Channel-last (D,H,W,Cin): Cin / 8;
  // chunk-planar [Cin/8][D][H][W][8] (x_planar: what the plane-sweep warp writes for this kernel -- a chunk's halo rows are
  // then dense 32-byte voxels instead of 32 of every 128 bytes): 8 / D*H*W*8
  int vstride;
  long cstride;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

// a - b.  hipcc has no packed form for an f32x2 subtraction (no v_pk_sub_f32; it emits two v_sub_f32).  An inline-asm
// v_pk_add_f32 with neg modifiers is NOT usable here: with it the first row pair of wavefronts 1-3 came out wrong whenever
// the MFMA that reads the result followed closely (the hazard recogniser / scheduler do not see through inline asm) --
// found by tests/test_gpu_conv.py::test_conv_siblings_one_launch, kept out on purpose.
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) { return a - b; }
__device__ __forceinline__ float4 f4sub(const float4& a, const float4& b) {
  const f32x2 lo = pk_sub((f32x2){a.x, a.y}, (f32x2){b.x, b.y}), hi = pk_sub((f32x2){a.z, a.w}, (f32x2){b.z, b.w});
  return make_float4(lo.x, lo.y, hi.x, hi.y);
}
__device__ __forceinline__ float4 f4add(const float4& a, const float4& b) {
  const f32x2 lo = (f32x2){a.x, a.y} + (f32x2){b.x, b.y}, hi = (f32x2){a.z, a.w} + (f32x2){b.z, b.w};
  return make_float4(lo.x, lo.y, hi.x, hi.y);
}

// SIB: additionally the stride-2 sibling (16 output channels) from the same staged image: 2(z) x 4(y) x 16(x) outputs per
// tile, wavefront w owns output plane w >> 1, rows 2 (w & 1), 2 (w & 1) + 1; K step i carries the two taps 2 i + (q >> 1)
// of the 27 (kd, kh, kw) (tap 27 = zero weights), channel group q & 1: 14 steps instead of the 18 of a (kd, kh, kw pair)
// enumeration.
// PRO: the convolution's input is act_a(bn_a(x)) [+ act_b(bn_b(x2))], formed between a halo slot's arrival and its LDS
// write (0 none, 1 one source, 2 two sources).
template <bool SIB, int PRO>
__global__ __launch_bounds__(256, 1) void conv_xw_kernel(XwArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int MAXS = XW_MAXS, JC = XW_JC, J2 = SIB ? XW_J2 : 0, VB = XW_VB, ROWB = XW_ROWB;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;

  // LDS read base of this lane: its fragment of halo row 0 of the wavefront's first plane (kd = 0) at x offset
  // xl = (q >> 1) (xh = 0); everything else is an immediate: (kd * HY + halo row) * ROWB + xh * VB
  const int rbase = ((wave * XW_HY) * XW_HXP + (q >> 1) * XW_ODD + r) * VB + (q & 1) * 16;
  // sibling read bases.  Lane half q >> 1 = h reads tap 2 i + h; with m = (2 i) % 3 its kw is (m + h) % 3 and its (kd, kh)
  // row differs from the even tap's by 0 rows, by 1 row (m == 2: kh + 1) or by HY - 2 rows (tap 8 -> 9: kd + 1, kh 2 -> 0):
  // five per-lane bases cover every step, the rest are immediates.
  int pb2[5] = {0, 0, 0, 0, 0};        // m = 0, m = 1, m = 2 (+1 row for the odd half), m = 2 (+HY-2 rows), last step
  if (SIB) {
    const int h = q >> 1;
    const int row0 = (2 * (wave >> 1) + 1 - p.pbz) * XW_HY + (4 * (wave & 1) + 1 - p.pby);
    auto colbase = [&](int kw) __attribute__((always_inline)) {
      const int xh = kw + 1 - p.pbx;                                  // halo x of output column 0
      return (row0 * XW_HXP + (xh & 1) * XW_ODD + (xh >> 1) + r) * VB + (q & 1) * 16;
    };
    pb2[0] = colbase(h ? 1 : 0);
    pb2[1] = colbase(h ? 2 : 1);
    pb2[2] = colbase(h ? 0 : 2) + h * ROWB;
    pb2[3] = colbase(h ? 0 : 2) + h * (XW_HY - 2) * ROWB;
    pb2[4] = colbase(2);               // tap 26 | tap 27 (zero weights): the odd half re-reads tap 26's fragment
  }

  // per-slot constants of this thread (conv_xp.hip): global element offset from the halo origin, LDS byte address, packed
  // halo coordinate for the bounds test
  int goff[MAXS], laddr[MAXS];
  unsigned pg[MAXS];
#pragma unroll
  for (int i = 0; i < MAXS; ++i) {
    int s = tid + i * 256;
    const bool live = s < XW_SLOTS;
    s = min(s, XW_SLOTS - 1);
    const int c4 = s & 1, v = s >> 1;
    const int xx = v % XW_HX, v2 = v / XW_HX;
    const int yy = v2 % XW_HY, zz = v2 / XW_HY;
    goff[i] = ((zz * p.Hi + yy) * p.Wi + xx) * p.vstride + c4 * 4;
    laddr[i] = ((zz * XW_HY + yy) * XW_HXP + (xx & 1) * XW_ODD + (xx >> 1)) * VB + c4 * 16;
    pg[i] = 0x808080u | (unsigned)(live ? zz : 0x7f) | ((unsigned)yy << 8) | ((unsigned)xx << 16);
  }
  const bool last_live = tid + (MAXS - 1) * 256 < XW_SLOTS;

  // persistent tile list (conv_xp.hip)
  const int G = p.wg;
  const int grp = p.sample_major ? (int)(blockIdx.x & 7) : (int)(blockIdx.x / p.wg);
  const int lbk = p.sample_major ? (int)(blockIdx.x >> 3) : (int)(blockIdx.x - grp * p.wg);
  const int xcd = p.sample_major ? 0 : (lbk & 7), tslot = p.sample_major ? lbk : (lbk >> 3);
  const unsigned srow = (unsigned)(grp * p.wg + lbk);
  const float* __restrict__ xg = p.x + (size_t)grp * p.gx;
  const float* __restrict__ xg2 = (PRO == 2) ? p.x2 + (size_t)grp * p.gx : nullptr;
  const float* __restrict__ ipa = (PRO >= 1 && p.in_pa) ? p.in_pa + (size_t)grp * 3 * p.Cin : nullptr;
  const float* __restrict__ ipb = (PRO == 2 && p.in_pb) ? p.in_pb + (size_t)grp * 3 * p.Cin : nullptr;
  const int c4t = tid & 1;                        // every slot of this thread is channel group c4t of the chunk
  float* __restrict__ yg = p.y + (size_t)grp * p.gy;
  float* __restrict__ y2g = p.y2 + (size_t)grp * p.gy2;
  const float* __restrict__ pbg = p.pbias ? p.pbias + (size_t)grp * p.gpb : nullptr;
  const float* __restrict__ pb2g = p.pbias2 ? p.pbias2 + (size_t)grp * p.gpb2 : nullptr;
  const int per_xcd = p.sample_major ? p.ntiles : ((p.ntiles + 7) >> 3);
  const int slots_per_xcd = p.sample_major ? G : (G >> 3);
  int my_tiles = 0;
  {
    int last = min(per_xcd, p.ntiles - xcd * per_xcd);
    if (tslot < last) my_tiles = (last - tslot + slots_per_xcd - 1) / slots_per_xcd;
  }
  const int nstage = my_tiles * p.nchunk;

  auto tile_origin = [&](int k, int* z0, int* y0, int* x0) __attribute__((always_inline)) {
    int tl = xcd * per_xcd + tslot + k * slots_per_xcd;
    int bx = tl % p.tiles_x;
    int rest = tl / p.tiles_x;
    *x0 = bx * XW_TXV;
    *y0 = (rest % p.tiles_y) * XW_TY;
    *z0 = (rest / p.tiles_y) * XW_TZ;
  };

  struct PfTile {
    const float* xb;
    const float* xb2;
    int org;
    unsigned lo, hi1;
  };
  auto pf_tile = [&](int stage) __attribute__((always_inline)) {
    PfTile T;
    int k = stage / p.nchunk, ch = stage - k * p.nchunk;
    int z0, y0, x0;
    tile_origin(k, &z0, &y0, &x0);
    const int gz0 = z0 - 1, gy0 = y0 - 1, gx0 = x0 - 1;
    T.xb = xg + (size_t)ch * p.cstride;
    T.xb2 = (PRO == 2) ? xg2 + (size_t)ch * p.cstride : nullptr;
    T.org = ((gz0 * p.Hi + gy0) * p.Wi + gx0) * p.vstride;
    T.lo = (unsigned)(gz0 < 0) | ((unsigned)(gy0 < 0) << 8) | ((unsigned)(gx0 < 0) << 16);
    T.hi1 = (unsigned)(min(p.Di - 1 - gz0, 0x7e) + 1) | ((unsigned)(min(p.Hi - 1 - gy0, 0x7e) + 1) << 8) |
            ((unsigned)(min(p.Wi - 1 - gx0, 0x7e) + 1) << 16);
    return T;
  };
  float4 pf[MAXS];
  float4 pf2[PRO == 2 ? MAXS : 1];
  unsigned vmask = 0;
  auto pf_slot = [&](const PfTile& T, int i) __attribute__((always_inline)) {
    const unsigned t1 = pg[i] - T.lo;
    const unsigned t2 = T.hi1 + ~pg[i];
    const bool ok = ((t1 & t2) & 0x808080u) == 0x808080u;
    pf[i] = ld4(ok ? (T.xb + (T.org + goff[i])) : p.zeros);
    if (PRO >= 1) vmask = (vmask & ~(1u << i)) | ((ok ? 1u : 0u) << i);
    if (PRO == 2) pf2[i] = ld4(ok ? (T.xb2 + (T.org + goff[i])) : p.zeros);
  };

  // prologue transform: arithmetic of bn_apply / bn_add (norm.hip), identical to conv_xp.hip's
  struct Par { float4 ma, sa, ba, mb, sb, bb; };
  const bool has_a = PRO >= 1 && ipa != nullptr, has_b = PRO == 2 && ipb != nullptr;
  const float floor_a = p.relu_a ? 0.f : -__builtin_huge_valf(), floor_b = p.relu_b ? 0.f : -__builtin_huge_valf();
  auto load_par = [&](int chunk) __attribute__((always_inline)) {
    Par P;
    const int cch = chunk * 8 + c4t * 4;
    const int sa = has_a ? p.Cin : 0, sb = has_b ? p.Cin : 0;
    const float* a = has_a ? ipa + cch : p.zeros;
    P.ma = ld4(a); P.sa = ld4(a + sa); P.ba = ld4(a + 2 * sa);
    if (PRO == 2) {
      const float* b = has_b ? ipb + cch : p.zeros;
      P.mb = ld4(b); P.sb = ld4(b + sb); P.bb = ld4(b + 2 * sb);
    }
    return P;
  };
  auto bn2 = [&](f32x2 v, f32x2 m, f32x2 sc, f32x2 be, float lo, bool has) __attribute__((always_inline)) {
    f32x2 t = __builtin_elementwise_fma(v, sc, __builtin_elementwise_fma(-m, sc, be));      // atvs_bn1 (common.h) on a pair
    t.x = fmaxf(t.x, lo);
    t.y = fmaxf(t.y, lo);
    t.x = has ? t.x : v.x;
    t.y = has ? t.y : v.y;
    return t;
  };
  auto xform = [&](int i, const Par& P) __attribute__((always_inline)) {
    const bool ok = (vmask >> i) & 1u;
    f32x2 lo = {pf[i].x, pf[i].y}, hi = {pf[i].z, pf[i].w};
    lo = bn2(lo, (f32x2){P.ma.x, P.ma.y}, (f32x2){P.sa.x, P.sa.y}, (f32x2){P.ba.x, P.ba.y}, floor_a, has_a);
    hi = bn2(hi, (f32x2){P.ma.z, P.ma.w}, (f32x2){P.sa.z, P.sa.w}, (f32x2){P.ba.z, P.ba.w}, floor_a, has_a);
    if (PRO == 2) {
      f32x2 ul = {pf2[i].x, pf2[i].y}, uh = {pf2[i].z, pf2[i].w};
      ul = bn2(ul, (f32x2){P.mb.x, P.mb.y}, (f32x2){P.sb.x, P.sb.y}, (f32x2){P.bb.x, P.bb.y}, floor_b, has_b);
      uh = bn2(uh, (f32x2){P.mb.z, P.mb.w}, (f32x2){P.sb.z, P.sb.w}, (f32x2){P.bb.z, P.bb.w}, floor_b, has_b);
      lo += ul;
      hi += uh;
    }
    pf[i] = make_float4(ok ? lo.x : 0.f, ok ? lo.y : 0.f, ok ? hi.x : 0.f, ok ? hi.y : 0.f);
  };

  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias) bv = ld4(p.bias + (q & 1) * 4);
  f32x4 acc[4][XW_NP];                 // [product][row pair]
  f32x4 acc2[2];
  float ssum2[4] = {0.f, 0.f, 0.f, 0.f}, ssq2[4] = {0.f, 0.f, 0.f, 0.f};
  const float4* __restrict__ wp = reinterpret_cast<const float4*>(p.wp);
  const float4* __restrict__ wp2 = reinterpret_cast<const float4*>(p.wp2);

  // weights of a stage's first XW_LOOK steps: requested before the barrier in front of it
  float4 wpre[XW_LOOK];
#pragma unroll
  for (int jj = 0; jj < XW_LOOK; ++jj) wpre[jj] = wp[jj * 64 + lane];

  // ---- stage 0: its halo as one burst into buffer 0
  if (nstage > 0) {
    const PfTile T0 = pf_tile(0);
#pragma unroll
    for (int i = 0; i < MAXS; ++i) pf_slot(T0, i);
    if (PRO >= 1) {
      const Par P0 = load_par(0);
#pragma unroll
      for (int i = 0; i < MAXS; ++i) xform(i, P0);
    }
#pragma unroll
    for (int i = 0; i < MAXS; ++i)
      if (i < MAXS - 1 || last_live) *reinterpret_cast<float4*>(smem + laddr[i]) = pf[i];
  }
  __syncthreads();

#ifdef ATVS_XW_DEBUG
  unsigned long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long dbg_t = clock64();
#endif
  for (int stage = 0; stage < nstage; ++stage) {
    const int k = stage / p.nchunk, ch = stage - k * p.nchunk;
    XDBG(0)
    if (ch == 0) {
#pragma unroll
      for (int pp = 0; pp < 4; ++pp)
#pragma unroll
        for (int i = 0; i < XW_NP; ++i) acc[pp][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc2[0] = acc2[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const float4* wch = wp + (size_t)ch * JC * 64 + lane;
    const float4* wch2 = wp2 + (size_t)ch * J2 * 64 + lane;
    const float4* wnx = wp + (size_t)((ch + 1 == p.nchunk) ? 0 : ch + 1) * JC * 64 + lane;      // the next stage's chunk
    float4 w[JC];
#pragma unroll
#ifdef XW_NOPRE
    for (int jj = 0; jj < XW_LOOK; ++jj) w[jj] = wch[jj * 64];
#else
    for (int jj = 0; jj < XW_LOOK; ++jj) w[jj] = wpre[jj];         // requested during the previous stage
#endif

    const unsigned char* rd = smem + (stage & 1) * XW_IMG;              // this stage's image
    unsigned char* wr = smem + ((stage + 1) & 1) * XW_IMG;              // the next stage's
    const PfTile T = pf_tile(min(stage + 1, nstage - 1));               // last stage: harmless re-read of its own halo
    const bool last_chunk = (ch == p.nchunk - 1);
    int tz0, ty0, tx0;
    tile_origin(k, &tz0, &ty0, &tx0);
    const int zo = tz0 + wave, xo = tx0 + 2 * r + (q >> 1), co = (q & 1) * 4;
    const bool evox_ok = zo < p.Di && xo < p.Wi;
    const size_t erow = (size_t)p.Wi * p.ldy;
    const size_t eo = (((size_t)zo * p.Hi + ty0) * p.Wi + xo) * (size_t)p.ldy + p.ycoff + co;
    const size_t epb_off = ((size_t)ty0 * p.Wi + xo) * 24 + plane_variant(zo - 1, p.Di) * 8 + co;
    auto erow_ok = [&](int t) __attribute__((always_inline)) { return evox_ok && ty0 + t < p.Hi; };
    float4 epb[XW_TY], epb2[2];      // depth-plane biases of the epilogue: requested near the end of the main K loop (every
                                     // stage: from the 16 zero bytes when there is none -- no live range before)
    const int zo2 = (tz0 >> 1) + (wave >> 1), yo2 = (ty0 >> 1) + 2 * (wave & 1), xo2 = (tx0 >> 1) + r;
    const bool evox2_ok = SIB && zo2 < p.Do2 && xo2 < p.Wo2;
    auto erow2_ok = [&](int t) __attribute__((always_inline)) { return evox2_ok && yo2 + t < p.Ho2; };
    const size_t erow2 = (size_t)p.Wo2 * p.ldy2;
    const size_t eo2 = (((size_t)zo2 * p.Ho2 + yo2) * p.Wo2 + xo2) * (size_t)p.ldy2 + p.ycoff2 + 4 * q;

    // raw image rows of group g = (kd, xh), even and odd halo rows apart: product 0 reads the even rows only, product 3 the
    // odd rows only, so the next group's even rows are requested during product 2 (into the registers the group before
    // last used) and its odd rows at its own product 0 -- 15 instead of 20 row fragments live at the peak
    float4 Re[2][XW_HY / 2], Ro[2][XW_HY / 2], b2[2][2];
    auto request_R = [&](int g, int par) __attribute__((always_inline)) {
      const int off = (g >> 1) * XW_HY * ROWB + (g & 1) * VB;
#pragma unroll
      for (int hr = 0; hr < XW_HY / 2; ++hr) {
        const float4 v = *reinterpret_cast<const float4*>(rd + rbase + (off + (2 * hr + par) * ROWB));
        if (par) Ro[g & 1][hr] = v;
        else Re[g & 1][hr] = v;
      }
    };
    auto request_b2 = [&](int i) __attribute__((always_inline)) {
      const int ta = 2 * i, m = ta % 3;                               // the even tap of the step
      const int rowoff = (ta / 9) * XW_HY + (ta / 3) % 3;
      const int pbi = (m < 2) ? m : ((ta == 8) ? 3 : ((ta == 26) ? 4 : 2));
#pragma unroll
      for (int t = 0; t < 2; ++t)
        b2[i & 1][t] = *reinterpret_cast<const float4*>(rd + pb2[pbi] + (rowoff + 2 * t) * ROWB);
    };
    float4 w2[SIB ? J2 : 1];
    Par Pn;
    if (PRO >= 1) Pn = load_par(min(stage + 1, nstage - 1) % p.nchunk);
    XDBG(1)
    // Main K loop.  (Tried, same time within the run-to-run spread: the transform formed one step ahead inside the previous
    // step's MFMA region, with and without sched_group_barrier interleaving; plain v_sub_f32 instead of packed ops: 10 %
    // slower.  The loop runs at ~75 % MFMA density: ~1.2 other VALU instructions per MFMA, 0.4 of them the transform.)
    request_R(0, 0);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < JC; ++j) {
      const int g = j >> 2, pp = j & 3;
      if (j + XW_LOOK < JC) w[j + XW_LOOK] = wch[(j + XW_LOOK) * 64];
      else if (SIB) w2[j + XW_LOOK - JC] = wch2[(j + XW_LOOK - JC) * 64];
      else wpre[j + XW_LOOK - JC] = wnx[(j + XW_LOOK - JC) * 64];
      if (pp == 0) request_R(g, 1);
      if (pp == 2) {
        if (g + 1 < 6) request_R(g + 1, 0);
        else if (SIB) request_b2(0);
      }
      if (j < MAXS) pf_slot(T, j);
      if (!SIB && j == JC - 4) {
        const bool use = last_chunk && pbg;
#pragma unroll
        for (int t = 0; t < XW_TY; ++t) epb[t] = ld4((use && erow_ok(t)) ? pbg + (epb_off + (size_t)t * p.Wi * 24) : p.zeros);
      }
      // this product's B operands: the F(2,3) input transform along y of the raw rows (halo rows 2i .. 2i+3 feed row pair i)
      float4 t[XW_NP];
#pragma unroll
      for (int i = 0; i < XW_NP; ++i) {
        const float4 &d0 = Re[g & 1][i], &d1 = Ro[g & 1][i], &d2 = Re[g & 1][i + 1], &d3 = Ro[g & 1][i + 1];
        t[i] = (pp == 0) ? f4sub(d0, d2) : (pp == 1) ? f4add(d1, d2) : (pp == 2) ? f4sub(d2, d1) : f4sub(d1, d3);
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < XW_NP; ++i)
          acc[pp][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(w[j], s), f4get(t[i], s), acc[pp][i], 0, 0, 0);
      if (j >= XW_LAG && j - XW_LAG < MAXS) {
        // the slot requested XW_LAG steps ago has arrived: (transform,) write it into the next stage's image
        const int i = j - XW_LAG;
        if (PRO >= 1) xform(i, Pn);
        if (i < MAXS - 1 || last_live) *reinterpret_cast<float4*>(wr + laddr[i]) = pf[i];
      }
    }
    XDBG(2)
    if (SIB) {
#pragma unroll
      for (int i = 0; i < J2; ++i) {
        if (i + XW_LOOK < J2) w2[i + XW_LOOK] = wch2[(i + XW_LOOK) * 64];
        else wpre[i + XW_LOOK - J2] = wnx[(i + XW_LOOK - J2) * 64];
        if (i + 1 < J2) request_b2(i + 1);
        if (i == J2 - 4) {        // the epilogues' depth-plane biases: behind the last weight request of the stage (vmcnt retires in order)
          const bool use = last_chunk && pbg;
#pragma unroll
          for (int t = 0; t < XW_TY; ++t) epb[t] = ld4((use && erow_ok(t)) ? pbg + (epb_off + (size_t)t * p.Wi * 24) : p.zeros);
          const bool use2 = last_chunk && pb2g;
          const size_t o = ((size_t)yo2 * p.Wo2 + xo2) * 48 + plane_variant(2 * zo2 - p.pbz, p.Di) * 16 + 4 * q;
#pragma unroll
          for (int t = 0; t < 2; ++t) epb2[t] = ld4((use2 && erow2_ok(t)) ? pb2g + (o + (size_t)t * p.Wo2 * 48) : p.zeros);
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(w2[i], s), f4get(b2[i & 1][t], s), acc2[t], 0, 0, 0);
      }
    }
    XDBG(3)
    if (last_chunk) {
      // ---- epilogue: the F(2,3) output transform, then as conv_xp.hip (this lane holds channels co..co+3 of voxel xo for
      // the 8 rows of plane zo)
      auto store_rows = [&](auto relu_tag) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < XW_TY; ++t) {
          if (!erow_ok(t)) continue;
          const int i = t >> 1;
          f32x4 m;
          if ((t & 1) == 0) m = (acc[0][i] + acc[1][i]) + acc[2][i];
          else m = (acc[1][i] - acc[2][i]) - acc[3][i];
          float4 v;
          v.x = (m[0] + bv.x) + epb[t].x;
          v.y = (m[1] + bv.y) + epb[t].y;
          v.z = (m[2] + bv.z) + epb[t].z;
          v.w = (m[3] + bv.w) + epb[t].w;
          if (decltype(relu_tag)::value) {          // NaN passes through, as in tf.nn.relu
            v.x = (v.x < 0.f) ? 0.f : v.x;
            v.y = (v.y < 0.f) ? 0.f : v.y;
            v.z = (v.z < 0.f) ? 0.f : v.z;
            v.w = (v.w < 0.f) ? 0.f : v.w;
          }
          st4(yg + (eo + (size_t)t * erow), v);
          ssum[0] += v.x; ssum[1] += v.y; ssum[2] += v.z; ssum[3] += v.w;
          ssq[0] += v.x * v.x; ssq[1] += v.y * v.y; ssq[2] += v.z * v.z; ssq[3] += v.w * v.w;
        }
      };
      if (p.relu) store_rows(std::true_type{});
      else store_rows(std::false_type{});
      if (SIB) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if (!erow2_ok(t)) continue;
          const float4 v = make_float4(acc2[t][0] + epb2[t].x, acc2[t][1] + epb2[t].y, acc2[t][2] + epb2[t].z, acc2[t][3] + epb2[t].w);
          st4(y2g + (eo2 + (size_t)t * erow2), v);
          ssum2[0] += v.x; ssum2[1] += v.y; ssum2[2] += v.z; ssum2[3] += v.w;
          ssq2[0] += v.x * v.x; ssq2[1] += v.y * v.y; ssq2[2] += v.z * v.z; ssq2[3] += v.w * v.w;
        }
      }
    }
    XDBG(4)
    __syncthreads();      // this stage's image is read, the next stage's is written
    XDBG(5)
  }
#ifdef ATVS_XW_DEBUG
  if (lane == 0 && blockIdx.x < 1024) {
    dbg_acc[7] = (unsigned long long)nstage;
    for (int i = 0; i < 8; ++i) atvs_dbg_xw[(blockIdx.x * 4 + wave) * 8 + i] = dbg_acc[i];
  }
#endif

  // ---- per-workgroup partial moments -> row `srow` of stats: [2][16] doubles (columns 0..7 = channels), as conv_xp.hip
  if (p.stats) {
    double* s_red = reinterpret_cast<double*>(smem);   // [4 waves][2][8]
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      double a = (double)ssum[kk], bq = (double)ssq[kk];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        a += __shfl_xor(a, o);
        bq += __shfl_xor(bq, o);
      }
      a += __shfl_xor(a, 32);      // lanes q and q^2 hold the same channels (the two x parities)
      bq += __shfl_xor(bq, 32);
      if (r == 0 && q < 2) {
        s_red[(wave * 2 + 0) * 8 + q * 4 + kk] = a;
        s_red[(wave * 2 + 1) * 8 + q * 4 + kk] = bq;
      }
    }
    __syncthreads();
    if (tid < 32) {
      const int which = tid >> 4, col = tid & 15;
      double v = 0.0;
      if (col < 8)
        v = (s_red[(0 * 2 + which) * 8 + col] + s_red[(1 * 2 + which) * 8 + col]) +
            (s_red[(2 * 2 + which) * 8 + col] + s_red[(3 * 2 + which) * 8 + col]);
      p.stats[((size_t)srow * 2 + which) * 16 + col] = v;
    }
  }
  if (SIB && p.stats2) {
    __syncthreads();
    double* s_red = reinterpret_cast<double*>(smem);   // [4 waves][2][16]
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      double a = (double)ssum2[kk], bq = (double)ssq2[kk];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        a += __shfl_xor(a, o);
        bq += __shfl_xor(bq, o);
      }
      if (r == 0) {
        s_red[(wave * 2 + 0) * 16 + q * 4 + kk] = a;
        s_red[(wave * 2 + 1) * 16 + q * 4 + kk] = bq;
      }
    }
    __syncthreads();
    if (tid < 32) {
      const int which = tid >> 4, col = tid & 15;
      p.stats2[((size_t)srow * 2 + which) * 16 + col] =
          (s_red[(0 * 2 + which) * 16 + col] + s_red[(1 * 2 + which) * 16 + col]) +
          (s_red[(2 * 2 + which) * 16 + col] + s_red[(3 * 2 + which) * 16 + col]);
    }
  }
}

long xw_ntiles(int D, int H, int W) {
  return (long)((D + XW_TZ - 1) / XW_TZ) * ((H + XW_TY - 1) / XW_TY) * ((W + XW_TXV - 1) / XW_TXV);
}

template <bool SIB, int PRO>
int launch_xw(const XwArgs& a, long blocks, hipStream_t s) {
  const size_t lds = 2 * (size_t)XW_IMG;
  static AtvsAttrOnce lds_once;                   // per kernel instantiation (this function is a template / has one kernel)
  if (const int rc_ = atvs_set_max_lds_once(lds_once, reinterpret_cast<const void*>(conv_xw_kernel<SIB, PRO>), 160 * 1024)) return rc_;
  hipLaunchKernelGGL((conv_xw_kernel<SIB, PRO>), dim3((unsigned)blocks), dim3(256), lds, s, a);
  return ATVS_OK;
}

}  // namespace

// workgroups PER SAMPLE of an x-pair launch (atvs_conv_xw_f32 / atvs_conv_xb_f32) over `groups` independent samples (rows of the
// statistics buffer = groups * this): one workgroup per CU in all, shared out among the samples, a multiple of 8 each
extern "C" long atvs_conv_xpair_grid(int D, int H, int W, int groups) {
  if (groups < 1) groups = 1;
  long nt = xw_ntiles(D, H, W);
  long share = 256 / groups / 8 * 8;
  if (share < 8) share = 8;
  long g = nt < share ? nt : share;
  return (g + 7) / 8 * 8;
}

// Floats of the packed form of a [3,3,3,Cin,8] kernel (Cin % 8 == 0), including 4 trailing zeros.
extern "C" int atvs_conv_xw_pack_size(int Cin, long* packed_floats) {
  if (Cin <= 0 || (Cin % 8) || !packed_floats) return ATVS_ERR_SHAPE;
  *packed_floats = (long)(Cin / 8) * XW_JC * 64 * 4 + 4;
  return ATVS_OK;
}

// HOST function.  w: TF kernel [3,3,3,Cin,8].  packed[chunk][K step j = (kd*2 + xh)*4 + p][lane = q*16 + (jx*8 + co)][s] =
// U_p[kd][kw = xl - jx][ci = chunk*8 + (q&1)*4 + s][co] (0 for kw outside 0..2), xl = 2 xh + (q >> 1), where U_p is the
// F(2,3) filter transform along kh: U_0 = g0, U_1 = (g0 + g1 + g2) / 2, U_2 = (g0 - g1 + g2) / 2, U_3 = g2 (in double,
// rounded to float once).
extern "C" int atvs_conv_xw_pack(const float* w, int Cin, float* packed) {
  if (!w || !packed) return ATVS_ERR_NULL;
  long pf;
  int rc = atvs_conv_xw_pack_size(Cin, &pf);
  if (rc) return rc;
  for (long i = 0; i < pf; ++i) packed[i] = 0.f;
  for (int ch = 0; ch < Cin / 8; ++ch)
    for (int j = 0; j < XW_JC; ++j)
      for (int q = 0; q < 4; ++q) {
        const int kd = j >> 3, xh = (j >> 2) & 1, pp = j & 3;
        const int xl = 2 * xh + (q >> 1);
        for (int jx = 0; jx < 2; ++jx) {
          const int kw = xl - jx;
          if (kw < 0 || kw > 2) continue;
          for (int co = 0; co < 8; ++co)
            for (int s = 0; s < 4; ++s) {
              const int ci = ch * 8 + (q & 1) * 4 + s;
              double g[3];
              for (int kh = 0; kh < 3; ++kh) g[kh] = (double)w[((((size_t)kd * 3 + kh) * 3 + kw) * Cin + ci) * 8 + co];
              const double u = (pp == 0) ? g[0] : (pp == 1) ? (g[0] + g[1] + g[2]) * 0.5 : (pp == 2) ? (g[0] - g[1] + g[2]) * 0.5 : g[2];
              packed[((((size_t)ch * XW_JC + j) * 64) + q * 16 + jx * 8 + co) * 4 + s] = (float)u;
            }
        }
      }
  return ATVS_OK;
}

extern "C" int atvs_conv_xw_pack_sibling_size(int Cin, long* packed_floats) {
  if (Cin <= 0 || (Cin % 8) || !packed_floats) return ATVS_ERR_SHAPE;
  *packed_floats = (long)(Cin / 8) * XW_J2 * 64 * 4;
  return ATVS_OK;
}

// HOST function.  w2: TF kernel [3,3,3,Cin,16] of the stride-2 sibling.  packed[chunk][K step i][lane = q*16 + co][s] =
// w2[tap = 2 i + (q >> 1)][chunk*8 + (q&1)*4 + s][co], tap = (kd*3 + kh)*3 + kw (0 for tap 27).
extern "C" int atvs_conv_xw_pack_sibling(const float* w2, int Cin, float* packed) {
  if (!w2 || !packed) return ATVS_ERR_NULL;
  long pf;
  int rc = atvs_conv_xw_pack_sibling_size(Cin, &pf);
  if (rc) return rc;
  for (long i = 0; i < pf; ++i) packed[i] = 0.f;
  for (int ch = 0; ch < Cin / 8; ++ch)
    for (int j = 0; j < XW_J2; ++j)
      for (int q = 0; q < 4; ++q) {
        const int tap = 2 * j + (q >> 1);
        if (tap > 26) continue;
        for (int co = 0; co < 16; ++co)
          for (int s = 0; s < 4; ++s) {
            const int ci = ch * 8 + (q & 1) * 4 + s;
            packed[((((size_t)ch * XW_J2 + j) * 64) + q * 16 + co) * 4 + s] = w2[((size_t)tap * Cin + ci) * 16 + co];
          }
      }
  return ATVS_OK;
}

// The x-pair contract of include/atvsnet_hip.h (atvs_conv_xw_f32) with weights packed by atvs_conv_xw_pack[_sibling]; grid and
// statistics rows = atvs_conv_xpair_grid.  Results differ from the direct sum by fp32 rounding only (F(2,3) along y).
extern "C" int atvs_conv_xw_f32(const float* x, const float* packed_w, const float* bias, const float* plane_bias,
                                float* y, double* stats_partial, int groups, int D, int H, int W, int Cin, int ldy, int y_coff,
                                int relu, const float* packed_w2, const float* plane_bias2, float* y2,
                                double* stats_partial2, int ldy2, int y_coff2, const float* x2, const float* in_params,
                                const float* in_params2, int in_relu, int in_relu2, long x_planar, atvs_stream_t stream) {
  if (!x || !packed_w || !y) return ATVS_ERR_NULL;
  if (in_params2 && !x2) return ATVS_ERR_ARG;
  if (groups <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || (Cin % 8)) return ATVS_ERR_SHAPE;
  if (x_planar && (x2 || in_params || x_planar < (long)D * H * W * 8)) return ATVS_ERR_ARG;          // the planar form is the plain (cost-volume) launch's
  if (y_coff < 0 || y_coff + 8 > ldy || (ldy % 4) || (y_coff % 4)) return ATVS_ERR_SHAPE;
  if (plane_bias && D < 2) return ATVS_ERR_ARG;
  if ((double)D * H * W * Cin >= 2147483648.0) return ATVS_ERR_SHAPE;   // 31-bit element offsets
  if (packed_w2) {
    if (!y2) return ATVS_ERR_NULL;
    if (y_coff2 < 0 || y_coff2 + 16 > ldy2 || (ldy2 % 4) || (y_coff2 % 4)) return ATVS_ERR_SHAPE;
  } else if (plane_bias2 || y2 || stats_partial2) {
    return ATVS_ERR_ARG;
  }
  long pf;
  atvs_conv_xw_pack_size(Cin, &pf);
  XwArgs a;
  a.x = x; a.wp = packed_w; a.zeros = packed_w + (pf - 4); a.bias = bias; a.pbias = plane_bias;
  a.y = y; a.stats = stats_partial;
  a.Di = D; a.Hi = H; a.Wi = W; a.Cin = Cin; a.ldy = ldy; a.ycoff = y_coff;
  a.nchunk = Cin / 8;
  a.tiles_y = (H + XW_TY - 1) / XW_TY; a.tiles_x = (W + XW_TXV - 1) / XW_TXV;
  a.ntiles = (int)xw_ntiles(D, H, W);
  a.relu = relu;
  a.wp2 = packed_w2; a.pbias2 = plane_bias2; a.y2 = y2; a.stats2 = stats_partial2;
  a.Do2 = (D + 1) / 2; a.Ho2 = (H + 1) / 2; a.Wo2 = (W + 1) / 2; a.ldy2 = ldy2; a.ycoff2 = y_coff2;
  a.pbz = D & 1; a.pby = H & 1; a.pbx = W & 1;
  a.wg = (int)atvs_conv_xpair_grid(D, H, W, groups);
  a.gx = x_planar ? x_planar * (Cin / 8) : (long)D * H * W * Cin; a.gy = (long)D * H * W * ldy; a.gpb = (long)H * W * 24;
  a.gy2 = (long)a.Do2 * a.Ho2 * a.Wo2 * ldy2; a.gpb2 = (long)a.Ho2 * a.Wo2 * 48;
  const long blocks = (long)a.wg * groups;
  hipStream_t st = as_stream(stream);
  a.sample_major = (groups == 8) ? 1 : 0;
  a.x2 = x2; a.in_pa = in_params; a.in_pb = in_params2; a.relu_a = in_relu; a.relu_b = in_relu2;
  a.vstride = x_planar ? 8 : Cin;
  a.cstride = x_planar ? x_planar : 8;
  const int pro = x2 ? 2 : (in_params ? 1 : 0);
  int rc;
  if (pro == 0) rc = packed_w2 ? launch_xw<true, 0>(a, blocks, st) : launch_xw<false, 0>(a, blocks, st);
  else if (pro == 1 && packed_w2) rc = launch_xw<true, 1>(a, blocks, st);
  else if (pro == 2 && packed_w2) rc = launch_xw<true, 2>(a, blocks, st);
  else return ATVS_ERR_ARG;
  if (rc) return rc;
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}
