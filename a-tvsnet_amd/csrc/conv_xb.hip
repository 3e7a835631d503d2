// 3x3x3 SAME stride-1 convolution to EIGHT output channels in x-pair form on the 16-bit matrix cores with SPLIT operands, one
// wavefront per SIMD (gfx950): conv_xp.hip's layers (conv_b*_0_1, global_refine_3dconv0_1, the photo stem; stride-2 sibling
// conv_b*_1_0 / 3dconv1_0 from the same staged image; /root/reference/cnn_wrapper/atvsnet.py StackedUNet / CostVolRefineNet,
// layer code network.py:165-215).
//
// Arithmetic (round 4): every fp32 operand is split into TWO fp16 pieces, x = h0 + h1 / 2048 with h0 = f16(x) and
// h1 = f16((x - h0) * 2048) (the residual scaled into fp16's normal range: 22 significant bits, no denormal loss), w = g0 + g1 /
// 2048 likewise (split by the host packer); THREE products on v_mfma_f32_16x16x32_f16 with fp32 accumulation: h0 g0 into the
// main accumulator, h0 g1 + h1 g0 into a second one that is scaled by 2^-11 once in the epilogue (the dropped h1 g1 is 2^-22 of
// a product).  Measured on MI355X (tools_dev/micro/f16_split_probe.hip): error of a 864-term dot product against a double sum
// 2.8e-7 of the output maximum, against 6.3e-7 for the fp32 matrix cores and 8.0e-7 for the round-3 form (three bf16 pieces, six
// products): fewer accumulation steps.  fp16 DENORMAL inputs are honoured by the instruction (same probe).  Range: |x| or |w| >
// 65504 becomes infinity and the output NaN (loud, not silent): the layers' inputs are batch-normalised activations / features,
// |x| <= sqrt(voxels) + |beta|; the packer refuses such weights.
//
// One K = 32 instruction covers the FOUR x offsets a voxel pair touches x 8 channels: lane group q = x offset xl, so a
// (kd, kh) tap row of an 8-channel chunk is ONE K step (36 fp32 16x16x4 steps of conv_xp become 9), rows = (x parity,
// channel) as in conv_xp (3/4 of the MFMA work useful).  Per 8-channel chunk and 8 rows: 9 steps x 3 products x 8 = 216
// instructions of 16 cycles (round 3: 432; conv_xw: 384 of 32 cycles).
//
// Structure: tile 4(z) x 8(y) x 32(x), wavefront w owns plane z0 + w (8 + 8 accumulator tiles); the input is staged in 8-channel
// chunks as TWO piece images [6][10][even / odd x interleaved in runs of eight][8 fp16] (46 KB each, single-buffered: the next
// stage's halo waits in registers as in conv_xp, two barriers per stage); the split happens once per staged element, after
// the optional prologue (batch norm + ReLU of the producers, skip add).  Weights (two pieces per step, split on the host): the
// NEXT chunk's 32 KB are fetched at the top of the K loop (eight 16-byte loads per thread, L2 hits) and written into the other
// of two LDS weight buffers a few phases later; the K loop reads its A fragments from LDS one step ahead.  (Round 3 streamed
// them from L2 into registers inside the loop: vector-memory operations retire in order, so every weight request queued behind
// the HBM halo loads of the phases before it and each K step waited for memory latency -- halving the MFMA count alone bought
// 30 %, not 2 x.)  Sibling: K step i = taps 4 i + q of the 27 (7 steps), 2 rows per wavefront.
#include <cstring>
#include <type_traits>

#include "conv_common.h"

// Development build (-DATVS_XB_DEBUG): per-wavefront cycle counts of the phases (tools_dev/phase_xw.py xb ...).
#ifdef ATVS_XB_DEBUG
__device__ unsigned long long atvs_dbg_xb[4096 * 8];
extern "C" int atvs_debug_read_xw(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(atvs_dbg_xb), sizeof(atvs_dbg_xb));
}
#define XDBG(i) { unsigned long long t_ = clock64(); dbg_acc[i] += t_ - dbg_t; dbg_t = t_; }
#else
#define XDBG(i)
#endif

namespace {

constexpr int XB_TZ = 4, XB_TY = 8, XB_TXV = 32;
constexpr int XB_HZ = XB_TZ + 2, XB_HY = XB_TY + 2, XB_HX = XB_TXV + 2;
constexpr int XB_ROWB = 3 * 256;                            // an image row: 17 even-x + 17 odd-x voxels, see xb_col
constexpr int XB_IMG = XB_HZ * XB_HY * XB_ROWB;             // 46,080 bytes per piece
constexpr int XB_SLOTS = XB_HZ * XB_HY * XB_HX * 2;         // float4 slots of the fp32 halo of a chunk
constexpr int XB_MAXS = (XB_SLOTS + 255) / 256;             // 16 per thread
constexpr int XB_JC = 9;                                    // main K steps per chunk: (kd, kh)
constexpr int XB_J2 = 7;                                    // sibling K steps per chunk: taps 4 i + q
constexpr int XB_NP = 2;                                    // operand pieces
constexpr int XB_WMAIN = XB_JC * XB_NP * 1024;              // bytes of a chunk's main weight pieces: [step][piece][lane][8 fp16]
constexpr int XB_WSIB = XB_J2 * XB_NP * 1024;               // ... of its sibling weight pieces
constexpr int XB_WBUF = XB_WMAIN + XB_WSIB;                 // one LDS weight buffer (32 KB); two of them behind the images
constexpr int XB_WOFF = XB_NP * XB_IMG;
constexpr int XB_LDS = XB_WOFF + 2 * XB_WBUF;               // 157,696 of the CU's 163,840 bytes
static_assert(XB_LDS <= 160 * 1024, "one workgroup per CU");
static_assert((XB_WMAIN / 16) % 128 == 0 && (XB_WBUF / 16) % 256 == 0, "weight slots: whole wavefronts per source");
constexpr float XB_RS = 2048.f, XB_IRS = 1.f / 2048.f;      // scale of the residual piece and its inverse (exact powers of two)
static_assert(XB_MAXS <= 18, "three halo slots per phase of the main K loop");
static_assert((XB_NP - 1) * XB_IMG + (2 * XB_HY + 2 + XB_TY) * XB_ROWB < XB_NP * XB_IMG, "fragment reads stay inside the images");

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// buffer_load_dwordx4 (offen).  hipcc 7.2's __builtin_amdgcn_raw_buffer_load_b128 compiles to a ONE-dword load whose value is
// splat over the four components (checked in the ISA), so the LLVM intrinsic is bound by name instead.
__device__ f32x4 xb_buffer_load_x4(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.ptr.buffer.load.v4f32");

struct XbArgs {
  const float* x;
  const f16x8* wp;       // packed weight pieces, see atvs_conv_xb_pack
  const float* zeros;    // 16 bytes of zeros (tail of the packed weights)
  const float* bias;
  const float* pbias;    // (H, W, 24) or nullptr
  float* y;
  double* stats;
  int Di, Hi, Wi, Cin;
  int ldy, ycoff;
  int nchunk;
  int tiles_y, tiles_x, ntiles;
  int relu;
  const f16x8* wp2;      // packed sibling weight pieces (atvs_conv_xb_pack_sibling) or nullptr
  const float* pbias2;   // (Ho2, Wo2, 48) or nullptr
  float* y2;
  double* stats2;
  int Do2, Ho2, Wo2, ldy2, ycoff2;
  int pbz, pby, pbx;
  int wg;
  long gx, gy, gpb, gy2, gpb2;
  const float* x2;
  const float* in_pa;
  const float* in_pb;
  int relu_a, relu_b;
  int sample_major;
  int vstride;           // floats between voxels / between 8-channel chunks of x (channel-last or chunk-planar: conv_xw.hip)
  long cstride;
};

// Byte offset inside an image row of the voxel with x parity `par` and index i = x / 2 (0..16): even and odd voxels
// alternate in 128-byte runs of eight.  A fragment read takes lane group q to parity q & 1, index (q >> 1) + r; with this
// interleave the 16 lanes of every ds_read_b128 lane group ({0-3,12-15 | 20-27}, ...: MI355X_MICROARCH.md) cover all 64
// banks once -- with the even | odd column split of conv_xp.hip (16-byte voxels) every fragment read was 2-way conflicted
// (PMC: half of the LDS cycles).
__device__ __forceinline__ constexpr int xb_col(int par, int i) { return (i >> 3) * 256 + par * 128 + (i & 7) * 16; }

// the two fp16 pieces of four fp32 values: h0 = f16(x), h1 = f16((x - h0) * 2^11)  (x - h0 is exact in fp32)
__device__ __forceinline__ void xb_split(const float4& v, f16x4* p0, f16x4* p1) {
  const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const _Float16 a = (_Float16)x[i];
    (*p0)[i] = a;
    (*p1)[i] = (_Float16)((x[i] - (float)a) * XB_RS);
  }
}

template <bool SIB, int PRO>
__global__ __launch_bounds__(256, 1) void conv_xb_kernel(XbArgs p) {
  // own the SIMD's whole register file (512 per lane): no wavefront of ANOTHER kernel runs beside this one's bf16 MFMAs --
  // beside them other kernels' wavefronts computed wrong lane quarters (tools_dev/micro/pk_beside_mfma.hip)
  asm volatile("" ::: "v255", "a255");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int MAXS = XB_MAXS, JC = XB_JC, J2 = SIB ? XB_J2 : 0, ROWB = XB_ROWB;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave: a scalar (uniform branches)
  const int r = lane & 15, q = lane >> 4;

  // this lane's fragment (the 8 channels of one voxel of a piece image) of halo row 0 of the wavefront's plane at its x
  // offset xl = q; piece images XB_IMG apart (one base each: the displacements exceed the 16-bit immediate otherwise)
  int fb[XB_NP];
#pragma unroll
  for (int pc = 0; pc < XB_NP; ++pc) fb[pc] = pc * XB_IMG + (wave * XB_HY) * XB_ROWB + xb_col(q & 1, (q >> 1) + r);
  // sibling: this lane's tap of step i is 4 i + q (taps past 26: zero weights, tap 26's fragment); per-step byte offsets
  int sd[SIB ? XB_J2 : 1];
  if (SIB) {
    const int row0 = (2 * (wave >> 1) + 1 - p.pbz) * XB_HY + (4 * (wave & 1) + 1 - p.pby);
#pragma unroll
    for (int i = 0; i < XB_J2; ++i) {
      const int tap = min(4 * i + q, 26);
      const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
      const int xh = kw + 1 - p.pbx;                                  // halo x of output column 0
      sd[i] = (row0 + kd * XB_HY + kh) * XB_ROWB + xb_col(xh & 1, (xh >> 1) + r);
    }
  }

  // per-slot constants: float4 = channels 4 c4 .. of a halo voxel of the fp32 chunk -> 8 bytes of each piece image
  int goff[MAXS], laddr[MAXS];
  unsigned pg[MAXS];
#pragma unroll
  for (int i = 0; i < MAXS; ++i) {
    int s = tid + i * 256;
    const bool live = s < XB_SLOTS;
    s = min(s, XB_SLOTS - 1);
    const int c4 = s & 1, v = s >> 1;
    const int xx = v % XB_HX, v2 = v / XB_HX;
    const int yy = v2 % XB_HY, zz = v2 / XB_HY;
    goff[i] = (((zz * p.Hi + yy) * p.Wi + xx) * p.vstride + c4 * 4) * 4;       // BYTES from the halo's first voxel
    laddr[i] = (zz * XB_HY + yy) * XB_ROWB + xb_col(xx & 1, xx >> 1) + c4 * 8;
    pg[i] = 0x808080u | (unsigned)(live ? zz : 0x7f) | ((unsigned)yy << 8) | ((unsigned)xx << 16);
  }
  const bool last_live = tid + (MAXS - 1) * 256 < XB_SLOTS;

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
    *x0 = bx * XB_TXV;
    *y0 = (rest % p.tiles_y) * XB_TY;
    *z0 = (rest / p.tiles_y) * XB_TZ;
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
  // The halo of the next stage is fetched with BUFFER loads from a descriptor whose base is the halo's first voxel of the
  // chunk (a scalar add per stage): a slot's address is then a per-kernel constant (goff) and a slot outside the volume gets
  // an offset beyond the descriptor's range -- the load returns zeros, no pointer select.  Which slots are inside is a property
  // of the TILE: the 16-bit mask is recomputed only when the prefetched stage starts a new tile (every nchunk-th stage).
  // (Round 3 formed a 64-bit address and the bounds test per slot and stage: ~16 vector instructions per slot, 257 of the
  // kernel's 480 per stage, against 240 MFMAs.)
  unsigned vinv = 0;                     // bit i set: slot i of the prefetched stage lies OUTSIDE the volume
  auto pf_mask = [&](const PfTile& T) __attribute__((always_inline)) {
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < MAXS; ++i) {
      const unsigned t1 = pg[i] - T.lo;
      const unsigned t2 = T.hi1 + ~pg[i];
      m |= ((((t1 & t2) & 0x808080u) == 0x808080u) ? 0u : 1u) << i;
    }
    return m;
  };
  auto pf_rsrc = [&](const float* base, int org) __attribute__((always_inline)) {
#ifdef ATVS_XB_HOT       // development: every tile fetches the same halo (cache hits) -- is the launch bound by the memory path?
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (org & 1023)), 0, 0x7ffffff0, 0x00020000);
#else
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + org), 0, 0x7ffffff0, 0x00020000);
#endif
  };
  auto pf_slot = [&](__amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb, int i) __attribute__((always_inline)) {
    const unsigned voff = (unsigned)goff[i] | (unsigned)__builtin_amdgcn_sbfe(vinv, i, 1);      // all ones when outside
    const f32x4 a = xb_buffer_load_x4(ra, (int)voff, 0, 0);
    pf[i] = make_float4(a[0], a[1], a[2], a[3]);
    if (PRO == 2) {
      const f32x4 b = xb_buffer_load_x4(rb, (int)voff, 0, 0);
      pf2[i] = make_float4(b[0], b[1], b[2], b[3]);
    }
  };

  // prologue transform: arithmetic of bn_apply / bn_add (norm.hip), as conv_xp.hip / conv_xw.hip
  struct Par { float4 ma, sa, ba, mb, sb, bb; };
  const bool has_a = PRO >= 1 && ipa != nullptr, has_b = PRO == 2 && ipb != nullptr;
  const float floor_a = p.relu_a ? 0.f : -__builtin_huge_valf(), floor_b = p.relu_b ? 0.f : -__builtin_huge_valf();
  // a source WITHOUT a pending batch norm gets the identity (mean 0, scale 1, beta 0, no floor) once per chunk, not a select per
  // staged value: (v - 0) * 1 + 0 == v (a -0 becomes +0: the same sums)
  auto load_par = [&](int chunk) __attribute__((always_inline)) {
    Par P;
    const int cch = chunk * 8 + c4t * 4;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f), one = make_float4(1.f, 1.f, 1.f, 1.f);
    P.ma = has_a ? ld4(ipa + cch) : zero; P.sa = has_a ? ld4(ipa + p.Cin + cch) : one; P.ba = has_a ? ld4(ipa + 2 * p.Cin + cch) : zero;
    if (PRO == 2) {
      P.mb = has_b ? ld4(ipb + cch) : zero; P.sb = has_b ? ld4(ipb + p.Cin + cch) : one; P.bb = has_b ? ld4(ipb + 2 * p.Cin + cch) : zero;
    }
    return P;
  };
  const float lo_a = has_a ? floor_a : -__builtin_huge_valf(), lo_b = has_b ? floor_b : -__builtin_huge_valf();
  auto bn1 = [&](float v, float m, float sc, float be, float lo) __attribute__((always_inline)) {
    return fmaxf((v - m) * sc + be, lo);               // bn_apply's arithmetic (norm.hip): sub, mul, add -- no contraction
  };
  auto xform = [&](int i, const Par& P) __attribute__((always_inline)) {
    const bool ok = !((vinv >> i) & 1u);            // vinv still describes the stage being staged: it is renewed in the K loop
    float4 v = pf[i];
    v.x = bn1(v.x, P.ma.x, P.sa.x, P.ba.x, lo_a); v.y = bn1(v.y, P.ma.y, P.sa.y, P.ba.y, lo_a);
    v.z = bn1(v.z, P.ma.z, P.sa.z, P.ba.z, lo_a); v.w = bn1(v.w, P.ma.w, P.sa.w, P.ba.w, lo_a);
    if (PRO == 2) {
      const float4 u = pf2[i];
      v.x += bn1(u.x, P.mb.x, P.sb.x, P.bb.x, lo_b); v.y += bn1(u.y, P.mb.y, P.sb.y, P.bb.y, lo_b);
      v.z += bn1(u.z, P.mb.z, P.sb.z, P.bb.z, lo_b); v.w += bn1(u.w, P.mb.w, P.sb.w, P.bb.w, lo_b);
    }
    pf[i] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
  };

  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias) bv = ld4(p.bias + (q & 1) * 4);
  f32x4 acc[XB_TY], accx[XB_TY];       // h0 g0 | (h0 g1 + h1 g0) * 2^11
  f32x4 acc2[2], acc2x[2];
  float ssum2[4] = {0.f, 0.f, 0.f, 0.f}, ssq2[4] = {0.f, 0.f, 0.f, 0.f};

  // weight pieces of a chunk: global -> registers -> LDS buffer `buf` ([main steps][sibling steps], XB_WBUF bytes)
  constexpr int NW = SIB ? XB_WBUF / 4096 : (XB_WMAIN + 4095) / 4096;        // float4 slots per thread: 8 | 5
  float4 wreg[NW];
  auto w_request = [&](int chunk) __attribute__((always_inline)) {
    const unsigned char* gm = reinterpret_cast<const unsigned char*>(p.wp) + (size_t)chunk * XB_WMAIN;
    const unsigned char* gs = SIB ? reinterpret_cast<const unsigned char*>(p.wp2) + (size_t)chunk * XB_WSIB : nullptr;
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int idx = tid + 256 * j;                         // float4 index inside the buffer; main | sibling is wave-uniform
      const unsigned char* src = idx < XB_WMAIN / 16 ? gm + (size_t)idx * 16
                                                     : (SIB ? gs + (size_t)(idx - XB_WMAIN / 16) * 16 : reinterpret_cast<const unsigned char*>(p.zeros));
      wreg[j] = ld4(reinterpret_cast<const float*>(src));
    }
  };
  auto w_land = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int idx = tid + 256 * j;
      if (SIB || idx < XB_WMAIN / 16) *reinterpret_cast<float4*>(smem + XB_WOFF + buf * XB_WBUF + idx * 16) = wreg[j];
    }
  };
  if (nstage > 0) {
    w_request(0);
    w_land(0);                             // read after the two barriers of stage 0
    const PfTile T0 = pf_tile(0);
    vinv = pf_mask(T0);
    const __amdgpu_buffer_rsrc_t r0 = pf_rsrc(T0.xb, T0.org), r02 = pf_rsrc(PRO == 2 ? T0.xb2 : T0.xb, T0.org);
#pragma unroll
    for (int i = 0; i < MAXS; ++i) pf_slot(r0, r02, i);
  }

#ifdef ATVS_XB_DEBUG
  unsigned long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long dbg_t = clock64();
#endif
  for (int stage = 0; stage < nstage; ++stage) {
    const int k = stage / p.nchunk, ch = stage - k * p.nchunk;
    XDBG(0)
    if (ch == 0) {
#pragma unroll
      for (int t = 0; t < XB_TY; ++t) acc[t] = accx[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc2[0] = acc2[1] = acc2x[0] = acc2x[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // this chunk's weight pieces sit in LDS buffer wbuf (one chunk: resident in buffer 0 for the whole launch)
    const bool wstream = p.nchunk > 1;
    const int wbuf = wstream ? (stage & 1) : 0;
    const int wb = XB_WOFF + wbuf * XB_WBUF + lane * 16;

    __syncthreads();                       // every wavefront is done reading the previous stage's images
    XDBG(5)
    if (PRO >= 1) {
      const Par P0 = load_par(ch);
#pragma unroll
      for (int i = 0; i < MAXS; ++i) xform(i, P0);
    }
#pragma unroll
    for (int i = 0; i < MAXS; ++i)
      if (i < MAXS - 1 || last_live) {
        f16x4 p0, p1;
        xb_split(pf[i], &p0, &p1);
        *reinterpret_cast<f16x4*>(smem + laddr[i]) = p0;
        *reinterpret_cast<f16x4*>(smem + XB_IMG + laddr[i]) = p1;
      }
    XDBG(1)
    __syncthreads();
    XDBG(6)

    // The next stage's traffic goes out in the FIRST phases of the K loop: the next chunk's weights first (their write into the
    // other LDS buffer, at phase 6, then waits for L2 hits only -- vector-memory operations retire in order), then two halo
    // slots per phase (two or three instructions each) in phases 0..7, so that the last ones have ten phases and the sibling
    // loop to arrive.  (One per phase up to phase 15, round 3: the next stage's split waited for HBM.  All 24 loads in one
    // burst in front of the loop: the wavefronts stall issuing them, 2,700 cycles per stage.)
    const PfTile T = pf_tile(min(stage + 1, nstage - 1));      // last stage: harmless re-read of its own halo
    const bool last_chunk = (ch == p.nchunk - 1);
    if (last_chunk) vinv = pf_mask(T);                         // the prefetched stage starts a new tile
    const __amdgpu_buffer_rsrc_t rsa = pf_rsrc(T.xb, T.org), rsb = pf_rsrc(PRO == 2 ? T.xb2 : T.xb, T.org);

    int tz0, ty0, tx0;
    tile_origin(k, &tz0, &ty0, &tx0);
    const int zo = tz0 + wave, xo = tx0 + 2 * r + (q >> 1), co = (q & 1) * 4;
    const bool evox_ok = zo < p.Di && xo < p.Wi;
    const size_t erow = (size_t)p.Wi * p.ldy;
    const size_t eo = (((size_t)zo * p.Hi + ty0) * p.Wi + xo) * (size_t)p.ldy + p.ycoff + co;
    const size_t epb_off = ((size_t)ty0 * p.Wi + xo) * 24 + plane_variant(zo - 1, p.Di) * 8 + co;
    auto erow_ok = [&](int t) __attribute__((always_inline)) { return evox_ok && ty0 + t < p.Hi; };
    float4 epb[XB_TY], epb2[2];
    const int zo2 = (tz0 >> 1) + (wave >> 1), yo2 = (ty0 >> 1) + 2 * (wave & 1), xo2 = (tx0 >> 1) + r;
    const bool evox2_ok = SIB && zo2 < p.Do2 && xo2 < p.Wo2;
    auto erow2_ok = [&](int t) __attribute__((always_inline)) { return evox2_ok && yo2 + t < p.Ho2; };
    const size_t erow2 = (size_t)p.Wo2 * p.ldy2;
    const size_t eo2 = (((size_t)zo2 * p.Ho2 + yo2) * p.Wo2 + xo2) * (size_t)p.ldy2 + p.ycoff2 + 4 * q;

    // ---- main K loop: per kd TWO phases -- the TEN halo rows of input piece h0 at depth kd serve the three kh steps x both
    // weight pieces (48 MFMAs), then the ten rows of h1 serve kh x g0 (24 MFMAs): 20 fragment reads per kd instead of 48.
    // Why: with a fragment read per MFMA the four wavefronts asked the LDS for 4 x 4 cycles of ds_read_b128 per 16-cycle MFMA
    // -- the array was saturated in every h1 phase and each MFMA waited for its fragment (phase timers: 5,300 cycles for
    // 3,456 of MFMA work; counters: LDS active 45 % of the launch ON AVERAGE, MFMA pipe 38 %).  Rows of one piece are requested
    // during the other piece's phase (every second MFMA slot), the six weight fragments of the next kd during the h0 phase.
    f16x8 Bq[XB_NP][XB_HY], B2[2][XB_NP][2], A[2][3][XB_NP], A2[2][XB_NP];
    auto request_A = [&](int kd) __attribute__((always_inline)) {           // the weight fragments of steps 3 kd .. 3 kd + 2
#ifdef ATVS_XB_BARE
      if (kd > 0) return;
#endif
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int pc = 0; pc < XB_NP; ++pc)
          A[kd & 1][kh][pc] = *reinterpret_cast<const f16x8*>(smem + wb + ((kd * 3 + kh) * XB_NP + pc) * 1024);
    };
    auto request_A2 = [&](int i) __attribute__((always_inline)) {
#pragma unroll
      for (int pc = 0; pc < XB_NP; ++pc) A2[i & 1][pc] = *reinterpret_cast<const f16x8*>(smem + wb + XB_WMAIN + (i * XB_NP + pc) * 1024);
    };
    auto request_B = [&](int kd, int pc) __attribute__((always_inline)) {
#ifdef ATVS_XB_NOREAD      // development: what does the K loop cost without its fragment reads?
      if (kd > 0 || pc > 0) return;
#endif
#pragma unroll
      for (int y = 0; y < XB_HY; ++y) Bq[pc][y] = *reinterpret_cast<const f16x8*>(smem + fb[pc] + (kd * XB_HY + y) * ROWB);
    };
    auto request_B2 = [&](int i) __attribute__((always_inline)) {         // the two pieces of sibling step i (6 MFMAs)
#pragma unroll
      for (int pc = 0; pc < XB_NP; ++pc)
#pragma unroll
        for (int t = 0; t < 2; ++t)
          B2[i & 1][pc][t] = *reinterpret_cast<const f16x8*>(smem + pc * XB_IMG + sd[i] + 2 * t * ROWB);
    };
    request_A(0);
    request_B(0, 0);
    asm volatile("" ::: "memory");
    XDBG(0)
#pragma unroll
    for (int ph = 0; ph < 2 * 3; ++ph) {
      const int kd = ph >> 1, pc = ph & 1;
      if (pc == 0) {
        request_B(kd, 1);
        if (kd + 1 < 3) request_A(kd + 1);
        else if (SIB) request_A2(0);
      } else {
        if (kd + 1 < 3) request_B(kd + 1, 0);
        else if (SIB) request_B2(0);
      }
      // the next stage's traffic: the next chunk's weights first (their write into the other LDS buffer, two phases on, then
      // waits for L2 hits only -- vector-memory operations retire in order), then three halo slots per phase.  Tried and
      // dropped: all 24 loads in one burst in front of the loop (the wavefronts stall issuing them: +2,700 cycles per stage),
      // one wavefront's 16 loads per phase, staggered (+700).  In-loop vector-memory instructions are what the K loop pays
      // for: 5,000 cycles with them, 3,100 = the bare MFMA stream without (phase timers, development builds).
      // (The two-source form -- one 8-channel chunk on the path, weights resident -- has no registers to hold them across phases.)
#ifndef ATVS_XB_BARE       // development: the bare MFMA stream
      if (ph == (PRO == 2 ? 2 : 0) && wstream) w_request(ch + 1 < p.nchunk ? ch + 1 : 0);
      if (ph == 2 && wstream) w_land(wbuf ^ 1);
#pragma unroll
      for (int i = 3 * ph; i < 3 * ph + 3; ++i)
        if (i < MAXS) pf_slot(rsa, rsb, i);
#endif
      if (!SIB && ph == 4 && last_chunk) {       // (uniform branch: only the tile's last stage has an epilogue)
#pragma unroll
        for (int t = 0; t < XB_TY; ++t) epb[t] = ld4((pbg && erow_ok(t)) ? pbg + (epb_off + (size_t)t * p.Wi * 24) : p.zeros);
      }
      if (pc == 0) {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
          for (int t = 0; t < XB_TY; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[kd & 1][kh][0], Bq[0][t + kh], acc[t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < XB_TY; ++t) accx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[kd & 1][kh][1], Bq[0][t + kh], accx[t], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int t = 0; t < XB_TY; ++t) accx[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[kd & 1][kh][0], Bq[1][t + kh], accx[t], 0, 0, 0);
      }
      // the phase's requests do not depend on its MFMAs: two MFMAs, then at most one LDS read / LDS write / global load and a
      // few other vector instructions in their shadow (bunched in front of the MFMAs they left the matrix pipe dry)
#pragma unroll
      for (int g = 0; g < (pc == 0 ? 24 : 12); ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);      // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // DS read
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // DS write
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // VMEM read
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);      // VALU
      }
      asm volatile("" ::: "memory");                            // this phase's requests stay in this phase
      __builtin_amdgcn_sched_barrier(0);
    }
    XDBG(2)
    if (SIB) {
      // 7 steps of 6 MFMAs, one phase per step (with two rows per wavefront a per-piece phase would be 2-4 MFMAs long)
#pragma unroll
      for (int i = 0; i < J2; ++i) {
        if (i + 1 < J2) {
          request_A2(i + 1);
          request_B2(i + 1);
        }
        if (i == J2 - 3 && last_chunk) {   // the epilogues' depth-plane biases (uniform branch: the tile's last stage only)
#pragma unroll
          for (int t = 0; t < XB_TY; ++t) epb[t] = ld4((pbg && erow_ok(t)) ? pbg + (epb_off + (size_t)t * p.Wi * 24) : p.zeros);
          const size_t o = ((size_t)yo2 * p.Wo2 + xo2) * 48 + plane_variant(2 * zo2 - p.pbz, p.Di) * 16 + 4 * q;
#pragma unroll
          for (int t = 0; t < 2; ++t) epb2[t] = ld4((pb2g && erow2_ok(t)) ? pb2g + (o + (size_t)t * p.Wo2 * 48) : p.zeros);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) acc2[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A2[i & 1][0], B2[i & 1][0][t], acc2[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc2x[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A2[i & 1][1], B2[i & 1][0][t], acc2x[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t) acc2x[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A2[i & 1][0], B2[i & 1][1][t], acc2x[t], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 6; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    XDBG(3)
    if (!last_chunk) continue;

    // ---- epilogue (conv_xp.hip): this lane holds channels co..co+3 of voxel xo for the 8 rows of plane zo
    auto store_rows = [&](auto relu_tag) __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < XB_TY; ++t) {
        if (!erow_ok(t)) continue;
        float4 v;
        // fmaf(cross, 2^-11, main): the product by a power of two is exact, i.e. the same value as main + cross * 2^-11 in one
        // instruction
        v.x = (__builtin_fmaf(accx[t][0], XB_IRS, acc[t][0]) + bv.x) + epb[t].x;
        v.y = (__builtin_fmaf(accx[t][1], XB_IRS, acc[t][1]) + bv.y) + epb[t].y;
        v.z = (__builtin_fmaf(accx[t][2], XB_IRS, acc[t][2]) + bv.z) + epb[t].z;
        v.w = (__builtin_fmaf(accx[t][3], XB_IRS, acc[t][3]) + bv.w) + epb[t].w;
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
        const float4 v = make_float4(__builtin_fmaf(acc2x[t][0], XB_IRS, acc2[t][0]) + epb2[t].x, __builtin_fmaf(acc2x[t][1], XB_IRS, acc2[t][1]) + epb2[t].y,
                                     __builtin_fmaf(acc2x[t][2], XB_IRS, acc2[t][2]) + epb2[t].z, __builtin_fmaf(acc2x[t][3], XB_IRS, acc2[t][3]) + epb2[t].w);
        st4(y2g + (eo2 + (size_t)t * erow2), v);
        ssum2[0] += v.x; ssum2[1] += v.y; ssum2[2] += v.z; ssum2[3] += v.w;
        ssq2[0] += v.x * v.x; ssq2[1] += v.y * v.y; ssq2[2] += v.z * v.z; ssq2[3] += v.w * v.w;
      }
    }
    XDBG(4)
  }
#ifdef ATVS_XB_DEBUG
  if (lane == 0 && blockIdx.x < 1024) {
    dbg_acc[7] = (unsigned long long)nstage;
    for (int i = 0; i < 8; ++i) atvs_dbg_xb[(blockIdx.x * 4 + wave) * 8 + i] = dbg_acc[i];
  }
#endif
  __syncthreads();

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


long xb_ntiles(int D, int H, int W) {
  return (long)((D + XB_TZ - 1) / XB_TZ) * ((H + XB_TY - 1) / XB_TY) * ((W + XB_TXV - 1) / XB_TXV);
}

template <bool SIB, int PRO>
int launch_xb(const XbArgs& a, long blocks, hipStream_t s) {
  const size_t lds = XB_LDS;
  static bool attr_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return ATVS_ERR_LAUNCH;
  if (!attr_set[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_xb_kernel<SIB, PRO>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return ATVS_ERR_LAUNCH;
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL((conv_xb_kernel<SIB, PRO>), dim3((unsigned)blocks), dim3(256), lds, s, a);
  return ATVS_OK;
}

// HOST: the two fp16 pieces of v (round to nearest even; the kernel's xb_split) at out[base + piece * 64 * 8]; false if v does
// not fit fp16's range
bool xb_put(uint16_t* out, size_t base, float v) {
  const _Float16 h0 = (_Float16)v;
  const _Float16 h1 = (_Float16)((v - (float)h0) * XB_RS);
  std::memcpy(&out[base], &h0, 2);
  std::memcpy(&out[base + 64 * 8], &h1, 2);
  const float back = (float)h0;
  return back - back == 0.f;                             // finite
}

}  // namespace

// Bytes of the packed form of a [3,3,3,Cin,8] kernel (Cin % 8 == 0) for atvs_conv_xb_f32, including 16 trailing zero bytes.
extern "C" int atvs_conv_xb_pack_size(int Cin, long* packed_bytes) {
  if (Cin <= 0 || (Cin % 8) || !packed_bytes) return ATVS_ERR_SHAPE;
  *packed_bytes = (long)(Cin / 8) * XB_JC * XB_NP * 1024 + 16;
  return ATVS_OK;
}

// HOST function.  w: TF kernel [3,3,3,Cin,8].  packed[chunk][step s = kd*3 + kh][piece][lane = q*16 + (jx*8 + co)][e] =
// piece of w[kd][kh][kw = q - jx][ci = chunk*8 + e][co] (0 for kw outside 0..2): lane group q = x offset of the pair window.
extern "C" int atvs_conv_xb_pack(const float* w, int Cin, unsigned char* packed) {
  if (!w || !packed) return ATVS_ERR_NULL;
  long pb;
  int rc = atvs_conv_xb_pack_size(Cin, &pb);
  if (rc) return rc;
  std::memset(packed, 0, (size_t)pb);
  uint16_t* out = reinterpret_cast<uint16_t*>(packed);
  bool fits = true;
  for (int ch = 0; ch < Cin / 8; ++ch)
    for (int s = 0; s < XB_JC; ++s)
      for (int q = 0; q < 4; ++q)
        for (int jx = 0; jx < 2; ++jx) {
          const int kd = s / 3, kh = s % 3, kw = q - jx;
          if (kw < 0 || kw > 2) continue;
          for (int co = 0; co < 8; ++co)
            for (int e = 0; e < 8; ++e)
              fits &= xb_put(out, ((((size_t)ch * XB_JC + s) * XB_NP) * 64 + q * 16 + jx * 8 + co) * 8 + e,
                             w[((((size_t)kd * 3 + kh) * 3 + kw) * Cin + ch * 8 + e) * 8 + co]);
        }
  return fits ? ATVS_OK : ATVS_ERR_ARG;                  // a weight beyond fp16's range (|w| > 65504)
}

extern "C" int atvs_conv_xb_pack_sibling_size(int Cin, long* packed_bytes) {
  if (Cin <= 0 || (Cin % 8) || !packed_bytes) return ATVS_ERR_SHAPE;
  *packed_bytes = (long)(Cin / 8) * XB_J2 * XB_NP * 1024;
  return ATVS_OK;
}

// HOST function.  w2: TF kernel [3,3,3,Cin,16] of the stride-2 sibling.  packed[chunk][step i][piece][lane = q*16 + co][e] =
// piece of w2[tap = 4 i + q][chunk*8 + e][co] (0 for taps past 26).
extern "C" int atvs_conv_xb_pack_sibling(const float* w2, int Cin, unsigned char* packed) {
  if (!w2 || !packed) return ATVS_ERR_NULL;
  long pb;
  int rc = atvs_conv_xb_pack_sibling_size(Cin, &pb);
  if (rc) return rc;
  std::memset(packed, 0, (size_t)pb);
  uint16_t* out = reinterpret_cast<uint16_t*>(packed);
  bool fits = true;
  for (int ch = 0; ch < Cin / 8; ++ch)
    for (int i = 0; i < XB_J2; ++i)
      for (int q = 0; q < 4; ++q) {
        const int tap = 4 * i + q;
        if (tap > 26) continue;
        for (int co = 0; co < 16; ++co)
          for (int e = 0; e < 8; ++e)
            fits &= xb_put(out, ((((size_t)ch * XB_J2 + i) * XB_NP) * 64 + q * 16 + co) * 8 + e,
                           w2[((size_t)tap * Cin + ch * 8 + e) * 16 + co]);
      }
  return fits ? ATVS_OK : ATVS_ERR_ARG;
}

// Same contract as atvs_conv_xw_f32 (x_planar included), weights packed by atvs_conv_xb_pack[_sibling]; grid and statistics
// rows = atvs_conv_xp_grid.  fp32-class results (split-bf16 operands, fp32 accumulation); rounding differs from the fp32 forms.
// Beyond atvs_conv_xw_f32: x_planar may come WITH in_params (one pending batch norm, the refinement's chunk-planar concat;
// not with x2), and y_group_stride != 0 = floats between the samples of y (default D*H*W*ldy): the photo stem writes
// plane 0 of each sample's chunk-planar concat (ldy = 8, y_group_stride = 4 * plane stride).
extern "C" int atvs_conv_xb_f32(const float* x, const unsigned char* packed_w, const float* bias, const float* plane_bias,
                                float* y, double* stats_partial, int groups, int D, int H, int W, int Cin, int ldy, int y_coff,
                                int relu, const unsigned char* packed_w2, const float* plane_bias2, float* y2,
                                double* stats_partial2, int ldy2, int y_coff2, const float* x2, const float* in_params,
                                const float* in_params2, int in_relu, int in_relu2, long x_planar, long y_group_stride,
                                atvs_stream_t stream) {
  if (!x || !packed_w || !y) return ATVS_ERR_NULL;
  if (in_params2 && !x2) return ATVS_ERR_ARG;
  if (groups <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || (Cin % 8)) return ATVS_ERR_SHAPE;
  if (x_planar && (x2 || x_planar < (long)D * H * W * 8)) return ATVS_ERR_ARG;
  if (y_group_stride && y_group_stride < (long)D * H * W * ldy) return ATVS_ERR_ARG;
  if (y_coff < 0 || y_coff + 8 > ldy || (ldy % 4) || (y_coff % 4)) return ATVS_ERR_SHAPE;
  if (plane_bias && D < 2) return ATVS_ERR_ARG;
  if ((double)D * H * W * Cin >= 2147483648.0) return ATVS_ERR_SHAPE;
  if (24.0 * ((double)H * W + W + 64) * (x_planar ? 8 : Cin) >= 2147483648.0) return ATVS_ERR_SHAPE;   // a halo's byte offsets (buffer loads)
  if (packed_w2) {
    if (!y2) return ATVS_ERR_NULL;
    if (y_coff2 < 0 || y_coff2 + 16 > ldy2 || (ldy2 % 4) || (y_coff2 % 4)) return ATVS_ERR_SHAPE;
  } else if (plane_bias2 || y2 || stats_partial2) {
    return ATVS_ERR_ARG;
  }
  long pb;
  atvs_conv_xb_pack_size(Cin, &pb);
  XbArgs a;
  a.x = x; a.wp = reinterpret_cast<const f16x8*>(packed_w); a.zeros = reinterpret_cast<const float*>(packed_w + (pb - 16));
  a.bias = bias; a.pbias = plane_bias; a.y = y; a.stats = stats_partial;
  a.Di = D; a.Hi = H; a.Wi = W; a.Cin = Cin; a.ldy = ldy; a.ycoff = y_coff;
  a.nchunk = Cin / 8;
  a.tiles_y = (H + XB_TY - 1) / XB_TY; a.tiles_x = (W + XB_TXV - 1) / XB_TXV;
  a.ntiles = (int)xb_ntiles(D, H, W);
  a.relu = relu;
  a.wp2 = reinterpret_cast<const f16x8*>(packed_w2); a.pbias2 = plane_bias2; a.y2 = y2; a.stats2 = stats_partial2;
  a.Do2 = (D + 1) / 2; a.Ho2 = (H + 1) / 2; a.Wo2 = (W + 1) / 2; a.ldy2 = ldy2; a.ycoff2 = y_coff2;
  a.pbz = D & 1; a.pby = H & 1; a.pbx = W & 1;
  a.wg = (int)atvs_conv_xp_grid(D, H, W, groups);
  a.gx = x_planar ? x_planar * (Cin / 8) : (long)D * H * W * Cin; a.gy = y_group_stride ? y_group_stride : (long)D * H * W * ldy; a.gpb = (long)H * W * 24;
  a.gy2 = (long)a.Do2 * a.Ho2 * a.Wo2 * ldy2; a.gpb2 = (long)a.Ho2 * a.Wo2 * 48;
  const long blocks = (long)a.wg * groups;
  hipStream_t st = as_stream(stream);
  a.sample_major = (groups == 8) ? 1 : 0;
  a.x2 = x2; a.in_pa = in_params; a.in_pb = in_params2; a.relu_a = in_relu; a.relu_b = in_relu2;
  a.vstride = x_planar ? 8 : Cin;
  a.cstride = x_planar ? x_planar : 8;
  const int pro = x2 ? 2 : (in_params ? 1 : 0);
  int rc;
  if (pro == 0) rc = packed_w2 ? launch_xb<true, 0>(a, blocks, st) : launch_xb<false, 0>(a, blocks, st);
  else if (pro == 1 && packed_w2) rc = launch_xb<true, 1>(a, blocks, st);
  else if (pro == 2 && packed_w2) rc = launch_xb<true, 2>(a, blocks, st);
  else return ATVS_ERR_ARG;
  if (rc) return rc;
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}
