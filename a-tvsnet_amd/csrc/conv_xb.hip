// 3x3x3 SAME stride-1 convolution to EIGHT output channels in x-pair form on the 16-bit matrix cores with SPLIT operands (gfx950):
// the layers conv_b*_0_1, global_refine_3dconv0_1, the photo stem, and from the same staged image their stride-2 sibling
// conv_b*_1_0 / 3dconv1_0 (/root/reference/cnn_wrapper/atvsnet.py StackedUNet / CostVolRefineNet, layer code network.py:165-215).
//
// Arithmetic (round 4): every fp32 operand is split into TWO fp16 pieces, x = h0 + h1 / 2048 with h0 = f16(x) and
// h1 = f16((x - h0) * 2048) (the residual scaled towards fp16's normal range: 22 significant bits for |x| >= 2^-14; smaller values
// keep an ABSOLUTE precision of 2^-36 -- both pieces are then fp16 subnormals, which the instruction honours), w = g0 + g1 /
// 2048 likewise (split by the host packer); THREE products on v_mfma_f32_16x16x32_f16 with fp32 accumulation: h0 g0 into the
// main accumulator, h0 g1 + h1 g0 into a second one that is scaled by 2^-11 once in the epilogue (the dropped h1 g1 is 2^-22 of
// a product).  Measured on MI355X (tools_dev/micro/f16_split_probe.hip): error of a 864-term dot product against a double sum
// 2.8e-7 of the output maximum, against 6.3e-7 for the fp32 matrix cores and 8.0e-7 for the round-3 form (three bf16 pieces, six
// products): fewer accumulation steps.  fp16 DENORMAL inputs are honoured by the instruction (same probe).  Range: |x| or |w| >
// 65504 becomes infinity and the output NaN (loud, not silent): the layers' inputs are batch-normalised activations / features,
// |x| <= sqrt(voxels) + |beta|; the packer refuses such weights.
//
// One K = 32 instruction covers the FOUR x offsets a voxel pair touches x 8 channels: lane group q = x offset xl, so a
// (kd, kh) tap row of an 8-channel chunk is ONE K step, rows = (x parity, channel): 3/4 of the MFMA work useful.  Per 8-channel
// chunk and 4 rows: 9 steps x 3 products x 4 = 108 instructions of 16 cycles per multiplying wavefront, + 21 for the sibling
// (K step i = taps 4 i + q of the 27: 7 steps x 3 products on the wavefront's one sibling row).
//
// Structure (round 4, second half): ONE workgroup of EIGHT wavefronts per CU, two roles.  Wavefronts 0..3 (one per SIMD) MULTIPLY:
// wavefront w owns plane z0 + w of a 4(z) x 4(y) x 32(x) tile (4 main + 4 cross accumulator tiles) and sibling row (w >> 1, w & 1);
// their loop holds LDS reads and MFMAs only.  Wavefronts 4..7 (the second wavefront of each SIMD) STAGE the next stage: the fp32
// halo of an 8-channel chunk (6 x 6 x 34 voxels, buffer loads with out-of-volume slots addressed out of range = zeros), the
// optional prologue (batch norm + ReLU of the producers, skip add), the split, the two piece images [6][6][even / odd x
// interleaved in runs of eight][8 fp16] (23 KB each) of the OTHER of two image buffers, and the chunk's 32 KB of weight pieces
// into the other of two weight buffers; the prologue's parameters of all chunks sit in the last 3.8 KB of LDS.  One barrier per
// stage.  The staging wavefronts' loads are issued by inline assembly and waited for with exact s_waitcnt vmcnt counts (the
// comment at the loads says why; tools_dev/check_xb_inflight.py checks that the compiler moves no loaded register).  Its vector
// arithmetic is scalar fp32 (built with -fno-slp-vectorize): the staging wavefronts compute beside the kernel's own MFMA
// wavefronts on every SIMD.  What bounds the launches is the package's power cap.
#include <cstring>
#include <type_traits>

#include "conv_common.h"

// Development build (-DATVS_XB_DEBUG): per-wavefront cycle counts of the phases (tools_dev/phase_xw.py xb ...).
#ifdef ATVS_XB_DEBUG
__device__ unsigned long long atvs_dbg_xb[512 * 8 * 8];
extern "C" int atvs_debug_read_xw(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(atvs_dbg_xb), sizeof(atvs_dbg_xb));
}
#define XDBG(i) { unsigned long long t_ = clock64(); dbg_acc[i] += t_ - dbg_t; dbg_t = t_; }
#else
#define XDBG(i)
#endif

namespace {

constexpr int XB_TZ = 4, XB_TY = 4, XB_TXV = 32;
constexpr int XB_HZ = XB_TZ + 2, XB_HY = XB_TY + 2, XB_HX = XB_TXV + 2;
constexpr int XB_ROWB = 656;                                // an image row: 17 even-x + 17 odd-x voxels, see xb_col
constexpr int XB_IMG = XB_HZ * XB_HY * XB_ROWB;             // 23,616 bytes per piece
constexpr int XB_SLOTS = XB_HZ * XB_HY * XB_HX * 2;         // float4 slots of the fp32 halo of a chunk
constexpr int XB_MAXS = (XB_SLOTS + 255) / 256;             // 10 per producer thread
constexpr int XB_JC = 9;                                    // main K steps per chunk: (kd, kh)
constexpr int XB_J2 = 7;                                    // sibling K steps per chunk: taps 4 i + q
constexpr int XB_NP = 2;                                    // operand pieces
constexpr int XB_WMAIN = XB_JC * XB_NP * 1024;              // bytes of a chunk's main weight pieces: [step][piece][lane][8 fp16]
constexpr int XB_WSIB = XB_J2 * XB_NP * 1024;               // ... of its sibling weight pieces
constexpr int XB_WBUF = XB_WMAIN + XB_WSIB;                 // one LDS weight buffer (32 KB)
constexpr int XB_IBUF = XB_NP * XB_IMG;                     // one image buffer: the two piece images of a stage
constexpr int XB_WOFF = 2 * XB_IBUF;                        // two image buffers, then two weight buffers
constexpr int XB_POFF = XB_WOFF + 2 * XB_WBUF;              // 160,000: the prologue's parameters, [2 sources][3][Cin] floats
constexpr int XB_PBYTES = 160 * 1024 - XB_POFF;             // 3,840 bytes: Cin <= 160 with a prologue
constexpr int XB_LDS = XB_POFF + XB_PBYTES;                 // all of the CU's 163,840 bytes
static_assert(XB_LDS <= 160 * 1024 && XB_PBYTES >= 2 * 3 * 64 * 4, "one workgroup per CU");
static_assert((XB_WMAIN / 16) % 128 == 0 && (XB_WBUF / 16) % 256 == 0, "weight slots: whole wavefronts per source");
constexpr float XB_RS = 2048.f, XB_IRS = 1.f / 2048.f;      // scale of the residual piece and its inverse (exact powers of two)
static_assert(((XB_TZ - 1) * XB_HY + 2 * XB_HY + XB_HY) * XB_ROWB <= XB_IMG, "fragment reads stay inside the images");
static_assert((XB_IMG % 16) == 0 && (XB_ROWB % 16) == 0, "16-byte fragments");

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// buffer_load_dwordx4 (offen).  hipcc 7.2's __builtin_amdgcn_raw_buffer_load_b128 compiles to a ONE-dword load whose value is
// splat over the four components (checked in the ISA), so the LLVM intrinsic is bound by name instead.

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
  long piece_bytes;      // x in pieces: bytes between the two piece planes of a chunk (D * H * W * 16)
};

// Byte offset inside an image row of the voxel with x parity `par` and index i = x / 2 (0..16): even and odd voxels
// alternate in 128-byte runs of eight; the two voxels of index 16 sit behind the two full runs at 512 (even) and 640 (odd),
// the bytes between them are unused.  A fragment read takes lane group q to parity q & 1, index (q >> 1) + r; with this
// interleave the 16 lanes of every ds_read_b128 lane group ({0-3,12-15 | 20-27}, ...: MI355X_MICROARCH.md) cover all 64
// banks once (index 16 of the odd run must land on banks 32..35: byte 128 modulo 256) -- with the even | odd column split of
// the round-2 kernel conv_xp (16-byte voxels; removed in round 4) every fragment read was 2-way conflicted (PMC: half of the LDS cycles).
__device__ __forceinline__ constexpr int xb_col(int par, int i) {
  return i < 16 ? (i >> 3) * 256 + par * 128 + (i & 7) * 16 : 512 + par * 128;
}

// (the operand split itself: atvs_split2_f16, common.h -- five vector instructions per two values)

// WS: the launch streams its weights (more than one 8-channel chunk); else the one chunk's pieces stay resident in buffer 0
// PIECES: x already holds the two fp16 pieces of every value (atvs_warp_planes(pieces)): the staging wavefronts only MOVE them,
// global -> LDS by LDS-DMA (no registers, no arithmetic); no prologue
template <bool SIB, int PRO, bool WS, bool PIECES>
__global__ __launch_bounds__(512, 1) void conv_xb_kernel(XbArgs p) {
  static_assert(!PIECES || PRO == 0, "pieces come finished: no prologue");
  // the workgroup's two wavefronts per SIMD take the SIMD's whole register file (256 each): no wavefront of ANOTHER kernel runs
  // beside this one's 16-bit MFMAs
  asm volatile("" ::: "v255");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int MAXS = XB_MAXS, J2 = SIB ? XB_J2 : 0, ROWB = XB_ROWB;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave: a scalar (uniform branches)
  const bool producer = wave >= 4;
  const int r = lane & 15, q = lane >> 4;

  // persistent tile list (conv_xw.hip)
  const int G = p.wg;
  const int grp = p.sample_major ? (int)(blockIdx.x & 7) : (int)(blockIdx.x / p.wg);
  const int lbk = p.sample_major ? (int)(blockIdx.x >> 3) : (int)(blockIdx.x - grp * p.wg);
  const int xcd = p.sample_major ? 0 : (lbk & 7), tslot = p.sample_major ? lbk : (lbk >> 3);
  const unsigned srow = (unsigned)(grp * p.wg + lbk);
  const int per_xcd = p.sample_major ? p.ntiles : ((p.ntiles + 7) >> 3);
  const int slots_per_xcd = p.sample_major ? G : (G >> 3);
  int my_tiles = 0;
  {
    int last = min(per_xcd, p.ntiles - xcd * per_xcd);
    if (tslot < last) my_tiles = (last - tslot + slots_per_xcd - 1) / slots_per_xcd;
  }
  const int nstage = my_tiles * p.nchunk;
  constexpr bool wstream = WS;

  // The workgroup's tiles are tl0, tl0 + step, ...: their (x, y, z) block coordinates are walked by carries -- the scalar unit is
  // shared by the CU's four SIMDs and a division chain per stage cost the staging wavefronts ~800 cycles of dependent scalar
  // instructions (phase timers).
  struct TileWalk { int bx, by, bz; };
  const int tstep = slots_per_xcd;
  const int step_x = tstep % p.tiles_x, step_r = tstep / p.tiles_x, step_y = step_r % p.tiles_y, step_z = step_r / p.tiles_y;
  TileWalk tw0;
  {
    const int tl = xcd * per_xcd + tslot, rest = tl / p.tiles_x;
    tw0.bx = tl % p.tiles_x; tw0.by = rest % p.tiles_y; tw0.bz = rest / p.tiles_y;
  }
  auto tile_next = [&](TileWalk* w) __attribute__((always_inline)) {
    w->bx += step_x;
    int c = w->bx >= p.tiles_x;
    w->bx -= c ? p.tiles_x : 0;
    w->by += step_y + c;
    c = w->by >= p.tiles_y;
    w->by -= c ? p.tiles_y : 0;
    w->bz += step_z + c;
  };

  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
  float ssum2[4] = {0.f, 0.f, 0.f, 0.f}, ssq2[4] = {0.f, 0.f, 0.f, 0.f};
#ifdef ATVS_XB_DEBUG
  unsigned long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long dbg_t = clock64();
  const unsigned long long dbg_w0 = wall_clock64(), dbg_c0 = dbg_t;
#endif

  if (PRO >= 1) {
    // the prologue's parameters of every chunk -> LDS [a: 3][Cin] [b: 3][Cin] (identity where a source has no pending batch norm)
    float* par = reinterpret_cast<float*>(smem + XB_POFF);
    const float* ga = p.in_pa ? p.in_pa + (size_t)grp * 3 * p.Cin : nullptr;
    const float* gb = (PRO == 2 && p.in_pb) ? p.in_pb + (size_t)grp * 3 * p.Cin : nullptr;
    for (int i = tid; i < 3 * p.Cin; i += 512) {
      const float ident = (i >= p.Cin && i < 2 * p.Cin) ? 1.f : 0.f;
      par[i] = ga ? ga[i] : ident;
      if (PRO == 2) par[3 * p.Cin + i] = gb ? gb[i] : ident;
    }
    __syncthreads();
  }
  if (producer && PIECES) {
    // ================= PRODUCER wavefronts, input in pieces: per stage 2 x 6 LDS-DMA instructions per wavefront copy the halo's
    // records (16 bytes = the 8 fp16 of one voxel of one piece) from the chunk's two piece planes straight into the two piece
    // images, 8 more the weight pieces.  An image is a sequence of 16-byte records in LDS order (41 per row: xb_col); a DMA
    // instruction fills 64 consecutive records, each lane FETCHING the voxel its record holds (the gather is on the global
    // side; holes of the row layout and voxels outside the volume get an out-of-range offset = zeros).  Stage t is requested at
    // the start of the iteration in which the multiplying wavefronts work on stage t - 1 and must have landed at that iteration's
    // barrier: a whole stage of their time, nothing to keep in registers.
    const int pw = wave - 4;
    constexpr int RPR = XB_ROWB / 16;                      // records per image row (41)
    constexpr int NREC = XB_HZ * XB_HY * RPR;              // 1,476 per piece image
    constexpr int NCHK = (NREC + 63) / 64;                 // 24 instructions per piece image
    constexpr int CPW = (NCHK + 3) / 4;                    // 6 per wavefront
    int goffp[CPW];
    unsigned pgp[CPW];
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
      const int rec = (pw + 4 * j) * 64 + lane;
      const int row = min(rec / RPR, XB_HZ * XB_HY - 1), pos = rec % RPR;
      const int zz = row / XB_HY, yy = row % XB_HY;
      // inverse of xb_col: records 0..31 = two runs of (8 even, 8 odd) voxels, record 32 = voxel 32, record 40 = voxel 33
      const int xx = pos < 32 ? 2 * ((pos >> 4) * 8 + (pos & 7)) + ((pos >> 3) & 1) : (pos == 32 ? 32 : 33);
      const bool voxel = rec < NREC && (pos <= 32 || pos == 40);
      goffp[j] = ((zz * p.Hi + yy) * p.Wi + xx) * 16;                         // BYTES from the halo's first voxel in a piece plane
      pgp[j] = 0x808080u | (unsigned)(voxel ? zz : 0x7f) | ((unsigned)yy << 8) | ((unsigned)xx << 16);
    }
    const unsigned char* __restrict__ xgb = reinterpret_cast<const unsigned char*>(p.x) + (size_t)grp * p.gx * 4;
    auto w_dma = [&](int chunk, int buf) __attribute__((always_inline)) {
      const unsigned char* gm = reinterpret_cast<const unsigned char*>(p.wp) + (size_t)chunk * XB_WMAIN + lane * 16;
      const unsigned char* gs = SIB ? reinterpret_cast<const unsigned char*>(p.wp2) + (size_t)chunk * XB_WSIB + lane * 16 : gm;
      constexpr int NWP = SIB ? XB_WBUF / 1024 / 4 : (XB_WMAIN / 1024 + 3) / 4;        // 8 | 5 kilobyte pieces per wavefront
#pragma unroll
      for (int j = 0; j < NWP; ++j) {
        const int pk = pw + 4 * j;                             // wave-uniform
        if (!SIB && pk >= XB_WMAIN / 1024) continue;
        const unsigned char* src = pk < XB_WMAIN / 1024 ? gm + pk * 1024 : gs + (pk - XB_WMAIN / 1024) * 1024;
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                         (void __attribute__((address_space(3)))*)(smem + XB_WOFF + buf * XB_WBUF + pk * 1024), 16, 0, 0);
      }
    };
    TileWalk twn = tw0;
    int chn = 0, tiles_left = my_tiles;
    unsigned vmask = 0;
    for (int t = 0; t < nstage; ++t) {
      const int gz0 = twn.bz * XB_TZ - 1, gy0 = twn.by * XB_TY - 1, gx0 = twn.bx * XB_TXV - 1;
      if (chn == 0) {                                          // a new tile: which records lie outside the volume
        const unsigned lo = (unsigned)(gz0 < 0) | ((unsigned)(gy0 < 0) << 8) | ((unsigned)(gx0 < 0) << 16);
        const unsigned hi1 = (unsigned)(min(p.Di - 1 - gz0, 0x7e) + 1) | ((unsigned)(min(p.Hi - 1 - gy0, 0x7e) + 1) << 8) |
                             ((unsigned)(min(p.Wi - 1 - gx0, 0x7e) + 1) << 16);
        vmask = 0;
#pragma unroll
        for (int j = 0; j < CPW; ++j) {
          const unsigned t1 = pgp[j] - lo, t2 = hi1 + ~pgp[j];
          vmask |= ((((t1 & t2) & 0x808080u) == 0x808080u) ? 0u : 1u) << j;
        }
      }
      const long org = ((long)(gz0 * p.Hi + gy0) * p.Wi + gx0) * 16;       // bytes; the halo's first voxel (may lie in front of the plane)
      const unsigned char* base = xgb + (size_t)chn * p.cstride * 4 + org;
      const int ib = (t & 1) * XB_IBUF;
      XDBG(2)
#pragma unroll
      for (int pc = 0; pc < XB_NP; ++pc) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(base + (size_t)pc * p.piece_bytes), 0,
                                                                             0x7ffffff0, 0x00020000);
#pragma unroll
        for (int j = 0; j < CPW; ++j) {
          const int c = pw + 4 * j;                            // wave-uniform: this instruction's 64 records
          if (c >= NCHK) continue;
          const unsigned voff = (unsigned)goffp[j] | (unsigned)__builtin_amdgcn_sbfe(vmask, j, 1);
          if (c * 64 + lane < NREC)                            // (the last instruction of an image: 4 records)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (void __attribute__((address_space(3)))*)(smem + ib + pc * XB_IMG + c * 1024), 16,
                                                     (int)voff, 0, 0, 0);
        }
      }
      // the weight pieces BEHIND the halo: L2 hits, they land while the halo's HBM round trip is still under way (in front of
      // it: the same time, measured)
      if (WS || t == 0) w_dma(chn, WS ? (t & 1) : 0);
      XDBG(3)
      // -> stage t + 1
      if (chn + 1 < p.nchunk) {
        ++chn;
      } else if (tiles_left > 1) {
        --tiles_left;
        chn = 0;
        tile_next(&twn);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // stage t has landed
      XDBG(1)
      __syncthreads();
      XDBG(5)
    }
    __syncthreads();                                           // the consumers' last stage
  } else if (producer) {
    // ================= PRODUCER wavefronts (4..7): global -> (prologue, split) -> the piece images and weight buffers of the
    // NEXT stage, while the consumers multiply the current one.  One barrier per stage.
    const int ptid = tid - 256;
#ifdef ATVS_XB_PRIO
    __builtin_amdgcn_s_setprio(ATVS_XB_PRIO);
#endif
    // per-slot constants: float4 = channels 4 c4 .. of a halo voxel of the fp32 chunk -> 8 bytes of each piece image
    int goff[MAXS], laddr[MAXS];
    unsigned pg[MAXS];
#pragma unroll
    for (int i = 0; i < MAXS; ++i) {
      int s = ptid + i * 256;
      const bool live = s < XB_SLOTS;
      s = min(s, XB_SLOTS - 1);
      const int c4 = s & 1, v = s >> 1;
      const int xx = v % XB_HX, v2 = v / XB_HX;
      const int yy = v2 % XB_HY, zz = v2 / XB_HY;
      goff[i] = (((zz * p.Hi + yy) * p.Wi + xx) * p.vstride + c4 * 4) * 4;       // BYTES from the halo's first voxel
      laddr[i] = (zz * XB_HY + yy) * XB_ROWB + xb_col(xx & 1, xx >> 1) + c4 * 8;
      pg[i] = 0x808080u | (unsigned)(live ? zz : 0x7f) | ((unsigned)yy << 8) | ((unsigned)xx << 16);
    }
    const bool last_live = ptid + (MAXS - 1) * 256 < XB_SLOTS;
    const float* __restrict__ xg = p.x + (size_t)grp * p.gx;
    const float* __restrict__ xg2 = (PRO == 2) ? p.x2 + (size_t)grp * p.gx : nullptr;
    const int c4t = ptid & 1;                        // every slot of this thread is channel group c4t of the chunk

    struct PfTile {
      const float* xb;
      const float* xb2;
      int org;
      unsigned lo, hi1;
    };
    auto pf_tile = [&](const TileWalk& w, int ch) __attribute__((always_inline)) {
      PfTile T;
      const int gz0 = w.bz * XB_TZ - 1, gy0 = w.by * XB_TY - 1, gx0 = w.bx * XB_TXV - 1;
      T.xb = xg + (size_t)ch * p.cstride;
      T.xb2 = (PRO == 2) ? xg2 + (size_t)ch * p.cstride : nullptr;
      T.org = ((gz0 * p.Hi + gy0) * p.Wi + gx0) * p.vstride;
      T.lo = (unsigned)(gz0 < 0) | ((unsigned)(gy0 < 0) << 8) | ((unsigned)(gx0 < 0) << 16);
      T.hi1 = (unsigned)(min(p.Di - 1 - gz0, 0x7e) + 1) | ((unsigned)(min(p.Hi - 1 - gy0, 0x7e) + 1) << 8) |
              ((unsigned)(min(p.Wi - 1 - gx0, 0x7e) + 1) << 16);
      return T;
    };
    // The halo is fetched with BUFFER loads from a descriptor whose base is the halo's first voxel of the chunk (a scalar add
    // per stage): a slot's address is then a per-kernel constant (goff) and a slot outside the volume gets an offset beyond
    // the descriptor's range -- the load returns zeros, no pointer select.  Which slots are inside is a property of the TILE:
    // the mask is recomputed only when the fetched stage starts a new tile (every nchunk-th stage).
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
    // ---- The producers' vector-memory operations are issued by INLINE ASSEMBLY and waited for by hand.  Vector-memory operations
    // retire in order; a stage's staging consumes loads that are a whole stage old while it issues the next stage's, so every
    // wait must name exactly how many YOUNGER operations may stay in flight.  The compiler's own counts inside this loop are
    // conservative ("all but the newest nine"): with compiler-visible loads each stage waited for requests it had just issued
    // (weights in front of the slot loop: an L2 round trip at the first slots; behind it: the halo's HBM round trip at the weight
    // write; LDS-DMA: vmcnt(0) in front of the next LDS write -- 1,400 to 1,900 cycles per stage, phase timers).  The loaded
    // registers are only read through xb_wait (a "+v" operand ties them to the wait), never copied while in flight (checked in
    // the ISA: tools_dev/check_xb_inflight.py); the queue of a producer wavefront holds nothing else (the prologue's
    // parameters come from LDS).
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    f32x4 pf[MAXS];
    f32x4 pf2[PRO == 2 ? MAXS : 1];
    constexpr int LPS = PRO == 2 ? 2 : 1;                             // halo loads per slot
    auto make_rsrc = [&](const void* base) __attribute__((always_inline)) {       // raw buffer descriptor: no stride, no bounds but 2 GB
      const unsigned long long a = reinterpret_cast<unsigned long long>(base);
      i32x4 d;
      d[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
      d[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));
      d[2] = 0x7ffffff0;
      d[3] = 0x00020000;
      return d;
    };
    auto pf_rsrc = [&](const float* base, int org) __attribute__((always_inline)) {
#ifdef ATVS_XB_HOT       // development: every tile fetches the same halo (cache hits) -- is the launch bound by the memory path?
      return make_rsrc(p.x + (org & 1023));
#else
      return make_rsrc(base + org);
#endif
    };
    auto pf_slot = [&](const i32x4& ra, const i32x4& rb, int i, unsigned vmask) __attribute__((always_inline)) {
      const unsigned voff = (unsigned)goff[i] | (unsigned)__builtin_amdgcn_sbfe(vmask, i, 1);      // all ones when outside
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(pf[i]) : "v"(voff), "s"(ra));
      if (PRO == 2) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(pf2[i]) : "v"(voff), "s"(rb));
    };
    // slot i's registers are valid afterwards; N = the younger operations that may stay in flight (macros: inline assembly
    // inside a generic lambda cannot name the enclosing function's variables)
#define XB_PF_WAIT(i, N)                                                                                   \
  do {                                                                                                     \
    if (PRO == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(pf[i]), "+v"(pf2[i]) : "n"(N));                \
    else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(pf[i]) : "n"(N));                                       \
  } while (0)

    // prologue transform: arithmetic of bn_apply / bn_add (norm.hip), as conv_xw.hip.  The parameters of all
    // chunks sit in LDS behind the weight buffers (copied there before the first barrier; a source WITHOUT a pending batch norm
    // gets the identity -- mean 0, scale 1, beta 0, no floor: (v - 0) * 1 + 0 == v, a -0 becomes +0: the same sums).
    struct Par { float4 ma, sa, ba, mb, sb, bb; };
    const bool has_a = PRO >= 1 && p.in_pa != nullptr, has_b = PRO == 2 && p.in_pb != nullptr;
    const float floor_a = p.relu_a ? 0.f : -__builtin_huge_valf(), floor_b = p.relu_b ? 0.f : -__builtin_huge_valf();
    auto load_par = [&](int chunk) __attribute__((always_inline)) {
      Par P;
      const float* pa = reinterpret_cast<const float*>(smem + XB_POFF) + chunk * 8 + c4t * 4;
      P.ma = ld4(pa); P.sa = ld4(pa + p.Cin); P.ba = ld4(pa + 2 * p.Cin);
      if (PRO == 2) {
        const float* pb = pa + 3 * p.Cin;
        P.mb = ld4(pb); P.sb = ld4(pb + p.Cin); P.bb = ld4(pb + 2 * p.Cin);
      }
      return P;
    };
    // (canonical once: fmaxf below then needs no per-use quieting of its bound)
    const float lo_a = __builtin_canonicalizef(has_a ? floor_a : -__builtin_huge_valf()),
                lo_b = __builtin_canonicalizef(has_b ? floor_b : -__builtin_huge_valf());
    auto bn1 = [&](float v, float m, float sc, float be, float lo) __attribute__((always_inline)) {
      return fmaxf(atvs_bn1(v, sc, atvs_bn_shift(m, sc, be)), lo);      // bn_apply's arithmetic (norm.hip, common.h): one fused multiply-add on the shift
    };
    // slot i of the stage in the registers (prologue, split) -> image buffer at byte `ib`
    auto stage_slot = [&](int i, const Par& P, unsigned vmask, int ib) __attribute__((always_inline)) {
      float4 v = make_float4(pf[i][0], pf[i][1], pf[i][2], pf[i][3]);
      if (PRO >= 1) {
        v.x = bn1(v.x, P.ma.x, P.sa.x, P.ba.x, lo_a); v.y = bn1(v.y, P.ma.y, P.sa.y, P.ba.y, lo_a);
        v.z = bn1(v.z, P.ma.z, P.sa.z, P.ba.z, lo_a); v.w = bn1(v.w, P.ma.w, P.sa.w, P.ba.w, lo_a);
        if (PRO == 2) {
          const float4 u = make_float4(pf2[i][0], pf2[i][1], pf2[i][2], pf2[i][3]);
          v.x += bn1(u.x, P.mb.x, P.sb.x, P.bb.x, lo_b); v.y += bn1(u.y, P.mb.y, P.sb.y, P.bb.y, lo_b);
          v.z += bn1(u.z, P.mb.z, P.sb.z, P.bb.z, lo_b); v.w += bn1(u.w, P.mb.w, P.sb.w, P.bb.w, lo_b);
        }
      }
      if (i < MAXS - 1 || last_live) {
        uint2 p0, p1;
        atvs_split2_f16(v.x, v.y, XB_RS, &p0.x, &p1.x);
        atvs_split2_f16(v.z, v.w, XB_RS, &p0.y, &p1.y);
        if (PRO >= 1) {
          // a slot OUTSIDE the volume must stage zeros (the transform of the zeros the load returned is not zero): the four
          // packed words are masked (pieces of 0 are 0) -- one bit-field extract + four ANDs instead of selects on the values
          const unsigned keep = ~(unsigned)__builtin_amdgcn_sbfe(vmask, i, 1);
          p0.x &= keep; p0.y &= keep; p1.x &= keep; p1.y &= keep;
        }
        *reinterpret_cast<uint2*>(smem + ib + laddr[i]) = p0;
        *reinterpret_cast<uint2*>(smem + ib + XB_IMG + laddr[i]) = p1;
      }
    };

    // weight pieces of a chunk: global -> registers -> LDS buffer ([main steps][sibling steps], XB_WBUF bytes): requested at the
    // END of a stage's staging (behind its halo requests), written into LDS at the end of the next one's -- a whole stage to
    // arrive, and the wait in front of the write names exactly the halo requests that are younger.
    constexpr int NW = SIB ? XB_WBUF / 4096 : (XB_WMAIN + 4095) / 4096;        // float4 slots per thread: 8 | 5
    f32x4 wreg[NW];
    // (buffer loads: ONE per-thread byte offset serves all NW loads, the piece and the chunk go into the scalar offset)
    // (the sibling pieces follow the main ones in the LDS buffer: their descriptor starts XB_WMAIN bytes in front of them, the
    // scalar offset -- an UNSIGNED addend -- stays positive)
    const i32x4 wrs_main = make_rsrc(p.wp),
                wrs_sib = make_rsrc(SIB ? reinterpret_cast<const unsigned char*>(p.wp2) - XB_WMAIN : reinterpret_cast<const unsigned char*>(p.wp));
    const int woff = ptid * 16;
    auto w_request = [&](int chunk) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const bool is_main = j < 4 || (j == 4 && wave < 6);    // XB_WMAIN / 16 = 4.5 x 256: main | sibling is wave-uniform
        // (no sibling: the lanes past the main pieces re-read the chunk's first ones -- every wavefront issues NW loads, the
        // waits count them)
        const int so_main = chunk * XB_WMAIN + ((SIB || is_main) ? j * 4096 : 0);
        const int so_sib = chunk * XB_WSIB + j * 4096;
        if (is_main || !SIB) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(wreg[j]) : "v"(woff), "s"(wrs_main), "s"(so_main));
        else asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(wreg[j]) : "v"(woff), "s"(wrs_sib), "s"(so_sib));
      }
    };
#define XB_W_WAIT(N)                                                                                                        \
  do {                                                                                                                      \
    if (SIB) asm volatile("s_waitcnt vmcnt(%8)" : "+v"(wreg[0]), "+v"(wreg[1]), "+v"(wreg[2]), "+v"(wreg[3]), "+v"(wreg[4]),  \
                          "+v"(wreg[NW > 5 ? 5 : 0]), "+v"(wreg[NW > 6 ? 6 : 0]), "+v"(wreg[NW > 7 ? 7 : 0]) : "n"(N));      \
    else asm volatile("s_waitcnt vmcnt(%5)" : "+v"(wreg[0]), "+v"(wreg[1]), "+v"(wreg[2]), "+v"(wreg[3]), "+v"(wreg[4]) : "n"(N)); \
  } while (0)
    auto w_land = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const int idx = ptid + 256 * j;
        if (SIB || idx < XB_WMAIN / 16) *reinterpret_cast<f32x4*>(smem + XB_WOFF + buf * XB_WBUF + idx * 16) = wreg[j];
      }
    };

    // Stage t is staged during iteration t - 1 (stage 0 in front of the loop).  ROLLING prefetch: as soon as slot i of stage
    // t has left its registers for LDS, the same registers receive slot i of stage t + 1 -- every load has a whole stage to
    // arrive and one register set suffices.  vcur / vnext: the outside-the-volume masks of the stage in the registers / of the
    // stage being requested.  Queue of a wavefront when a stage's staging starts, oldest first: that stage's halo (MAXS x LPS
    // loads), then its weights (NW loads, if the launch streams them).
    unsigned vcur = 0, vnext = 0;
    TileWalk twn = tw0;                    // tile of the stage being REQUESTED
    int chn = 0, tiles_left = my_tiles;    // its chunk; tiles not yet requested
    auto advance = [&]() __attribute__((always_inline)) {            // -> the stage after the requested one; false at the end
      if (chn + 1 < p.nchunk) { ++chn; return false; }
      if (tiles_left <= 1) return false;                                // past the end: a harmless re-read of the last stage
      --tiles_left;
      chn = 0;
      tile_next(&twn);
      return true;
    };
    // ONE loop, one inline-assembly site per load: iteration t = -1 only requests (stage 0's halo and weights), iteration t >= 0
    // stages stage t and requests stage t + 1.  (A separate copy of the body in front of the loop made the compiler shuffle
    // the loaded registers between the two register assignments -- v_mov of registers whose loads were still in flight.)
    if (!wstream && nstage > 0) {        // one chunk: its weights stay resident in buffer 0 (compiler-visible loads: nothing else in flight)
      const unsigned char* gm = reinterpret_cast<const unsigned char*>(p.wp);
      const unsigned char* gs = SIB ? reinterpret_cast<const unsigned char*>(p.wp2) - XB_WMAIN : gm;
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const int idx = ptid + 256 * j;
        if (SIB || idx < XB_WMAIN / 16)
          *reinterpret_cast<float4*>(smem + XB_WOFF + idx * 16) = ld4(reinterpret_cast<const float*>((idx < XB_WMAIN / 16 ? gm : gs) + (size_t)idx * 16));
      }
    }
    // (the two-source form stages twice the halo registers: it is built for ONE chunk -- Cin = 8, the U-Net's stack inputs --
    // whose weights stay resident; the host entry refuses more)
    static_assert(!(PRO == 2 && WS), "the two-source form does not stream weights");
    for (int t = -1; t < nstage; ++t) {
      const bool live = t >= 0;                                      // (uniform)
      Par P;
      if (PRO >= 1 && live) P = load_par(chn);
      const bool newtile = live ? advance() : true;                  // (twn, chn): stage t + 1
      const PfTile T = pf_tile(twn, chn);
      if (newtile) vnext = pf_mask(T);
      if (!live) vcur = vnext;
      const i32x4 rsa = pf_rsrc(T.xb, T.org), rsb = pf_rsrc(PRO == 2 ? T.xb2 : T.xb, T.org);
      const int ib = (t & 1) * XB_IBUF;
      XDBG(2)
#pragma unroll
      for (int i = 0; i < MAXS; ++i) {
        if (live) {
          // younger than slot i of stage t: its slots i+1.., its weights, slots ..i-1 of stage t + 1 -- the same count for every i
          XB_PF_WAIT(i, (MAXS - 1) * LPS + (WS ? NW : 0));
          stage_slot(i, P, vcur, ib);
        }
        pf_slot(rsa, rsb, i, vnext);
      }
      XDBG(3)
      if (WS) {
        if (live) {
          XB_W_WAIT(MAXS * LPS);                                     // younger than stage t's weights: stage t + 1's halo
          w_land(t & 1);
        }
        w_request(chn);                                              // stage t + 1's
      }
      vcur = vnext;
      XDBG(0)
      if (live) {
        XDBG(1)
        __syncthreads();
        XDBG(5)
      }
    }
    __syncthreads();                                                 // the consumers' last stage
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the requests past the last stage
  } else {
    // ================= CONSUMER wavefronts (0..3): wavefront w owns plane z0 + w of the tile (4 rows: 4 main + 4 cross
    // accumulator tiles) and sibling row (w >> 1, w & 1)
    // this lane's fragment (the 8 channels of one voxel of a piece image) of halo row 0 of the wavefront's plane at its x
    // offset xl = q; piece images XB_IMG apart
    int fb[XB_NP];
#pragma unroll
    for (int pc = 0; pc < XB_NP; ++pc) fb[pc] = pc * XB_IMG + (wave * XB_HY) * XB_ROWB + xb_col(q & 1, (q >> 1) + r);
    // sibling: this lane's tap of step i is 4 i + q (taps past 26: zero weights, tap 26's fragment); per-step byte offsets
    int sd[SIB ? XB_J2 : 1];
    if (SIB) {
      const int row0 = (2 * (wave >> 1) + 1 - p.pbz) * XB_HY + (2 * (wave & 1) + 1 - p.pby);
#pragma unroll
      for (int i = 0; i < XB_J2; ++i) {
        const int tap = min(4 * i + q, 26);
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        const int xh = kw + 1 - p.pbx;                                  // halo x of output column 0
        sd[i] = (row0 + kd * XB_HY + kh) * XB_ROWB + xb_col(xh & 1, (xh >> 1) + r);
      }
    }
    float* __restrict__ yg = p.y + (size_t)grp * p.gy;
    float* __restrict__ y2g = p.y2 + (size_t)grp * p.gy2;
    const float* __restrict__ pbg = p.pbias ? p.pbias + (size_t)grp * p.gpb : nullptr;
    const float* __restrict__ pb2g = p.pbias2 ? p.pbias2 + (size_t)grp * p.gpb2 : nullptr;
    const bool any_pb = pbg != nullptr || pb2g != nullptr;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bv = ld4(p.bias + (q & 1) * 4);
    f32x4 acc[XB_TY], accx[XB_TY];       // h0 g0 | (h0 g1 + h1 g0) * 2^11
    f32x4 acc2, acc2x;

    XDBG(0)
    __syncthreads();                       // stage 0 is in LDS
    XDBG(5)
    int ch = 0;
    TileWalk tw = tw0;
    for (int stage = 0; stage < nstage; ++stage) {
      if (ch == 0) {
#pragma unroll
        for (int t = 0; t < XB_TY; ++t) acc[t] = accx[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc2 = acc2x = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      const int ib = (stage & 1) * XB_IBUF;
      const int wb = XB_WOFF + (wstream ? (stage & 1) : 0) * XB_WBUF + lane * 16;
      const bool last_chunk = (ch == p.nchunk - 1);

      const int tz0 = tw.bz * XB_TZ, ty0 = tw.by * XB_TY, tx0 = tw.bx * XB_TXV;
      const int zo = tz0 + wave, xo = tx0 + 2 * r + (q >> 1), co = (q & 1) * 4;
      const bool evox_ok = zo < p.Di && xo < p.Wi;
      const size_t erow = (size_t)p.Wi * p.ldy;
      const size_t eo = (((size_t)zo * p.Hi + ty0) * p.Wi + xo) * (size_t)p.ldy + p.ycoff + co;
      const size_t epb_off = ((size_t)ty0 * p.Wi + xo) * 24 + plane_variant(zo - 1, p.Di) * 8 + co;
      auto erow_ok = [&](int t) __attribute__((always_inline)) { return evox_ok && ty0 + t < p.Hi; };
      float4 epb[XB_TY], epb2;
      const int zo2 = (tz0 >> 1) + (wave >> 1), yo2 = (ty0 >> 1) + (wave & 1), xo2 = (tx0 >> 1) + r;
      const bool e2_ok = SIB && zo2 < p.Do2 && yo2 < p.Ho2 && xo2 < p.Wo2;
      const size_t eo2 = (((size_t)zo2 * p.Ho2 + yo2) * p.Wo2 + xo2) * (size_t)p.ldy2 + p.ycoff2 + 4 * q;

      // ---- main K loop: per kd TWO phases -- the SIX halo rows of input piece h0 at depth kd serve the three kh steps x both
      // weight pieces (24 MFMAs), then the six rows of h1 serve kh x g0 (12 MFMAs).  Rows of one piece are requested during
      // the other piece's phase, the six weight fragments of the next kd during the h0 phase; nothing but LDS reads and MFMAs
      // in the loop (the staging runs in the producer wavefronts of the same SIMDs).
      f16x8 Bq[XB_NP][XB_HY], B2[3][XB_NP], A[2][3][XB_NP], A2[3][XB_NP];
      auto request_A = [&](int kd) __attribute__((always_inline)) {           // the weight fragments of steps 3 kd .. 3 kd + 2
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int pc = 0; pc < XB_NP; ++pc)
            A[kd & 1][kh][pc] = *reinterpret_cast<const f16x8*>(smem + wb + ((kd * 3 + kh) * XB_NP + pc) * 1024);
      };
      auto request_B = [&](int kd, int pc) __attribute__((always_inline)) {
#pragma unroll
        for (int y = 0; y < XB_HY; ++y) Bq[pc][y] = *reinterpret_cast<const f16x8*>(smem + ib + fb[pc] + (kd * XB_HY + y) * ROWB);
      };
      auto request_2 = [&](int i) __attribute__((always_inline)) {           // sibling step i: two weight pieces, two input pieces
#pragma unroll
        for (int pc = 0; pc < XB_NP; ++pc) {
          A2[i % 3][pc] = *reinterpret_cast<const f16x8*>(smem + wb + XB_WMAIN + (i * XB_NP + pc) * 1024);
          B2[i % 3][pc] = *reinterpret_cast<const f16x8*>(smem + ib + pc * XB_IMG + sd[i]);
        }
      };
      request_A(0);
      request_B(0, 0);
      // the epilogue's depth-plane biases: a whole K loop to arrive (uniform branches).  A launch WITHOUT plane biases issues no load
      // here at all: loads and stores share one in-order counter, so a load of zeros would make every epilogue wait for the
      // acknowledgement of the previous tile's stores (one-chunk launches: an HBM write round trip per stage).  (No zero
      // initialisation in front of the loads either: the compiler turns "zero or loaded" into a select and waits for the load on
      // the spot.)
      if (last_chunk && any_pb) {
#pragma unroll
        for (int t = 0; t < XB_TY; ++t) epb[t] = ld4((pbg && erow_ok(t)) ? pbg + (epb_off + (size_t)t * p.Wi * 24) : p.zeros);
        if (SIB) {
          const size_t o = ((size_t)yo2 * p.Wo2 + xo2) * 48 + plane_variant(2 * zo2 - p.pbz, p.Di) * 16 + 4 * q;
          epb2 = ld4((pb2g && e2_ok) ? pb2g + o : p.zeros);
        }
      }
      asm volatile("" ::: "memory");
      XDBG(0)
#pragma unroll
      for (int ph = 0; ph < 2 * 3; ++ph) {
        const int kd = ph >> 1, pc = ph & 1;
        if (pc == 0) {
          request_B(kd, 1);
          if (kd + 1 < 3) request_A(kd + 1);
          else if (SIB) request_2(0);
        } else {
          if (kd + 1 < 3) request_B(kd + 1, 0);
          else if (SIB) request_2(1);
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
        // the phase's requests do not depend on its MFMAs: two MFMAs, then one LDS read (and the few epilogue loads) in their shadow
#pragma unroll
        for (int g = 0; g < (pc == 0 ? 12 : 6); ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);      // MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // DS read
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // VMEM read
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      // VALU
        }
        asm volatile("" ::: "memory");                            // this phase's requests stay in this phase
        __builtin_amdgcn_sched_barrier(0);
      }
      XDBG(2)
      if (SIB) {
        // 7 steps of 3 MFMAs on the wavefront's one sibling row; fragments requested two steps ahead
#pragma unroll
        for (int i = 0; i < J2; ++i) {
          if (i + 2 < J2) request_2(i + 2);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A2[i % 3][0], B2[i % 3][0], acc2, 0, 0, 0);
          acc2x = __builtin_amdgcn_mfma_f32_16x16x32_f16(A2[i % 3][1], B2[i % 3][0], acc2x, 0, 0, 0);
          acc2x = __builtin_amdgcn_mfma_f32_16x16x32_f16(A2[i % 3][0], B2[i % 3][1], acc2x, 0, 0, 0);
          asm volatile("" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      XDBG(3)
      if (last_chunk) {
        // ---- epilogue (conv_xw.hip's layout): this lane holds channels co..co+3 of voxel xo for the 4 rows of plane zo
        // Every loaded epilogue operand passes through an (empty) assembly statement FIRST: the compiler waits for them here,
        // once.  Otherwise its loop-carried bookkeeping puts s_waitcnt vmcnt(0) in front of each row's first use of the bias --
        // loads and stores share the counter, so every row waited for the previous row's store to be acknowledged (an HBM
        // write round trip per row: 3,700-5,200 cycles per epilogue, phase timers).
        asm volatile("" : "+v"(bv.x), "+v"(bv.y), "+v"(bv.z), "+v"(bv.w));
        if (any_pb) {
          asm volatile("" : "+v"(epb2.x), "+v"(epb2.y), "+v"(epb2.z), "+v"(epb2.w));
#pragma unroll
          for (int t = 0; t < XB_TY; ++t) asm volatile("" : "+v"(epb[t].x), "+v"(epb[t].y), "+v"(epb[t].z), "+v"(epb[t].w));
        } else {                             // (+ 0 as a launch with all-zero biases adds it: the same bits, -0 included)
#pragma unroll
          for (int t = 0; t < XB_TY; ++t) epb[t] = make_float4(0.f, 0.f, 0.f, 0.f);
          epb2 = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        auto store_rows = [&](auto relu_tag) __attribute__((always_inline)) {
#pragma unroll
          for (int t = 0; t < XB_TY; ++t) {
            if (!erow_ok(t)) continue;
            float4 v;
            // fmaf(cross, 2^-11, main): the product by a power of two is exact, i.e. the same value as main + cross * 2^-11 in
            // one instruction
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
            st4_stream(yg + (eo + (size_t)t * erow), v);
            ssum[0] += v.x; ssum[1] += v.y; ssum[2] += v.z; ssum[3] += v.w;
            ssq[0] += v.x * v.x; ssq[1] += v.y * v.y; ssq[2] += v.z * v.z; ssq[3] += v.w * v.w;
          }
        };
        if (p.relu) store_rows(std::true_type{});
        else store_rows(std::false_type{});
        if (SIB && e2_ok) {
          const float4 v = make_float4(__builtin_fmaf(acc2x[0], XB_IRS, acc2[0]) + epb2.x, __builtin_fmaf(acc2x[1], XB_IRS, acc2[1]) + epb2.y,
                                       __builtin_fmaf(acc2x[2], XB_IRS, acc2[2]) + epb2.z, __builtin_fmaf(acc2x[3], XB_IRS, acc2[3]) + epb2.w);
          st4_stream(y2g + eo2, v);
          ssum2[0] += v.x; ssum2[1] += v.y; ssum2[2] += v.z; ssum2[3] += v.w;
          ssq2[0] += v.x * v.x; ssq2[1] += v.y * v.y; ssq2[2] += v.z * v.z; ssq2[3] += v.w * v.w;
        }
        XDBG(4)
      }
      if (++ch == p.nchunk) { ch = 0; tile_next(&tw); }
      __syncthreads();                     // the next stage is in LDS; the producers may overwrite this one's buffers
      XDBG(5)
    }
  }
#ifdef ATVS_XB_DEBUG
  if (lane == 0 && blockIdx.x < 512) {
    dbg_acc[7] = (unsigned long long)nstage;
    dbg_acc[6] = ((wall_clock64() - dbg_w0) << 32) | ((clock64() - dbg_c0) >> 8);   // 100 MHz ticks | shader cycles / 256
    for (int i = 0; i < 8; ++i) atvs_dbg_xb[(blockIdx.x * 8 + wave) * 8 + i] = dbg_acc[i];
  }
#endif

  // ---- per-workgroup partial moments -> row `srow` of stats: [2][16] doubles (columns 0..7 = channels), as conv_xw.hip
  // (after the last barrier nobody reads the images: the reduction borrows their first bytes; the producers hold zeros)
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
      if (!producer && r == 0 && q < 2) {
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
      if (!producer && r == 0) {
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


#undef XB_PF_WAIT
#undef XB_W_WAIT

long xb_ntiles(int D, int H, int W) {
  return (long)((D + XB_TZ - 1) / XB_TZ) * ((H + XB_TY - 1) / XB_TY) * ((W + XB_TXV - 1) / XB_TXV);
}

template <bool SIB, int PRO, bool WS, bool PIECES = false>
int launch_xb(const XbArgs& a, long blocks, hipStream_t s) {
  const size_t lds = XB_LDS;
  static AtvsAttrOnce lds_once;                   // per kernel instantiation (this function is a template / has one kernel)
  if (const int rc_ = atvs_set_max_lds_once(lds_once, reinterpret_cast<const void*>(conv_xb_kernel<SIB, PRO, WS, PIECES>), 160 * 1024)) return rc_;
  hipLaunchKernelGGL((conv_xb_kernel<SIB, PRO, WS, PIECES>), dim3((unsigned)blocks), dim3(512), lds, s, a);
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
// rows = atvs_conv_xpair_grid.  fp32-class results (split-fp16 operands, fp32 accumulation); rounding differs from the fp32 forms.
// Beyond atvs_conv_xw_f32: x_planar may come WITH in_params (one pending batch norm, the refinement's chunk-planar concat;
// not with x2), and y_group_stride != 0 = floats between the samples of y (default D*H*W*ldy): the photo stem writes
// plane 0 of each sample's chunk-planar concat (ldy = 8, y_group_stride = 4 * plane stride).
extern "C" int atvs_conv_xb_f32(const float* x, const unsigned char* packed_w, const float* bias, const float* plane_bias,
                                float* y, double* stats_partial, int groups, int D, int H, int W, int Cin, int ldy, int y_coff,
                                int relu, const unsigned char* packed_w2, const float* plane_bias2, float* y2,
                                double* stats_partial2, int ldy2, int y_coff2, const float* x2, const float* in_params,
                                const float* in_params2, int in_relu, int in_relu2, long x_planar, long y_group_stride,
                                int x_pieces, atvs_stream_t stream) {
  if (!x || !packed_w || !y) return ATVS_ERR_NULL;
  if (in_params2 && !x2) return ATVS_ERR_ARG;
  if (x2 && Cin != 8) return ATVS_ERR_SHAPE;             // the two-source form: one 8-channel chunk (resident weights)
  if (groups <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || (Cin % 8)) return ATVS_ERR_SHAPE;
  if (x_planar && (x2 || x_planar < (long)D * H * W * 8)) return ATVS_ERR_ARG;
  if (x_pieces && (!x_planar || in_params || x2)) return ATVS_ERR_ARG;      // pieces: chunk-planar, finished values (no prologue)
  if (y_group_stride && y_group_stride < (long)D * H * W * ldy) return ATVS_ERR_ARG;
  if (y_coff < 0 || y_coff + 8 > ldy || (ldy % 4) || (y_coff % 4)) return ATVS_ERR_SHAPE;
  if (plane_bias && D < 2) return ATVS_ERR_ARG;
  if ((in_params || x2) && 2 * 3 * Cin * 4 > XB_PBYTES) return ATVS_ERR_SHAPE;          // the prologue's parameters live in LDS
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
  a.wg = (int)atvs_conv_xpair_grid(D, H, W, groups);
  a.gx = x_planar ? x_planar * (Cin / 8) : (long)D * H * W * Cin; a.gy = y_group_stride ? y_group_stride : (long)D * H * W * ldy; a.gpb = (long)H * W * 24;
  a.gy2 = (long)a.Do2 * a.Ho2 * a.Wo2 * ldy2; a.gpb2 = (long)a.Ho2 * a.Wo2 * 48;
  const long blocks = (long)a.wg * groups;
  hipStream_t st = as_stream(stream);
  a.sample_major = (groups == 8) ? 1 : 0;
  a.x2 = x2; a.in_pa = in_params; a.in_pb = in_params2; a.relu_a = in_relu; a.relu_b = in_relu2;
  a.vstride = x_planar ? 8 : Cin;
  a.cstride = x_planar ? x_planar : 8;
  a.piece_bytes = (long)D * H * W * 16;
  const int pro = x2 ? 2 : (in_params ? 1 : 0);
  int rc;
  const bool ws = a.nchunk > 1;
  if (x_pieces && packed_w2) rc = ws ? launch_xb<true, 0, true, true>(a, blocks, st) : launch_xb<true, 0, false, true>(a, blocks, st);
  else if (x_pieces) rc = ws ? launch_xb<false, 0, true, true>(a, blocks, st) : launch_xb<false, 0, false, true>(a, blocks, st);
  else if (pro == 0 && packed_w2) rc = ws ? launch_xb<true, 0, true>(a, blocks, st) : launch_xb<true, 0, false>(a, blocks, st);
  else if (pro == 0) rc = ws ? launch_xb<false, 0, true>(a, blocks, st) : launch_xb<false, 0, false>(a, blocks, st);
  else if (pro == 1 && packed_w2) rc = ws ? launch_xb<true, 1, true>(a, blocks, st) : launch_xb<true, 1, false>(a, blocks, st);
  else if (pro == 2 && packed_w2) rc = launch_xb<true, 2, false>(a, blocks, st);
  else return ATVS_ERR_ARG;
  if (rc) return rc;
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}
