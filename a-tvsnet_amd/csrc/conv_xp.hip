// 3x3x3 SAME stride-1 convolution to EIGHT output channels in x-pair form, one wavefront per SIMD (gfx950).
//
// These are the widest layers of the stacked U-Nets / refinement net (conv_b*_0_1, global_refine_3dconv0_1,
// the refinement stems: cnn_wrapper/atvsnet.py StackedUNet / CostVolRefineNet, layer code
// /root/reference/cnn_wrapper/network.py:165-215): full-resolution volumes, 8 output channels.  They are
// MFMA-bound and dominate the depth-map time, so this kernel is built around what measurements on MI355X
// showed about keeping the fp32 MFMA pipe busy:
//   * a second wavefront on the SIMD does not help (its non-MFMA instructions issue about once per MFMA of
//     the first) -- what counts is the number of non-MFMA instructions the MFMA-issuing wavefront itself
//     executes per MFMA, and that it never waits for a load younger than ~2000 cycles;
//   * so: ONE workgroup of 4 wavefronts per CU with the whole 512-entry register file and 160 KB of LDS,
//     a fully unrolled K loop whose LDS reads are base register + immediate (no address arithmetic),
//     weights requested 4 steps ahead, and the next tile's halo loads (one 16-byte slot per K step, address
//     = per-slot register + tile origin) issued between the MFMAs instead of as a burst in front of them.
//
// x-pair: 8 output channels fill half of a 16-row MFMA tile, so the rows are (x parity, channel): lane column
// i holds the voxel PAIR (2i, 2i+1) and the K axis runs over the 4 x-offsets -1..2 the pair touches, the kernel
// zero-padded accordingly (36 virtual taps instead of 27: 3/4 of the MFMA work is useful instead of 1/2).
//
// Tile: 4(z) x 8(y) x 32(x) output voxels per workgroup step, wavefront w owns plane z = w (8 accumulator
// tiles of 16 voxel pairs).  LDS image [6][10][40 voxels][chunk]: even x in columns 0..16, odd x in columns
// 18..34 so that a tap read is 16 consecutive voxels; the row pitch (40 voxels) is a multiple
// of 512 bytes, which makes every (dz, dy, row) displacement an immediate offset that commutes with the bank
// swizzle (bit 5 ^= bit 8).
#include <type_traits>

#include "conv_common.h"

// Development build (-DATVS_XP_DEBUG): per-wavefront cycle counts of the phases, read back with atvs_debug_read
// (tools_dev/phase_times.py).
#ifdef ATVS_XP_DEBUG
__device__ unsigned long long atvs_dbg[4096 * 8];
extern "C" int atvs_debug_read(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(atvs_dbg), sizeof(atvs_dbg));
}
#define DBG_T(i) { unsigned long long t_ = clock64(); dbg_acc[i] += t_ - dbg_t; dbg_t = t_; }
#else
#define DBG_T(i)
#endif

namespace {

constexpr int XP_TZ = 4, XP_TY = 8, XP_TXV = 32;
constexpr int XP_HZ = XP_TZ + 2, XP_HY = XP_TY + 2, XP_HX = XP_TXV + 2;
constexpr int XP_HXP = 40;       // voxels per image row: even x in columns 0..16, odd x in columns XP_ODD..XP_ODD+16
constexpr int XP_ODD = 18;       // 18 * 64 B = 128 (mod 256): the even and the odd voxels that one 16-lane group
                                 // of a halo write touches fall into different bank halves
constexpr int XP_LOOK = 4;                           // weight look-ahead in K steps

struct XpArgs {
  const float* x;
  const float* wp;       // packed weights, see atvs_conv_xp_pack
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
  // sibling: a second 3x3x3 convolution of the SAME input, stride 2, 16 output channels (the U-Net's encoder
  // branch conv_b*_1_0 next to conv_b*_0_1), computed from the staged image after the main K loop
  const float* wp2;      // packed sibling weights (atvs_conv_xp_pack_sibling) or nullptr
  const float* pbias2;   // (Ho2, Wo2, 48) or nullptr
  float* y2;
  double* stats2;
  int Do2, Ho2, Wo2, ldy2, ycoff2;
  int pbz, pby, pbx;     // SAME padding in front of each axis (0 or 1)
  // groups: independent samples stacked on the leading axis of every tensor; `wg` workgroups per sample
  // (gridDim.x = groups * wg) sweep that sample's tiles, statistics rows are (sample, workgroup)
  int wg;
  long gx, gy, gpb, gy2, gpb2;
  // prologue (template PRO): the convolution's input is  act_a(bn_a(x)) [+ act_b(bn_b(x2))]  formed while the halo is
  // staged -- the batch norm (+ ReLU) of the producing layer(s) and the U-Net's skip add (network.py:206-212, 695-697)
  // without a pass of their own.  in_pa / in_pb: (groups, 3, Cin) = mean, rstd, beta, or nullptr = identity;
  // out-of-volume lanes stay zero (the reference pads the finished tensor).
  const float* x2;
  const float* in_pa;
  const float* in_pb;
  int relu_a, relu_b;
  int sample_major;      // 8 samples: sample = blockIdx % 8 = XCD (all its workgroups behind one L2), tiles dealt round-robin
};

__device__ __forceinline__ int xp_swz(int a) { return a ^ (((a >> 8) & 1) << 5); }

// C4 = float4 channel groups per voxel of a chunk: 4 (16-channel chunks) or 2 (8-channel chunks).
// K step j of a chunk:  C4 == 4: one virtual tap (dz,dy,xl) = (j/12, j/4%3, j%4), lane group q = channel group;
//                       C4 == 2: two taps (dz,dy) = (j/6, j/2%3), xl = 2*(j%2) + (q>>1), channel group q&1.
//
// SIB: additionally the stride-2 sibling.  Its 2(z) x 4(y) x 16(x) outputs per tile read the same image (the
// even / odd x runs make a stride-2 tap read 16 consecutive voxels as well); wavefront w owns output plane
// w>>1, rows 2(w&1), 2(w&1)+1.  K steps:  C4 == 4: tap (kd,kh,kw) = (i/9, i/3%3, i%3), q = channel group;
//                                        C4 == 2: (kd,kh) = (i/6, i/2%3), kw = 2*(i%2) + (q>>1) (kw 3 = zero), q&1.
template <int C4, bool SIB, int PRO>
__global__ __launch_bounds__(256, 1) void conv_xp_kernel(XpArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int VB = C4 * 16;                      // bytes per voxel in LDS
  constexpr int ROWB = XP_HXP * VB;                // bytes per image row
  constexpr int SLOTS = XP_HZ * XP_HY * XP_HX * C4;
  constexpr int MAXS = (SLOTS + 255) / 256;        // 16-byte halo slots per thread
  constexpr int TPS = 4 / C4;                      // taps per K step
  constexpr int JC = 36 / TPS;                     // K steps per chunk
  constexpr int NB = 4 / TPS;                      // distinct x displacements per lane group
  constexpr bool SWZ = (C4 == 4);
  constexpr int CC = C4 * 4;                       // channels per chunk
  constexpr int J2 = SIB ? ((C4 == 4) ? 27 : 18) : 0;   // sibling K steps per chunk
  constexpr int NB2 = (C4 == 4) ? 3 : 2;
  static_assert(MAXS <= JC, "one halo slot per K step");
  static_assert(((2 * XP_HY + 2) + 3) * ROWB < 65536, "ds_read immediate offset");

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;

  // ---- LDS read bases: (x displacement set, row group of 4) -> byte address of this lane's fragment for
  // (dz, dy, row) = (-1, -1, 0); everything else is an immediate.
  int base[NB][2];
#pragma unroll
  for (int xs = 0; xs < NB; ++xs) {
    const int xl = (TPS == 1) ? xs : 2 * xs + (q >> 1);
    const int cg = (TPS == 1) ? q : (q & 1);
    const int xcol = (xl & 1) * XP_ODD + (xl >> 1);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      int a = ((wave * XP_HY + 4 * g) * XP_HXP + xcol + r) * VB + cg * 16;
      base[xs][g] = SWZ ? xp_swz(a) : a;
    }
  }

  // sibling read bases: this lane's fragment for tap (kd, kh) = (0, 0), output row 0 of the wavefront
  int base2[NB2];
  if (SIB) {
#pragma unroll
    for (int xs = 0; xs < NB2; ++xs) {
      const int kw = (C4 == 4) ? xs : min(2 * xs + (q >> 1), 2);
      const int cg = (C4 == 4) ? q : (q & 1);
      const int xh = kw + 1 - p.pbx;                                  // halo x of output column 0
      const int col = (xh & 1) * XP_ODD + (xh >> 1) + r;
      const int row0 = (2 * (wave >> 1) + 1 - p.pbz) * XP_HY + (4 * (wave & 1) + 1 - p.pby);
      int a = (row0 * XP_HXP + col) * VB + cg * 16;
      base2[xs] = SWZ ? xp_swz(a) : a;
    }
  }

  // ---- per-slot constants of this thread: global element offset from the halo origin, LDS byte address,
  // and the packed halo coordinate (zz | yy<<8 | xx<<16, each byte with its top bit set) for the bounds test
  int goff[MAXS], laddr[MAXS];
  unsigned pg[MAXS];
#pragma unroll
  for (int i = 0; i < MAXS; ++i) {
    int s = tid + i * 256;
    const bool live = s < SLOTS;
    s = min(s, SLOTS - 1);
    const int c4 = s % C4, v = s / C4;
    const int xx = v % XP_HX, v2 = v / XP_HX;
    const int yy = v2 % XP_HY, zz = v2 / XP_HY;
    goff[i] = ((zz * p.Hi + yy) * p.Wi + xx) * p.Cin + c4 * 4;
    int a = ((zz * XP_HY + yy) * XP_HXP + (xx & 1) * XP_ODD + (xx >> 1)) * VB + c4 * 16;
    laddr[i] = SWZ ? xp_swz(a) : a;
    pg[i] = 0x808080u | (unsigned)(live ? zz : 0x7f) | ((unsigned)yy << 8) | ((unsigned)xx << 16);
  }

  // persistent tile list, dealt so that the workgroups of one XCD (blockIdx % 8) sweep one contiguous eighth
  // of the tile range (halo re-use in that XCD's L2)
  const int G = p.wg;
  const int grp = p.sample_major ? (int)(blockIdx.x & 7) : (int)(blockIdx.x / p.wg);
  const int lbk = p.sample_major ? (int)(blockIdx.x >> 3) : (int)(blockIdx.x - grp * p.wg);
  const int xcd = p.sample_major ? 0 : (lbk & 7), tslot = p.sample_major ? lbk : (lbk >> 3);
  const unsigned srow = (unsigned)(grp * p.wg + lbk);        // this workgroup's row of the statistics buffers
  const float* __restrict__ xg = p.x + (size_t)grp * p.gx;
  const float* __restrict__ xg2 = (PRO == 2) ? p.x2 + (size_t)grp * p.gx : nullptr;
  const float* __restrict__ ipa = (PRO >= 1 && p.in_pa) ? p.in_pa + (size_t)grp * 3 * p.Cin : nullptr;
  const float* __restrict__ ipb = (PRO == 2 && p.in_pb) ? p.in_pb + (size_t)grp * 3 * p.Cin : nullptr;
  const int c4t = tid % C4;                       // 256 % C4 == 0: every slot of this thread is channel group c4t
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
    *x0 = bx * XP_TXV;
    *y0 = (rest % p.tiles_y) * XP_TY;
    *z0 = (rest / p.tiles_y) * XP_TZ;
  };

  // uniform description of the (tile, chunk) whose halo is being fetched
  struct PfTile {
    const float* xb;      // p.x + first channel of the chunk
    const float* xb2;     // second source (PRO == 2)
    int org;              // element offset of the halo origin (may be negative: first layer of the halo is outside)
    unsigned lo, hi1;     // packed bounds: valid iff lo_f <= f <= hi_f in every field (hi1 = hi + 1 per byte)
  };
  auto pf_tile = [&](int stage) __attribute__((always_inline)) {
    PfTile T;
    int k = stage / p.nchunk, ch = stage - k * p.nchunk;
    int z0, y0, x0;
    tile_origin(k, &z0, &y0, &x0);
    const int gz0 = z0 - 1, gy0 = y0 - 1, gx0 = x0 - 1;
    T.xb = xg + ch * CC;
    T.xb2 = (PRO == 2) ? xg2 + ch * CC : nullptr;
    T.org = ((gz0 * p.Hi + gy0) * p.Wi + gx0) * p.Cin;
    T.lo = (unsigned)(gz0 < 0) | ((unsigned)(gy0 < 0) << 8) | ((unsigned)(gx0 < 0) << 16);
    T.hi1 = (unsigned)(min(p.Di - 1 - gz0, 0x7e) + 1) | ((unsigned)(min(p.Hi - 1 - gy0, 0x7e) + 1) << 8) |
            ((unsigned)(min(p.Wi - 1 - gx0, 0x7e) + 1) << 16);
    return T;
  };
  float4 pf[MAXS];
  float4 pf2[PRO == 2 ? MAXS : 1];
  unsigned vmask = 0;                          // PRO: which slots of the halo in flight lie inside the volume
  // slot i of the halo: lanes outside the volume (or past the last slot) read the 16 zero bytes instead
  auto pf_slot = [&](const PfTile& T, int i) __attribute__((always_inline)) {
    const unsigned t1 = pg[i] - T.lo;          // byte f keeps its top bit iff f >= lo_f
    const unsigned t2 = T.hi1 + ~pg[i];        // byte f has its top bit iff f <= hi_f
    const bool ok = ((t1 & t2) & 0x808080u) == 0x808080u;
    const float* src = ok ? (T.xb + (T.org + goff[i])) : p.zeros;
    pf[i] = ld4(src);
    if (PRO >= 1) vmask = (vmask & ~(1u << i)) | ((ok ? 1u : 0u) << i);
    if (PRO == 2) pf2[i] = ld4(ok ? (T.xb2 + (T.org + goff[i])) : p.zeros);
  };

  // PRO: the producers' batch norm (+ ReLU) and the skip add, applied to a halo slot between its arrival and its LDS
  // write.  Every slot of this thread carries channel group c4t of the chunk, so one set of parameters per stage serves
  // them; out-of-volume slots stay zero.  Arithmetic = bn_apply / bn_add (norm.hip): (x - mean) * rstd + beta, then max.
  struct Par { float4 ma, sa, ba, mb, sb, bb; };
  // branch-free throughout (a branch inside the unrolled K loop would split its scheduling region): absent parameters
  // read the 16 zero bytes and are deselected below; no ReLU = a floor of -inf
  const bool has_a = PRO >= 1 && ipa != nullptr, has_b = PRO == 2 && ipb != nullptr;
  const float floor_a = p.relu_a ? 0.f : -__builtin_huge_valf(), floor_b = p.relu_b ? 0.f : -__builtin_huge_valf();
  auto load_par = [&](int chunk) __attribute__((always_inline)) {
    Par P;
    const int cch = chunk * CC + c4t * 4;
    const int sa = has_a ? p.Cin : 0, sb = has_b ? p.Cin : 0;
    const float* a = has_a ? ipa + cch : p.zeros;
    P.ma = ld4(a); P.sa = ld4(a + sa); P.ba = ld4(a + 2 * sa);
    if (PRO == 2) {
      const float* b = has_b ? ipb + cch : p.zeros;
      P.mb = ld4(b); P.sb = ld4(b + sb); P.bb = ld4(b + 2 * sb);
    }
    return P;
  };
  // two channels per instruction (v_pk_add_f32 / v_pk_mul_f32: the same IEEE operations as the scalar forms)
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  auto bn2 = [&](f32x2 v, f32x2 m, f32x2 sc, f32x2 be, float lo, bool has) __attribute__((always_inline)) {
    f32x2 t = (v - m) * sc + be;
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
  // K steps between a slot's request and its transform (the steps of the sibling loop count on after the main loop's)
  constexpr int XLAG = (C4 == 4) ? 4 : 5;
  static_assert(PRO == 0 || (MAXS - 1 + XLAG <= JC + J2 - 1), "every slot is transformed inside the K loops");
  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);        // bias of this lane's 4 output channels
  if (p.bias) bv = ld4(p.bias + (q & 1) * 4);
  f32x4 acc[XP_TY];
  f32x4 acc2[2];
  float ssum2[4] = {0.f, 0.f, 0.f, 0.f}, ssq2[4] = {0.f, 0.f, 0.f, 0.f};
  const float4* __restrict__ wp = reinterpret_cast<const float4*>(p.wp);
  const float4* __restrict__ wp2 = reinterpret_cast<const float4*>(p.wp2);

  if (nstage > 0) {
    const PfTile T = pf_tile(0);
#pragma unroll
    for (int i = 0; i < MAXS; ++i) pf_slot(T, i);
  }

#ifdef ATVS_XP_DEBUG
  unsigned long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long dbg_t = clock64();
#endif
  for (int stage = 0; stage < nstage; ++stage) {
    const int k = stage / p.nchunk, ch = stage - k * p.nchunk;
    DBG_T(6)
    if (ch == 0) {
#pragma unroll
      for (int t = 0; t < XP_TY; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc2[0] = acc2[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // weights of the first steps: on their way while the image is written
    const float4* wch = wp + (size_t)ch * JC * 64 + lane;
    const float4* wch2 = wp2 + (size_t)ch * J2 * 64 + lane;
    float4 w[JC];
#pragma unroll
    for (int jj = 0; jj < XP_LOOK; ++jj) w[jj] = wch[jj * 64];

    __syncthreads();                       // every wave is done reading the previous stage's image
    DBG_T(0)
    // PRO: the halo of stage 0 is transformed here; every later one during the K loop of the stage before it
    if (PRO >= 1 && stage == 0) {
      const Par P0 = load_par(0);
#pragma unroll
      for (int i = 0; i < MAXS; ++i) xform(i, P0);
    }
#pragma unroll
    for (int i = 0; i < MAXS; ++i)
      if (i < MAXS - 1 || tid + i * 256 < SLOTS) *reinterpret_cast<float4*>(smem + laddr[i]) = pf[i];
    DBG_T(1)
    __syncthreads();
    DBG_T(2)

    // ---- K loop, fully unrolled
    const PfTile T = pf_tile(min(stage + 1, nstage - 1));      // last stage: harmless re-read of its own halo
    // output addressing of this tile: this lane holds channels (q&1)*4..+3 of voxel x0 + 2r + (q>>1) for the
    // 8 rows y0..y0+7 of plane z0 + wave (accumulator rows = (x parity, channel))
    const bool last_chunk = (ch == p.nchunk - 1);
    int tz0, ty0, tx0;
    tile_origin(k, &tz0, &ty0, &tx0);
    const int zo = tz0 + wave, xo = tx0 + 2 * r + (q >> 1), co = (q & 1) * 4;
    const bool evox_ok = zo < p.Di && xo < p.Wi;
    const size_t erow = (size_t)p.Wi * p.ldy;                                          // floats per output row
    const size_t eo = (((size_t)zo * p.Hi + ty0) * p.Wi + xo) * (size_t)p.ldy + p.ycoff + co;
    const size_t epb_off = ((size_t)ty0 * p.Wi + xo) * 24 + plane_variant(zo - 1, p.Di) * 8 + co;
    auto erow_ok = [&](int t) __attribute__((always_inline)) { return evox_ok && ty0 + t < p.Hi; };
    float4 epb[XP_TY], epb2[2];
#pragma unroll
    for (int t = 0; t < XP_TY; ++t) epb[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    epb2[0] = epb2[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    // sibling outputs of this lane: channels 4q..4q+3 of voxel (tz0/2 + (wave>>1), ty0/2 + 2(wave&1) + t, tx0/2 + r)
    const int zo2 = (tz0 >> 1) + (wave >> 1), yo2 = (ty0 >> 1) + 2 * (wave & 1), xo2 = (tx0 >> 1) + r;
    const bool evox2_ok = SIB && zo2 < p.Do2 && xo2 < p.Wo2;
    auto erow2_ok = [&](int t) __attribute__((always_inline)) { return evox2_ok && yo2 + t < p.Ho2; };
    const size_t erow2 = (size_t)p.Wo2 * p.ldy2;
    const size_t eo2 = (((size_t)zo2 * p.Ho2 + yo2) * p.Wo2 + xo2) * (size_t)p.ldy2 + p.ycoff2 + 4 * q;
    float4 b[2][XP_TY], b2[2][2];
    auto request_b = [&](int j) __attribute__((always_inline)) {
      const int dzdy = j / NB, xs = j % NB;
      const int rowoff = (dzdy / 3) * XP_HY + (dzdy % 3);
#pragma unroll
      for (int t = 0; t < XP_TY; ++t)
        b[j & 1][t] = *reinterpret_cast<const float4*>(smem + base[xs][t >> 2] + (rowoff + (t & 3)) * ROWB);
    };
    auto request_b2 = [&](int i) __attribute__((always_inline)) {
      const int kdkh = i / NB2, xs = i % NB2;
      const int rowoff = (kdkh / 3) * XP_HY + (kdkh % 3);
#pragma unroll
      for (int t = 0; t < 2; ++t)
        b2[i & 1][t] = *reinterpret_cast<const float4*>(smem + base2[xs] + (rowoff + 2 * t) * ROWB);
    };
    float4 w2[SIB ? J2 : 1];
    Par Pn;
    if (PRO >= 1) Pn = load_par(min(stage + 1, nstage - 1) % p.nchunk);
    request_b(0);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < JC; ++j) {
      if (j + XP_LOOK < JC) w[j + XP_LOOK] = wch[(j + XP_LOOK) * 64];
      else if (SIB) w2[j + XP_LOOK - JC] = wch2[(j + XP_LOOK - JC) * 64];
      if (j + 1 < JC) request_b(j + 1);
      else if (SIB) request_b2(0);
      if (j < MAXS) pf_slot(T, j);
      // operands of the epilogues (depth-plane biases) of the tile's last chunk: requested a few steps before
      // the end of the main K loop, so the epilogue never waits for them
      if (j == JC - 4 && last_chunk && pbg) {
#pragma unroll
        for (int t = 0; t < XP_TY; ++t) epb[t] = ld4(erow_ok(t) ? pbg + (epb_off + (size_t)t * p.Wi * 24) : p.zeros);
      }
      if (SIB && j == JC - 3 && last_chunk && pb2g) {
        const size_t o = ((size_t)yo2 * p.Wo2 + xo2) * 48 + plane_variant(2 * zo2 - p.pbz, p.Di) * 16 + 4 * q;
#pragma unroll
        for (int t = 0; t < 2; ++t) epb2[t] = ld4(erow2_ok(t) ? pb2g + (o + (size_t)t * p.Wo2 * 48) : p.zeros);
      }
      // compiler barrier (keeps InstCombine / the scheduler from sinking the requests to their uses) +
      // scheduling barrier (keeps them in front of the MFMAs that cover their latency)
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int t = 0; t < XP_TY; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(w[j], s), f4get(b[j & 1][t], s), acc[t], 0, 0, 0);
      // PRO: the slot requested XLAG steps ago has arrived; its transform shares this step's scheduling region with the
      // MFMAs above (VALU in their shadow)
      if (PRO >= 1 && j >= XLAG && j - XLAG < MAXS) xform(j - XLAG, Pn);
    }
    if (SIB) {
      // ---- the sibling's K steps on the same image (weights and the first fragments are already on their way)
#pragma unroll
      for (int i = 0; i < J2; ++i) {
        if (i + XP_LOOK < J2) w2[i + XP_LOOK] = wch2[(i + XP_LOOK) * 64];
        if (i + 1 < J2) request_b2(i + 1);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            acc2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(w2[i], s), f4get(b2[i & 1][t], s), acc2[t], 0, 0, 0);
        if (PRO >= 1 && JC + i - XLAG < MAXS) xform(JC + i - XLAG, Pn);
      }
    }
    DBG_T(4)
    if (ch != p.nchunk - 1) continue;

    // ---- epilogue (no loads, no branches per row besides the bounds test)
    auto store_rows = [&](auto relu_tag) __attribute__((always_inline)) {
#pragma unroll
      for (int t = 0; t < XP_TY; ++t) {
        if (!erow_ok(t)) continue;
        float4 v;
        v.x = (acc[t][0] + bv.x) + epb[t].x;
        v.y = (acc[t][1] + bv.y) + epb[t].y;
        v.z = (acc[t][2] + bv.z) + epb[t].z;
        v.w = (acc[t][3] + bv.w) + epb[t].w;
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
        float4 v = make_float4(acc2[t][0] + epb2[t].x, acc2[t][1] + epb2[t].y, acc2[t][2] + epb2[t].z, acc2[t][3] + epb2[t].w);
        st4(y2g + (eo2 + (size_t)t * erow2), v);
        ssum2[0] += v.x; ssum2[1] += v.y; ssum2[2] += v.z; ssum2[3] += v.w;
        ssq2[0] += v.x * v.x; ssq2[1] += v.y * v.y; ssq2[2] += v.z * v.z; ssq2[3] += v.w * v.w;
      }
    }
    DBG_T(7)
  }

#ifdef ATVS_XP_DEBUG
  DBG_T(5)
  if (lane == 0 && blockIdx.x < 1024)
    for (int i = 0; i < 8; ++i) atvs_dbg[(blockIdx.x * 4 + wave) * 8 + i] = dbg_acc[i];
#endif
  // ---- per-workgroup partial moments (sum, sum of squares) per output channel -> row blockIdx of stats:
  // [2][16] doubles, columns 0..7 = channels, 8..15 = 0 (the layout of the tiled kernel's x-pair form)
  if (p.stats) {
    __syncthreads();
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

int xp_c4(int Cin) { return (Cin % 16 == 0) ? 4 : ((Cin % 8 == 0) ? 2 : 0); }

long xp_ntiles(int D, int H, int W) {
  return (long)((D + XP_TZ - 1) / XP_TZ) * ((H + XP_TY - 1) / XP_TY) * ((W + XP_TXV - 1) / XP_TXV);
}

}  // namespace

// Floats of the packed form of a [3,3,3,Cin,8] kernel (Cin % 8 == 0), including 4 trailing zeros.
extern "C" int atvs_conv_xp_pack_size(int Cin, long* packed_floats) {
  const int C4 = xp_c4(Cin);
  if (Cin <= 0 || !C4 || !packed_floats) return ATVS_ERR_SHAPE;
  const int nch = Cin / (4 * C4), JC = 36 * C4 / 4;
  *packed_floats = (long)nch * JC * 64 * 4 + 4;
  return ATVS_OK;
}

// HOST function.  w: TF kernel [3,3,3,Cin,8].  packed[chunk][K step][lane = q*16 + (jx*8 + co)][s]:
// the x-pair virtual kernel Wv[(kd,kh,xl)][ci][jx][co] = W[kd][kh][kw = xl - jx][ci][co] (0 outside 0..2),
// ordered as the kernel's K steps consume it.
extern "C" int atvs_conv_xp_pack(const float* w, int Cin, float* packed) {
  if (!w || !packed) return ATVS_ERR_NULL;
  long pf;
  int rc = atvs_conv_xp_pack_size(Cin, &pf);
  if (rc) return rc;
  const int C4 = xp_c4(Cin), TPS = 4 / C4, NB = 4 / TPS, JC = 36 / TPS, CC = 4 * C4, nch = Cin / CC;
  for (long i = 0; i < pf; ++i) packed[i] = 0.f;
  for (int ch = 0; ch < nch; ++ch)
    for (int j = 0; j < JC; ++j)
      for (int q = 0; q < 4; ++q) {
        const int dzdy = j / NB, xs = j % NB;
        const int kd = dzdy / 3, kh = dzdy % 3;
        const int xl = (TPS == 1) ? xs : 2 * xs + (q >> 1);
        const int cg = (TPS == 1) ? q : (q & 1);
        for (int jx = 0; jx < 2; ++jx) {
          const int kw = xl - jx;
          if (kw < 0 || kw > 2) continue;
          for (int co = 0; co < 8; ++co)
            for (int s = 0; s < 4; ++s) {
              const int ci = ch * CC + cg * 4 + s;
              packed[((((size_t)ch * JC + j) * 64) + q * 16 + jx * 8 + co) * 4 + s] =
                  w[((((size_t)kd * 3 + kh) * 3 + kw) * Cin + ci) * 8 + co];
            }
        }
      }
  return ATVS_OK;
}

// workgroups PER SAMPLE of a launch over `groups` independent samples (rows of the statistics buffer = groups * this):
// one workgroup per CU in all, shared out among the samples, a multiple of 8 each
extern "C" long atvs_conv_xp_grid(int D, int H, int W, int groups) {
  if (groups < 1) groups = 1;
  long nt = xp_ntiles(D, H, W);
  long share = 256 / groups / 8 * 8;
  if (share < 8) share = 8;
  long g = nt < share ? nt : share;
  return (g + 7) / 8 * 8;
}

// Floats of the packed sibling kernel [3,3,3,Cin,16].
extern "C" int atvs_conv_xp_pack_sibling_size(int Cin, long* packed_floats) {
  const int C4 = xp_c4(Cin);
  if (Cin <= 0 || !C4 || !packed_floats) return ATVS_ERR_SHAPE;
  const int nch = Cin / (4 * C4), J2 = (C4 == 4) ? 27 : 18;
  *packed_floats = (long)nch * J2 * 64 * 4;
  return ATVS_OK;
}

// HOST function.  w2: TF kernel [3,3,3,Cin,16] of the stride-2 sibling.  packed[chunk][K step][lane = q*16 + co][s].
extern "C" int atvs_conv_xp_pack_sibling(const float* w2, int Cin, float* packed) {
  if (!w2 || !packed) return ATVS_ERR_NULL;
  long pf;
  int rc = atvs_conv_xp_pack_sibling_size(Cin, &pf);
  if (rc) return rc;
  const int C4 = xp_c4(Cin), NB2 = (C4 == 4) ? 3 : 2, J2 = (C4 == 4) ? 27 : 18, CC = 4 * C4, nch = Cin / CC;
  for (long i = 0; i < pf; ++i) packed[i] = 0.f;
  for (int ch = 0; ch < nch; ++ch)
    for (int j = 0; j < J2; ++j)
      for (int q = 0; q < 4; ++q) {
        const int kdkh = j / NB2, xs = j % NB2;
        const int kd = kdkh / 3, kh = kdkh % 3;
        const int kw = (C4 == 4) ? xs : 2 * xs + (q >> 1);
        const int cg = (C4 == 4) ? q : (q & 1);
        if (kw > 2) continue;
        for (int co = 0; co < 16; ++co)
          for (int s = 0; s < 4; ++s) {
            const int ci = ch * CC + cg * 4 + s;
            packed[((((size_t)ch * J2 + j) * 64) + q * 16 + co) * 4 + s] =
                w2[((((size_t)kd * 3 + kh) * 3 + kw) * Cin + ci) * 16 + co];
          }
      }
  return ATVS_OK;
}

template <int C4, bool SIB, int PRO>
static int launch_xp1(const XpArgs& a, long blocks, hipStream_t s) {
  size_t lds = (size_t)XP_HZ * XP_HY * XP_HXP * C4 * 16;
  // the attribute is per device: one flag per device ordinal of this process
  static bool attr_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return ATVS_ERR_LAUNCH;
  if (!attr_set[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_xp_kernel<C4, SIB, PRO>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return ATVS_ERR_LAUNCH;
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL((conv_xp_kernel<C4, SIB, PRO>), dim3((unsigned)blocks), dim3(256), lds, s, a);
  return ATVS_OK;
}

// y[z,y,x, y_coff + co] = sum_taps W * x (+ bias, + plane_bias, ReLU), co < 8; x (D,H,W,Cin) with Cin % 8 == 0;
// y (D,H,W,ldy).  stats_partial: atvs_conv_xp_grid rows of [2][16] doubles (or NULL).
// Sibling (packed_w2 != NULL): y2 (ceil(D/2), ceil(H/2), ceil(W/2), ldy2)[..., y_coff2 + co] = the 3x3x3 SAME
// stride-2 convolution of the same x with a [3,3,3,Cin,16] kernel (+ plane_bias2 (Ho2, Wo2, 48)), its partial
// moments in stats_partial2 (same rows, 16 channels).
extern "C" int atvs_conv_xp_f32(const float* x, const float* packed_w, const float* bias, const float* plane_bias,
                                float* y, double* stats_partial, int groups, int D, int H, int W, int Cin, int ldy, int y_coff,
                                int relu, const float* packed_w2, const float* plane_bias2, float* y2,
                                double* stats_partial2, int ldy2, int y_coff2, const float* x2, const float* in_params,
                                const float* in_params2, int in_relu, int in_relu2, atvs_stream_t stream) {
  if (!x || !packed_w || !y) return ATVS_ERR_NULL;
  if (in_params2 && !x2) return ATVS_ERR_ARG;
  const int C4 = xp_c4(Cin);
  if (groups <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || !C4) return ATVS_ERR_SHAPE;
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
  atvs_conv_xp_pack_size(Cin, &pf);
  XpArgs a;
  a.x = x; a.wp = packed_w; a.zeros = packed_w + (pf - 4); a.bias = bias; a.pbias = plane_bias;
  a.y = y; a.stats = stats_partial;
  a.Di = D; a.Hi = H; a.Wi = W; a.Cin = Cin; a.ldy = ldy; a.ycoff = y_coff;
  a.nchunk = Cin / (4 * C4);
  a.tiles_y = (H + XP_TY - 1) / XP_TY; a.tiles_x = (W + XP_TXV - 1) / XP_TXV;
  a.ntiles = (int)xp_ntiles(D, H, W);
  a.relu = relu;
  a.wp2 = packed_w2; a.pbias2 = plane_bias2; a.y2 = y2; a.stats2 = stats_partial2;
  a.Do2 = (D + 1) / 2; a.Ho2 = (H + 1) / 2; a.Wo2 = (W + 1) / 2; a.ldy2 = ldy2; a.ycoff2 = y_coff2;
  a.pbz = D & 1; a.pby = H & 1; a.pbx = W & 1;      // tf SAME, kernel 3, stride 2: one leading pad iff the size is odd
  a.wg = (int)atvs_conv_xp_grid(D, H, W, groups);
  a.gx = (long)D * H * W * Cin; a.gy = (long)D * H * W * ldy; a.gpb = (long)H * W * 24;
  a.gy2 = (long)a.Do2 * a.Ho2 * a.Wo2 * ldy2; a.gpb2 = (long)a.Ho2 * a.Wo2 * 48;
  const long blocks = (long)a.wg * groups;
  hipStream_t st = as_stream(stream);
  // eight samples (the eight U-Nets of a 5-view depth map): one sample per XCD -- its workgroups sweep neighbouring tiles
  // behind one L2 (measured: -0.8 % on the dominant launch against spreading every sample over all XCDs)
  a.sample_major = (groups == 8) ? 1 : 0;
  a.x2 = x2; a.in_pa = in_params; a.in_pb = in_params2; a.relu_a = in_relu; a.relu_b = in_relu2;
  const int pro = x2 ? 2 : (in_params ? 1 : 0);
  int rc;
  // instantiated prologue forms: none (every shape); single-source normalise (the refinement's concat -> 3dconv0_1 |
  // 3dconv1_0, 32 channels + sibling); two-source normalise + add (the U-Net's stack inputs, 8 channels + sibling)
  if (pro == 0) {
    if (packed_w2) rc = (C4 == 4) ? launch_xp1<4, true, 0>(a, blocks, st) : launch_xp1<2, true, 0>(a, blocks, st);
    else rc = (C4 == 4) ? launch_xp1<4, false, 0>(a, blocks, st) : launch_xp1<2, false, 0>(a, blocks, st);
  } else if (pro == 1 && packed_w2 && C4 == 4) {
    rc = launch_xp1<4, true, 1>(a, blocks, st);
  } else if (pro == 2 && packed_w2 && C4 == 2) {
    rc = launch_xp1<2, true, 2>(a, blocks, st);
  } else {
    return ATVS_ERR_ARG;
  }
  if (rc) return rc;
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}
