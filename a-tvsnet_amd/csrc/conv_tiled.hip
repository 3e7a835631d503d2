// LDS-tiled fp32-MFMA convolution for halo-1 stencils (gfx950): the 3x3x3 stride-1 SAME
// convolutions of the stacked U-Nets / refinement net / AANet and the stride-2 transposed
// convolution (all 8 output parity classes from one staged tile).
//
// Replaces tf.layers.conv3d / tf.nn.conv3d / tf.layers.conv3d_transpose (+ bias-free batch-norm
// statistics) at /root/reference/cnn_wrapper/network.py:165-167,198-200,304,331,534-536.
//
// Structure (one workgroup = 4 wavefronts, persistent over output tiles):
//   tile     4(z) x TY(y) x 16(x) output voxels; wavefront w owns plane z = w = TY MFMA tiles of 16
//            x-consecutive voxels (GEMM: M = output channels, N = 16 voxels, K = (tap, channel)).
//   stage    = (tile, <=16-channel chunk).  The (4+2) x (TY+2) x 18 input halo of the stage is
//            PREFETCHED into registers (global_load_dwordx4, issued before the previous stage's MFMAs
//            so HBM/L2 latency hides under them), then written to LDS as [z][y][x][chunk].
//   LDS      a wavefront's tap read covers 16 voxels x 64 B; bit 5 of the byte address is XOR-ed with
//            bit 8 so that each 16-lane group of ds_read_b128 hits 16 distinct 16-byte slots
//            (conflict-free for every tap alignment; the plain image is 2-way conflicted).
//   K loop   27 taps x chunk/4 channel groups; a lane's float4 (4 consecutive input channels of its
//            tap) feeds 4 MFMAs; packed weights stream from L2; operands of step j+1 are loaded
//            before the MFMAs of step j.
//   epilogue bias / depth-plane bias / residual / ReLU, 16-byte channel-last stores, per-channel
//            (sum, sum of squares) accumulated over the workgroup's tiles -> one partial row per
//            workgroup for the deterministic batch-norm reduction.
//   x-pair   (XP, Cout == 8 only) 8 output channels would fill half of the 16-row MFMA tile.  The rows
//            become (x parity, channel): lane column i holds the voxel PAIR (2i, 2i+1), the K axis runs
//            over the 4 x-offsets -1..2 that the pair touches with the kernel zero-padded accordingly
//            (36 virtual taps instead of 27): 3/4 of the MFMA work is useful instead of 1/2, a third fewer
//            MFMAs per voxel.  The LDS image keeps even and odd x in separate runs so a tap read is
//            still 16 voxels x 64 B contiguous.
#include <type_traits>

#include "conv_common.h"

// Development build (-DATVS_TILED_DEBUG): per-wavefront cycle counts of the phases, read back with
// atvs_debug_read_tiled (tools_dev/phase_times.py --tiled).
#ifdef ATVS_TILED_DEBUG
__device__ unsigned long long atvs_dbg_tiled[4096 * 8];
extern "C" int atvs_debug_read_tiled(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(atvs_dbg_tiled), sizeof(atvs_dbg_tiled));
}
#define TDBG(i) { unsigned long long t_ = clock64(); dbg_acc[i] += t_ - dbg_t; dbg_t = t_; }
#else
#define TDBG(i)
#endif

#define TILE_TZ 4
#define TILE_TX 16
#define TILED_MAX_WG 512     // persistent grid: 2 workgroups per CU

struct TiledArgs {
  const float* x;
  const float* wp;
  const int* tab;      // per K group: LDS byte offset of (tap, channel group) relative to the voxel's base
  const float* bias;
  const float* res;
  float* y;
  double* stats;
  int Di, Hi, Wi, Cin;
  int Hy, Wy;
  int oS, offz, offy, offx;
  int ldy, ycoff, Cout;
  int nchunk, Jc;            // chunks; K steps per chunk
  int tiles_z, tiles_y, tiles_x, ntiles;
  int relu, vec_out;
  const float* pbias;        // (H, W, 3*Cout) or nullptr
  // fused stride-2 transposed convolution: the GEMM's M axis is (parity class, output channel);
  // cls_cout = real Cout (0 = ordinary convolution), cls_base = first class of this launch
  int cls_cout, cls_base;
  // N-split: the 16-channel MFMA tiles of the output are dealt to `nsplit` workgroups per spatial tile
  // (more workgroups for small volumes); nt_total = tiles in the packed weights, this kernel's NT = nt_total / nsplit
  int nsplit, nt_total;
  // batch-norm moments finished in-launch by the last workgroup to arrive (no separate finalize launch):
  // fin_counter = one zero-initialised device word (left at zero again), fin_params = (3, fin_c) floats
  // (mean, rstd, 0), fin_rows = rows of `stats` to reduce (all launches of the layer), fin_arrivals =
  // workgroups of all those launches, fin_fold = columns per channel, fin_count = elements per channel.
  unsigned* fin_counter;
  float* fin_params;
  int fin_rows, fin_arrivals, fin_c, fin_fold;
  double fin_count;
  float fin_eps;
  double* fin_stats;     // first row of the layer's statistics buffer
  // groups: independent samples stacked on the leading axis of x / y / residual / plane bias; `wg` workgroups
  // per sample sweep that sample's tiles (gridDim.x = groups * wg), statistics rows are (sample, workgroup)
  int wg, ngroups;
  long gx, gy, gpb;
};

static inline long a_groups(const TiledArgs& a) { return a.ngroups; }

__device__ __forceinline__ int lds_swz(int a) { return a ^ (((a >> 8) & 1) << 5); }

// Wavefronts per SIMD the register budget of an instantiation allows (accumulators + prefetch
// registers + operands): 2 workgroups per CU when it fits in 256 VGPRs, else 1 (512).  The bound is set where the compiler's
// allocation stops spilling: the forms estimated at 168-188 (<1,4,4,XP>, <1,8,2,XP>, <1,8,4>: fallback paths behind
// conv_xw / conv_c16) needed 4-40 spilled registers at two workgroups per CU and get the whole register file instead.
__host__ __device__ constexpr int tiled_wps(int NT, int TY, int C4, bool XP = false) {
  return (NT * TY * 4 + ((TILE_TZ + 2) * (TY + 2) * ((XP ? 2 : 1) * TILE_TX + 2) * C4 + 255) / 256 * 4 + TY * 4 + NT * 16 + 40 <= 165) ? 2 : 1;
}
// statistics in the epilogue cost 8*NT registers; the widest variant leaves them to a separate pass
__host__ __device__ constexpr bool tiled_has_stats(int NT) { return NT < 8; }

// C4 = channel groups (float4) per voxel of a chunk in LDS: 4 (16 channels), 2 (8), 1 (<= 4).
// FULL = every chunk has exactly 4*C4 real channels and Cin % 4 == 0 (vector loads, no tail).
template <int NT, int TY, int C4, bool FULL, bool XP>
__global__ __launch_bounds__(256, tiled_wps(NT, TY, C4, XP)) void conv_tiled_f32_kernel(TiledArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TXV = XP ? 2 * TILE_TX : TILE_TX;          // voxels per tile row
  constexpr int HZ = TILE_TZ + 2, HY = TY + 2, HX = TXV + 2;
  constexpr int SLOTS = HZ * HY * HX * C4;
  constexpr int MAXS = (SLOTS + 255) / 256;
  constexpr int VB = C4 * 16;                     // bytes per voxel in LDS
  constexpr bool SWZ = (C4 == 4);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  int* s_tab = reinterpret_cast<int*>(smem + SLOTS * 16);
  for (int i = tid; i < p.Jc * 5; i += 256) s_tab[i] = p.tab[i];   // Jc*4 offsets, then Jc tile masks

  // persistent tile list of this workgroup; tiles are dealt so that workgroups sharing an XCD
  // (blockIdx % 8) sweep one contiguous eighth of the tile range (halo re-use in that XCD's L2).
  // With N-split, workgroup (xcd, local) owns output-channel tiles nsi = local % nsplit of every
  // spatial tile it visits (gridDim.x is a multiple of 8 * nsplit).
  const int G = p.wg;
  const int grp = blockIdx.x / p.wg, lbk = blockIdx.x - grp * p.wg;
  const int xcd = lbk & 7, local = lbk >> 3;
  const int per_xcd = (p.ntiles + 7) >> 3;                 // spatial tiles per XCD range
  const int nsi = local % p.nsplit, tslot = local / p.nsplit;
  const int slots_per_xcd = (G >> 3) / p.nsplit;           // workgroups per (XCD, nsi)
  int my_tiles = 0;
  {
    int last = min(per_xcd, p.ntiles - xcd * per_xcd);     // tiles in this XCD's range
    if (tslot < last) my_tiles = (last - tslot + slots_per_xcd - 1) / slots_per_xcd;
  }
  const int nstage = my_tiles * p.nchunk;

  constexpr bool STATS = tiled_has_stats(NT);
  constexpr int SN = STATS ? NT : 1;
  f32x4 acc[TY][NT];
  float ssum[SN][4], ssq[SN][4];
#pragma unroll
  for (int n = 0; n < SN; ++n)
#pragma unroll
    for (int k = 0; k < 4; ++k) ssum[n][k] = ssq[n][k] = 0.f;

  // channel part of this lane's output offsets, per 16-channel tile: plain = first channel + y_coff; transposed
  // convolution = parity-class displacement + real channel + y_coff; ~0 = no such channel
  unsigned ycoff[NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int co = (nsi * NT + n) * 16 + 4 * q;
    if (co >= p.Cout) {
      ycoff[n] = ~0u;
    } else if (p.cls_cout) {
      const int cls = p.cls_base + co / p.cls_cout, cr = co % p.cls_cout;
      ycoff[n] = (((unsigned)(cls >> 2) * p.Hy + ((cls >> 1) & 1)) * p.Wy + (cls & 1)) * p.ldy + p.ycoff + cr;
    } else {
      ycoff[n] = (unsigned)(p.ycoff + co);
    }
  }

  int vbase[TY];
#pragma unroll
  for (int t = 0; t < TY; ++t) vbase[t] = ((wave * HY + t) * HX + r) * VB;   // XP: r = pair index within the even/odd run

  const float4* __restrict__ wp = reinterpret_cast<const float4*>(p.wp);
  const unsigned char* lds = smem;
  const int Cc = C4 * 4;

  auto tile_origin = [&](int k, int* z0, int* y0, int* x0) {
    int tl = xcd * per_xcd + tslot + k * slots_per_xcd;
    int bx = tl % p.tiles_x;
    int rest = tl / p.tiles_x;
    *x0 = bx * TXV;
    *y0 = (rest % p.tiles_y) * TY;
    *z0 = (rest / p.tiles_y) * TILE_TZ;
  };

  // Halo slot s = tid + 256 i of this thread = (zz, yy, xx, channel group c4) of the halo; the coordinates are packed once
  // (15 bits: zz << 12 | yy << 8 | xx << 2 | c4, two slots per register) so that a stage's address generation is a few
  // shifts and 24-bit multiply-adds per slot instead of six divisions by constants and three full multiplies.
  unsigned slot_pk[(MAXS + 1) / 2];
#pragma unroll
  for (int i = 0; i < MAXS; ++i) {
    const int s = min(tid + i * 256, SLOTS - 1);            // past the end: a duplicate (never written to LDS)
    const int c4 = s % C4, v = s / C4;
    const int xx = v % HX, v2 = v / HX;
    const int yy = v2 % HY, zz = v2 / HY;
    const unsigned code = (unsigned)(zz << 12 | yy << 8 | xx << 2 | c4);
    if (i & 1) slot_pk[i >> 1] |= code << 16;
    else slot_pk[i >> 1] = code;
  }
  const unsigned SY = (unsigned)p.Wi * p.Cin, SZ = (unsigned)p.Hi * SY;     // elements per row / plane

  float4 pf[MAXS];
  auto prefetch = [&](int stage) {
    int k = stage / p.nchunk, ch = stage - k * p.nchunk;
    int z0, y0, x0;
    tile_origin(k, &z0, &y0, &x0);
    const int cbase = ch * Cc;
    // one uniform base pointer + 32-bit per-lane element offsets (the launcher refuses inputs of 2^31
    // elements or more).  Tiles whose halo lies inside the volume skip every bounds test.
    const bool interior = FULL && z0 >= 1 && y0 >= 1 && x0 >= 1 && z0 + TILE_TZ < p.Di && y0 + TY < p.Hi && x0 + TXV < p.Wi;
    const float* xb = p.x + (size_t)grp * p.gx + cbase;
    const int gz0 = z0 - 1, gy0 = y0 - 1, gx0 = x0 - 1;
    // wrap-around arithmetic: org may be "negative" for border tiles, the sum is right whenever the slot is in bounds
    const unsigned org = (unsigned)gz0 * SZ + (unsigned)gy0 * SY + (unsigned)gx0 * (unsigned)p.Cin;
#pragma unroll
    for (int i = 0; i < MAXS; ++i) {
      const unsigned code = (i & 1) ? (slot_pk[i >> 1] >> 16) : (slot_pk[i >> 1] & 0xffffu);
      const unsigned zz = code >> 12, yy = (code >> 8) & 15u, xx = (code >> 2) & 63u, c4 = code & 3u;
      const unsigned off = org + zz * SZ + __umul24(yy, SY) + __umul24(xx, (unsigned)p.Cin) + c4 * 4u;
      if (interior) {
        pf[i] = ld4(xb + off);
        continue;
      }
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      const bool ok = ((unsigned)(gz0 + (int)zz) < (unsigned)p.Di) && ((unsigned)(gy0 + (int)yy) < (unsigned)p.Hi) &&
                      ((unsigned)(gx0 + (int)xx) < (unsigned)p.Wi);
      if (ok) {
        const float* src = xb + off;
        if (FULL) {
          val = ld4(src);
        } else {
          int left = p.Cin - cbase - (int)c4 * 4;        // real channels from here
          if (left > 0) val.x = src[0];
          if (left > 1) val.y = src[1];
          if (left > 2) val.z = src[2];
          if (left > 3) val.w = src[3];
        }
      }
      pf[i] = val;
    }
  };

  if (nstage > 0) prefetch(0);
  __syncthreads();   // s_tab
#ifdef ATVS_TILED_DEBUG
  unsigned long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long dbg_t = clock64();
#endif

  for (int stage = 0; stage < nstage; ++stage) {
    const int k = stage / p.nchunk, ch = stage - k * p.nchunk;
    TDBG(6)
    if (ch == 0) {
#pragma unroll
      for (int t = 0; t < TY; ++t)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[t][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();                       // every wave is done reading the previous stage's image
    TDBG(0)
#pragma unroll
    for (int i = 0; i < MAXS; ++i) {
      int s = tid + i * 256;
      if (s < SLOTS) {
        int a = s * 16;
        if (XP) {
          // even / odd x in separate runs of 17 voxels: (row, xl) -> row*34 + (xl&1)*17 + (xl>>1)
          int c4 = s % C4, v = s / C4;
          int xl = v % HX, row = v / HX;
          a = ((row * HX + (xl & 1) * (HX / 2) + (xl >> 1)) * C4 + c4) * 16;
        }
        if (SWZ) a = lds_swz(a);
        *reinterpret_cast<float4*>(smem + a) = pf[i];
      }
    }
    TDBG(1)
    __syncthreads();
    TDBG(2)

    // ---- K steps of this chunk.  Operands are software-pipelined by hand through register RINGS: weights W[j % 3]
    // (L2, ~600+ cycles), LDS fragments B[j % 2].  Order of the memory requests (the vector-memory counter retires in
    // order, so a wait for one load waits for every older one):
    //   weights of steps 0, 1, 2  ->  the next stage's halo (HBM / L2, the longest latency)  ->  in step j, AFTER its
    //   MFMAs are issued, the weights of step j + 3 into the ring slot step j just freed.
    // Steps 0..2 therefore never wait for the halo, and step 3 waits for loads that are three steps old.  The loop body
    // is written out for 6 consecutive steps (lcm of the ring periods) so that every ring index is a compile-time
    // constant: no register copies (a copy of a register with a load in flight is a wait), no moves.  The trip count is
    // a run-time value and MFMAs are convergent operations, so the compiler may not unroll the loop itself.
    const float4* wch = wp + ((size_t)ch * p.Jc * p.nt_total + nsi * NT) * 64;
    const int wstep = p.nt_total * 64;                                          // float4 per K step
    const int jlast = p.Jc - 1;
    if constexpr (TY == 4) {
      float4 W[3][NT], B[2][TY];
      int msk_n = 0;
      {
        const int j1 = min(1, jlast), j2 = min(2, jlast);
  #pragma unroll
        for (int n = 0; n < NT; ++n) {
          W[0][n] = wch[n * 64 + lane];
          W[1][n] = wch[(size_t)j1 * wstep + n * 64 + lane];
          W[2][n] = wch[(size_t)j2 * wstep + n * 64 + lane];
        }
      }
      if (stage + 1 < nstage) prefetch(stage + 1);   // in flight during this stage's MFMAs
      TDBG(3)
      {
        int off = s_tab[q];
        if (NT >= 2) msk_n = s_tab[p.Jc * 4];
  #pragma unroll
        for (int t = 0; t < TY; ++t) {
          int a = vbase[t] + off;
          if (SWZ) a = lds_swz(a);
          B[0][t] = *reinterpret_cast<const float4*>(lds + a);
        }
      }
      auto kstep = [&](int j, auto WC, auto BC) __attribute__((always_inline)) {
        constexpr int wc = decltype(WC)::value, bc = decltype(BC)::value, bn = bc ^ 1;
        const int msk_c = msk_n;
        // the next step's tile mask is requested ahead of the fragments (LDS returns in order)
        if (NT >= 2) msk_n = s_tab[p.Jc * 4 + min(j + 1, jlast)];
        const int off_n = s_tab[min(j + 1, jlast) * 4 + q];
  #pragma unroll
        for (int t = 0; t < TY; ++t) {
          int a = vbase[t] + off_n;
          if (SWZ) a = lds_swz(a);
          B[bn][t] = *reinterpret_cast<const float4*>(lds + a);
        }
        // output-channel tiles whose weights of this K step are all zero are skipped (the fused transposed
        // convolution's (offset, parity class) blocks: 42-56 % of its tile-steps are non-zero); the mask is uniform
        // (only instantiations with >= 2 tiles test it: the narrow ones have no structural zeros)
        if constexpr (NT >= 2) {
          const unsigned msk = (unsigned)__builtin_amdgcn_readfirstlane(msk_c) >> (nsi * NT);
  #pragma unroll
          for (int n = 0; n < NT; ++n) {
            if (!((msk >> n) & 1u)) continue;
  #pragma unroll
            for (int s = 0; s < 4; ++s)
  #pragma unroll
              for (int t = 0; t < TY; ++t)
                acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(W[wc][n], s), f4get(B[bc][t], s), acc[t][n], 0, 0, 0);
          }
        } else {
  #pragma unroll
          for (int s = 0; s < 4; ++s)
  #pragma unroll
            for (int n = 0; n < NT; ++n)
  #pragma unroll
              for (int t = 0; t < TY; ++t)
                acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(W[wc][n], s), f4get(B[bc][t], s), acc[t][n], 0, 0, 0);
        }
        // the ring slot of this step is free: weights of step j + 3 (clamped: a harmless re-read past the end)
        const int j3 = min(j + 3, jlast);
  #pragma unroll
        for (int n = 0; n < NT; ++n) W[wc][n] = wch[(size_t)j3 * wstep + n * 64 + lane];
      };
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      int j = 0;
      for (; j + 6 <= p.Jc; j += 6) {
        kstep(j, I0{}, I0{});
        kstep(j + 1, I1{}, I1{});
        kstep(j + 2, I2{}, I0{});
        kstep(j + 3, I0{}, I1{});
        kstep(j + 4, I1{}, I0{});
        kstep(j + 5, I2{}, I1{});
      }
      {
        const int rem = p.Jc - j;            // 0..5 trailing steps, same ring positions
        if (rem > 0) kstep(j, I0{}, I0{});
        if (rem > 1) kstep(j + 1, I1{}, I1{});
        if (rem > 2) kstep(j + 2, I2{}, I0{});
        if (rem > 3) kstep(j + 3, I0{}, I1{});
        if (rem > 4) kstep(j + 4, I1{}, I0{});
      }
    } else {
      // 8-row tiles: the ring form does not fit 256 registers (two workgroups per CU); their K loops are long (27 steps
      // of 32 MFMAs), so the rotating-copy form's one-step effective look-ahead is amortised
      float4 w_cur[NT], w_nx1[NT], b_cur[TY];
      {
        const int j1 = min(1, jlast);
  #pragma unroll
        for (int n = 0; n < NT; ++n) {
          w_cur[n] = wch[n * 64 + lane];
          w_nx1[n] = wch[(size_t)j1 * wstep + n * 64 + lane];
        }
      }
      if (stage + 1 < nstage) prefetch(stage + 1);   // in flight during this stage's MFMAs (after the first weights: the
      TDBG(3)                                        // vector-memory counter retires in order)
      {
        int off = s_tab[q];
  #pragma unroll
        for (int t = 0; t < TY; ++t) {
          int a = vbase[t] + off;
          if (SWZ) a = lds_swz(a);
          b_cur[t] = *reinterpret_cast<const float4*>(lds + a);
        }
      }
      for (int j = 0; j < p.Jc; ++j) {
        float4 w_nx2[NT], b_nxt[TY];
        const int off_n = s_tab[min(j + 1, jlast) * 4 + q];
        const int j2 = min(j + 2, jlast);
  #pragma unroll
        for (int n = 0; n < NT; ++n) w_nx2[n] = wch[(size_t)j2 * wstep + n * 64 + lane];
  #pragma unroll
        for (int t = 0; t < TY; ++t) {
          int a = vbase[t] + off_n;
          if (SWZ) a = lds_swz(a);
          b_nxt[t] = *reinterpret_cast<const float4*>(lds + a);
        }
        // output-channel tiles whose weights of this K step are all zero are skipped (the fused transposed
        // convolution's (offset, parity class) blocks: 42-56 % of its tile-steps are non-zero); the mask is uniform
        // (only instantiations with >= 2 tiles test it: the narrow ones have no structural zeros and no registers to spare)
        if constexpr (NT >= 2) {
          const unsigned msk = (unsigned)__builtin_amdgcn_readfirstlane(s_tab[p.Jc * 4 + j]) >> (nsi * NT);
  #pragma unroll
          for (int n = 0; n < NT; ++n) {
            if (!((msk >> n) & 1u)) continue;
  #pragma unroll
            for (int s = 0; s < 4; ++s)
  #pragma unroll
              for (int t = 0; t < TY; ++t)
                acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(w_cur[n], s), f4get(b_cur[t], s), acc[t][n], 0, 0, 0);
          }
        } else {
  #pragma unroll
          for (int s = 0; s < 4; ++s)
  #pragma unroll
            for (int n = 0; n < NT; ++n)
  #pragma unroll
              for (int t = 0; t < TY; ++t)
                acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4get(w_cur[n], s), f4get(b_cur[t], s), acc[t][n], 0, 0, 0);
        }
  #pragma unroll
        for (int n = 0; n < NT; ++n) {
          w_cur[n] = w_nx1[n];
          w_nx1[n] = w_nx2[n];
        }
  #pragma unroll
        for (int t = 0; t < TY; ++t) b_cur[t] = b_nxt[t];
      }
    }
    TDBG(4)
    if (ch != p.nchunk - 1) continue;

    // ---- epilogue of this tile: lane holds channels n*16 + 4q .. +3 of voxel (z0+wave, y0+t, x0+r)
    int z0, y0, x0;
    tile_origin(k, &z0, &y0, &x0);
    const int zo = z0 + wave;
    if (XP) {
      // rows of the accumulator = (x parity, channel): this lane holds channels (q&1)*4..+3 of voxel 2r + (q>>1)
      const int xo = x0 + 2 * r + (q >> 1), co = (q & 1) * 4;
#pragma unroll
      for (int t = 0; t < TY; ++t) {
        const int yo = y0 + t;
        if (zo >= p.Di || yo >= p.Hi || xo >= p.Wi) continue;
        size_t base = (size_t)grp * p.gy + (((size_t)zo * p.Hy + yo) * p.Wy + xo) * (size_t)p.ldy + p.ycoff + co;
        float4 v = make_float4(acc[t][0][0], acc[t][0][1], acc[t][0][2], acc[t][0][3]);
        if (p.pbias) {
          float4 b = ld4(p.pbias + (size_t)grp * p.gpb + ((size_t)yo * p.Wi + xo) * 24 + plane_variant(zo - 1, p.Di) * 8 + co);
          v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        if (p.bias) {
          float4 b = ld4(p.bias + co);
          v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        if (p.res) {
          float4 b = ld4(p.res + base);
          v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        if (p.relu) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        st4(p.y + base, v);
        ssum[0][0] += v.x; ssum[0][1] += v.y; ssum[0][2] += v.z; ssum[0][3] += v.w;
        ssq[0][0] += v.x * v.x; ssq[0][1] += v.y * v.y; ssq[0][2] += v.z * v.z; ssq[0][3] += v.w * v.w;
      }
      continue;
    }
    // 32-bit element offsets inside the sample (the launcher refuses outputs of 2^32 elements or more); everything that
    // does not depend on the row t is formed once per tile, the per-lane channel part (ycoff) once per kernel
    const int xo = x0 + r;
    const bool vox_ok = zo < p.Di && xo < p.Wi;
    float* __restrict__ ys = p.y + (size_t)grp * p.gy;
    if (p.cls_cout) {
      // fused transposed convolution: virtual channel -> (parity class, real channel), output voxel (2z+pz, 2y+py, 2x+px)
      const unsigned row0 = (((unsigned)(2 * zo) * p.Hy + 2 * y0) * p.Wy + 2 * xo) * p.ldy;
      const unsigned rstep = 2u * p.Wy * p.ldy;
#pragma unroll
      for (int t = 0; t < TY; ++t) {
        if (!vox_ok || y0 + t >= p.Hi) continue;
        const unsigned rowt = row0 + t * rstep;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          if (ycoff[n] == ~0u) continue;
          float v[4] = {acc[t][n][0], acc[t][n][1], acc[t][n][2], acc[t][n][3]};
          if (p.relu) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) v[kk] = fmaxf(v[kk], 0.f);
          }
          st4(ys + rowt + ycoff[n], make_float4(v[0], v[1], v[2], v[3]));
          if (STATS) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
              ssum[n % SN][kk] += v[kk];
              ssq[n % SN][kk] += v[kk] * v[kk];
            }
          }
        }
      }
      TDBG(5)
      continue;
    }
    const unsigned row0 = ((((unsigned)(zo * p.oS + p.offz)) * p.Hy + (y0 * p.oS + p.offy)) * p.Wy + (xo * p.oS + p.offx)) * p.ldy;
    const unsigned rstep = (unsigned)p.oS * p.Wy * p.ldy;
    const float* __restrict__ rs = p.res ? p.res + (size_t)grp * p.gy : nullptr;
    const float* __restrict__ pbs = p.pbias ? p.pbias + (size_t)grp * p.gpb : nullptr;
    const unsigned pb0 = (((unsigned)y0 * p.Wi + xo) * 3u + plane_variant(zo - 1, p.Di)) * p.Cout, pbstep = (unsigned)p.Wi * 3u * p.Cout;
#pragma unroll
    for (int t = 0; t < TY; ++t) {
      if (!vox_ok || y0 + t >= p.Hi) continue;
      const unsigned base = row0 + t * rstep;
      const float* pb = pbs ? pbs + (pb0 + t * pbstep) : nullptr;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        if (ycoff[n] == ~0u) continue;
        const int co = (nsi * NT + n) * 16 + 4 * q;
        float v[4] = {acc[t][n][0], acc[t][n][1], acc[t][n][2], acc[t][n][3]};
        if (pb) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)
            if (co + kk < p.Cout) v[kk] += pb[co + kk];
        }
        if (p.vec_out) {
          if (p.bias) {
            float4 b = ld4(p.bias + co);
            v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
          }
          if (rs) {
            float4 rr = ld4(rs + base + ycoff[n]);
            v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
          }
          if (p.relu) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) v[kk] = fmaxf(v[kk], 0.f);
          }
          st4(ys + base + ycoff[n], make_float4(v[0], v[1], v[2], v[3]));
          if (STATS) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
              ssum[n % SN][kk] += v[kk];
              ssq[n % SN][kk] += v[kk] * v[kk];
            }
          }
        } else {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            if (co + kk < p.Cout) {
              float u = v[kk];
              if (p.bias) u += p.bias[co + kk];
              if (rs) u += rs[base + ycoff[n] + kk];
              if (p.relu) u = fmaxf(u, 0.f);
              ys[base + ycoff[n] + kk] = u;
              if (STATS) {
                ssum[n % SN][kk] += u;
                ssq[n % SN][kk] += u * u;
              }
            }
          }
        }
      }
    }
    TDBG(5)
  }
#ifdef ATVS_TILED_DEBUG
  if (lane == 0 && blockIdx.x < 1024)
    for (int i = 0; i < 8; ++i) atvs_dbg_tiled[(blockIdx.x * 4 + wave) * 8 + i] = dbg_acc[i];
#endif

  if (STATS && p.stats) {
    __syncthreads();     // the tile image is dead: reuse LDS for the cross-wave reduction
    double* s_red = reinterpret_cast<double*>(smem);   // [4][2][NT*16]
#pragma unroll
    for (int n = 0; n < SN; ++n)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        double a = (double)ssum[n][k], b = (double)ssq[n][k];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          a += __shfl_xor(a, o);
          b += __shfl_xor(b, o);
        }
        if (XP) {   // lanes q and q^2 hold the same channels (the two x parities)
          a += __shfl_xor(a, 32);
          b += __shfl_xor(b, 32);
          if (q >= 2) a = b = 0.0;
        }
        if (r == 0) {
          s_red[(wave * 2 + 0) * (NT * 16) + n * 16 + 4 * q + k] = a;
          s_red[(wave * 2 + 1) * (NT * 16) + n * 16 + 4 * q + k] = b;
        }
      }
    __syncthreads();
    // this workgroup's row of the statistics buffer: width nt_total*16; columns of the output-channel
    // tiles it does not own (N-split) are written as zeros, so the buffer needs no pre-clearing.
    // Written through to memory (relaxed agent-scope store = global_store sc1) for the in-launch finalize.
    for (int i = tid; i < 2 * p.nt_total * 16; i += 256) {
      int which = i / (p.nt_total * 16), col = i % (p.nt_total * 16);
      int c = col - nsi * NT * 16;
      double v = 0.0;
      if (c >= 0 && c < NT * 16)
        v = (s_red[(0 * 2 + which) * (NT * 16) + c] + s_red[(1 * 2 + which) * (NT * 16) + c]) +
            (s_red[(2 * 2 + which) * (NT * 16) + c] + s_red[(3 * 2 + which) * (NT * 16) + c]);
      __hip_atomic_store(p.stats + ((size_t)blockIdx.x * 2 + which) * (p.nt_total * 16) + col, v, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    if (p.fin_counter) {
      // ---- last-arriver finalize: write-through partial rows, every storing wave drains its stores, one
      // relaxed agent-scope ticket; only the last arriver pays an acquire (L1 invalidate) before reading
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      int* s_flag = reinterpret_cast<int*>(smem + 4 * 2 * NT * 16 * sizeof(double));
      if (tid == 0) {
        unsigned prev = __hip_atomic_fetch_add(p.fin_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int last = (prev == (unsigned)(p.fin_arrivals - 1)) ? 1 : 0;
        if (last) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *s_flag = last;
      }
      __syncthreads();
      if (*s_flag) {
        const int C = p.fin_c, cpad = p.nt_total * 16;
        const int pairs = 2 * C;                 // (which, channel)
        const int lanes = 256 / pairs;           // row lanes per pair (C <= 64 -> >= 2)
        double* red = reinterpret_cast<double*>(smem);
        double a = 0.0;
        if (tid < lanes * pairs) {
          const int pc = tid % pairs, rl = tid / pairs;
          const int which = pc / C, c = pc % C;
          for (int r = rl; r < p.fin_rows; r += lanes)
            for (int f = 0; f < p.fin_fold; ++f) a += p.fin_stats[((size_t)r * 2 + which) * cpad + c + f * C];
        }
        __syncthreads();
        red[tid] = a;
        __syncthreads();
        if (tid < C) {
          double sm = 0.0, sq = 0.0;
          for (int l = 0; l < lanes; ++l) {
            sm += red[l * pairs + tid];
            sq += red[l * pairs + C + tid];
          }
          double mean = sm / p.fin_count;
          double var = sq / p.fin_count - mean * mean;
          if (var < 0.0) var = 0.0;
          p.fin_params[tid] = (float)mean;
          p.fin_params[C + tid] = (float)(1.0 / sqrt(var + (double)p.fin_eps));
          p.fin_params[2 * C + tid] = 0.f;
        }
        if (tid == 0) __hip_atomic_store(p.fin_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// chunks of <= 16 input channels: 16 when Cin > 8 (last chunk zero-padded), else 8 or 4
static void tiled_chunks(int Cin, int* nch, int* Cc) {
  if (Cin > 8) {
    *Cc = 16;
    *nch = (Cin + 15) / 16;
  } else {
    *Cc = (Cin > 4) ? 8 : 4;
    *nch = 1;
  }
}

extern "C" int atvs_conv_tiled_pack_size(int ntaps, int Cin, int Cout, int* nchunk, int* chunk_pad, int* ksteps_per_chunk,
                                         int* ntiles, long* packed_floats, long* table_ints) {
  if (ntaps <= 0 || Cin <= 0 || Cout <= 0 || Cout > 128) return ATVS_ERR_SHAPE;
  int nch, Ccp;
  tiled_chunks(Cin, &nch, &Ccp);
  int Jc = (ntaps * (Ccp / 4) + 3) / 4;
  int NT = pow2_tiles(Cout);
  if (nchunk) *nchunk = nch;
  if (chunk_pad) *chunk_pad = Ccp;
  if (ksteps_per_chunk) *ksteps_per_chunk = Jc;
  if (ntiles) *ntiles = NT;
  if (packed_floats) *packed_floats = (long)nch * Jc * NT * 64 * 4;
  if (table_ints) *table_ints = (long)Jc * 5;      // Jc*4 LDS offsets + Jc masks of the non-zero output tiles
  return ATVS_OK;
}

// HOST function.  Same inputs as atvs_conv_pack, taps restricted to offsets in [-1, 1]^3; `tile_y` is the
// TY the launch will use (the table holds LDS byte offsets of the (4+2) x (tile_y+2) x 18 x chunk image).
extern "C" int atvs_conv_tiled_pack(const float* w, int w_transposed, const int32_t* taps, int ntaps, int Cin, int Cout,
                                    int tile_y, int xpair, float* packed, int32_t* table) {
  if (!w || !taps || !packed || !table) return ATVS_ERR_NULL;
  int nch, Ccp, Jc, NT;
  long pf, ti;
  int rc = atvs_conv_tiled_pack_size(ntaps, Cin, Cout, &nch, &Ccp, &Jc, &NT, &pf, &ti);
  if (rc) return rc;
  const int c4n = Ccp / 4;
  const int G = ntaps * c4n;
  const int HY = tile_y + 2, HX = (xpair ? 2 : 1) * TILE_TX + 2;
  for (int t = 0; t < ntaps; ++t)
    for (int a = 1; a < 4; ++a)
      if (taps[t * 4 + a] < -1 || taps[t * 4 + a] > ((xpair && a == 3) ? 2 : 1)) return ATVS_ERR_ARG;
  for (long i = 0; i < pf; ++i) packed[i] = 0.f;
  for (int j = 0; j < Jc; ++j) table[Jc * 4 + j] = 0;
  for (int j = 0; j < Jc; ++j)
    for (int q = 0; q < 4; ++q) {
      int g = j * 4 + q;
      if (g >= G) {
        table[g] = 0;
        continue;
      }
      int t = g / c4n, c4 = g % c4n;
      int dz = taps[t * 4 + 1], dy = taps[t * 4 + 2], dx = taps[t * 4 + 3];
      int xl = dx + 1;          // x position in the halo row for column 0 of the tile
      int xcol = xpair ? ((xl & 1) * (HX / 2) + (xl >> 1)) : xl;
      table[g] = ((((dz + 1) * HY + (dy + 1)) * HX + xcol) * Ccp + c4 * 4) * 4;
      int wt = taps[t * 4 + 0];
      for (int ch = 0; ch < nch; ++ch)
        for (int n = 0; n < NT; ++n)
          for (int co16 = 0; co16 < 16; ++co16) {
            int co = n * 16 + co16;
            if (co >= Cout) continue;
            for (int s = 0; s < 4; ++s) {
              int ci = ch * Ccp + c4 * 4 + s;
              if (ci >= Cin) continue;
              float val = w_transposed ? w[((size_t)wt * Cout + co) * Cin + ci] : w[((size_t)wt * Cin + ci) * Cout + co];
              packed[(((((size_t)ch * Jc + j) * NT + n) * 64) + q * 16 + co16) * 4 + s] = val;
              if (val != 0.f) table[Jc * 4 + j] |= 1 << n;
            }
          }
    }
  return ATVS_OK;
}

static long tiled_ntiles(int Do, int Ho, int Wo, int tile_y, int xpair = 0) {
  int tx = (xpair ? 2 : 1) * TILE_TX;
  return (long)((Do + TILE_TZ - 1) / TILE_TZ) * ((Ho + tile_y - 1) / tile_y) * ((Wo + tx - 1) / tx);
}

// N-split factor: deal the 16-channel output tiles of a spatial tile to `ns` workgroups when that
// shortens the launch: estimated time = rounds of the persistent grid x cost of one work item
// (MFMA work ~ tiles per item, plus a fixed staging / epilogue share).  ntiles counts every sample's tiles.
static int tiled_nsplit(long ntiles, int NT, int tile_y, int C4, bool xp = false) {
  int best = 1;
  double best_t = 1e30;
  for (int ns = 1; ns <= NT; ns <<= 1) {
    long cap = 256L * tiled_wps(NT / ns, tile_y, C4, xp);
    long rounds = (ntiles * ns + cap - 1) / cap;
    double t = (double)rounds * ((double)(NT / ns) + 0.5);
    if (t < best_t - 1e-9) {
      best_t = t;
      best = ns;
    }
  }
  return best;
}

// workgroups PER SAMPLE of a launch over `groups` independent samples (rows of stats_partial = groups * this):
// the persistent grid (256 CUs x resident workgroups) is shared out among the samples, a multiple of 8 * nsplit each.
extern "C" long atvs_conv_tiled_grid(int Do, int Ho, int Wo, int tile_y, int Cin, int Cout, int xpair, int groups,
                                     int* nsplit_out) {
  int nch, Ccp;
  tiled_chunks(Cin, &nch, &Ccp);
  if (groups < 1) groups = 1;
  long nt = tiled_ntiles(Do, Ho, Wo, tile_y, xpair);
  int NT = xpair ? 1 : pow2_tiles(Cout);
  int ns = tiled_nsplit(nt * groups, NT, tile_y, Ccp / 4, xpair != 0);
  long cap = 256L * tiled_wps(NT / ns, tile_y, Ccp / 4, xpair != 0);
  long unit = 8L * ns;
  long share = cap / groups / unit * unit;           // this sample's share of the resident workgroups
  if (share < unit) share = unit;
  long want = (nt * ns + unit - 1) / unit * unit;
  long g = want < share ? want : share;
  if (nsplit_out) *nsplit_out = ns;
  return g;
}
extern "C" long atvs_conv_tiled_num_blocks(int Do, int Ho, int Wo, int tile_y, int Cin, int Cout, int xpair, int groups) {
  return atvs_conv_tiled_grid(Do, Ho, Wo, tile_y, Cin, Cout, xpair, groups, nullptr);
}
extern "C" int atvs_conv_tiled_has_stats(int Do, int Ho, int Wo, int tile_y, int Cin, int Cout, int groups) {
  int nch, Ccp;
  tiled_chunks(Cin, &nch, &Ccp);
  if (groups < 1) groups = 1;
  int NT = pow2_tiles(Cout);
  return tiled_has_stats(NT / tiled_nsplit(tiled_ntiles(Do, Ho, Wo, tile_y) * groups, NT, tile_y, Ccp / 4)) ? 1 : 0;
}

template <int NT, int TY, int C4, bool FULL, bool XP = false>
static int launch_tiled(const TiledArgs& a, long blocks, hipStream_t s) {
  size_t lds = (size_t)(TILE_TZ + 2) * (TY + 2) * ((XP ? 2 : 1) * TILE_TX + 2) * C4 * 16 + (size_t)a.Jc * 5 * sizeof(int);
  size_t red = (size_t)4 * 2 * NT * 16 * sizeof(double) + 16;
  if (red < 256 * sizeof(double)) red = 256 * sizeof(double);
  if (lds < red) lds = red;
  static AtvsAttrOnce lds_once;                   // per kernel instantiation (this function is a template / has one kernel)
  if (const int rc_ = atvs_set_max_lds_once(lds_once, reinterpret_cast<const void*>(conv_tiled_f32_kernel<NT, TY, C4, FULL, XP>), 160 * 1024)) return rc_;
  hipLaunchKernelGGL((conv_tiled_f32_kernel<NT, TY, C4, FULL, XP>), dim3((unsigned)(blocks * a_groups(a))), dim3(256), lds, s, a);
  return ATVS_OK;
}

template <int TY>
static int launch_xp(const TiledArgs& a, int C4, bool full, long blocks, hipStream_t s) {
  if (C4 == 4) return full ? launch_tiled<1, TY, 4, true, true>(a, blocks, s) : launch_tiled<1, TY, 4, false, true>(a, blocks, s);
  if (C4 == 2) return full ? launch_tiled<1, TY, 2, true, true>(a, blocks, s) : launch_tiled<1, TY, 2, false, true>(a, blocks, s);
  if (C4 == 1) return full ? launch_tiled<1, TY, 1, true, true>(a, blocks, s) : launch_tiled<1, TY, 1, false, true>(a, blocks, s);
  return ATVS_ERR_ARG;
}

template <int NT, int TY>
static int launch_c4(const TiledArgs& a, int C4, bool full, long blocks, hipStream_t s) {
  if (C4 == 4) return full ? launch_tiled<NT, TY, 4, true>(a, blocks, s) : launch_tiled<NT, TY, 4, false>(a, blocks, s);
  if (C4 == 2) return full ? launch_tiled<NT, TY, 2, true>(a, blocks, s) : launch_tiled<NT, TY, 2, false>(a, blocks, s);
  if (C4 == 1) return full ? launch_tiled<NT, TY, 1, true>(a, blocks, s) : launch_tiled<NT, TY, 1, false>(a, blocks, s);
  return ATVS_ERR_ARG;
}

// 3-D stride-1 halo-1 stencil on the LDS-tiled kernel.  x (D,H,W,Cin); the logical output grid equals
// the input grid (D,H,W); output voxel = o*out_stride + off inside the full tensor (Dy,Hy,Wy,ldy).
// stats_partial rows = atvs_conv_tiled_num_blocks(D,H,W,tile_y), width 16*ntiles.  tile_y in {4, 8}.
extern "C" int atvs_conv_tiled_f32(const float* x, const float* packed_w, const int32_t* table, const float* bias,
                                   const float* residual, const float* plane_bias, float* y, double* stats_partial,
                                   int groups, int D, int H, int W, int Cin, int Dy, int Hy, int Wy, int out_stride, int off_z,
                                   int off_y, int off_x, int ldy, int y_coff, int Cout, int ntaps, int tile_y, int relu,
                                   int class_cout, int class_base, int xpair, uint32_t* fin_counter, float* fin_params,
                                   double* fin_stats, int fin_rows, int fin_arrivals, int fin_channels, int fin_fold,
                                   long fin_count, float fin_eps, atvs_stream_t stream) {
  if (!x || !packed_w || !table || !y) return ATVS_ERR_NULL;
  if (groups <= 0 || D <= 0 || H <= 0 || W <= 0 || out_stride <= 0) return ATVS_ERR_SHAPE;
  if (fin_counter && groups != 1) return ATVS_ERR_ARG;       // the in-launch finalize is single-sample
  if (class_cout) {
    // fused transposed convolution: Cout = classes_in_this_launch * class_cout virtual channels
    if (class_cout % 4 || Cout % class_cout || class_base < 0 || class_base + Cout / class_cout > 8) return ATVS_ERR_SHAPE;
    if (y_coff < 0 || y_coff + class_cout > ldy || (ldy % 4) || (y_coff % 4)) return ATVS_ERR_SHAPE;
    if (2 * D > Dy || 2 * H > Hy || 2 * W > Wy || bias || residual || plane_bias) return ATVS_ERR_ARG;
  } else {
    if (y_coff < 0 || y_coff + Cout > ldy) return ATVS_ERR_SHAPE;
    if ((D - 1) * out_stride + off_z >= Dy || (H - 1) * out_stride + off_y >= Hy || (W - 1) * out_stride + off_x >= Wy)
      return ATVS_ERR_SHAPE;
  }
  if (residual && y_coff != 0) return ATVS_ERR_ARG;
  if (plane_bias && (out_stride != 1 || D < 2)) return ATVS_ERR_ARG;
  if ((double)D * H * W * Cin >= 2147483648.0) return ATVS_ERR_SHAPE;   // 31-bit tile-relative element offsets
  if ((double)Dy * Hy * Wy * ldy >= 4294967296.0 || (double)W * Cin >= 16777216.0) return ATVS_ERR_SHAPE;   // 32-bit output offsets, 24-bit row pitch
  if (xpair && (Cout != 8 || class_cout || out_stride != 1 || (ldy % 4) || (y_coff % 4) || ntaps != 36)) return ATVS_ERR_ARG;
  int nch, Ccp, Jc, NT;
  int rc = atvs_conv_tiled_pack_size(ntaps, Cin, xpair ? 16 : Cout, &nch, &Ccp, &Jc, &NT, nullptr, nullptr);
  if (rc) return rc;
  TiledArgs a;
  a.x = x; a.wp = packed_w; a.tab = table; a.bias = bias; a.res = residual; a.y = y; a.stats = stats_partial;
  a.Di = D; a.Hi = H; a.Wi = W; a.Cin = Cin; a.Hy = Hy; a.Wy = Wy;
  a.oS = out_stride; a.offz = off_z; a.offy = off_y; a.offx = off_x; a.ldy = ldy; a.ycoff = y_coff; a.Cout = Cout;
  a.nchunk = nch; a.Jc = Jc;
  const int txv = (xpair ? 2 : 1) * TILE_TX;
  a.tiles_z = (D + TILE_TZ - 1) / TILE_TZ; a.tiles_y = (H + tile_y - 1) / tile_y; a.tiles_x = (W + txv - 1) / txv;
  a.ntiles = a.tiles_z * a.tiles_y * a.tiles_x;
  a.relu = relu;
  a.vec_out = (Cout % 4 == 0) && (ldy % 4 == 0) && (y_coff % 4 == 0);
  a.pbias = plane_bias;
  a.cls_cout = class_cout; a.cls_base = class_base;
  a.fin_counter = fin_counter; a.fin_params = fin_params; a.fin_stats = fin_stats; a.fin_rows = fin_rows;
  a.fin_arrivals = fin_arrivals; a.fin_c = fin_channels; a.fin_fold = fin_fold; a.fin_count = (double)fin_count;
  a.fin_eps = fin_eps;
  if (fin_counter && (!stats_partial || !fin_params || !fin_stats || fin_rows <= 0 || fin_arrivals <= 0 ||
                      fin_channels <= 0 || fin_channels > 64 || fin_fold < 1 || fin_count <= 0))
    return ATVS_ERR_ARG;
  const int C4 = Ccp / 4;
  const bool full = (Cin % Ccp == 0);
  int ns = 1;
  long blocks = atvs_conv_tiled_grid(D, H, W, tile_y, Cin, Cout, xpair, groups, &ns);     // per sample
  a.nsplit = ns; a.nt_total = NT;
  a.wg = (int)blocks; a.ngroups = groups;
  a.gx = (long)D * H * W * Cin; a.gy = (long)Dy * Hy * Wy * ldy; a.gpb = (long)H * W * 3 * Cout;
  if (blocks * groups > 0x7fffffffL) return ATVS_ERR_SHAPE;
  const int NTg = NT / ns;
  if (stats_partial && !tiled_has_stats(NTg)) return ATVS_ERR_ARG;
  hipStream_t s = as_stream(stream);
  if (xpair) {
    if (tile_y == 8 && C4 < 4) rc = launch_xp<8>(a, C4, full, blocks, s);
    else if (tile_y == 4) rc = launch_xp<4>(a, C4, full, blocks, s);
    else return ATVS_ERR_ARG;
  } else if (tile_y == 8) {
    if (NTg == 1) rc = launch_c4<1, 8>(a, C4, full, blocks, s);
    else if (NTg == 2) rc = launch_c4<2, 8>(a, C4, full, blocks, s);
    else return ATVS_ERR_ARG;
  } else if (tile_y == 4) {
    if (NTg == 1) rc = launch_c4<1, 4>(a, C4, full, blocks, s);
    else if (NTg == 2) rc = launch_c4<2, 4>(a, C4, full, blocks, s);
    else if (NTg == 4) rc = launch_c4<4, 4>(a, C4, full, blocks, s);
    else if (NTg == 8) rc = launch_c4<8, 4>(a, C4, full, blocks, s);
    else return ATVS_ERR_ARG;
  } else {
    return ATVS_ERR_ARG;
  }
  if (rc) return rc;
  ATVS_LAUNCH_CHECK();
  return ATVS_OK;
}
