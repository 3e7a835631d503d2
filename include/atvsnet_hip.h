/*
 * atvsnet_hip.h -- C-ABI of the MI355X (gfx950) kernels for the A-TVSNet
 * two-view cost-volume inference path.
 *
 * This is the drop-in boundary: plain pointers, sizes and
 * scalars, no framework types.  Every pointer is a DEVICE pointer to float32
 * data unless stated otherwise; tensors are dense, row-major, channel-last
 * (the reference's NDHWC / NHWC), batch size 1 (FLAGS.batch_size, example.py:45).
 * The caller owns every buffer (outputs and workspaces included); kernels never
 * allocate.  Calls are asynchronous on `stream` (a hipStream_t; NULL = default
 * stream), re-entrant per stream, hold no global state, and return ATVS_OK or a
 * negative error code (no exceptions cross the ABI).
 *
 * Each entry point names the reference TensorFlow op chain it replaces
 * (paths relative to the reference repository root).
 */
#ifndef ATVSNET_HIP_H
#define ATVSNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* atvs_stream_t; /* hipStream_t */

enum {
  ATVS_OK = 0,
  ATVS_ERR_NULL = -1,   /* a required pointer is NULL */
  ATVS_ERR_SHAPE = -2,  /* inconsistent or unsupported sizes */
  ATVS_ERR_ARG = -3,    /* bad enum / flag value */
  ATVS_ERR_LAUNCH = -4  /* the HIP runtime rejected the launch */
};

/* ABI version of this header (bumped on any signature change).  atvs_abi_version() returns the value the library
 * was compiled with; the loader (a-tvsnet_amd/_lib.py) refuses a library whose version differs from this header's. */
#define ATVS_ABI_VERSION 43
int atvs_abi_version(void);
/* "gfx950" -- the only code object in the library. */
const char* atvs_target_arch(void);

/* ------------------------------------------------------------------------- *
 * Geometry  (atvsnet/homography_warping.py)
 * ------------------------------------------------------------------------- */

/* get_homographies, homography_warping.py:179-227.
 * left_cam/right_cam: (2,4,4) [extrinsic; K + depth range].  depth_start,
 * depth_interval: 1 float each.  homographies out: (depth_num,3,3). */
int atvs_get_homographies(const float* left_cam, const float* right_cam, const float* depth_start,
                          const float* depth_interval, float* homographies, int depth_num,
                          int inverse_depth, atvs_stream_t stream);

/* D x homography_warping(bilinear) + tf.stack, with the caller's epilogue fused:
 * homography_warping.py:230-271 + :31-104, unrolled over d at model.py:190-194,
 * 272-279, 292-297.
 *   src (h,w,C); homographies (D,3,3); out (D,h,w,ld_out) written at channels
 *   [c_off, c_off+C) (mode 2: [c_off, c_off+rep)); mask_out (D,h,w) or NULL.
 *   mode 0: warp.   mode 1: |warp - ref| * mask, ref (h,w,C) (photo volume).
 *   mode 2: C == 1; (|warp - delta_d| / interval / D) * mask replicated to `rep`
 *           channels (geo view volume, reference quirk: 16 identical channels).
 *   mode 3: nearest-neighbour warp (homography_warping.py:45-56: tf.round half-to-even,
 *           out-of-range pixels read source pixel (0,0) and are NOT zeroed; mask_out tells).
 *   planar != 0 (mode 0 or 1, C in {16, 32, 64}, ld_out == C, c_off == 0): out is written chunk-planar,
 *           [C/8] planes of [D][h][w][8], `planar` floats apart (>= D*h*w*8; pad it so that the planes do not start
 *           on the same HBM channel) -- the layout atvs_conv_xw_f32 / _xb_f32 read with x_planar (dense 32-byte
 *           voxels per 8-channel chunk); same values, another place.
 *   pieces != 0 (with planar): every value is written as its two fp16 pieces, h0 = fp16(x), h1 = fp16((x - h0) * 2048) -- the
 *           operand split of the split-operand convolutions, done once by the producer: a chunk plane then holds
 *           [2 pieces][D][h][w][8 fp16] (the same D*h*w*32 bytes), which atvs_conv_xb_f32 reads with x_pieces and stages
 *           by LDS-DMA. */
int atvs_warp_planes(const float* src, const float* homographies, const float* ref,
                     const float* depth_start, const float* depth_interval, float* out, float* mask_out,
                     int D, int h, int w, int C, int ld_out, int c_off, int mode, int rep, long planar, int pieces,
                     atvs_stream_t stream);

/* build_cost_volume, model.py:157-200: tf.tile(ref) ++ stack_d(warp_d(view)) ->
 * cost_volume (D,h,w,2C).  C % 4 == 0. */
int atvs_build_cost_volume(const float* ref_feature, const float* view_feature, const float* homographies,
                           float* cost_volume, int D, int h, int w, int C, atvs_stream_t stream);

/* tf.tile(tf.expand_dims(x,1),[1,D,1,1,1]) into a channel slice: model.py:311,316,329-330. */
int atvs_tile_planes(const float* src, float* out, int D, int h, int w, int C, int ld_out, int c_off,
                     atvs_stream_t stream);

/* cost_volume_geo_ref, model.py:289-290: |depth_ref - delta_d| / interval / D into channel c_off. */
int atvs_geo_ref_planes(const float* depth_ref, const float* depth_start, const float* depth_interval,
                        float* out, int D, int h, int w, int ld_out, int c_off, atvs_stream_t stream);
/* geo_ref and geo_view of the refinement (model.py:285-300) in one launch: out (D,h,w,ld)[..., c_off] = atvs_geo_ref_planes,
 * [..., c_off + 1 .. c_off + rep] = atvs_warp_planes(mode 2, rep) of the one-channel map view_depth (h,w) -- the same values. */
int atvs_geo_volume(const float* depth_ref, const float* view_depth, const float* homographies, const float* depth_start,
                    const float* depth_interval, float* out, int D, int h, int w, int ld_out, int c_off, int rep,
                    atvs_stream_t stream);

/* get_visual_hull with view_num = 2, homography_warping.py:329-387: out (D,h,w).
 * view_depth_in_ref = transform_depth(view depth, view_cam, ref_cam). */
int atvs_visual_hull(const float* ref_depth, const float* view_depth_in_ref, const float* homographies,
                     const float* depth_start, const float* depth_interval, float* out, int D, int h, int w,
                     int inverse_depth, atvs_stream_t stream);

/* homography_warping_by_depth, homography_warping.py:108-176.  method 0 bilinear,
 * 1 nearest.  mask_out (h,w) or NULL.  pose_ws: 12 floats of scratch. */
int atvs_warp_by_depth(const float* src, const float* left_cam, const float* right_cam, const float* depth,
                       float* out, float* mask_out, float* pose_ws, int h, int w, int C, int method,
                       int inverse_depth, atvs_stream_t stream);

/* |homography_warping_by_depth(src) - ref| * mask into channels [c_off, c_off + C) of out (h,w,ld_out): photo_err / geo_err of the
 * refinement (model.py:309-316) in one launch; the operations and order of atvs_warp_by_depth + atvs_absdiff_mask.
 * copy_ref != 0: ref itself is written to the C channels behind the error map (the tiled map that follows it, :329-334). */
int atvs_warp_by_depth_err(const float* src, const float* ref, const float* left_cam, const float* right_cam,
                           const float* depth, float* out, int ld_out, int c_off, int h, int w, int C, int method,
                           int inverse_depth, int copy_ref, atvs_stream_t stream);

/* interpolate, homography_warping.py:31-104, with caller-supplied coordinates: src (h,w,C), x / y (n) in the
 * reference's texture coordinates (pixel centres at +0.5), out (n,C), mask_out (n) 1.f / 0.f or NULL.  method 0
 * bilinear (invalid points give 0), 1 nearest (tf.round; invalid points read pixel (0,0), not masked). */
int atvs_interpolate(const float* src, const float* x, const float* y, float* out, float* mask_out, long n,
                     int h, int w, int C, int method, atvs_stream_t stream);

/* get_pixel_grids, homography_warping.py:8-17: out (3*h*w) = [x + 0.5 | y + 0.5 | 1], row-major pixels. */
int atvs_pixel_grids(float* out, int h, int w, atvs_stream_t stream);

/* transform_depth, homography_warping.py:275-326.  ws14: 14 floats of scratch. */
int atvs_transform_depth(const float* depth, const float* left_cam, const float* right_cam, float* out,
                         float* ws14, int h, int w, int inverse_depth, atvs_stream_t stream);
/* n <= 16 maps of one size in one launch (a map per workgroup; h * w <= 32,768: atvs_transform_depth_batch_supported): the values of n
 * calls of atvs_transform_depth (the refinement transforms every source view's depth map twice, model.py:289,321-324).  The four
 * arguments are HOST arrays of n device pointers. */
int atvs_transform_depth_batch_supported(int h, int w);
int atvs_transform_depth_batch(const float* const* depth, const float* const* left_cam, const float* const* right_cam,
                               float* const* out, int n, int h, int w, int inverse_depth, atvs_stream_t stream);

/* tf.abs(a - b) * tile(mask): photo_err / geo_err, model.py:310,315.
 * a, b, out (npix, C); mask (npix). */
int atvs_absdiff_mask(const float* a, const float* b, const float* mask, float* out, int npix, int C,
                      atvs_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Soft-argmin  (atvsnet/model.py)
 * ------------------------------------------------------------------------- */

/* prob2depth, model.py:80-109: cost (groups,D,h,w) -> depth (groups,h,w); the volumes of a launch share the depth sweep. */
int atvs_softargmin(const float* cost, const float* depth_start, const float* depth_interval,
                    float* depth_out, int groups, int D, int h, int w, atvs_stream_t stream);

/* upsample_prob_vol + prob2depth, model.py:68-76,121-127: cost (D,h,w) ->
 * depth (h*up, w*up) without materialising the upsampled volume. */
int atvs_upsample_softargmin(const float* cost, const float* depth_start, const float* depth_interval,
                             float* depth_up_out, int D, int h, int w, int up_scale, atvs_stream_t stream);

/* get_propability_map (model.py:13-65) as used by prob2depth / prob2depth_upsample(out_prob_map=True)
 * (:104-107, :122-125; the ETH3D driver eval_pointcloud.py:232,269): per pixel the sum of the probabilities of
 * the four depth planes around the estimated depth, l0 = clip(floor(d)), l1 = clip(l0-1), r0 = clip(ceil(d)),
 * r1 = clip(r0+1), d = (depth - depth_start) / depth_interval.  vol (D,h,w); depth_map, prob_out
 * (h*up_scale, w*up_scale).  softmax != 0: vol is the pre-softmax cost and P = softmax(-vol) is evaluated on
 * the fly; up_scale > 1: vol is read through the align_corners bilinear interpolation of upsample_prob_vol
 * (:66-75) -- neither the probability volume nor its upsampled copy is materialised. */
int atvs_probability_map(const float* vol, const float* depth_map, const float* depth_start,
                         const float* depth_interval, float* prob_out, int D, int h, int w, int up_scale, int softmax,
                         atvs_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Convolutions  (cnn_wrapper/network.py)
 * ------------------------------------------------------------------------- */

/* Sizes of the packed weights / group table for `ntaps` taps (see atvs_conv_pack).
 * vec = 4 when Cin % 4 == 0 else 1; ksteps = ceil(ntaps*Cin/vec / 4); ntiles = Cout
 * rounded up to 16, 32, 64 or 128, over 16.  Host function, no GPU work. */
int atvs_conv_pack_size(int ntaps, int Cin, int Cout, int* vec, int* ksteps, int* ntiles,
                        long* packed_floats, long* table_ints);

/* HOST function (plain CPU loops, host pointers): arrange a TF-layout kernel for
 * atvs_conv_mfma_f32.  w: [n_w_taps][Cin][Cout] (w_transposed = 0, tf.layers.conv2d/3d
 * kernels [k.., Cin, Cout]) or [n_w_taps][Cout][Cin] (w_transposed = 1,
 * conv3d_transpose kernels [k.., Cout, Cin]).  taps: ntaps x 4 int32 =
 * (index of the tap in w, dz, dy, dx), (dz,dy,dx) = input offset relative to
 * output_index * in_stride (SAME / explicit padding and dilation folded in; the 8
 * parity classes of a stride-2 transposed convolution are 8 tap lists).
 * The caller uploads `packed` and `table` to the device. */
int atvs_conv_pack(const float* w, int w_transposed, const int32_t* taps, int ntaps, int Cin, int Cout,
                   float* packed, int32_t* table);

/* Workgroups a launch with M = Do*Ho*Wo output voxels and `tile_m` uses
 * (= rows of stats_partial): ceil(M / (64 * tile_m)). */
long atvs_conv_num_blocks(long M, int tile_m);

/* tf.layers.conv2d / conv3d / conv3d_transpose (one parity class per call),
 * slim.conv2d, tf.nn.conv3d + bias_add + residual add + relu, on the fp32 matrix
 * cores: network.py:141-215, 282-351, 510-602.
 *   x (Di,Hi,Wi,Cin) (2-D: Di = 1); y rows of width ldy, channels [y_coff, y_coff+Cout)
 *   of the full output tensor (Dy,Hy,Wy); this launch writes output voxels
 *   o*out_stride + off for o in the logical grid (Do,Ho,Wo) and reads inputs at
 *   o*in_stride + tap offset.  bias (Cout) / residual (same addressing as y,
 *   y_coff must be 0) may be NULL.  relu != 0 applies max(.,0) last.
 *   plane_bias: NULL, or (Ho,Wo,3*Cout): the contribution of input channels that are
 *   constant along depth (the tf.tile'd halves of model.py:186,311,316,329-330), i.e. the
 *   2-D convolution of those channels with the kd-summed kernel, for the three sets of
 *   in-range kd taps [kd=0 missing | all | kd=2 missing]; added per output voxel according
 *   to its plane (pad_z = z padding before).  3x3x3 kernels, out_stride 1 only.
 *   stats_partial: NULL, or [atvs_conv_num_blocks][2][16*ntiles] doubles receiving
 *   per-workgroup (sum, sum of squares) of the values written, per channel, for
 *   training-mode batch norm (network.py:206-212).  tile_m in {1,2,4,8},
 *   tile_m * ntiles <= 16.
 *   groups >= 1: x, y, residual and plane_bias are `groups` independent samples stacked on a leading axis (one
 *   network call each in the reference); a workgroup never spans samples and stats_partial has
 *   groups * atvs_conv_num_blocks rows, sample-major (atvs_bn_finalize reduces them per sample). */
int atvs_conv_mfma_f32(const float* x, const float* packed_w, const int32_t* group_table, const float* bias,
                       const float* residual, const float* plane_bias, int pad_z, float* y,
                       double* stats_partial, int groups, int Di, int Hi, int Wi,
                       int Cin, int Do, int Ho, int Wo, int in_stride, int Dy, int Hy, int Wy,
                       int out_stride, int off_z, int off_y, int off_x, int ldy, int y_coff, int Cout,
                       int ntaps, int tile_m, int relu, atvs_stream_t stream);

/* LDS-tiled variant of the same convolution for halo-1 stencils: 3-D, input step 1, every
 * tap offset in [-1,1]^3 (3x3x3 stride-1 SAME convolutions; each parity class of the
 * stride-2 transposed convolution).  A workgroup computes a 4 x tile_y x 16 block of the
 * logical output grid (= the input grid D,H,W) from an input halo tile staged once in LDS.
 * atvs_conv_tiled_pack_size / _pack are the HOST packing functions for it (same inputs
 * as atvs_conv_pack; the table depends on tile_y in {4, 8}).  The grid is persistent
 * (atvs_conv_tiled_num_blocks workgroups sweep the tiles); stats_partial has that many rows of
 * 16*ntiles doubles x 2 (one per workgroup).  Small volumes deal the 16-channel output tiles of a
 * spatial tile to several workgroups (atvs_conv_tiled_grid reports nsplit): each then fills its own
 * columns of its row and zeros in the others.  Launches whose per-workgroup
 * width is 128 channels produce no statistics (atvs_conv_tiled_has_stats == 0: pass NULL and
 * use atvs_channel_stats on the output).
 * Fused stride-2 transposed convolution (class_cout != 0): the 8 output parity classes of
 * conv3d_transpose(3, stride 2, SAME) are computed from ONE staged tile -- the GEMM's N axis is
 * (class, channel): Cout = n_classes * class_cout "virtual" channels over the 8 taps
 * {-1,0}^3 with the dense virtual kernel Wv[off][ci][(class, co)] = W[k(off, class)] or 0; virtual
 * channel (c, co) of grid voxel j is written to output voxel 2j + parity(class_base + c); the
 * statistics columns (c, co) fold onto co in atvs_bn_finalize(fold = n_classes).
 * x-pair form (xpair != 0, Cout == 8 only): the 16 GEMM rows are (x parity, channel) and a tile column
 * is the voxel pair (2i, 2i+1), so no row of the MFMA tile is padding; the packed kernel is the dense
 * virtual kernel Wv[(kd,kh,ox)][ci][(jx,co)] = W[kd][kh][ox-jx+1] (or 0) over the 36 taps
 * ox in -1..2 (pack with xpair = 1 and 16 output channels); tile_y = 4 for 16-channel chunks.
 * In-launch finalize (fin_counter != NULL): the last workgroup to arrive (one agent-scope atomic ticket on
 * *fin_counter, which must be zero before the layer's first launch and is left at zero) reduces the
 * fin_rows rows of fin_stats (the layer's whole statistics buffer; fin_arrivals = workgroups of all the
 * layer's launches) into fin_params = (3, fin_channels) floats (mean, rsqrt(var + fin_eps), 0), exactly
 * what atvs_bn_finalize(beta = NULL) would write -- no separate finalize launch.  fin_channels <= 64.
 * `table` (atvs_conv_tiled_pack): per K step the 4 LDS offsets of its (tap, channel group) entries, then per
 * K step a bit mask of the 16-channel output tiles whose packed weights are not all zero -- the kernel skips
 * the others (the structural zeros of the fused transposed convolution: 44-58 % of its tile-steps). */
int atvs_conv_tiled_pack_size(int ntaps, int Cin, int Cout, int* nchunk, int* chunk_pad, int* ksteps_per_chunk,
                              int* ntiles, long* packed_floats, long* table_ints);
int atvs_conv_tiled_pack(const float* w, int w_transposed, const int32_t* taps, int ntaps, int Cin, int Cout,
                         int tile_y, int xpair, float* packed, int32_t* table);
/* groups >= 1 independent samples stacked on the leading axis of x / y / residual / plane_bias: the persistent grid is
 * shared out among the samples (atvs_conv_tiled_num_blocks / _grid return the workgroups PER SAMPLE; stats_partial has
 * groups times that many rows, sample-major); the in-launch finalize needs groups == 1. */
long atvs_conv_tiled_num_blocks(int Do, int Ho, int Wo, int tile_y, int Cin, int Cout, int xpair, int groups);
long atvs_conv_tiled_grid(int Do, int Ho, int Wo, int tile_y, int Cin, int Cout, int xpair, int groups, int* nsplit_out);
int atvs_conv_tiled_has_stats(int Do, int Ho, int Wo, int tile_y, int Cin, int Cout, int groups);
int atvs_conv_tiled_f32(const float* x, const float* packed_w, const int32_t* table, const float* bias,
                        const float* residual, const float* plane_bias, float* y, double* stats_partial,
                        int groups, int D, int H, int W, int Cin,
                        int Dy, int Hy, int Wy, int out_stride, int off_z, int off_y, int off_x, int ldy,
                        int y_coff, int Cout, int ntaps, int tile_y, int relu, int class_cout, int class_base,
                        int xpair, uint32_t* fin_counter, float* fin_params, double* fin_stats, int fin_rows,
                        int fin_arrivals, int fin_channels, int fin_fold, long fin_count, float fin_eps,
                        atvs_stream_t stream);

/* conv / conv_bn(3, 8, 1) on a full-resolution volume -- 3x3x3 SAME stride-1 convolution to EIGHT output
 * channels (conv_b*_0_1, global_refine_3dconv0_1, the refinement stems; cnn_wrapper/atvsnet.py, layer code
 * cnn_wrapper/network.py:165-215), Cin % 8 == 0.  x-pair form (two x-adjacent voxels fill the 16 MFMA rows),
 * ONE workgroup per CU with the whole register file and LDS.  Two kernels share the contract: conv_xb.hip (split fp16
 * operands on the 16-bit matrix cores: the product's kernel) and conv_xw.hip (fp32 matrix cores, Winograd F(2,3) along y:
 * the A/B form, ops.configure(split16=False)).
 *   atvs_conv_x{w,b}_pack_size / _pack  HOST: pack the TF kernel [3,3,3,Cin,8] (upload the result)
 *   atvs_conv_xpair_grid                   workgroups of a launch PER SAMPLE = rows of stats_partial per sample ([2][16] doubles
 *                                       each, columns 0..7 = channels, the layout atvs_bn_finalize takes with cpad 16)
 *   atvs_conv_x{w,b}_f32                y (D,H,W,ldy)[..., y_coff + co] = conv(x) (+ bias, + plane_bias (H,W,24), ReLU)
 * Sibling: the U-Nets feed the same tensor to conv_b*_0_1 (8 channels, stride 1) and to the encoder branch
 * conv_b*_1_0 (16 channels, stride 2; cnn_wrapper/atvsnet.py StackedUNet, CostVolRefineNet 0_1 / 1_0).  With
 * packed_w2 != NULL (atvs_conv_x{w,b}_pack_sibling of the TF kernel [3,3,3,Cin,16]) the launch also writes
 * y2 (ceil(D/2), ceil(H/2), ceil(W/2), ldy2)[..., y_coff2 + co] = that stride-2 SAME convolution (+ plane_bias2
 * (Ho2, Wo2, 48)) from the tile already staged in LDS, and its moments into stats_partial2 (same rows).
 * groups >= 1 independent samples stacked on the leading axis of every tensor.
 * Prologue (normalise-on-load / add-on-load): with in_params != NULL the convolution's input is relu?((x - mean) * rstd +
 * beta) per sample, parameters (groups, 3, Cin) -- the producer's training-mode batch norm, network.py:206-212 -- and with
 * x2 != NULL the SUM of two such terms (in_params2 for x2; either parameter block may be NULL = that term as is): the
 * U-Net's skip add (network.py:695-697) formed while the halo is staged.  Out-of-volume taps stay zero.  Built for the
 * shapes the path has: in_params with Cin % 16 == 0 and a sibling; x2 with a sibling and Cin % 16 == 8 (conv_xw) /
 * Cin == 8 (conv_xb) -- else ATVS_ERR_ARG / ATVS_ERR_SHAPE. */
long atvs_conv_xpair_grid(int D, int H, int W, int groups);

/* The fp32 form, conv_xw.hip: x-pair rows x minimal filtering F(2,3) along y -- two output rows from 4 products per
 * (kd, x offset, channel) instead of 6.  The filter transform
 * U = [g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2] is applied by the HOST packer in double; the input transform
 * [d0-d2, d1+d2, d2-d1, d1-d3] in registers from the raw rows staged in LDS (8-channel chunks, double-buffered image).
 * Results differ from the direct sum by fp32 rounding only (one 32 -> 8 layer: 7.7e-7 of the output maximum).
 * x_planar != 0 (no prologue): x is chunk-planar per sample, Cin/8 planes of [D][H][W][8] x_planar floats apart, as
 * atvs_warp_planes writes it with the same `planar` (sample stride = x_planar * Cin / 8). */
int atvs_conv_xw_pack_size(int Cin, long* packed_floats);
int atvs_conv_xw_pack(const float* w, int Cin, float* packed);
int atvs_conv_xw_pack_sibling_size(int Cin, long* packed_floats);
int atvs_conv_xw_pack_sibling(const float* w2, int Cin, float* packed);
int atvs_conv_xw_f32(const float* x, const float* packed_w, const float* bias, const float* plane_bias, float* y,
                     double* stats_partial, int groups, int D, int H, int W, int Cin, int ldy, int y_coff, int relu,
                     const float* packed_w2, const float* plane_bias2, float* y2, double* stats_partial2, int ldy2,
                     int y_coff2, const float* x2, const float* in_params, const float* in_params2, int in_relu,
                     int in_relu2, long x_planar, atvs_stream_t stream);

/* The same layers (same contract as atvs_conv_xw_f32, x_planar included) on the 16-bit matrix cores with SPLIT operands
 * (conv_xb.hip): every fp32 operand = TWO fp16 pieces, x = h0 + h1 / 2048 (h1 = fp16((x - h0) * 2048)), the three products
 * h0 g0 + (h0 g1 + h1 g0) / 2048 accumulated in fp32 by v_mfma_f32_16x16x32_f16 -- fp32-class results (22 significant bits per
 * operand; |x| or |w| beyond 65504 gives inf / NaN, the packers return ATVS_ERR_ARG for such weights), 9 K steps of 16-cycle
 * instructions per (8-channel chunk, kd, kh) row instead of 36 fp32 steps of 32 cycles.  Weights: atvs_conv_xb_pack /
 * _pack_sibling (HOST; sizes in BYTES).  Beyond atvs_conv_xw_f32: x_planar may come with in_params (a pending batch norm
 * over a chunk-planar input: the refinement's concat; not with x2); y_group_stride != 0 = floats between the samples of y
 * (>= D*H*W*ldy; 0 = dense): with ldy = 8 the output lands in one 8-channel plane of each sample's chunk-planar buffer.
 * x2 (two sources) needs Cin == 8; a prologue needs Cin <= 160 (its parameters live in LDS).
 * x_pieces != 0 (with x_planar, no prologue): the chunk planes hold the two fp16 pieces of every value, [2][D][H][W][8 fp16] as
 * atvs_warp_planes(pieces) writes them; the launch then only moves them into LDS (LDS-DMA), the same products follow. */
int atvs_conv_xb_pack_size(int Cin, long* packed_bytes);
int atvs_conv_xb_pack(const float* w, int Cin, unsigned char* packed);
int atvs_conv_xb_pack_sibling_size(int Cin, long* packed_bytes);
int atvs_conv_xb_pack_sibling(const float* w2, int Cin, unsigned char* packed);
int atvs_conv_xb_f32(const float* x, const unsigned char* packed_w, const float* bias, const float* plane_bias, float* y,
                     double* stats_partial, int groups, int D, int H, int W, int Cin, int ldy, int y_coff, int relu,
                     const unsigned char* packed_w2, const float* plane_bias2, float* y2, double* stats_partial2, int ldy2,
                     int y_coff2, const float* x2, const float* in_params, const float* in_params2, int in_relu,
                     int in_relu2, long x_planar, long y_group_stride, int x_pieces, atvs_stream_t stream);

/* 3x3 stride-1 SAME 2-D convolution (dilation 1, 2 or 4) of wide feature maps, LDS-tiled (conv2d_lds.hip): the
 * heavy layers of the feature towers -- the bottlenecks' conv2 (slim.conv2d, network.py:585-587), conv0_1 / conv0_2 /
 * fusion0 (tf.layers.conv2d, network.py:198-200; cnn_wrapper/atvsnet.py:254-292).  x (G,H,W,Cin): G independent
 * images (one tower call each in the reference), Cin % 16 == 0, Cout in {32, 64, 128} (dilation 2 / 4: Cout 128).
 *   atvs_conv2d_lds_supported   1 if (Cin, Cout, dilation) is served by this kernel
 *   atvs_conv2d_lds_pack_size / _pack   HOST: pack the TF kernel [3,3,Cin,Cout] (upload the result)
 *   atvs_conv2d_lds_rows        workgroups per image = rows per image of stats_partial ([2][Cout] doubles each)
 *   atvs_conv2d_lds_f32         y (G,H,W,ldy)[..., y_coff + co] = conv(x) (+ bias, + residual, ReLU); in_params
 *                               (G,3,Cin) != NULL: x is a raw convolution output and its batch norm (mean, rstd,
 *                               beta per image) [+ ReLU if in_relu] is applied while the tile is staged
 *                               (normalise-on-load; the SAME padding stays zero). */
int atvs_conv2d_lds_supported(int Cin, int Cout, int dilation);
int atvs_conv2d_lds_pack_size(int Cin, int Cout, long* packed_floats);
int atvs_conv2d_lds_pack(const float* w, int Cin, int Cout, float* packed);
long atvs_conv2d_lds_rows(int H, int W, int Cout);
int atvs_conv2d_lds_f32(const float* x, const float* packed_w, const float* bias, const float* residual,
                        const float* in_params, int in_relu, float* y, double* stats_partial, int G, int H, int W,
                        int Cin, int Cout, int dilation, int ldy, int y_coff, int relu, atvs_stream_t stream);

/* The same layers (same contract, shapes with Cin % 32 == 0, statistics rows = atvs_conv2d_lds_rows) with SPLIT operands
 * (conv2d_b.hip: every fp32 operand = two fp16 pieces, three products, fp32 accumulation on v_mfma_f32_16x16x32_f16; the
 * arithmetic of atvs_conv_c16b_f32).  Weights: atvs_conv2d_b_pack (HOST; size in BYTES; ATVS_ERR_ARG for a weight beyond
 * fp16's range). */
int atvs_conv2d_b_pack_size(int Cin, int Cout, long* packed_bytes);
int atvs_conv2d_b_pack(const float* w, int Cin, int Cout, unsigned char* packed);
int atvs_conv2d_b_f32(const float* x, const unsigned char* packed_w, const float* bias, const float* residual,
                      const float* in_params, int in_relu, float* y, double* stats_partial, int G, int H, int W, int Cin,
                      int Cout, int dilation, int ldy, int y_coff, int relu, atvs_stream_t stream);

/* A residual unit's conv2 AND conv3 in one launch (reference cnn_wrapper/network.py:585-601; conv2d_b.hip, TAIL form):
 *   y = conv3_1x1(relu(conv2_3x3_dil(x) + b2)) + b3 + residual
 * x (G,H,W,C) = the unit's conv1 output; packed_w2 = atvs_conv2d_b_pack of [3][3][C][C], packed_w3 = atvs_conv1x1_b_pack of
 * [C][C]; residual (G,H,W,C) = the shortcut or NULL; stats_partial: moments of y (rows: atvs_conv2d_lds_rows) or NULL.  r2 crosses
 * the wavefronts through LDS as fp16 pieces.  For the 128-channel dilated units (atvs_conv2d_b_tail_supported: C = 128, dilation
 * 2 / 4), whose conv1 halo does not fit the fully fused unit (atvs_bottleneck_b_f32).  Bit for bit atvs_conv2d_b_f32 followed by
 * atvs_conv1x1_b_f32. */
int atvs_conv2d_b_tail_supported(int C, int dilation);
int atvs_conv2d_b_tail_f32(const float* x, const unsigned char* packed_w2, const float* b2, const unsigned char* packed_w3,
                           const float* b3, const float* residual, float* y, double* stats_partial, int G, int H, int W, int C,
                           int dilation, atvs_stream_t stream);

/* The strided conv2 of a residual unit's first block (reference cnn_wrapper/network.py:588-595: explicit symmetric padding 1 +
 * VALID, stride 2 -- taps centred on input pixel 2 i, quirk C17) on the split-operand kernel (conv2d_b.hip, STRIDE form):
 * x (G,H,W,Cin), H and W even -> y (G,H/2,W/2,Cout) (+ bias, ReLU).  Built for Cout = 64, Cin % 32 == 0 (conv1_x_0/conv2 of
 * ResNetDS2SPP); weights atvs_conv2d_b_pack; stats_partial: rows atvs_conv2d_lds_rows(H/2, W/2, Cout) or NULL. */
int atvs_conv2d_b_s2_supported(int Cin, int Cout);
int atvs_conv2d_b_s2_f32(const float* x, const unsigned char* packed_w, const float* bias, float* y, double* stats_partial, int G,
                         int H, int W, int Cin, int Cout, int relu, atvs_stream_t stream);

/* 1x1 convolution of feature maps (the bottlenecks' conv1 / conv3 / shortcut, fusion1: slim.conv2d 1x1,
 * network.py:573-601; cnn_wrapper/atvsnet.py:254-292) as a tall GEMM: weights staged once per workgroup in LDS, pixels
 * streamed once (conv1x1.hip).  x (groups, pixels, Cin), Cin % 16 == 0, Cin <= 128, Cout in {32, 64, 128}.
 *   atvs_conv1x1_pack_size / _pack   HOST: pack the TF kernel [1,1,Cin,Cout]
 *   atvs_conv1x1_rows                workgroups per image = rows per image of stats_partial ([2][Cout] doubles each)
 *   atvs_conv1x1_f32                 y (groups,pixels,ldy)[..., y_coff + co] = x W (+ bias, + residual, ReLU); in_params
 *                                    (groups,3,Cin) != NULL: batch norm (+ ReLU if in_relu) of x applied on load -- the
 *                                    bottleneck's pre-activation (network.py:570-571) is never materialised. */
int atvs_conv1x1_supported(int Cin, int Cout);
int atvs_conv1x1_pack_size(int Cin, int Cout, long* packed_floats);
int atvs_conv1x1_pack(const float* w, int Cin, int Cout, float* packed);
long atvs_conv1x1_rows(long pixels);
int atvs_conv1x1_f32(const float* x, const float* packed_w, const float* bias, const float* residual,
                     const float* in_params, int in_relu, float* y, double* stats_partial, int groups, long pixels,
                     int Cin, int Cout, int ldy, int y_coff, int relu, atvs_stream_t stream);

/* The same layers (network.py:573-601) with SPLIT fp16 operands on the 16-bit matrix cores (conv1x1_b.hip; the arithmetic of
 * atvs_conv_c16b_f32): Cin % 32 == 0, Cin <= 1024, Cout in {32, 64, 128}.  Same contract as atvs_conv1x1_f32 except the
 * packed weights (bytes of fp16 pieces) and the statistics rows (128 pixels per workgroup: atvs_conv1x1_b_rows). */
int atvs_conv1x1_b_supported(int Cin, int Cout);
int atvs_conv1x1_b_pack_size(int Cin, int Cout, long* packed_bytes);
int atvs_conv1x1_b_pack(const float* w, int Cin, int Cout, unsigned char* packed);
long atvs_conv1x1_b_rows(long pixels);
int atvs_conv1x1_b_f32(const float* x, const unsigned char* packed_w, const float* bias, const float* residual,
                       const float* in_params, int in_relu, float* y, double* stats_partial, int groups, long pixels,
                       int Cin, int Cout, int ldy, int y_coff, int relu, atvs_stream_t stream);

/* One launch per pre-activation residual unit (bottleneck_b.hip; replaces Network.bottleneck, reference
 * cnn_wrapper/network.py:552-602, for depth == depth_in, stride 1 -- the identity-shortcut units of res_block, :604-616):
 *   y = x + conv3_1x1(relu(conv2_3x3_dil(relu(conv1_1x1(relu(bn(x))) + b1)) + b2)) + b3
 * for G images (H,W,C) channel-last fp32.  The unit has ONE global reduction (the moments of x, in_params (G,3,C) from
 * atvs_bn_finalize); conv1 is evaluated for the tile and its dilation halo, r1 and r2 stay in LDS as fp16 pieces.  Split fp16
 * operands, the K order and packed weights of the unfused kernels: w1 / w3 = atvs_conv1x1_b_pack of [C][C], w2 =
 * atvs_conv2d_b_pack of [3][3][C][C]; y is bit for bit what atvs_conv1x1_b_f32 -> atvs_conv2d_b_f32 -> atvs_conv1x1_b_f32 give.
 * stats_partial: NULL or (G, atvs_bottleneck_b_rows(C, H, W), 2, C) doubles: per-workgroup moments of y (the next unit's batch
 * norm).  x and y may not alias.  Shapes: atvs_bottleneck_b_supported (C in {32, 64}, dilation 1). */
int atvs_bottleneck_b_supported(int C, int dilation);
long atvs_bottleneck_b_rows(int C, int H, int W);
int atvs_bottleneck_b_f32(const float* x, const float* in_params, const unsigned char* w1, const float* b1,
                          const unsigned char* w2, const float* b2, const unsigned char* w3, const float* b3, float* y,
                          double* stats_partial, int G, int H, int W, int C, int dilation, atvs_stream_t stream);

/* 3x3x3 SAME stride-1 convolutions with Cin % 16 == 0 and 32 / 64 output channels (conv_b*_2_1, conv_b*_3_1,
 * global_refine_3dconv{2,3}_1: cnn_wrapper/atvsnet.py StackedUNet / CostVolRefineNet, network.py:172-215) with SPLIT fp16
 * operands on the 16-bit matrix cores (conv3d_b.hip; the arithmetic of atvs_conv_c16b_f32).  x (groups,D,H,W,Cin) ->
 * y (groups,D,H,W,ldy)[..., y_coff : y_coff + Cout]; stats_partial: groups * atvs_conv_c16_grid rows of [2][Cout] doubles or NULL.
 * Weights: atvs_conv3d_b_pack (HOST; size in BYTES). */
int atvs_conv3d_b_supported(int Cin, int Cout);
int atvs_conv3d_b_pack_size(int Cin, int Cout, long* packed_bytes);
int atvs_conv3d_b_pack(const float* w, int Cin, int Cout, unsigned char* packed);
int atvs_conv3d_b_f32(const float* x, const unsigned char* packed_w, const float* bias, float* y, double* stats_partial,
                      int groups, int D, int H, int W, int Cin, int Cout, int ldy, int y_coff, int relu, atvs_stream_t stream);
/* The same convolution of relu?((x - mean) * scale + beta): x a raw convolution output, in_params (groups,3,Cin) its pending
 * batch norm (conv_b*_{2,3}_1 read conv_b*_{2,3}_0, cnn_wrapper/atvsnet.py:20-26), applied per staged halo voxel -- the
 * normalised tensor is never written.  Bit for bit atvs_bn_apply followed by atvs_conv3d_b_f32. */
int atvs_conv3d_b_norm_f32(const float* x, const float* in_params, int in_relu, const unsigned char* packed_w, const float* bias,
                           float* y, double* stats_partial, int groups, int D, int H, int W, int Cin, int Cout, int ldy,
                           int y_coff, int relu, atvs_stream_t stream);

/* The STRIDE-2 form (conv_b*_2_0: 16 -> 32, conv_b*_3_0: 32 -> 64, global_refine_3dconv{2,3}_0; network.py:172-215) with split
 * fp16 operands (conv3d_s2b.hip): x (groups,D,H,W,Cin) -> y (groups,Do,Ho,Wo,ldy)[..., y_coff : y_coff + Cout], Do = ceil(D / 2) ...,
 * TF SAME padding (pad_before = (2 (Do - 1) + 3 - D) / 2 per axis).  stats_partial: groups * atvs_conv3d_s2b_grid(Do,Ho,Wo,groups)
 * rows of [2][Cout] doubles or NULL.  Weights: atvs_conv3d_s2b_pack (HOST; size in BYTES). */
int atvs_conv3d_s2b_supported(int Cin, int Cout);
long atvs_conv3d_s2b_grid(int Do, int Ho, int Wo, int groups);
int atvs_conv3d_s2b_pack_size(int Cin, int Cout, long* packed_bytes);
int atvs_conv3d_s2b_pack(const float* w, int Cin, int Cout, unsigned char* packed);
int atvs_conv3d_s2b_f32(const float* x, const unsigned char* packed_w, const float* bias, float* y, double* stats_partial,
                        int groups, int D, int H, int W, int Cin, int Cout, int ldy, int y_coff, int relu, atvs_stream_t stream);
/* ... of relu?((x - mean) * scale + beta), in_params (groups,3,Cin) the pending batch norm of the raw convolution output x (the
 * encoders conv_b*_{2,3}_0 read conv_b*_{1,2}_0, cnn_wrapper/atvsnet.py:10-12).  Bit for bit atvs_bn_apply followed by
 * atvs_conv3d_s2b_f32. */
int atvs_conv3d_s2b_norm_f32(const float* x, const float* in_params, int in_relu, const unsigned char* packed_w,
                             const float* bias, float* y, double* stats_partial, int groups, int D, int H, int W, int Cin,
                             int Cout, int ldy, int y_coff, int relu, atvs_stream_t stream);

/* atvs_deconv_up_f32's layers and contract with SPLIT operands on the 16-bit matrix cores (deconv_up_b.hip; the arithmetic of
 * atvs_conv_c16b_f32: two fp16 pieces per operand, three products, the cross terms in an accumulator of their own).  Grid /
 * statistics rows = atvs_deconv_up_b_grid (two workgroups per CU; NOT atvs_deconv_up_grid).  The packed weights of all chunks
 * stay in LDS beside two piece images where they fit; Cout 16 with more input channels re-reads one chunk's weights per stage
 * (atvs_deconv_up_b_supported).  stats_ld / stats_coff: a statistics row is [2][stats_ld] doubles and this
 * launch's channels start at column stats_coff (16, 0 = atvs_deconv_up_f32's layout): a 32-channel layer (conv_b*_4_0) runs as two
 * 16-channel launches with y_coff / stats_coff 0 and 16, ldy = stats_ld = 32.  Weights: atvs_deconv_up_b_pack (HOST; size in BYTES;
 * ATVS_ERR_ARG for a weight beyond fp16's range). */
long atvs_deconv_up_b_grid(int D, int H, int W, int Cout, int groups);
int atvs_deconv_up_b_supported(int Cin, int Cout);
int atvs_deconv_up_b_pack_size(int Cin, int Cout, long* packed_bytes);
int atvs_deconv_up_b_pack(const float* w, int Cin, int Cout, unsigned char* packed);
int atvs_deconv_up_b_f32(const float* x, const unsigned char* packed_w, float* y, double* stats_partial, int groups, int D, int H,
                         int W, int Cin, int Cout, int ldy, int y_coff, int relu, int stats_ld, int stats_coff,
                         atvs_stream_t stream);

/* atvs_deconv_up_b_f32 over the SUM of two or three volumes that is never written (the U-Net's skip adds in front of
 * conv_b*_6_0 / global_refine_3dconv6_0, reference cnn_wrapper/atvsnet.py:156-158,186-188,332-334):
 *   x_in = t(x0, params0, bit 0) + t(x1, params1, bit 1) [+ t(x2, params2, bit 2)],
 *   t(v, par, relu) = par ? relu?((v - mean) * scale + beta) : v                  (relu = that bit of relu_mask)
 * -- atvs_bn_add's arithmetic and order, formed per staged halo voxel; params_i (groups,3,Cin) or NULL (a finished tensor),
 * x2 NULL: two terms.  Bit for bit atvs_bn_add followed by atvs_deconv_up_b_f32.  Shapes: atvs_deconv_up_b_sum_supported
 * (Cin = 16, Cout = 8: the full-resolution decoder; one workgroup of eight wavefronts per CU, four multiply and store, four stage
 * the next tile's sources). */
int atvs_deconv_up_b_sum_supported(int Cin, int Cout);
int atvs_deconv_up_b_sum_f32(const float* x0, const float* params0, const float* x1, const float* params1, const float* x2,
                             const float* params2, int relu_mask, const unsigned char* packed_w, float* y, double* stats_partial,
                             int groups, int D, int H, int W, int Cin, int Cout, int ldy, int y_coff, int relu, int stats_ld,
                             int stats_coff, atvs_stream_t stream);
/* conv_bn(3, 8, 1) on a volume with ONE or TWO channels: the probability / visual-hull / geometric stems of the
 * refinement network (cnn_wrapper/atvsnet.py:300-311).  HBM-bound (432 FLOP per 36 B at one channel): FMA kernel with a
 * sliding register window along z, not MFMA (conv_stem.hip).  x (groups,D,H,W,Cin), Cin in {1,2}; w = the TF kernel
 * [3,3,3,Cin,8] on the device; y (groups,D,H,W,ldy)[..., y_coff + co]; plane_bias (groups,H,W,24) or NULL (see
 * atvs_conv_mfma_f32); stats_partial: groups * atvs_conv_stem_rows rows of [2][16] doubles (columns 0..7) or NULL. */
long atvs_conv_stem_rows(int D, int H, int W);
int atvs_conv_stem_f32(const float* x, const float* w, const float* plane_bias, float* y, double* stats_partial,
                       int groups, int D, int H, int W, int Cin, int ldy, int y_coff, int relu, atvs_stream_t stream);

/* The geo | prob | vishull stems of CostVolRefineNet (cnn_wrapper/atvsnet.py:300-313) in ONE pass, stored together with
 * the RAW output of the photo stem as whole 128-byte rows of the 32-channel concat buffer (stem-by-stem slices are
 * partial-line HBM writes): y (groups,D,H,W,32) = [photo_raw (..,8) | conv(geo (..,2)) + geo_plane_bias (groups,H,W,24) |
 * conv(prob (..,1)) | conv(hull (..,1))], no activation.  atvs_refine_stems_pack (HOST) arranges the three TF kernels
 * [3,3,3,Cin,8] as [27][geo0, geo1, prob, hull][8] (upload the 864 floats).  stats_partial: groups * atvs_conv_stem_rows
 * rows of [2][24] doubles over the 24 computed channels, or NULL.
 * y_planar != 0: y is chunk-planar instead, (groups, 4, y_planar) floats with planes of [D][H][W][8] (the layout
 * atvs_conv_xb_f32 reads with x_planar): planes 1..3 = geo | prob | hull are written, plane 0 is the photo stem's own
 * output (atvs_conv_xb_f32 with ldy = 8 and y_group_stride = 4 * y_planar) and photo_raw is not read (may be NULL):
 * 16 B read + 96 B written per voxel instead of 48 + 128. */
int atvs_refine_stems_pack(const float* w_geo, const float* w_prob, const float* w_hull, float* packed);
int atvs_refine_stems_f32(const float* photo_raw, const float* geo, const float* geo_plane_bias, const float* prob,
                          const float* hull, const float* w, float* y, double* stats_partial, int groups, int D, int H,
                          int W, long y_planar, atvs_stream_t stream);

/* conv(3, 1, 1, relu=False) on an 8-channel volume: the probability heads conv_b2_6_2,
 * attention_prob_vol[_refine], global_refined_cost_vol (cnn_wrapper/atvsnet.py:192,213,220,226,
 * 242,336).  x (D,H,W,8); w = the TF kernel [3,3,3,8,1] (216 floats, device); y (D,H,W).
 * One output channel: packed-FMA kernel with four outputs per thread, not MFMA (conv8to1.hip). */
int atvs_conv3d_8to1(const float* x, const float* w, float* y, int groups, int D, int H, int W, atvs_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Batch norm with batch statistics, element-wise glue  (cnn_wrapper/network.py)
 * ------------------------------------------------------------------------- */

/* `groups` in this section: the tensor is `groups` INDEPENDENT samples stacked along its leading axis, each with its
 * own batch statistics -- the reference evaluates every network call with batch 1 and per-call statistics (quirk C1);
 * stacking the calls of a depth map (views, siamese directions) into one launch must not change them. */

/* Reduce per-workgroup partial sums [groups][num_blocks][2][cpad] (double) to
 * params [groups][3][C] = (mean, rsqrt(var+eps), beta): the moments of
 * tf.layers.batch_normalization(training=True) / slim.batch_norm, network.py:206-212,
 * 541-547, 570-571.  count = elements per channel of one sample.  beta (C, shared) or NULL.  fold >= 1: channel c
 * also sums columns c + C, c + 2C, ... (fold of them).  nonfinite_flag: NULL or a device word that is OR-ed with 1 when a
 * moment is not finite -- the layers upstream carried a value beyond the fp16 range of the split-operand kernels (or a
 * non-finite input); sticky, the host reads and clears it (a later ReLU can swallow the NaN, the flag keeps it). */
int atvs_bn_finalize(const double* stats_partial, int groups, long num_blocks, int cpad, int fold, long count,
                     const float* beta, float eps, float* params, int C, int* nonfinite_flag, atvs_stream_t stream);

/* Partial sums of an arbitrary (groups, rows, C) tensor, C <= 256, in the layout above with
 * cpad = C and atvs_channel_stats_num_blocks(rows) blocks per sample. */
long atvs_channel_stats_num_blocks(long rows);
int atvs_channel_stats(const float* x, int groups, long rows, int C, double* stats_partial, atvs_stream_t stream);

/* y = (x - mean) * rstd + beta [, relu]; y may alias x.  x and y are groups * rows rows of width ld of which the C
 * channels starting at c_off are touched (ld = C, c_off = 0 for a dense tensor); params (groups, 3, C).
 * Arithmetic (here and wherever an entry point of this header applies a pending batch norm while it loads -- the in_params /
 * params_i arguments): ONE fused multiply-add per value, fma(x, rstd, beta - mean * rstd), tf.nn.batch_normalization's own form
 * (x * inv + (offset - mean * inv)); every site uses the same two helpers, so a fused form equals the passes it replaces bit for bit. */
int atvs_bn_apply(const float* x, const float* params, float* y, int groups, long rows, int C, int ld, int c_off,
                  int relu, atvs_stream_t stream);

/* tf.add_n of two or three tensors of which any may still be a raw convolution output whose
 * batch norm (+ ReLU, relu_mask bit i) is pending: y = sum_i bn_relu_i(x_i), params_i (groups,3,C) or NULL
 * for an already-final tensor; x2 may be NULL.  C % 4 == 0; rows per sample.  (network.py:172-215 + :695-697 fused.) */
int atvs_bn_add(const float* x0, const float* params0, const float* x1, const float* params1, const float* x2,
                const float* params2, float* y, int groups, long rows, int C, int relu_mask, atvs_stream_t stream);
/* atvs_bn_add that also writes y2 = base + y, base (rows,C) ONE sample shared by the `groups` samples (global_refine_3dconv6_1 and
 * refined_cost = filtered_cost + cost_residual of every source view, model.py:438, in one pass): y / y2 bit for bit atvs_bn_add /
 * atvs_add_n(base, y). */
int atvs_bn_add_plus(const float* x0, const float* params0, const float* x1, const float* params1, const float* x2,
                     const float* params2, float* y, const float* base, float* y2, int groups, long rows, int C, int relu_mask,
                     atvs_stream_t stream);

/* tf.add_n of two or three tensors (c may be NULL), network.py:695-697. */
int atvs_add_n(const float* a, const float* b, const float* c, float* y, long n, atvs_stream_t stream);

/* tf.layers.average_pooling2d(padding='SAME'), network.py:665-671: x (H,W,C) ->
 * y (ceil(H/stride), ceil(W/stride), C), mean over the valid window elements; `groups` images per launch.
 * ws: groups * atvs_avg_pool_ws_floats(H, W, C, stride) floats of scratch. */
long atvs_avg_pool_ws_floats(int H, int W, int C, int stride);
int atvs_avg_pool_same(const float* x, float* y, float* ws, int groups, int H, int W, int C, int pool, int stride,
                       atvs_stream_t stream);

/* tf.image.resize_images(BILINEAR, align_corners=True), network.py:649-655:
 * x (H,W,C) -> y (Ho,Wo,ld_out)[..., c_off:c_off+C]. */
int atvs_resize_bilinear(const float* x, float* y, int groups, int H, int W, int C, int Ho, int Wo, int ld_out,
                         int c_off, atvs_stream_t stream);

/* tf.concat along channels / un-stacking a trailing axis, network.py:691-693:
 * dst[r, dst_off + c] = src[r, src_off + c], c < C. */
int atvs_copy_channels(const float* src, float* dst, long rows, int C, int ld_src, int src_off, int ld_dst,
                       int dst_off, atvs_stream_t stream);

/* tf.stack along a new leading axis (model.py:186,289: the per-pair reference features of a batch of cost volumes, the two
 * initial depth maps of the refinement): dst[i] = *srcs[i], i < n <= 16, each `elems` floats (a multiple of 4, 16-byte aligned).
 * srcs: HOST array of n device pointers. */
int atvs_stack(const float* const* srcs, int n, long elems, float* dst, atvs_stream_t stream);

/* ------------------------------------------------------------------------- *
 * AANet aggregation over views  (cnn_wrapper/network.py:282-351, 378-408)
 * ------------------------------------------------------------------------- */

/* sr_ptrs / x_ptrs: HOST arrays of nv (<= 16) device pointers.  SR_n (V,16) =
 * [relu(conv(X_n, W_shared)) | relu(conv(X_n, W_unique))], X_n (V,8).
 * out (V,8) = sum_n softmax_n(R_n - S_n + sum_m S_m) * X_n. */
int atvs_aanet_combine(const float* const* sr_ptrs, const float* const* x_ptrs, int nv, float* out, long V,
                       atvs_stream_t stream);

/* The whole AANet module in ONE launch (aanet_b.hip; round 6: eight wavefronts in two roles -- four multiply, four stage the next
 * halo and run the softmax of the previous tile beside them): the shared | unique 3x3x3 score convolutions of every view (split
 * fp16 operands, conv_c16b's K order), their results handed over in LDS, then the cross-view softmax and weighted sum --
 *   out (D,H,W,8) = sum_n softmax_n((R_n - S_n) + sum_m S_m) * X_n,   S_n | R_n = relu(conv3d(X_n, W_shared | W_unique, SAME))
 * (reference cnn_wrapper/network.py:282-351,378-408).  x: HOST array of nv <= 8 device pointers (D,H,W,8) (atvs_aanet_b_supported; more views: the two-launch form); packed_w:
 * atvs_aanet_b_pack(w_shared, w_unique) (HOST; [3,3,3,8,8] each; size in BYTES).  [S|R] is never written: bit for bit
 * atvs_conv_c16b_f32 (ReLU) per view followed by atvs_aanet_combine. */
int atvs_aanet_b_supported(int C, int nv);
int atvs_aanet_b_pack_size(long* packed_bytes);
int atvs_aanet_b_pack(const float* w_shared, const float* w_unique, unsigned char* packed);
int atvs_aanet_b_f32(const float* const* x, int nv, const unsigned char* packed_w, float* out, int D, int H, int W,
                     atvs_stream_t stream);

/* The same arithmetic split at its three reductions over views, for views sharded
 * across GPUs (all-reduce SUM / MAX / SUM between stages):
 *   stage 0: out (V,8) = sum_local S_n
 *   stage 1: out (V,8) = max_local U_n                (needs ssum)
 *   stage 2: out (2,V,8) = [sum e_n ; sum e_n * X_n]   (needs ssum, umax, x_ptrs) */
int atvs_aanet_partial(const float* const* sr_ptrs, const float* const* x_ptrs, int nv, int stage,
                       const float* ssum, const float* umax, float* out, long V, atvs_stream_t stream);

/* out = num / den, n % 4 == 0 (last step of the sharded AANet). */
int atvs_divide(const float* num, const float* den, float* out, long n, atvs_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Depth-map fusion  (fusibile/fusibile.cu, after the hot path: SURVEY.md 8 f-4)
 * ------------------------------------------------------------------------- */

/* The consistency-voting kernel `fusibile` (fusibile/fusibile.cu:138-277) for reference camera `ref`.
 * cams (nviews, 28) floats per camera: P[12] (3x4 projection, row-major) | M_inv[9] (inverse of P's left 3x3) |
 * C[3] (camera centre) | P_col34[3] (P's last column) | f (K[0,0]) -- cameraGeometryUtils.h:377-433.
 * normals_depths, images: (nviews, rows, cols, 4) floats = (nx, ny, nz, depth) and (b, g, r, unused), the two float4
 * textures of main.cpp:816-822, sampled bilinearly with clamped addressing and 8-bit weights (no texture unit).
 * Outputs, all for the pixels of `ref`: coord / normal / texture (rows, cols, 4) = the pixel's 3-D point, the normal and
 * the colour averaged over the agreeing views; created (rows, cols) = 1 when at least num_consistent views agree
 * (relative disparity difference < disp_thresh and normal angle < normal_thresh), else 0. */
int atvs_fusibile(const float* cams, const float* normals_depths, const float* images, int nviews, int ref, int rows,
                  int cols, float disp_thresh, float normal_thresh, int num_consistent, float* coord, float* normal,
                  float* texture, float* created, atvs_stream_t stream);


/* ---- 3x3x3 stride-2 SAME transposed convolution to 8 or 16 channels, all 8 output parity classes per staged input tile
 * (csrc/deconv_up.hip): the decoders conv_b*_6_0 / conv_b*_5_0, global_refine_3dconv6_0 / 5_0 (cnn_wrapper/atvsnet.py
 * StackedUNet / CostVolRefineNet; tf.layers.conv3d_transpose, cnn_wrapper/network.py:510-550).  Cout in {8, 16},
 * Cin % 16 == 0, Cin <= 64, output below 4 GiB per sample (else ATVS_ERR_SHAPE: callers use the class-fused form of
 * atvs_conv_tiled_f32).
 * w: TF layout [3,3,3,Cout,Cin].  x (groups, D,H,W, Cin) -> y (groups, 2D,2H,2W, ldy)[..., y_coff : y_coff + Cout].
 * stats_partial: groups * atvs_deconv_up_grid rows of [2][16] doubles (sum / sum of squares per channel) or NULL. */
int atvs_deconv_up_pack_size(int Cin, int Cout, long* packed_floats);
int atvs_deconv_up_pack(const float* w, int Cin, int Cout, float* packed);        /* host function */
long atvs_deconv_up_grid(int D, int H, int W, int Cout, int groups);              /* workgroups PER SAMPLE */
int atvs_deconv_up_f32(const float* x, const float* packed_w, float* y, double* stats_partial, int groups, int D, int H,
                       int W, int Cin, int Cout, int ldy, int y_coff, int relu, atvs_stream_t stream);

/* ---- 3x3x3 SAME stride-1 convolution to 16 channels (Cin in {8, 16, 32}) or 32 channels (Cin in {16, 32, 48, 64}), one
 * workgroup per CU (csrc/conv_c16.hip): the half- / quarter-resolution layers conv_b*_1_1, conv_b*_2_1,
 * global_refine_3dconv1_1 / 2_1 (cnn_wrapper/atvsnet.py StackedUNet / CostVolRefineNet; cnn_wrapper/network.py:165-215)
 * and the AANet modules' shared | unique convolution (network.py:282-351, 8 -> 8 + 8).  Other shapes: ATVS_ERR_SHAPE
 * (callers use atvs_conv_tiled_f32).
 * w: TF layout [3,3,3,Cin,Cout].  x (groups, D,H,W, Cin) -> y (groups, D,H,W, ldy)[..., y_coff : y_coff + Cout] (+ bias,
 * ReLU).  stats_partial: groups * atvs_conv_c16_grid rows of [2][Cout] doubles (sum / sum of squares per channel) or NULL. */
int atvs_conv_c16_pack_size(int Cin, int Cout, long* packed_floats);
int atvs_conv_c16_pack(const float* w, int Cin, int Cout, float* packed);         /* host function */
long atvs_conv_c16_grid(int D, int H, int W, int groups);                         /* workgroups PER SAMPLE */
int atvs_conv_c16_f32(const float* x, const float* packed_w, const float* bias, float* y, double* stats_partial,
                      int groups, int D, int H, int W, int Cin, int Cout, int ldy, int y_coff, int relu,
                      atvs_stream_t stream);

/* The 8 / 16 -> 16 channel forms of atvs_conv_c16_f32 on the 16-bit matrix cores with SPLIT operands (conv_c16b.hip;
 * BASELINE.json configs[1] names "bf16 conv3d MFMA"): x = h0 + h1 / 2048 and w = g0 + g1 / 2048 in fp16 (h0 = fp16(x),
 * h1 = fp16((x - h0) * 2048): 22 significant bits, the residual piece scaled into fp16's normal range), the three products
 * h0 g0 + (h0 g1 + h1 g0) / 2048 accumulated in fp32 by v_mfma_f32_16x16x32_f16, the cross terms in an accumulator of their
 * own that is scaled once -- fp32-class results (per-layer error against float64 below the fp32 MFMA form's; rounding
 * differs from it).  |x| or |w| beyond 65504 gives inf / NaN; the packers return ATVS_ERR_ARG for such weights.  Same grid /
 * statistics rows as atvs_conv_c16_f32.
 *   atvs_conv_c16b_pack_size / _pack   HOST: split and pack the TF kernel [3,3,3,Cin,16], Cin 8 or 16 (bytes; upload the result) */
int atvs_conv_c16b_pack_size(int Cin, long* packed_bytes);
int atvs_conv_c16b_pack(const float* w, int Cin, unsigned char* packed);
int atvs_conv_c16b_f32(const float* x, const unsigned char* packed_w, const float* bias, float* y, double* stats_partial,
                       int groups, int D, int H, int W, int Cin, int ldy, int y_coff, int relu, atvs_stream_t stream);
/* The 16 -> 16 form of an input that is never written: x_in = t(x0, params0, bit 0) [+ t(x1, params1, bit 1)] with
 * t(v, par, relu) = par ? relu?((v - mean) * scale + beta) : v -- one term: a pending batch norm applied while the halo is staged
 * (conv_b0_1_1 reads conv_b0_1_0; atvs_bn_apply's arithmetic); two terms: the U-Net's skip sum (conv_b{1,2}_1_1_concat,
 * cnn_wrapper/atvsnet.py:45-46,75-76,140-141,170-171; atvs_bn_add's arithmetic and order).  params_i (groups,3,16) or NULL (that
 * term is a finished tensor); x1 NULL: one term (params0 required).  relu_mask bit i: ReLU after term i's batch norm.  Bit for
 * bit atvs_bn_apply / atvs_bn_add followed by atvs_conv_c16b_f32. */
int atvs_conv_c16b_sum_supported(int Cin);
int atvs_conv_c16b_sum_f32(const float* x0, const float* params0, const float* x1, const float* params1, int relu_mask,
                           const unsigned char* packed_w, const float* bias, float* y, double* stats_partial, int groups, int D,
                           int H, int W, int Cin, int ldy, int y_coff, int relu, atvs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ATVSNET_HIP_H */
