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

/* ABI version of this header (bumped on any signature change). */
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
 *           channels (geo view volume, reference quirk: 16 identical channels). */
int atvs_warp_planes(const float* src, const float* homographies, const float* ref,
                     const float* depth_start, const float* depth_interval, float* out, float* mask_out,
                     int D, int h, int w, int C, int ld_out, int c_off, int mode, int rep,
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

/* transform_depth, homography_warping.py:275-326.  ws14: 14 floats of scratch. */
int atvs_transform_depth(const float* depth, const float* left_cam, const float* right_cam, float* out,
                         float* ws14, int h, int w, int inverse_depth, atvs_stream_t stream);

/* tf.abs(a - b) * tile(mask): photo_err / geo_err, model.py:310,315.
 * a, b, out (npix, C); mask (npix). */
int atvs_absdiff_mask(const float* a, const float* b, const float* mask, float* out, int npix, int C,
                      atvs_stream_t stream);

/* ------------------------------------------------------------------------- *
 * Soft-argmin  (atvsnet/model.py)
 * ------------------------------------------------------------------------- */

/* prob2depth, model.py:80-109: cost (D,h,w) -> depth (h,w). */
int atvs_softargmin(const float* cost, const float* depth_start, const float* depth_interval,
                    float* depth_out, int D, int h, int w, atvs_stream_t stream);

/* upsample_prob_vol + prob2depth, model.py:68-76,121-127: cost (D,h,w) ->
 * depth (h*up, w*up) without materialising the upsampled volume. */
int atvs_upsample_softargmin(const float* cost, const float* depth_start, const float* depth_interval,
                             float* depth_up_out, int D, int h, int w, int up_scale, atvs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ATVSNET_HIP_H */
