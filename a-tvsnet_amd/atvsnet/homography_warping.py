"""Plane-sweep geometry of the reference's ``atvsnet/homography_warping.py`` on the HIP kernels.

Same function names, argument order and tensor layouts (batch-first, channel-last,
B = 1) as /root/reference/atvsnet/homography_warping.py; each function is one or a few
launches through ``ops`` (include/atvsnet_hip.h).  ``FLAGS.inverse_depth`` is read
where the reference reads it (:149,215,301,321,369,378).

``get_pixel_grids`` and ``interpolate`` (reference :8-17, :31-104) are exported as functions too, although the warps
have them folded into their kernels: a caller that brings its own sampling coordinates binds to ``interpolate``.

Beyond the reference API, ``homography_warping`` accepts a whole (B,D,3,3) stack of
homographies and returns the (B,D,H,W,C) stack of warps in one launch -- the form
model.py's D-unrolled loops need.
"""
import torch

from .. import ops
from ..flags import FLAGS


def _cam(c):
    """(B,2,4,4) -> contiguous (2,4,4)."""
    if c.shape[0] != 1:
        raise ValueError('batch size must be 1 (FLAGS.batch_size)')
    return c[0].contiguous()


def _scalar(t):
    return t.reshape(-1)[:1].contiguous()


def get_pixel_grids(height, width, device=None):
    """Pixel-centre texture coordinates (reference :8-17): the flat (3*H*W,) tensor [x + 0.5 | y + 0.5 | 1].
    `device` (not in the reference, whose graph has one device): defaults to the current HIP device."""
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    return ops.pixel_grids(torch.empty(0, device=dev), int(height), int(width))


def interpolate(image, x, y, output_mask=False, method='bilinear'):
    """Sample image (B,H,W,C) at flat texture coordinates x, y (B*H*W,) -> (B*H*W, C) [, bool mask (B*H*W,)]
    (reference :31-104).  bilinear: points outside [0, W-1) x [0, H-1) (after the -0.5 shift) or NaN give 0; nearest:
    tf.round, such points read pixel (0,0) un-masked (quirk C4).  B = 1 like every caller on the path."""
    if method not in ('bilinear', 'nearest'):
        raise ValueError('interpolate: unknown method %r' % (method,))
    if image.shape[0] != 1:
        raise ValueError('batch size must be 1 (FLAGS.batch_size)')
    res = ops.interpolate(image[0].contiguous(), x.reshape(-1).contiguous(), y.reshape(-1).contiguous(),
                          method=method, want_mask=output_mask)
    if output_mask:
        return res[0], res[1] > 0
    return res


def get_homographies(left_cam, right_cam, depth_num, depth_start, depth_interval):
    """(B,2,4,4) x2 -> (B,D,3,3) (reference :179-227)."""
    H = ops.get_homographies(_cam(left_cam), _cam(right_cam), _scalar(depth_start), _scalar(depth_interval),
                             int(depth_num), FLAGS.inverse_depth)
    return H.unsqueeze(0)


def homography_warping(input_image, homography, method='bilinear', output_mask=False):
    """Warp (B,H,W,C) by (B,3,3) -> (B,H,W,C) [, bool mask (B,H,W,1)] (reference :230-271).

    With a (B,D,3,3) stack: -> (B,D,H,W,C) [, mask (B,D,H,W,1)].  method 'nearest' (reference :45-56):
    tf.round, out-of-range pixels read pixel (0,0) un-masked (quirk C4).  (get_visual_hull has its own fused
    kernel and does not come through here.)
    """
    if method not in ('bilinear', 'nearest'):
        raise ValueError('homography_warping: unknown method %r' % (method,))
    stack = homography.dim() == 4
    Hm = (homography[0] if stack else homography).contiguous()
    res = ops.warp_planes(input_image[0].contiguous(), Hm, want_mask=output_mask,
                          mode=ops.WARP_NEAREST if method == 'nearest' else 0)
    out, mask = res if output_mask else (res, None)
    if not stack:
        out = out[0]
        mask = mask[0] if mask is not None else None
    if output_mask:
        return out.unsqueeze(0), (mask.unsqueeze(0).unsqueeze(-1) > 0)
    return out.unsqueeze(0)


def homography_warping_by_depth(input_image, left_cam, right_cam, depth_image, output_mask=False, method='bilinear'):
    """Per-pixel-depth warp (reference :108-176): (B,H,W,C), depth (B,H,W,1) -> (B,H,W,C) [, mask (B,H,W,1)]."""
    h, w = input_image.shape[1:3]
    out, mask = ops.warp_by_depth(input_image[0].contiguous(), _cam(left_cam), _cam(right_cam),
                                  depth_image.reshape(h, w).contiguous(), method, FLAGS.inverse_depth)
    if output_mask:
        return out.unsqueeze(0), (mask.reshape(1, h, w, 1) > 0)
    return out.unsqueeze(0)


def transform_depth(left_depth, left_cam, right_cam):
    """Express a view's (inverse-)depth map in another camera's frame (reference :275-326).
    (B,H,W[,1]) -> same shape."""
    shape = left_depth.shape
    h, w = shape[1], shape[2]
    out = ops.transform_depth(left_depth.reshape(h, w).contiguous(), _cam(left_cam), _cam(right_cam),
                              FLAGS.inverse_depth)
    return out.reshape(shape)


def transform_depth_batch(jobs):
    """transform_depth(left_depth, left_cam, right_cam) for every job of `jobs`, in one launch where the maps allow it
    (ops.transform_depth_batch); the maps keep their shapes."""
    flat = []
    for left_depth, left_cam, right_cam in jobs:
        h, w = left_depth.shape[1], left_depth.shape[2]
        flat.append((left_depth.reshape(h, w).contiguous(), _cam(left_cam), _cam(right_cam)))
    outs = ops.transform_depth_batch(flat, FLAGS.inverse_depth)
    return [o.reshape(job[0].shape) for o, job in zip(outs, jobs)]


def get_visual_hull(depth_images, cams, depth_num, depth_start, depth_interval, ref_id=0, view_num=None):
    """(B,N,H,W) depths -> (B,D,H,W,1) (reference :329-387).  Only view_num == 2 (every call site:
    model.py:323-324 with num_depths=2) is built.  Quirk C6: the second map is paired with
    cams[:, id_reorder[1]], whatever the current source view is."""
    if view_num is None:
        view_num = FLAGS.view_num
    if view_num != 2:
        raise NotImplementedError('get_visual_hull: view_num=2 is the only form the path uses')
    ids = list(range(view_num))
    ids[0] = ref_id
    ids[ref_id] = 0
    vi = ids[1]
    ref_cam, view_cam = cams[:, ref_id], cams[:, vi]
    H = get_homographies(ref_cam, view_cam, depth_num, depth_start, depth_interval)
    h, w = depth_images.shape[2:4]
    vtrans = transform_depth(depth_images[:, vi], view_cam, ref_cam)
    hull = ops.visual_hull(depth_images[0, ref_id].contiguous(), vtrans.reshape(h, w), H[0], _scalar(depth_start),
                           _scalar(depth_interval), FLAGS.inverse_depth)
    return hull.reshape(1, depth_num, h, w, 1)
