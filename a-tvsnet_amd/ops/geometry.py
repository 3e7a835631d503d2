"""Geometry and soft-argmin entry points: homographies, plane-sweep warps, cost volume, refinement volumes
(reference atvsnet/homography_warping.py, model.py:13-129,157-200,270-336).
"""

import ctypes

from .. import _lib
from .base import _Timed, _call, _dev_ok, _new, _p, _ptr_array, _stream
from .packing import planar_stride


def get_homographies(left_cam, right_cam, depth_start, depth_interval, depth_num, inverse_depth=True):
    """cams (2,4,4); depth_start/interval 1-element tensors -> (D,3,3)."""
    out = _new(left_cam, (depth_num, 3, 3))
    if _dev_ok(left_cam, right_cam, depth_start, depth_interval):
        _call('atvs_get_homographies', _p(left_cam), _p(right_cam), _p(depth_start), _p(depth_interval), _p(out),
              int(depth_num), int(bool(inverse_depth)), _stream())
    return out


WARP_NEAREST = 3      # atvs_warp_planes mode: nearest-neighbour sampling (include/atvsnet_hip.h)


def warp_planes(src, homographies, out=None, ld_out=None, c_off=0, mode=0, ref=None, depth_start=None,
                depth_interval=None, rep=1, want_mask=False, planar=False, pieces=False):
    """src (h,w,C), homographies (D,3,3) -> out (D,h,w,ld_out) [, mask (D,h,w)].
    planar=True (plain warp, C in {16,32,64}): out is chunk-planar, a PlanarVolume-shaped (C/8, plane_floats(D,h,w)) buffer
    whose rows hold (D,h,w,8) -- the layout the x-pair kernels read as dense 32-byte voxels (SplitVolume(planar=True)).
    pieces=True (with planar): every value is written as its two fp16 pieces (the operand split of the split-operand
    convolutions, done once by this producer); a row of `out` then holds [2 pieces][D][h][w][8 fp16] -- the same bytes, to be
    read by conv_xb only (SplitVolume(pieces=True); planar_pieces_decode() for anything else)."""
    h, w, C = src.shape
    D = homographies.shape[0]
    width = rep if mode == 2 else C
    if pieces and not planar:
        raise ValueError('warp_planes(pieces=True) needs planar=True')
    if planar:
        if mode not in (0, 1) or C not in (16, 32, 64) or c_off != 0:
            raise ValueError('warp_planes(planar=True): plain warp / photo volume of 16 / 32 / 64 channels')
        pstride = planar_stride(D, h, w)
        if out is None:
            out = _new(src, (C // 8, pstride))
        elif tuple(out.shape) != (C // 8, pstride) or not out.is_contiguous():
            raise ValueError('warp_planes(planar=True): out must be a contiguous (C/8, planar_stride(D,h,w)) tensor')
        ld_out = C
    elif out is None:
        ld_out = width if ld_out is None else ld_out
        out = _new(src, (D, h, w, ld_out))
    else:
        ld_out = out.shape[-1]
    mask = _new(src, (D, h, w)) if want_mask else None
    if _dev_ok(src, homographies, out, ref, depth_start, depth_interval):
        with _Timed(('warp', int(mode)), (D, h, w, C), width):
            _call('atvs_warp_planes', _p(src), _p(homographies), _p(ref), _p(depth_start), _p(depth_interval),
                  _p(out), _p(mask), D, h, w, C, int(ld_out), int(c_off), int(mode), int(rep),
                  ctypes.c_long(planar_stride(D, h, w) if planar else 0), int(bool(pieces)), _stream())
    return (out, mask) if want_mask else out


def build_cost_volume(ref_feature, view_feature, homographies):
    """(h,w,C) x2, (D,3,3) -> (D,h,w,2C)."""
    h, w, C = ref_feature.shape
    D = homographies.shape[0]
    out = _new(ref_feature, (D, h, w, 2 * C))
    if _dev_ok(ref_feature, view_feature, homographies):
        _call('atvs_build_cost_volume', _p(ref_feature), _p(view_feature), _p(homographies), _p(out), D, h, w, C,
              _stream())
    return out


def tile_planes(src, out, c_off):
    """src (h,w,C) broadcast along D into out (D,h,w,ld)[..., c_off:c_off+C]."""
    h, w, C = src.shape
    D, ld = out.shape[0], out.shape[-1]
    if _dev_ok(src, out):
        _call('atvs_tile_planes', _p(src), _p(out), D, h, w, C, ld, int(c_off), _stream())
    return out


def geo_ref_planes(depth_ref, depth_start, depth_interval, out, c_off):
    """depth_ref (h,w) -> out (D,h,w,ld)[..., c_off]."""
    h, w = depth_ref.shape[:2]
    D, ld = out.shape[0], out.shape[-1]
    if _dev_ok(depth_ref, depth_start, depth_interval, out):
        _call('atvs_geo_ref_planes', _p(depth_ref), _p(depth_start), _p(depth_interval), _p(out), D, h, w, ld,
              int(c_off), _stream())
    return out


def geo_volume(depth_ref, view_depth, homographies, depth_start, depth_interval, out, c_off=0, rep=1):
    """geo_ref_planes(depth_ref) into out[..., c_off] and warp_planes(view_depth, mode=2, rep) into out[..., c_off + 1 ..] as one
    launch (the refinement's geo volume); out (D,h,w,ld)."""
    h, w = depth_ref.shape[:2]
    D, ld = out.shape[0], out.shape[-1]
    if _dev_ok(depth_ref, view_depth, homographies, depth_start, depth_interval, out):
        _call('atvs_geo_volume', _p(depth_ref), _p(view_depth), _p(homographies), _p(depth_start), _p(depth_interval), _p(out),
              D, h, w, ld, int(c_off), int(rep), _stream())
    return out


def visual_hull(ref_depth, view_depth_in_ref, homographies, depth_start, depth_interval, inverse_depth=True, out=None):
    """(h,w) x2 -> (D,h,w) (written into `out` when given)."""
    h, w = ref_depth.shape[:2]
    D = homographies.shape[0]
    out = _new(ref_depth, (D, h, w)) if out is None else out
    if _dev_ok(ref_depth, view_depth_in_ref, homographies, depth_start, depth_interval):
        _call('atvs_visual_hull', _p(ref_depth), _p(view_depth_in_ref), _p(homographies), _p(depth_start),
              _p(depth_interval), _p(out), D, h, w, int(bool(inverse_depth)), _stream())
    return out


def warp_by_depth(src, left_cam, right_cam, depth, method='bilinear', inverse_depth=True):
    """src (h,w,C), depth (h,w) -> (warped (h,w,C), mask (h,w))."""
    h, w, C = src.shape
    out = _new(src, (h, w, C))
    mask = _new(src, (h, w))
    ws = _new(src, (12,))
    if _dev_ok(src, left_cam, right_cam, depth):
        _call('atvs_warp_by_depth', _p(src), _p(left_cam), _p(right_cam), _p(depth), _p(out), _p(mask), _p(ws), h, w,
              C, 1 if method == 'nearest' else 0, int(bool(inverse_depth)), _stream())
    return out, mask


def warp_by_depth_err(src, ref, left_cam, right_cam, depth, out, c_off=0, method='bilinear', inverse_depth=True, copy_ref=False):
    """|warp_by_depth(src) - ref| * mask written into out[..., c_off:c_off + C] (out: (h,w,ld)): the refinement's photo_err /
    geo_err (warp, absolute difference, mask and the copy into the tiled-channel buffer) as one launch; copy_ref: ref itself goes
    to the C channels behind."""
    h, w, C = src.shape
    if tuple(ref.shape) != (h, w, C) or tuple(out.shape[:2]) != (h, w):
        raise ValueError('warp_by_depth_err: src / ref (h,w,C), out (h,w,ld)')
    if _dev_ok(src, ref, left_cam, right_cam, depth, out):
        _call('atvs_warp_by_depth_err', _p(src), _p(ref), _p(left_cam), _p(right_cam), _p(depth), _p(out), int(out.shape[-1]),
              int(c_off), h, w, C, 1 if method == 'nearest' else 0, int(bool(inverse_depth)), int(bool(copy_ref)), _stream())
    return out


def interpolate(src, x, y, method='bilinear', want_mask=False):
    """src (h,w,C), x / y (n,) texture coordinates -> (n,C) [, mask (n,) of 1.f / 0.f] (atvs_interpolate)."""
    h, w, C = src.shape
    n = x.numel()
    if y.numel() != n:
        raise ValueError('interpolate: x and y must have the same number of points')
    if method not in ('bilinear', 'nearest'):
        raise ValueError('interpolate: unknown method %r' % (method,))
    out = _new(src, (n, C))
    mask = _new(src, (n,)) if want_mask else None
    if _dev_ok(src, x, y):
        _call('atvs_interpolate', _p(src), _p(x), _p(y), _p(out), _p(mask), ctypes.c_long(n), h, w, C,
              1 if method == 'nearest' else 0, _stream())
    return (out, mask) if want_mask else out


def pixel_grids(ref, height, width):
    """-> (3*height*width,) = [x + 0.5 | y + 0.5 | 1] on ref's device (atvs_pixel_grids)."""
    out = _new(ref, (3 * int(height) * int(width),))
    if _dev_ok(out):
        _call('atvs_pixel_grids', _p(out), int(height), int(width), _stream())
    return out


def transform_depth(depth, left_cam, right_cam, inverse_depth=True):
    """depth (h,w) in left_cam -> same pixels, values expressed in right_cam's frame."""
    h, w = depth.shape[:2]
    out = _new(depth, (h, w))
    ws = _new(depth, (14,))
    if _dev_ok(depth, left_cam, right_cam):
        _call('atvs_transform_depth', _p(depth), _p(left_cam), _p(right_cam), _p(out), _p(ws), h, w,
              int(bool(inverse_depth)), _stream())
    return out


def transform_depth_batch(jobs, inverse_depth=True):
    """transform_depth of several maps of one size: jobs = [(depth (h,w), left_cam, right_cam), ...] -> [(h,w), ...].  One launch
    per 16 maps where a map fits a workgroup (atvs_transform_depth_batch_supported), else the single-map launches."""
    if not jobs:
        return []
    h, w = jobs[0][0].shape[:2]
    ok = all(tuple(d.shape[:2]) == (h, w) for d, _, _ in jobs) and not jobs[0][0].is_meta \
        and bool(_lib.lib().atvs_transform_depth_batch_supported(int(h), int(w)))
    if not ok or len(jobs) == 1:
        return [transform_depth(d, lc, rc, inverse_depth) for d, lc, rc in jobs]
    outs = [_new(d, (h, w)) for d, _, _ in jobs]
    for lo in range(0, len(jobs), 16):
        part = jobs[lo:lo + 16]
        if _dev_ok(*([t for job in part for t in job])):
            _call('atvs_transform_depth_batch', _ptr_array([d for d, _, _ in part]), _ptr_array([lc for _, lc, _ in part]),
                  _ptr_array([rc for _, _, rc in part]), _ptr_array(outs[lo:lo + 16]), len(part), int(h), int(w),
                  int(bool(inverse_depth)), _stream())
    return outs


def absdiff_mask(a, b, mask):
    """|a - b| * mask, a/b (h,w,C), mask (h,w)."""
    out = _new(a, a.shape)
    npix = mask.numel()
    if _dev_ok(a, b, mask):
        _call('atvs_absdiff_mask', _p(a), _p(b), _p(mask), _p(out), npix, a.numel() // npix, _stream())
    return out


def softargmin(cost, depth_start, depth_interval, groups=None):
    """cost (D,h,w) -> (h,w); groups=G: (G,D,h,w) -> (G,h,w), one depth sweep for all."""
    G = 1 if groups is None else int(groups)
    D, h, w = cost.shape[-3:]
    out = _new(cost, (h, w) if groups is None else (G, h, w))
    if _dev_ok(cost, depth_start, depth_interval):
        _call('atvs_softargmin', _p(cost), _p(depth_start), _p(depth_interval), _p(out), G, D, h, w, _stream())
    return out


def upsample_softargmin(cost, depth_start, depth_interval, up_scale=4):
    """cost (D,h,w) -> (h*up, w*up)."""
    D, h, w = cost.shape
    out = _new(cost, (h * up_scale, w * up_scale))
    if _dev_ok(cost, depth_start, depth_interval):
        _call('atvs_upsample_softargmin', _p(cost), _p(depth_start), _p(depth_interval), _p(out), D, h, w,
              int(up_scale), _stream())
    return out


def probability_map(vol, depth_map, depth_start, depth_interval, up_scale=1, softmax=True):
    """vol (D,h,w) + depth_map (h*up, w*up) -> (h*up, w*up): sum of the four plane probabilities around the depth."""
    D, h, w = vol.shape
    if tuple(depth_map.shape) != (h * up_scale, w * up_scale):
        raise ValueError('probability_map: depth map %s for a (%d,%d) volume x%d' % (tuple(depth_map.shape), h, w, up_scale))
    out = _new(vol, (h * up_scale, w * up_scale))
    if _dev_ok(vol, depth_map, depth_start, depth_interval):
        _call('atvs_probability_map', _p(vol), _p(depth_map), _p(depth_start), _p(depth_interval), _p(out), D, h, w,
              int(up_scale), int(bool(softmax)), _stream())
    return out
