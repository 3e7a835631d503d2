"""Python face of the C-ABI (include/atvsnet_hip.h): one function per entry point.

torch is plumbing here: it owns device memory (caching allocator) and the
current HIP stream; every computation happens in the gfx950 kernels reached
through ctypes.  Tensors must be float32, contiguous and on a ``cuda`` device;
``meta`` tensors run the same host code without launching (shape / memory
planning and CPU plumbing tests).  CPU tensors are refused: there is no CPU
fallback on the product path.
"""
import ctypes

import torch

from . import _lib

_ERR = {-1: 'null pointer', -2: 'bad shape', -3: 'bad argument', -4: 'launch failed'}


def _dev_ok(*ts):
    """True if the kernels must be launched, False for meta tensors."""
    meta = None
    for t in ts:
        if t is None:
            continue
        if t.dtype != torch.float32:
            raise TypeError('atvsnet ops take float32 tensors, got %s' % t.dtype)
        if not t.is_contiguous():
            raise ValueError('atvsnet ops take contiguous tensors')
        if t.device.type == 'cpu':
            raise RuntimeError('atvsnet ops run on the MI355X only: got a CPU tensor and there is no CPU fallback')
        m = t.device.type == 'meta'
        if meta is None:
            meta = m
        elif meta != m:
            raise RuntimeError('mixing meta and device tensors')
    return not meta


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _call(name, *args):
    rc = getattr(_lib.lib(), name)(*args)
    if rc != 0:
        raise RuntimeError('%s failed: %s (%d)' % (name, _ERR.get(rc, 'unknown'), rc))


def _new(ref, shape):
    return torch.empty(tuple(int(s) for s in shape), dtype=torch.float32, device=ref.device)


# --------------------------------------------------------------------------- geometry

def get_homographies(left_cam, right_cam, depth_start, depth_interval, depth_num, inverse_depth=True):
    """cams (2,4,4); depth_start/interval 1-element tensors -> (D,3,3)."""
    out = _new(left_cam, (depth_num, 3, 3))
    if _dev_ok(left_cam, right_cam, depth_start, depth_interval):
        _call('atvs_get_homographies', _p(left_cam), _p(right_cam), _p(depth_start), _p(depth_interval), _p(out),
              int(depth_num), int(bool(inverse_depth)), _stream())
    return out


def warp_planes(src, homographies, out=None, ld_out=None, c_off=0, mode=0, ref=None, depth_start=None,
                depth_interval=None, rep=1, want_mask=False):
    """src (h,w,C), homographies (D,3,3) -> out (D,h,w,ld_out) [, mask (D,h,w)]."""
    h, w, C = src.shape
    D = homographies.shape[0]
    width = rep if mode == 2 else C
    if out is None:
        ld_out = width if ld_out is None else ld_out
        out = _new(src, (D, h, w, ld_out))
    else:
        ld_out = out.shape[-1]
    mask = _new(src, (D, h, w)) if want_mask else None
    if _dev_ok(src, homographies, out, ref, depth_start, depth_interval):
        _call('atvs_warp_planes', _p(src), _p(homographies), _p(ref), _p(depth_start), _p(depth_interval), _p(out),
              _p(mask), D, h, w, C, int(ld_out), int(c_off), int(mode), int(rep), _stream())
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


def visual_hull(ref_depth, view_depth_in_ref, homographies, depth_start, depth_interval, inverse_depth=True):
    """(h,w) x2 -> (D,h,w)."""
    h, w = ref_depth.shape[:2]
    D = homographies.shape[0]
    out = _new(ref_depth, (D, h, w))
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


def transform_depth(depth, left_cam, right_cam, inverse_depth=True):
    """depth (h,w) in left_cam -> same pixels, values expressed in right_cam's frame."""
    h, w = depth.shape[:2]
    out = _new(depth, (h, w))
    ws = _new(depth, (14,))
    if _dev_ok(depth, left_cam, right_cam):
        _call('atvs_transform_depth', _p(depth), _p(left_cam), _p(right_cam), _p(out), _p(ws), h, w,
              int(bool(inverse_depth)), _stream())
    return out


def absdiff_mask(a, b, mask):
    """|a - b| * mask, a/b (h,w,C), mask (h,w)."""
    out = _new(a, a.shape)
    npix = mask.numel()
    if _dev_ok(a, b, mask):
        _call('atvs_absdiff_mask', _p(a), _p(b), _p(mask), _p(out), npix, a.numel() // npix, _stream())
    return out


# --------------------------------------------------------------------------- soft-argmin

def softargmin(cost, depth_start, depth_interval):
    """cost (D,h,w) -> (h,w)."""
    D, h, w = cost.shape
    out = _new(cost, (h, w))
    if _dev_ok(cost, depth_start, depth_interval):
        _call('atvs_softargmin', _p(cost), _p(depth_start), _p(depth_interval), _p(out), D, h, w, _stream())
    return out


def upsample_softargmin(cost, depth_start, depth_interval, up_scale=4):
    """cost (D,h,w) -> (h*up, w*up)."""
    D, h, w = cost.shape
    out = _new(cost, (h * up_scale, w * up_scale))
    if _dev_ok(cost, depth_start, depth_interval):
        _call('atvs_upsample_softargmin', _p(cost), _p(depth_start), _p(depth_interval), _p(out), D, h, w,
              int(up_scale), _stream())
    return out
