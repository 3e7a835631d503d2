"""AANet aggregation (one launch per module, or score convolution + combine, or the view-sharded partial forms) and the depth-map
fusion entry point (after the hot path).
"""

import ctypes

import torch

from .. import _lib
from .base import _Timed, _call, _dev_ok, _new, _p, _ptr_array, _stream, cfg
from .packing import _Packed, _pack_cache, split_on


def aanet_combine(srs, xs, out=None):
    """srs: list of (V..,16) [S|R] tensors, xs: list of (V..,8) -> sum_n softmax_n(U) X_n, shape of xs[0]
    (written into `out` when given)."""
    out = _new(xs[0], xs[0].shape) if out is None else out
    if _dev_ok(*(list(srs) + list(xs))):
        _call('atvs_aanet_combine', _ptr_array(srs), _ptr_array(xs), len(xs), _p(out),
              ctypes.c_long(out.numel() // 8), _stream())
    return out


def aanet_fused_ok(xs):
    """Does the whole AANet module over these views run as ONE launch (atvs_aanet_b_f32)?"""
    return (cfg.aanet_fused and cfg.conv_c16 and cfg.force_impl is None and split_on('c16b') and xs[0].dim() == 4
            and xs[0].shape[-1] == 8 and xs[0].shape[2] >= 12 and all(t.is_contiguous() and tuple(t.shape) == tuple(xs[0].shape) for t in xs)
            and bool(_lib.lib().atvs_aanet_b_supported(8, len(xs))))


def aanet_fused(xs, key, w_shared, w_unique):
    """AANet over the views xs (list of (D,H,W,8)): score convolutions + cross-view softmax + weighted sum in one launch ->
    (D,H,W,8).  w_shared / w_unique: host TF kernels [3,3,3,8,8]; key: pack-cache key."""
    import numpy as np
    dev = xs[0].device
    ck = ('aanet_b', key, str(dev))
    pk = _pack_cache.get(ck)
    if pk is None:
        L = _lib.lib()
        pf = ctypes.c_long()
        L.atvs_aanet_b_pack_size(ctypes.byref(pf))
        packed = np.empty(pf.value, np.uint8)
        ws = np.ascontiguousarray(w_shared, dtype=np.float32)
        wu = np.ascontiguousarray(w_unique, dtype=np.float32)
        rc = L.atvs_aanet_b_pack(ws.ctypes.data_as(ctypes.c_void_p), wu.ctypes.data_as(ctypes.c_void_p),
                                 packed.ctypes.data_as(ctypes.c_void_p))
        if rc:
            raise RuntimeError('atvs_aanet_b_pack failed (%d)' % rc)
        pk = _Packed()
        pk.key, pk.tab = key, None
        pk.wp = None if dev.type == 'meta' else torch.from_numpy(packed).to(dev)
        _pack_cache[ck] = pk
    D, H, W, _ = xs[0].shape
    out = _new(xs[0], xs[0].shape)
    if _dev_ok(out, *xs):
        with _Timed(key, (D, H, W, 8), 16, len(xs)):
            _call('atvs_aanet_b_f32', _ptr_array(xs), len(xs), _p(pk.wp), _p(out), D, H, W, _stream())
    return out


def aanet_partial(srs, xs, stage, ssum=None, umax=None):
    V8 = tuple(xs[0].shape)
    out = _new(xs[0], ((2,) + V8) if stage == 2 else V8)
    if _dev_ok(*(list(srs) + list(xs))):
        _call('atvs_aanet_partial', _ptr_array(srs), _ptr_array(xs), len(srs), int(stage), _p(ssum), _p(umax), _p(out),
              ctypes.c_long(xs[0].numel() // 8), _stream())
    return out


def divide(num, den):
    out = _new(num, num.shape)
    if _dev_ok(num, den):
        _call('atvs_divide', _p(num), _p(den), _p(out), ctypes.c_long(num.numel()), _stream())
    return out


def fusibile(cams, normals_depths, images, ref, disp_thresh, normal_thresh, num_consistent):
    """The consistency-voting kernel of the reference's fusibile for reference camera `ref` (atvs_fusibile).
    cams (N,28), normals_depths / images (N,rows,cols,4) -> coord, normal, texture (rows,cols,4), created (rows,cols)."""
    N, rows, cols, _ = normals_depths.shape
    coord, normal, tex = (_new(images, (rows, cols, 4)) for _ in range(3))
    created = _new(images, (rows, cols))
    if _dev_ok(cams, normals_depths, images):
        _call('atvs_fusibile', _p(cams), _p(normals_depths), _p(images), N, int(ref), rows, cols, ctypes.c_float(disp_thresh),
              ctypes.c_float(normal_thresh), int(num_consistent), _p(coord), _p(normal), _p(tex), _p(created), _stream())
    return coord, normal, tex, created
