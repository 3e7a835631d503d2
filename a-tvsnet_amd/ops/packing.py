"""Weights in the kernels' operand order (host packers of the C-ABI, cached per (variable, kernel family, device)), tap lists
of SAME / explicit padding, stride, dilation and transposed convolution, the chunk-planar volume layout.
"""

import ctypes
import os

import torch

from .. import _lib
from .base import cfg


def same_pad(in_size, k, s, d=1):
    """TF padding='SAME' -> (pad_before, out_size); end-heavy when the total is odd."""
    out = -(-in_size // s)
    total = max((out - 1) * s + (k - 1) * d + 1 - in_size, 0)
    return total // 2, out


def conv_taps(ksize, dilation, pad_before):
    """Tap list (index in the TF kernel, dz, dy, dx) of a forward convolution."""
    kd, kh, kw = ksize
    taps = []
    for a in range(kd):
        for b in range(kh):
            for c in range(kw):
                taps.append(((a * kh + b) * kw + c, a * dilation - pad_before[0], b * dilation - pad_before[1],
                             c * dilation - pad_before[2]))
    return tuple(taps)


def deconv_s2_class_taps(parity):
    """Taps of one output-parity class of conv3d_transpose(k=3, stride=2, SAME):
    out[2i+k] += in[i] W[k]  =>  even outputs 2j take (k=0, i=j), (k=2, i=j-1); odd 2j+1 take (k=1, i=j)."""
    per_axis = [((0, 0), (2, -1)) if p == 0 else ((1, 0),) for p in parity]
    taps = []
    for ka, oa in per_axis[0]:
        for kb, ob in per_axis[1]:
            for kc, oc in per_axis[2]:
                taps.append(((ka * 3 + kb) * 3 + kc, oa, ob, oc))
    return tuple(taps)


_pack_cache = {}


class _Packed(object):
    __slots__ = ('wp', 'tab', 'ntaps', 'vec', 'ksteps', 'ntiles', 'cin', 'cout', 'key', 'xw', 'kind')


def pack_conv_weights(key, w_host, taps, transposed, device):
    """Packed weights + group table on `device` for (variable, tap list); cached."""
    import numpy as np
    ck = (key, taps, bool(transposed), str(device))
    pk = _pack_cache.get(ck)
    if pk is not None:
        return pk
    w = np.ascontiguousarray(w_host, dtype=np.float32)
    cin, cout = (w.shape[-1], w.shape[-2]) if transposed else (w.shape[-2], w.shape[-1])
    ntaps = len(taps)
    L = _lib.lib()
    vec, ks, nt = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    pf, ti = ctypes.c_long(), ctypes.c_long()
    rc = L.atvs_conv_pack_size(ntaps, cin, cout, ctypes.byref(vec), ctypes.byref(ks), ctypes.byref(nt),
                               ctypes.byref(pf), ctypes.byref(ti))
    if rc:
        raise RuntimeError('atvs_conv_pack_size failed (%d) for Cin=%d Cout=%d' % (rc, cin, cout))
    packed = np.empty(pf.value, np.float32)
    table = np.empty(ti.value, np.int32)
    tp = np.ascontiguousarray(np.array(taps, dtype=np.int32).reshape(-1, 4))
    rc = L.atvs_conv_pack(w.ctypes.data_as(ctypes.c_void_p), int(bool(transposed)), tp.ctypes.data_as(ctypes.c_void_p),
                          ntaps, cin, cout, packed.ctypes.data_as(ctypes.c_void_p),
                          table.ctypes.data_as(ctypes.c_void_p))
    if rc:
        raise RuntimeError('atvs_conv_pack failed (%d)' % rc)
    pk = _Packed()
    pk.ntaps, pk.vec, pk.ksteps, pk.ntiles, pk.cin, pk.cout = ntaps, vec.value, ks.value, nt.value, cin, cout
    pk.key = key
    if torch.device(device).type == 'meta':
        pk.wp = pk.tab = None
    else:
        pk.wp = torch.from_numpy(packed).to(device)
        pk.tab = torch.from_numpy(table).to(device)
    _pack_cache[ck] = pk
    return pk


def pack_conv_weights_tiled(key, w_host, taps, transposed, device, tile_y, xpair=False):
    """Packed weights + LDS-offset table for the LDS-tiled kernel; cached."""
    import numpy as np
    ck = ('tiled', key, taps, bool(transposed), str(device), tile_y, bool(xpair))
    pk = _pack_cache.get(ck)
    if pk is not None:
        return pk
    w = np.ascontiguousarray(w_host, dtype=np.float32)
    cin, cout = (w.shape[-1], w.shape[-2]) if transposed else (w.shape[-2], w.shape[-1])
    ntaps = len(taps)
    L = _lib.lib()
    nch, ccp, jc, nt = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    pf, ti = ctypes.c_long(), ctypes.c_long()
    rc = L.atvs_conv_tiled_pack_size(ntaps, cin, cout, ctypes.byref(nch), ctypes.byref(ccp), ctypes.byref(jc),
                                     ctypes.byref(nt), ctypes.byref(pf), ctypes.byref(ti))
    if rc:
        raise RuntimeError('atvs_conv_tiled_pack_size failed (%d)' % rc)
    packed = np.empty(pf.value, np.float32)
    table = np.empty(ti.value, np.int32)
    tp = np.ascontiguousarray(np.array(taps, dtype=np.int32).reshape(-1, 4))
    rc = L.atvs_conv_tiled_pack(w.ctypes.data_as(ctypes.c_void_p), int(bool(transposed)),
                                tp.ctypes.data_as(ctypes.c_void_p), ntaps, cin, cout, int(tile_y), int(bool(xpair)),
                                packed.ctypes.data_as(ctypes.c_void_p), table.ctypes.data_as(ctypes.c_void_p))
    if rc:
        raise RuntimeError('atvs_conv_tiled_pack failed (%d)' % rc)
    pk = _Packed()
    pk.ntaps, pk.vec, pk.ksteps, pk.ntiles, pk.cin, pk.cout = ntaps, 4, jc.value * nch.value, nt.value, cin, cout
    pk.key = key
    if torch.device(device).type == 'meta':
        pk.wp = pk.tab = None
    else:
        pk.wp = torch.from_numpy(packed).to(device)
        pk.tab = torch.from_numpy(table).to(device)
    _pack_cache[ck] = pk
    return pk


def _xkind():
    """Which one-workgroup-per-CU x-pair kernel serves the 8-output-channel layers."""
    return 'xb' if cfg.xb else 'xw'


PLANAR_PAD = int(os.environ.get('ATVS_PLANAR_PAD', 4096 + 64))         # floats between chunk planes beyond D*h*w*8: 16.25 KiB, so that the C/8 write streams of the


def planar_stride(D, h, w):
    """Floats between the 8-channel chunk planes of a chunk-planar volume."""
    return D * h * w * 8 + PLANAR_PAD


def planar_view(buf, D, h, w):
    """(.., K, planar_stride) chunk-planar buffer -> the (.., K, D, h, w, 8) view of its planes."""
    return buf[..., :D * h * w * 8].unflatten(-1, (D, h, w, 8))


def planar_pieces_ok(shape, F):
    """Should build_cost_volumes write the warped half as pieces?  Only the split-operand x-pair kernel reads them."""
    return cfg.pieces and _xkind() == 'xb' and planar_cost_volume_ok(shape, F)


def planar_pieces_decode(buf, D, h, w):
    """(.., K, planar_stride) buffer written with pieces=True -> (.., K, D, h, w, 8) float32 values h0 + h1 / 2048 (what the
    products of the split-operand kernels see: equal to the fp32 value to 2^-22 relative; tests and fallbacks)."""
    n = D * h * w * 8
    halves = buf[..., :n].contiguous().view(torch.float16).unflatten(-1, (2, D, h, w, 8)).float()
    return halves[..., 0, :, :, :, :] + halves[..., 1, :, :, :, :] / 2048.0


def planar_cost_volume_ok(shape, F):
    """Should build_cost_volumes write the warped half chunk-planar?  When its one consumer is an x-pair launch of
    conv_b0_0_1 | conv_b0_1_0 (both x-pair kernels read the layout)."""
    from .convolution import siblings_ok          # (the policy of the launch that reads the planes)
    return (cfg.planar and F in (16, 32, 64) and siblings_ok(tuple(shape), F, 8, 16))


def pack_conv_xp(key, w_host, device):
    """Packed weights of the one-workgroup-per-CU x-pair kernels (atvs_conv_xb_f32 / atvs_conv_xw_f32); cached."""
    import numpy as np
    kind = _xkind()
    xw = kind == 'xw'
    ck = (kind, key, str(device))
    pk = _pack_cache.get(ck)
    if pk is not None:
        return pk
    w = np.ascontiguousarray(w_host, dtype=np.float32)           # [3,3,3,Cin,8]
    cin = w.shape[-2]
    L = _lib.lib()
    pf = ctypes.c_long()
    size_fn, pack_fn = getattr(L, 'atvs_conv_%s_pack_size' % kind), getattr(L, 'atvs_conv_%s_pack' % kind)
    rc = size_fn(cin, ctypes.byref(pf))
    if rc:
        raise RuntimeError('atvs_conv_%s_pack_size failed (%d) for Cin=%d' % (kind, rc, cin))
    packed = np.empty(pf.value, np.uint8 if kind == 'xb' else np.float32)      # xb: bytes (fp16 pieces)
    rc = pack_fn(w.ctypes.data_as(ctypes.c_void_p), cin, packed.ctypes.data_as(ctypes.c_void_p))
    if rc:
        raise RuntimeError('atvs_conv_%s_pack failed (%d)' % (kind, rc))
    pk = _Packed()
    pk.ntaps, pk.vec, pk.ksteps, pk.ntiles, pk.cin, pk.cout = 36, 4, 0, 1, cin, 8
    pk.xw, pk.kind = xw, kind
    pk.key = key
    pk.tab = None
    pk.wp = None if torch.device(device).type == 'meta' else torch.from_numpy(packed).to(device)
    _pack_cache[ck] = pk
    return pk


def pack_deconv_up(key, w_host, device, kind=''):
    """Packed weights of the 8- / 16-channel transposed-convolution kernels (atvs_deconv_up_f32; kind '_b': the split-fp16
    atvs_deconv_up_b_f32, bytes of fp16 pieces); cached."""
    import numpy as np
    ck = ('up' + kind, key, str(device))
    pk = _pack_cache.get(ck)
    if pk is not None:
        return pk
    w = np.ascontiguousarray(w_host, dtype=np.float32)           # [3,3,3,Cout,Cin]
    cout, cin = int(w.shape[-2]), int(w.shape[-1])
    L = _lib.lib()
    pf = ctypes.c_long()
    rc = getattr(L, 'atvs_deconv_up%s_pack_size' % kind)(cin, cout, ctypes.byref(pf))
    if rc:
        raise RuntimeError('atvs_deconv_up%s_pack_size failed (%d) for %d -> %d' % (kind, rc, cin, cout))
    packed = np.empty(pf.value, np.uint8 if kind else np.float32)
    rc = getattr(L, 'atvs_deconv_up%s_pack' % kind)(w.ctypes.data_as(ctypes.c_void_p), cin, cout,
                                                    packed.ctypes.data_as(ctypes.c_void_p))
    if rc:
        raise RuntimeError('atvs_deconv_up%s_pack failed (%d)' % (kind, rc))
    pk = _Packed()
    pk.ntaps, pk.vec, pk.ksteps, pk.ntiles, pk.cin, pk.cout = 27, 4, 0, 1, cin, cout
    pk.key = key
    pk.tab = None
    pk.wp = None if torch.device(device).type == 'meta' else torch.from_numpy(packed).to(device)
    _pack_cache[ck] = pk
    return pk


def pack_conv_c16(key, w_host, device):
    """Packed weights of the 16-output-channel 3x3x3 kernel (atvs_conv_c16_f32); cached."""
    import numpy as np
    ck = ('c16', key, str(device))
    pk = _pack_cache.get(ck)
    if pk is not None:
        return pk
    w = np.ascontiguousarray(w_host, dtype=np.float32)           # [3,3,3,Cin,Cout]
    cin, cout = int(w.shape[-2]), int(w.shape[-1])
    L = _lib.lib()
    pf = ctypes.c_long()
    rc = L.atvs_conv_c16_pack_size(cin, cout, ctypes.byref(pf))
    if rc:
        raise RuntimeError('atvs_conv_c16_pack_size failed (%d) for %d -> %d' % (rc, cin, cout))
    packed = np.empty(pf.value, np.float32)
    rc = L.atvs_conv_c16_pack(w.ctypes.data_as(ctypes.c_void_p), cin, cout, packed.ctypes.data_as(ctypes.c_void_p))
    if rc:
        raise RuntimeError('atvs_conv_c16_pack failed (%d)' % rc)
    pk = _Packed()
    pk.ntaps, pk.vec, pk.ksteps, pk.ntiles, pk.cin, pk.cout = 27, 4, 0, cout // 16, cin, cout
    pk.key = key
    pk.tab = None
    pk.wp = None if torch.device(device).type == 'meta' else torch.from_numpy(packed).to(device)
    _pack_cache[ck] = pk
    return pk


def split_on(name):
    """Is the split-operand kernel family `name` enabled?  (c16b, c3b, s2b, upb, c2b, c1b, btl; `ops.configure(split_off=(...))` /
    ATVS_SPLIT_OFF=a,b keep single families on the fp32 matrix cores -- testing / A-B hook; conv_xb has cfg.xb.)"""
    return cfg.split16 and name not in cfg.split_off


def pack_conv_c16b(key, w_host, device):
    """Packed fp16 pieces of a [3,3,3,Cin,16] kernel (Cin 8 or 16) for atvs_conv_c16b_f32; cached."""
    import numpy as np
    ck = ('c16b', key, str(device))
    pk = _pack_cache.get(ck)
    if pk is not None:
        return pk
    w = np.ascontiguousarray(w_host, dtype=np.float32)
    cin = int(w.shape[-2])
    L = _lib.lib()
    pb = ctypes.c_long()
    rc = L.atvs_conv_c16b_pack_size(cin, ctypes.byref(pb))
    if rc:
        raise RuntimeError('atvs_conv_c16b_pack_size failed (%d)' % rc)
    packed = np.empty(pb.value, np.uint8)
    rc = L.atvs_conv_c16b_pack(w.ctypes.data_as(ctypes.c_void_p), cin, packed.ctypes.data_as(ctypes.c_void_p))
    if rc:
        raise RuntimeError('atvs_conv_c16b_pack failed (%d)' % rc)
    pk = _Packed()
    pk.ntaps, pk.vec, pk.ksteps, pk.ntiles, pk.cin, pk.cout = 27, 4, 0, 1, cin, 16
    pk.key, pk.tab, pk.xw = key, None, False
    pk.wp = None if torch.device(device).type == 'meta' else torch.from_numpy(packed).to(device)
    _pack_cache[ck] = pk
    return pk


def pack_conv3d_b(key, w_host, device, kind='b'):
    """Packed fp16 pieces of a [3,3,3,Cin,Cout] kernel (Cin % 16 == 0, Cout 32 / 64) for atvs_conv3d_b_f32 (kind 'b') or the
    stride-2 atvs_conv3d_s2b_f32 (kind 's2b'); cached."""
    import numpy as np
    ck = ('c3' + kind, key, str(device))
    pk = _pack_cache.get(ck)
    if pk is not None:
        return pk
    w = np.ascontiguousarray(w_host, dtype=np.float32)
    cin, cout = int(w.shape[-2]), int(w.shape[-1])
    L = _lib.lib()
    pb = ctypes.c_long()
    rc = getattr(L, 'atvs_conv3d_%s_pack_size' % kind)(cin, cout, ctypes.byref(pb))
    if rc:
        raise RuntimeError('atvs_conv3d_%s_pack_size failed (%d) for %d -> %d' % (kind, rc, cin, cout))
    packed = np.empty(pb.value, np.uint8)
    rc = getattr(L, 'atvs_conv3d_%s_pack' % kind)(w.ctypes.data_as(ctypes.c_void_p), cin, cout,
                                                    packed.ctypes.data_as(ctypes.c_void_p))
    if rc:
        raise RuntimeError('atvs_conv3d_%s_pack failed (%d)' % (kind, rc))
    pk = _Packed()
    pk.key, pk.tab, pk.cin, pk.cout = key, None, cin, cout
    pk.wp = None if torch.device(device).type == 'meta' else torch.from_numpy(packed).to(device)
    _pack_cache[ck] = pk
    return pk


def deconv_up_ok(cin, cout):
    return cfg.deconv_up and cfg.force_impl is None and cout in (8, 16) and cin % 16 == 0 and 0 < cin <= 64


def pack_conv_xp_sibling(key, w_host, device):
    """Packed weights of the stride-2 sibling [3,3,3,Cin,16] of an x-pair launch; cached."""
    import numpy as np
    kind = _xkind()
    xw = kind == 'xw'
    ck = (kind + '2', key, str(device))
    pk = _pack_cache.get(ck)
    if pk is not None:
        return pk
    w = np.ascontiguousarray(w_host, dtype=np.float32)
    cin = w.shape[-2]
    if w.shape[-1] != 16:
        raise ValueError('x-pair sibling: 16 output channels, got %d' % w.shape[-1])
    L = _lib.lib()
    pf = ctypes.c_long()
    size_fn, pack_fn = getattr(L, 'atvs_conv_%s_pack_sibling_size' % kind), getattr(L, 'atvs_conv_%s_pack_sibling' % kind)
    rc = size_fn(cin, ctypes.byref(pf))
    if rc:
        raise RuntimeError('x-pair sibling pack size failed (%d) for Cin=%d' % (rc, cin))
    packed = np.empty(pf.value, np.uint8 if kind == 'xb' else np.float32)
    rc = pack_fn(w.ctypes.data_as(ctypes.c_void_p), cin, packed.ctypes.data_as(ctypes.c_void_p))
    if rc:
        raise RuntimeError('x-pair sibling pack failed (%d)' % rc)
    pk = _Packed()
    pk.ntaps, pk.vec, pk.ksteps, pk.ntiles, pk.cin, pk.cout = 27, 4, 0, 1, cin, 16
    pk.xw, pk.kind = xw, kind
    pk.key = key
    pk.tab = None
    pk.wp = None if torch.device(device).type == 'meta' else torch.from_numpy(packed).to(device)
    _pack_cache[ck] = pk
    return pk


_xp_cache = {}


def clear_pack_cache():
    """Forget every arranged form of the weights (packed device copies, folded split kernels, virtual x-pair and
    transposed-convolution kernels).  They are keyed by variable NAME, so the variable store calls this whenever a
    value changes (VariableStore.set / clear / load_*).  A captured HIP graph keeps the copies it was captured
    with alive (GraphedInference holds references) and goes on using them."""
    _pack_cache.clear()
    _fold_cache.clear()
    _xp_cache.clear()
    _virt_cache.clear()


invalidate_weights = clear_pack_cache


def cache_snapshot():
    """References to every cached device tensor (for owners of captured graphs)."""
    return [pk for pk in _pack_cache.values()]


_fold_cache = {}


_virt_cache = {}
