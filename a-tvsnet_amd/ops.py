"""Python face of the C-ABI (include/atvsnet_hip.h): one function per entry point.

torch is plumbing here: it owns device memory (caching allocator) and the
current HIP stream; every computation happens in the gfx950 kernels reached
through ctypes.  Tensors must be float32, contiguous and on a ``cuda`` device;
``meta`` tensors run the same host code without launching (shape / memory
planning and CPU plumbing tests).  CPU tensors are refused: there is no CPU
fallback on the product path.
"""
import ctypes
import os

import torch

from . import _lib

_ERR = {-1: 'null pointer', -2: 'bad shape', -3: 'bad argument', -4: 'launch failed'}


class Config(object):
    """Every dispatch switch of this module in ONE object (`ops.cfg`).  The product never changes them: the defaults below
    are what runs; tests and A/B measurements use `with ops.configure(name=value, ...):`, which restores the previous
    values on exit (also on an exception).  Unknown names raise.

    split16        the split-operand kernels (x = h0 + h1 / 2048 in fp16, three products on v_mfma_f32_16x16x32_f16, fp32
                   accumulation) in front of their fp32-MFMA forms.  Setting it sets `xb` too.  ATVS_SPLIT16=0 in
                   the environment: every convolution on the fp32 matrix cores.
    split_off      kernel families (c16b, c3b, s2b, upb, c2b, c1b, btl) kept on the fp32 matrix cores while split16 is on
    xb             the split-operand x-pair kernel (conv_xb.hip) in front of the fp32 one (conv_xw.hip)
    planar         the warped half of the cost volume chunk-planar;  pieces: ... as fp16 pieces written by the warp
    planar_concat  the refinement's 32-channel concat as four dense 8-channel planes
    conv_c16, deconv_up, stem, conv2d_lds, conv1x1, xp1w, xpair, siblings, bottleneck, aanet_fused
                   the dedicated kernel of that layer family in front of the generic ones
    prologue       normalise-on-load / add-on-load in the consumers (else pending batch norms / sums are materialised first)
    sum_on_load    the U-Net's skip sums formed inside their consumer's staging (transposed convolution, 16-channel convolution)
                   instead of a bn_add pass (needs prologue)
    norm3d         pending batch norms / two-term skip sums formed while conv_c16b, conv3d_b, conv3d_s2b stage their halo instead of
                   a bn_apply / bn_add pass (needs sum_on_load)
    force_impl     None (automatic) | 'tiled' | 'gather': the generic convolution kernel to use
    fused_finalize batch-norm moments finished inside the convolution launch (measured slower: off)
    side_streams   independent small launches of one layer on side streams (parallel branches of a captured graph)
    """
    _DEFAULTS = dict(
        split16=os.environ.get('ATVS_SPLIT16', '1') == '1',
        split_off=frozenset(v for v in os.environ.get('ATVS_SPLIT_OFF', '').split(',') if v),
        xb=os.environ.get('ATVS_SPLIT16', '1') == '1',
        planar=True,
        pieces=os.environ.get('ATVS_PIECES', '1') == '1',
        planar_concat=os.environ.get('ATVS_PLANAR_CONCAT', '1') != '0',
        conv_c16=True, deconv_up=True, stem=True, conv2d_lds=True, conv1x1=True, xp1w=True, xpair=True, siblings=True,
        aanet_fused=os.environ.get('ATVS_AANET_FUSED', '1') != '0',
        bottleneck=os.environ.get('ATVS_BOTTLENECK', '1') != '0',
        prologue=True, sum_on_load=os.environ.get('ATVS_SUM_ON_LOAD', '1') != '0',
        norm3d=os.environ.get('ATVS_NORM3D', '1') != '0', force_impl=None, fused_finalize=False,
        side_streams=os.environ.get('ATVS_SIDE_STREAMS', '1') != '0')

    def __init__(self):
        for k, v in self._DEFAULTS.items():
            object.__setattr__(self, k, v)

    def __setattr__(self, name, value):
        if name not in self._DEFAULTS:
            raise AttributeError('ops.cfg has no switch %r (known: %s)' % (name, ', '.join(sorted(self._DEFAULTS))))
        if name == 'split_off':
            value = frozenset(value)
        elif name == 'force_impl':
            if value not in (None, 'tiled', 'gather'):
                raise ValueError('force_impl: None | "tiled" | "gather"')
        else:
            value = bool(value)
        object.__setattr__(self, name, value)
        if name == 'split16':
            object.__setattr__(self, 'xb', value)

    def snapshot(self):
        return {k: getattr(self, k) for k in self._DEFAULTS}


cfg = Config()


class configure(object):
    """`with ops.configure(split16=False, clear_pack_cache=True): ...` -- set switches of `ops.cfg` for the block and restore
    them afterwards.  clear_pack_cache=True also drops the arranged-weight cache on entry and exit (for tests that reuse a
    weight key under two kernel families)."""

    def __init__(self, clear_pack_cache=False, **switches):
        for k in switches:
            if k not in Config._DEFAULTS:
                raise AttributeError('ops.cfg has no switch %r' % k)
        self._new, self._clear = switches, clear_pack_cache

    def __enter__(self):
        self._old = cfg.snapshot()
        # split16 first: it drags xb along, an explicit xb= in the same call wins
        for k in sorted(self._new, key=lambda n: n != 'split16'):
            setattr(cfg, k, self._new[k])
        if self._clear:
            clear_pack_cache()
        return cfg

    def __exit__(self, *exc):
        for k, v in self._old.items():
            object.__setattr__(cfg, k, v)
        if self._clear:
            clear_pack_cache()
        return False


def _dev_ok(*ts):
    """True if the kernels must be launched, False for meta tensors.  Device tensors must live on the CURRENT
    device (torch.cuda.set_device / FLAGS.gpu_id): the launch goes to that device's current stream."""
    meta = None
    for t in ts:
        if t is None:
            continue
        if t.device.type == 'cuda' and t.device.index != torch.cuda.current_device():
            raise RuntimeError('atvsnet ops launch on the current device (cuda:%d) but got a tensor on %s: call '
                               'torch.cuda.set_device first (example.py --gpu_id does)' %
                               (torch.cuda.current_device(), t.device))
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


_side_pool = {}




def _side_stream(device, i):
    key = (str(device), i)
    s = _side_pool.get(key)
    if s is None:
        s = _side_pool[key] = torch.cuda.Stream(device=device)
    return s


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


# --------------------------------------------------------------------------- soft-argmin

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


# --------------------------------------------------------------------------- convolutions

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
                               # warp (same offset in every plane) do not all start on the same HBM channel / bank


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


def norm_on_load_2d_ok(src, ksize, filters, stride=1, rate=1):
    """Can a 2-D convolution of this shape take a pending batch norm (PendingBN, channel-last) as it is -- the kernel
    normalises (+ ReLU) while staging (conv2d_b.hip / conv1x1_b.hip `in_params`)?"""
    if not (cfg.prologue and isinstance(src, PendingBN) and src._final is None and not src.planar and src.dim() == 4
            and src.raw.is_contiguous() and stride == 1):
        return False
    G, H, W, cin = src.shape
    if ksize == 3:
        return conv2d_lds_ok(cin, filters, rate, H, W) and split_on('c2b') and cin % 32 == 0
    if ksize == 1:
        return rate == 1 and conv1x1_ok(cin, filters)
    return False


def norm_on_load_3d_ok(src, ksize, filters, stride=1, rate=1):
    """Can a 3-D convolution take this lazy input as it is (its batch norm / its skip sum formed while the kernel stages
    the halo)?  Built forms: a pending batch norm in front of conv_c16b (16 -> 16), conv3d_b (Cin % 16 == 0 -> 32 / 64) and
    the stride-2 conv3d_s2b; a sum of two (dense or pending) in front of conv_c16b.  ops.conv falls back to the passes
    themselves for a shape its dispatch sends elsewhere."""
    if not (cfg.sum_on_load and cfg.norm3d) or cfg.force_impl is not None or not cfg.conv_c16 or ksize != 3 or rate != 1 \
            or src.dim() != 5:
        return False
    cin = int(src.shape[-1])
    if isinstance(src, PendingBN):
        if src._final is not None or src.planar or not src.raw.is_contiguous() or cin % 16:
            return False
        if stride == 2:
            return split_on('s2b') and filters in (32, 64)
        return stride == 1 and ((cin == 16 and filters == 16 and split_on('c16b')) or (filters in (32, 64) and split_on('c3b')))
    if isinstance(src, PendingSum):
        if src._final is not None or len(src.items) != 2 or stride != 1 or cin != 16 or filters != 16 or not split_on('c16b'):
            return False
        gs = set()
        for t in src.items:
            raw = t.raw if isinstance(t, PendingBN) else t
            if isinstance(t, PendingBN) and t._final is None:
                gs.add(_param_groups(t.params))
            if not raw.is_contiguous() or (isinstance(t, PendingBN) and t.planar):
                return False
        return len(gs) <= 1
    return False


def conv2d_lds_ok(cin, cout, dilation, H, W):
    """Is the LDS-tiled 2-D kernel (atvs_conv2d_lds_f32) used for a 3x3 stride-1 SAME convolution of this shape?"""
    # tiny maps (the pyramid branches' pooled maps, 2 x 3 ... 8 x 10 pixels): the split-operand kernel covers them with one masked
    # tile per image in ~20 us; the generic gather kernel needs ~49 us for its serial 9 x Cin K loop
    tiny_ok = split_on('c2b') and cin % 32 == 0 and H >= 2 and W >= 2
    return (cfg.force_impl != 'gather' and cfg.conv2d_lds and ((H >= 8 and W >= 16) or tiny_ok)
            and bool(_lib.lib().atvs_conv2d_lds_supported(int(cin), int(cout), int(dilation))))








def pack_conv2d_lds(key, w_host, device):
    """Packed weights of the LDS-tiled 2-D kernel for a TF kernel [3,3,Cin,Cout]; cached."""
    import numpy as np
    w = np.ascontiguousarray(w_host, dtype=np.float32)
    cin, cout = int(w.shape[-2]), int(w.shape[-1])
    split = split_on('c2b') and cin % 32 == 0      # conv2d_b.hip: split-fp16 operands (its chunk loop runs in pairs)
    kind = 'b' if split else 'lds'
    ck = ('c2' + kind, key, str(device))
    pk = _pack_cache.get(ck)
    if pk is not None:
        return pk
    L = _lib.lib()
    pf = ctypes.c_long()
    rc = getattr(L, 'atvs_conv2d_%s_pack_size' % kind)(cin, cout, ctypes.byref(pf))
    if rc:
        raise RuntimeError('atvs_conv2d_%s_pack_size failed (%d) for Cin=%d Cout=%d' % (kind, rc, cin, cout))
    packed = np.empty(pf.value, np.uint8 if split else np.float32)
    rc = getattr(L, 'atvs_conv2d_%s_pack' % kind)(w.ctypes.data_as(ctypes.c_void_p), cin, cout, packed.ctypes.data_as(ctypes.c_void_p))
    if rc:
        raise RuntimeError('atvs_conv2d_%s_pack failed (%d)' % (kind, rc))
    pk = _Packed()
    pk.ntaps, pk.vec, pk.ksteps, pk.ntiles, pk.cin, pk.cout = 9, 4, 0, cout // 16, cin, cout
    pk.key, pk.kind = key, kind
    pk.tab = None
    pk.wp = None if torch.device(device).type == 'meta' else torch.from_numpy(packed).to(device)
    _pack_cache[ck] = pk
    return pk


def conv2d_lds(x, key, w_host, dilation=1, bias=None, residual=None, relu=False, want_stats=False, out=None, y_coff=0,
               in_params=None, in_relu=False):
    """3x3 stride-1 SAME convolution of x (G,H,W,Cin) -> (G,H,W,Cout) on the LDS-tiled 2-D kernel.
    in_params (G,3,Cin): batch norm (+ ReLU if in_relu) of x applied on load.  want_stats: also returns the
    per-image moments (Stats with groups = G)."""
    G, H, W, cin = x.shape
    pk = pack_conv2d_lds(key, w_host, x.device)
    if pk.cin != cin:
        raise ValueError('conv %s: input has %d channels, kernel wants %d' % (key, cin, pk.cin))
    y = _new(x, (G, H, W, pk.cout)) if out is None else out
    if tuple(y.shape[:3]) != (G, H, W):
        raise ValueError('conv %s: output buffer %s does not match %s' % (key, tuple(y.shape), (G, H, W)))
    st, sbuf = None, None
    if want_stats:
        rows = int(_lib.lib().atvs_conv2d_lds_rows(H, W, pk.cout))
        sbuf = torch.empty((G, rows, 2, pk.cout), dtype=torch.float64, device=x.device)
        st = Stats()
        st.partial, st.blocks, st.cpad, st.count, st.groups = sbuf, rows, pk.cout, H * W, G
    if _dev_ok(x, y, bias, residual, in_params):
        with _Timed(pk.key, (1, H, W, cin), pk.cout, G):
            _call('atvs_conv2d_%s_f32' % pk.kind, _p(x), _p(pk.wp), _p(bias), _p(residual), _p(in_params), int(bool(in_relu)),
                  _p(y), ctypes.c_void_p(sbuf.data_ptr()) if sbuf is not None else ctypes.c_void_p(0), G, H, W, cin,
                  pk.cout, int(dilation), int(y.shape[-1]), int(y_coff), int(bool(relu)), _stream())
    return (y, st) if want_stats else y


def conv1x1_ok(cin, cout):
    """Is a GEMM kernel (atvs_conv1x1_b_f32 / atvs_conv1x1_f32) used for a stride-1 1x1 convolution of these channel counts?"""
    lib = _lib.lib()
    return (cfg.force_impl != 'gather' and cfg.conv1x1 and
            bool((split_on('c1b') and lib.atvs_conv1x1_b_supported(int(cin), int(cout))) or lib.atvs_conv1x1_supported(int(cin), int(cout))))






def pack_conv1x1(key, w_host, cin, device):
    """Packed weights of the 1x1 GEMM kernels for a TF kernel [1,1,Cin,Cout] (or [Cin,Cout]); cached.  pk.kind: '_b' = fp16
    pieces for conv1x1_b.hip (and the 1x1 stages of bottleneck_b.hip), '' = fp32 for conv1x1.hip."""
    import numpy as np
    lib = _lib.lib()
    kind = '_b' if (split_on('c1b') and lib.atvs_conv1x1_b_supported(int(cin), int(np.asarray(w_host).size // cin))) else ''
    ck = ('c1' + kind, key, str(device))
    pk = _pack_cache.get(ck)
    if pk is None:
        w = np.ascontiguousarray(w_host, dtype=np.float32).reshape(cin, -1)
        cout = int(w.shape[1])
        pf = ctypes.c_long()
        rc = getattr(lib, 'atvs_conv1x1%s_pack_size' % kind)(cin, cout, ctypes.byref(pf))
        if rc:
            raise RuntimeError('atvs_conv1x1%s_pack_size failed (%d) for Cin=%d Cout=%d' % (kind, rc, cin, cout))
        packed = np.empty(pf.value, np.uint8 if kind else np.float32)       # split kernel: bytes of fp16 pieces
        rc = getattr(lib, 'atvs_conv1x1%s_pack' % kind)(w.ctypes.data_as(ctypes.c_void_p), cin, cout,
                                                        packed.ctypes.data_as(ctypes.c_void_p))
        if rc:
            raise RuntimeError('atvs_conv1x1%s_pack failed (%d)' % (kind, rc))
        pk = _Packed()
        pk.key, pk.tab, pk.cin, pk.cout, pk.ntiles, pk.kind = key, None, cin, cout, cout // 16, kind
        pk.wp = None if torch.device(device).type == 'meta' else torch.from_numpy(packed).to(device)
        _pack_cache[ck] = pk
    return pk


def conv1x1(x, key, w_host, bias=None, residual=None, relu=False, want_stats=False, out=None, y_coff=0, in_params=None,
            in_relu=False):
    """1x1 stride-1 convolution of x (G, ..., Cin) -> (G, ..., Cout) (any spatial axes between) on the GEMM kernel.
    in_params (G,3,Cin): batch norm (+ ReLU if in_relu) of x applied on load."""
    G, cin = x.shape[0], x.shape[-1]
    pixels = x.numel() // G // cin
    lib = _lib.lib()
    pk = pack_conv1x1(key, w_host, cin, x.device)
    kind = pk.kind
    y = _new(x, tuple(x.shape[:-1]) + (pk.cout,)) if out is None else out
    st, sbuf = None, None
    if want_stats:
        rows = int(getattr(lib, 'atvs_conv1x1%s_rows' % kind)(ctypes.c_long(pixels)))
        sbuf = torch.empty((G, rows, 2, pk.cout), dtype=torch.float64, device=x.device)
        st = Stats()
        st.partial, st.blocks, st.cpad, st.count, st.groups = sbuf, rows, pk.cout, pixels, G
    if _dev_ok(x, y, bias, residual, in_params):
        with _Timed(pk.key, (1, 1, pixels, cin), pk.cout, G):
            _call('atvs_conv1x1%s_f32' % kind, _p(x), _p(pk.wp), _p(bias), _p(residual), _p(in_params), int(bool(in_relu)),
                  _p(y), ctypes.c_void_p(sbuf.data_ptr()) if sbuf is not None else ctypes.c_void_p(0), G,
                  ctypes.c_long(pixels), cin, pk.cout, int(y.shape[-1]), int(y_coff), int(bool(relu)), _stream())
    return (y, st) if want_stats else y


def bottleneck_ok(C, dilation, H, W):
    """Is the identity-shortcut residual unit of this shape ONE launch (atvs_bottleneck_b_f32)?"""
    return (cfg.bottleneck and cfg.force_impl is None and cfg.conv1x1 and cfg.conv2d_lds and split_on('btl') and split_on('c1b')
            and split_on('c2b') and H >= 8 and W >= 16 and bool(_lib.lib().atvs_bottleneck_b_supported(int(C), int(dilation))))


def bottleneck(x, in_params, keys, w1, b1, w2, b2, w3, b3, dilation=1, want_stats=True):
    """Network.bottleneck with an identity shortcut (reference cnn_wrapper/network.py:552-602) in one launch:
    y = x + conv3(relu(conv2(relu(conv1(relu(bn(x))) + b1)) + b2)) + b3 for x (G,H,W,C); in_params (G,3,C) = the pre-activation
    batch norm's parameters (bn_params of x's moments with the unit's beta).  keys = the pack-cache keys of the three kernels
    (the unfused path's: the arranged weights are shared).  Returns (y, Stats of y) -- the next unit's moments."""
    G, H, W, C = x.shape
    k1, k2, k3 = keys
    p1, p3 = pack_conv1x1(k1, w1, C, x.device), pack_conv1x1(k3, w3, C, x.device)
    p2 = pack_conv2d_lds(k2, w2, x.device)
    if p1.kind != '_b' or p3.kind != '_b' or p2.kind != 'b' or (p1.cout, p2.cout, p3.cout) != (C, C, C):
        raise ValueError('bottleneck: the fused unit takes the split-operand packs of three C -> C kernels')
    y = _new(x, x.shape)
    st, sbuf = None, None
    if want_stats:
        rows = int(_lib.lib().atvs_bottleneck_b_rows(int(C), int(H), int(W)))
        sbuf = torch.empty((G, rows, 2, C), dtype=torch.float64, device=x.device)
        st = Stats()
        st.partial, st.blocks, st.cpad, st.count, st.groups = sbuf, rows, C, H * W, G
    if _dev_ok(x, y, in_params, b1, b2, b3):
        if in_params.numel() != G * 3 * C:
            raise ValueError('bottleneck: in_params must be (groups, 3, C)')
        with _Timed(k2, (1, H, W, C), C, G):
            _call('atvs_bottleneck_b_f32', _p(x), _p(in_params), _p(p1.wp), _p(b1), _p(p2.wp), _p(b2), _p(p3.wp), _p(b3), _p(y),
                  ctypes.c_void_p(sbuf.data_ptr()) if sbuf is not None else ctypes.c_void_p(0), G, H, W, C, int(dilation),
                  _stream())
    return (y, st) if want_stats else y


def conv2d_tail_ok(C, dilation, H, W):
    """Do a residual unit's conv2 (3x3, dilated) and conv3 (1x1) of this shape run as ONE launch (atvs_conv2d_b_tail_f32)?"""
    return (cfg.bottleneck and cfg.force_impl is None and cfg.conv1x1 and cfg.conv2d_lds and split_on('btl') and split_on('c1b')
            and split_on('c2b') and H >= 8 and W >= 16 and bool(_lib.lib().atvs_conv2d_b_tail_supported(int(C), int(dilation))))


def conv2d_tail(x, keys, w2, b2, w3, b3, residual=None, dilation=1, want_stats=True):
    """y = conv3_1x1(relu(conv2_3x3_dil(x) + b2)) + b3 [+ residual] for x (G,H,W,C): conv2 and conv3 of Network.bottleneck
    (reference cnn_wrapper/network.py:585-601) in one launch.  keys = the pack-cache keys of the two kernels (the unfused path's).
    Returns (y, Stats of y)."""
    G, H, W, C = x.shape
    p2, p3 = pack_conv2d_lds(keys[0], w2, x.device), pack_conv1x1(keys[1], w3, C, x.device)
    if p2.kind != 'b' or p3.kind != '_b' or (p2.cout, p3.cout) != (C, C):
        raise ValueError('conv2d_tail: the split-operand packs of two C -> C kernels')
    y = _new(x, x.shape)
    st, sbuf = None, None
    if want_stats:
        rows = int(_lib.lib().atvs_conv2d_lds_rows(H, W, C))
        sbuf = torch.empty((G, rows, 2, C), dtype=torch.float64, device=x.device)
        st = Stats()
        st.partial, st.blocks, st.cpad, st.count, st.groups = sbuf, rows, C, H * W, G
    if _dev_ok(x, y, b2, b3, residual):
        with _Timed(keys[0], (1, H, W, C), C, G):
            _call('atvs_conv2d_b_tail_f32', _p(x), _p(p2.wp), _p(b2), _p(p3.wp), _p(b3), _p(residual), _p(y),
                  ctypes.c_void_p(sbuf.data_ptr()) if sbuf is not None else ctypes.c_void_p(0), G, H, W, C, int(dilation), _stream())
    return (y, st) if want_stats else y


def conv_xp_launch(x5, pk, y, y_coff, bias=None, relu=False, stats_buf=None, plane_bias=None, sibling=None,
                   prologue=None, planar=False, ldy=None, y_gstride=0, y_off=0, pieces=False):
    """One x-pair launch (atvs_conv_xb_f32 / atvs_conv_xw_f32): x5 (G,D,H,W,Cin) -> y (G,D,H,W,ldy)[..., y_coff:y_coff+8].
    sibling = (pk2, y2, y_coff2, stats_buf2, plane_bias2): the stride-2 16-channel convolution of the same x5.
    prologue = (x2 | None, params | None, params2 | None, relu, relu2): the input is formed on load as
    act(bn(x5)) [+ act(bn(x2))] (include/atvsnet_hip.h)."""
    if planar:
        (G, K), (D, H, W) = x5.shape[:2], planar
        Cin = K * 8
    else:
        G, D, H, W, Cin = x5.shape
    ldy = y.shape[-1] if ldy is None else int(ldy)      # ldy / y_gstride given: y is a plane of a chunk-planar buffer (xb only)
    null = ctypes.c_void_p(0)
    sp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else null     # noqa: E731
    pk2, y2, y_coff2, sbuf2, pb2 = sibling if sibling is not None else (None, None, 0, None, None)
    x2, ipa, ipb, relu_a, relu_b = prologue if prologue is not None else (None, None, None, False, False)
    if x2 is not None and (tuple(x2.shape) != tuple(x5.shape) or not x2.is_contiguous()):
        raise ValueError('conv_xp: the second source must have the shape of the first')
    for ip in (ipa, ipb):
        if ip is not None and (ip.numel() != G * 3 * Cin or not ip.is_contiguous()):
            raise ValueError('conv_xp: prologue parameters must be (groups, 3, Cin)')
    kind = pk.kind
    if pk2 is not None and pk2.kind != kind:
        raise ValueError('conv_xp: the main and the sibling weights are packed for different kernels')
    if (y_gstride or (planar and prologue is not None)) and kind != 'xb':
        raise ValueError('conv_xp: a strided output / a prologue over a chunk-planar input belong to the split-fp16 kernel')
    if _dev_ok(x5, y, bias, plane_bias, y2, pb2, x2, ipa, ipb):
        with _Timed(pk.key, (D, H, W, Cin), pk.cout + (16 if pk2 is not None else 0), G):
            yp = ctypes.c_void_p(y.data_ptr() + 4 * int(y_off))     # y_off: floats into a chunk-planar buffer (with ldy / y_gstride)
            args = [_p(x5), _p(pk.wp), _p(bias), _p(plane_bias), yp, sp(stats_buf), G, D, H, W, Cin,
                    ldy, int(y_coff), int(bool(relu)), _p(pk2.wp) if pk2 is not None else null, _p(pb2), _p(y2),
                    sp(sbuf2), int(y2.shape[-1]) if y2 is not None else 0, int(y_coff2), _p(x2), _p(ipa), _p(ipb),
                    int(bool(relu_a)), int(bool(relu_b))]
            if kind == 'xb':
                _call('atvs_conv_xb_f32', *(args + [ctypes.c_long(planar_stride(D, H, W) if planar else 0),
                                                    ctypes.c_long(int(y_gstride)), int(bool(pieces)), _stream()]))
            else:
                _call('atvs_conv_xw_f32', *(args + [ctypes.c_long(planar_stride(D, H, W) if planar else 0), _stream()]))


def xp_blocks(D, H, W, groups=1):
    """Workgroups per sample of an x-pair launch."""
    return int(_lib.lib().atvs_conv_xpair_grid(int(D), int(H), int(W), int(groups)))






def tiled_tile_y(H, W, cout):
    """tile_y for the LDS-tiled kernel, or 0 when the gather kernel should be used."""
    if cfg.force_impl == 'gather':
        return 0
    nt = 1
    while nt * 16 < cout:
        nt *= 2
    if W < 12 and cfg.force_impl != 'tiled':
        return 0
    if nt <= 2 and H >= 16:
        return 8
    if nt <= 4:
        return 4
    return 0






_fin_pool = {}
# Measured on MI355X (cfg3 pipeline, HIP-graph replay): 51.4 ms/depth-map with the separate 5-us finalize
# launches, 54.5 ms with the in-launch last-arriver finalize (the arrival drains every workgroup's output
# stores and the reducing workgroup runs alone at the tail) -> off by default, kept and tested.




def _fin_counter(device):
    """A zero device word for one layer's arrival ticket (the kernel leaves it at zero again)."""
    key = str(device)
    ent = _fin_pool.get(key)
    if ent is None:
        ent = [torch.zeros(8192, dtype=torch.int32, device=device), 0]
        _fin_pool[key] = ent
    ent[1] = (ent[1] + 1) % 8192
    return ent[0][ent[1]:ent[1] + 1]


class Fin(object):
    """In-launch finalize request for one layer (all its launches share it)."""
    __slots__ = ('counter', 'params', 'stats', 'rows', 'arrivals', 'channels', 'fold', 'count')


def conv_tiled_launch(x5, pk, y, out_stride, out_off, y_coff, tile_y, bias=None, residual=None, relu=False,
                      stats_buf=None, plane_bias=None, class_cout=0, class_base=0, xpair=False, fin=None):
    """One atvs_conv_tiled_f32 launch: logical output grid = input grid of x5 (G,D,H,W,Cin); y (G,Dy,Hy,Wy,ldy)."""
    G, D, H, W, Cin = x5.shape
    Dy, Hy, Wy, ldy = y.shape[-4:]
    if _dev_ok(x5, y, bias, residual, plane_bias):
        with _Timed(pk.key, x5.shape[1:], pk.cout, G):
            _call('atvs_conv_tiled_f32', _p(x5), _p(pk.wp), ctypes.c_void_p(pk.tab.data_ptr()), _p(bias), _p(residual),
                  _p(plane_bias), _p(y),
                  ctypes.c_void_p(stats_buf.data_ptr()) if stats_buf is not None else ctypes.c_void_p(0), G, D, H, W,
                  Cin, Dy, Hy, Wy, int(out_stride), int(out_off[0]), int(out_off[1]), int(out_off[2]), ldy, int(y_coff),
                  8 if xpair else pk.cout, pk.ntaps, int(tile_y), int(bool(relu)), int(class_cout), int(class_base),
                  int(bool(xpair)),
                  ctypes.c_void_p(fin.counter.data_ptr()) if fin is not None else ctypes.c_void_p(0),
                  _p(fin.params) if fin is not None else ctypes.c_void_p(0),
                  ctypes.c_void_p(fin.stats.data_ptr()) if fin is not None else ctypes.c_void_p(0),
                  fin.rows if fin is not None else 0, fin.arrivals if fin is not None else 0,
                  fin.channels if fin is not None else 0, fin.fold if fin is not None else 0,
                  ctypes.c_long(fin.count if fin is not None else 0), ctypes.c_float(1e-3), _stream())


def tiled_blocks(D, H, W, tile_y, cin, cout, xpair=False, groups=1):
    """Workgroups PER SAMPLE (= statistics rows per sample) of a tiled launch: its share of the persistent grid."""
    return int(_lib.lib().atvs_conv_tiled_num_blocks(int(D), int(H), int(W), int(tile_y), int(cin), int(cout),
                                                     int(bool(xpair)), int(groups)))


def tiled_nsplit(D, H, W, tile_y, cin, cout, xpair=False, groups=1):
    ns = ctypes.c_int()
    _lib.lib().atvs_conv_tiled_grid(int(D), int(H), int(W), int(tile_y), int(cin), int(cout), int(bool(xpair)),
                                    int(groups), ctypes.byref(ns))
    return ns.value


_xp_cache = {}


def _xpair_virtual_kernel(key, w_host):
    """Dense virtual kernel of the x-pair form: (36 taps (kd,kh,ox in -1..2), Cin, (jx, co)) with
    Wv[(kd,kh,ox)][ci][jx*8+co] = W[kd][kh][kw = ox - jx + 1][ci][co] (0 when kw is outside 0..2)."""
    import numpy as np
    hit = _xp_cache.get(key)
    if hit is not None:
        return hit
    w = np.asarray(w_host, np.float32)               # [3,3,3,Cin,8]
    cin = w.shape[3]
    wv = np.zeros((3, 3, 4, cin, 2, 8), np.float32)
    for oi in range(4):
        for jx in range(2):
            kw = (oi - 1) - jx + 1
            if 0 <= kw <= 2:
                wv[:, :, oi, :, jx, :] = w[:, :, kw]
    hit = wv.reshape(36, cin, 16)
    _xp_cache[key] = hit
    return hit


XPAIR_TAPS = tuple(((kd * 3 + kh) * 4 + oi, kd - 1, kh - 1, oi - 1) for kd in range(3) for kh in range(3)
                   for oi in range(4))




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


def _pick_tile_m(M, ntiles):
    """Largest voxel-tile count per wavefront that still leaves >= 1024 workgroups (4 per CU)."""
    for tm in (8, 4, 2):
        if tm * ntiles <= 16 and -(-M // (64 * tm)) >= 1024:
            return tm
    return 1


class Stats(object):
    """Per-workgroup partial sums of a tensor: feeds bn_finalize."""
    __slots__ = ('partial', 'blocks', 'cpad', 'count', 'fold', 'params', 'groups')

    def __init__(self):
        self.fold = 1
        self.params = None      # (3,C) moments already finished inside the producing launch
        self.groups = 1         # independent samples: partial is (groups, blocks, 2, cpad), count per sample


def conv_launch(x5, pk, y, out_grid, in_stride, out_stride, out_off, y_coff, bias=None, residual=None, relu=False,
                stats_buf=None, tile_m=None, plane_bias=None, pad_z=0):
    """One atvs_conv_mfma_f32 launch.  x5: (G,Di,Hi,Wi,Cin); y: full output (G,Dy,Hy,Wy,ldy)."""
    G, Di, Hi, Wi, Cin = x5.shape
    Dy, Hy, Wy, ldy = y.shape[-4:]
    Do, Ho, Wo = out_grid
    M = Do * Ho * Wo
    tm = tile_m or _pick_tile_m(M * G, pk.ntiles)
    if _dev_ok(x5, y, bias, residual, plane_bias):
        args = [_p(x5), _p(pk.wp), ctypes.c_void_p(pk.tab.data_ptr()), _p(bias), _p(residual), _p(plane_bias),
                int(pad_z), _p(y),
                ctypes.c_void_p(stats_buf.data_ptr()) if stats_buf is not None else ctypes.c_void_p(0),
                G, Di, Hi, Wi, Cin, Do, Ho, Wo, int(in_stride), Dy, Hy, Wy, int(out_stride), int(out_off[0]),
                int(out_off[1]), int(out_off[2]), ldy, int(y_coff), pk.cout, pk.ntaps, tm, int(bool(relu)), _stream()]
        with _Timed(pk.key, x5.shape[1:], pk.cout, G):
            _call('atvs_conv_mfma_f32', *args)
    return tm


_watch = {'tag': None, 'events': []}


def _watched(key):
    tag = _watch['tag']
    return tag is not None and (tag == '*' or key == tag or (isinstance(tag, list) and key in tag))


class _Timed(object):
    """HIP events around one launch on the launch stream (= torch's current stream), when `key` is watched."""

    def __init__(self, key, shape, cout, groups=1):
        self.on = _watched(key)
        self.info = (key, tuple(shape), cout, int(groups))

    def __enter__(self):
        if self.on:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.e1.record()
            _watch['events'].append((self.e0, self.e1) + self.info)
        return False


def watch(tag):
    """Time launches with HIP events on the launch stream.  tag: a convolution's weight key, ('warp', mode) for
    atvs_warp_planes, a LIST of such keys, or '*' (every convolution launch).  watch(None) stops and returns the
    durations in ms: a list for one key, {key: [(ms, samples in the launch)]} for a list of keys,
    [(key, input shape, Cout, ms)] for '*'."""
    out = None
    if tag is None:
        torch.cuda.synchronize()
        ev, old = _watch['events'], _watch['tag']
        if old == '*':
            out = [(e[2], e[3], e[4], e[0].elapsed_time(e[1])) for e in ev]
        elif isinstance(old, list):       # per key: (ms, independent samples in the launch)
            out = {k: [(e[0].elapsed_time(e[1]), e[5]) for e in ev if e[2] == k] for k in old}
        else:
            out = [e[0].elapsed_time(e[1]) for e in ev]
    _watch['tag'] = tag
    _watch['events'] = []
    return out


def conv_blocks(M, ntiles, tile_m=None, groups=1):
    """(workgroups per sample, tile_m) of a gather launch."""
    tm = tile_m or _pick_tile_m(M * groups, ntiles)
    return -(-M // (64 * tm)), tm


def _stats_buffer(ref, blocks, cpad, zero=False, groups=1):
    f = torch.zeros if zero else torch.empty
    return f((groups, blocks, 2, cpad), dtype=torch.float64, device=ref.device)


def _to5(x, groups, what='tensor'):
    """Canonical (G,D,H,W,C) view of a channel-last tensor.  groups=None: x is one sample, (H,W,C) or (D,H,W,C);
    groups=G: x is G independent samples stacked on a leading axis, (G,H,W,C) or (G,D,H,W,C).  -> (x5, nsp)."""
    if groups is None:
        nsp = x.dim() - 1
        lead = (1,)
        rest = tuple(x.shape)
    else:
        nsp = x.dim() - 2
        if x.shape[0] != groups:
            raise ValueError('%s: leading axis %d, groups %d' % (what, x.shape[0], groups))
        lead = (int(groups),)
        rest = tuple(x.shape[1:])
    if nsp not in (2, 3):
        raise ValueError('%s: %d spatial axes' % (what, nsp))
    return x.reshape(lead + (1,) * (3 - nsp) + rest), nsp


def _from5(y5, nsp, groups):
    shape = tuple(y5.shape[4 - nsp:])
    return y5.reshape(shape if groups is None else (y5.shape[0],) + shape)


def conv(x, key, w_host, stride=1, dilation=1, padding='SAME', explicit_pad=None, bias=None, residual=None,
         relu=False, want_stats=False, out=None, y_coff=0, plane_bias=None, groups=None, in_params=None,
         in_relu=False, in_sum=None):
    """Forward convolution of a channel-last tensor x: (H,W,C) or (D,H,W,C); with groups=G, G independent samples
    (G,H,W,C) / (G,D,H,W,C) in one launch (per-sample batch-norm moments).

    w_host: TF-layout numpy kernel [k.., Cin, Cout]; `key` names it for the pack cache.
    padding: 'SAME' | 'VALID'; explicit_pad = (before, after) per spatial axis overrides it
    (bottleneck conv2, network.py:589-595).  in_params (G,3,Cin) [+ in_relu]: x is a raw convolution output whose
    batch norm is applied on load (only where the kernel of this shape supports it: norm_on_load_2d_ok / norm_on_load_3d_ok;
    a 3-D shape without such a form gets the normalised tensor from a bn_apply pass here).  in_sum = (x1, params1 | None,
    relu1) (3-D 16 -> 16 only): the input is x [normalised by in_params] + x1 [normalised by params1], formed on load.
    Returns y or (y, Stats).
    """
    x5, nsp = _to5(x, groups, 'conv input')
    G = x5.shape[0]
    ks = (1,) * (3 - nsp) + tuple(int(k) for k in w_host.shape[:nsp])
    ins = tuple(x5.shape[1:4])
    pads, outs = [], []
    for i in range(3):
        if ks[i] == 1 and i < 3 - nsp:
            pads.append(0)
            outs.append(1)
            continue
        ax = i - (3 - nsp)
        if explicit_pad is not None:
            pb, pe = explicit_pad[ax]
            pads.append(pb)
            outs.append((ins[i] + pb + pe - ((ks[i] - 1) * dilation + 1)) // stride + 1)
        elif padding == 'SAME':
            pb, o = same_pad(ins[i], ks[i], stride, dilation)
            pads.append(pb)
            outs.append(o)
        else:
            pads.append(0)
            outs.append((ins[i] - ((ks[i] - 1) * dilation + 1)) // stride + 1)
    taps = conv_taps(ks, dilation, pads)
    cin, cout = int(w_host.shape[-2]), int(w_host.shape[-1])
    if cin != x5.shape[4]:
        raise ValueError('conv %s: input has %d channels, kernel wants %d' % (key, x5.shape[4], cin))
    y5 = None
    if out is not None:
        y5, _ = _to5(out, groups, 'conv output buffer')
        if tuple(y5.shape[:4]) != (G,) + tuple(outs):
            raise ValueError('conv %s: output buffer %s does not match %s' % (key, tuple(y5.shape), (G,) + tuple(outs)))
    res5 = _to5(residual, groups, 'residual')[0] if residual is not None else None
    if plane_bias is not None:
        pb_shape = ((G,) if groups is not None else ()) + (outs[1], outs[2], 3 * cout)
        if tuple(plane_bias.shape) != pb_shape:
            raise ValueError('conv %s: plane_bias %s, expected %s' % (key, tuple(plane_bias.shape), pb_shape))
    M = outs[0] * outs[1] * outs[2]

    # ---- 2-D, 3x3, stride 1, SAME on wide channels: the LDS-tiled tower kernel
    if nsp == 2 and stride == 1 and ks == (1, 3, 3) and tuple(pads[1:]) == (dilation, dilation) \
            and tuple(outs) == ins and plane_bias is None \
            and (y5 is None or (y5.shape[-1] % 4 == 0 and y_coff % 4 == 0)) \
            and conv2d_lds_ok(cin, cout, dilation, ins[1], ins[2]):
        r = conv2d_lds(x5[:, 0], key, w_host, dilation, bias, None if res5 is None else res5[:, 0], relu, want_stats,
                       None if y5 is None else y5[:, 0], y_coff, in_params, in_relu)
        y4, st = r if want_stats else (r, None)
        y = out if out is not None else _from5(y4.unsqueeze(1), nsp, groups)
        return (y, st) if want_stats else y
    # ---- 2-D, 3x3, stride 2 behind explicit symmetric padding 1 (the strided conv2 of a residual unit's first block)
    if nsp == 2 and stride == 2 and dilation == 1 and ks == (1, 3, 3) and explicit_pad is not None \
            and tuple(tuple(int(v) for v in pr) for pr in explicit_pad) == ((1, 1), (1, 1)) and ins[1] % 2 == 0 and ins[2] % 2 == 0 \
            and residual is None and plane_bias is None and in_params is None and y5 is None and cfg.force_impl is None \
            and cfg.conv2d_lds and split_on('c2b') and ins[1] >= 16 and ins[2] >= 32 \
            and bool(_lib.lib().atvs_conv2d_b_s2_supported(int(cin), int(cout))):
        pk = pack_conv2d_lds(key, w_host, x.device)
        if pk.kind == 'b':
            Ho, Wo = ins[1] // 2, ins[2] // 2
            y4 = _new(x, (G, Ho, Wo, cout))
            st, sbuf = None, None
            if want_stats:
                rows = int(_lib.lib().atvs_conv2d_lds_rows(Ho, Wo, cout))
                sbuf = torch.empty((G, rows, 2, cout), dtype=torch.float64, device=x.device)
                st = Stats()
                st.partial, st.blocks, st.cpad, st.count, st.groups = sbuf, rows, cout, Ho * Wo, G
            if _dev_ok(x5, y4, bias):
                with _Timed(key, (1, ins[1], ins[2], cin), cout, G):
                    _call('atvs_conv2d_b_s2_f32', _p(x5), _p(pk.wp), _p(bias), _p(y4),
                          ctypes.c_void_p(sbuf.data_ptr()) if sbuf is not None else ctypes.c_void_p(0), G, ins[1], ins[2], cin, cout,
                          int(bool(relu)), _stream())
            y = _from5(y4.unsqueeze(1), nsp, groups)
            return (y, st) if want_stats else y
    # ---- 2-D 1x1, stride 1: the LDS-staged GEMM kernel
    if nsp == 2 and stride == 1 and ks == (1, 1, 1) and plane_bias is None and conv1x1_ok(cin, cout) \
            and (y5 is None or (y5.shape[-1] % 4 == 0 and y_coff % 4 == 0)):
        r = conv1x1(x5[:, 0], key, w_host, bias, None if res5 is None else res5[:, 0], relu, want_stats,
                    None if y5 is None else y5[:, 0], y_coff, in_params, in_relu)
        y4, st = r if want_stats else (r, None)
        y = out if out is not None else _from5(y4.unsqueeze(1), nsp, groups)
        return (y, st) if want_stats else y
    if (in_params is not None or in_sum is not None) and nsp == 2:
        raise ValueError('conv %s: no normalise-on-load form for this shape' % (key,))
    # 3-D: which split-operand kernel (if any) takes this shape, and does it form a lazy input on load?
    c16_shape = nsp == 3 and stride == 1 and dilation == 1 and ks == (3, 3, 3) and tuple(pads) == (1, 1, 1) \
        and residual is None and plane_bias is None and cfg.conv_c16 and cfg.force_impl is None \
        and tuple(outs) == ins and ins[2] >= 12 and 4.0 * M * (cout if y5 is None else y5.shape[-1]) < 2.0 ** 32 \
        and (y5 is None or (y5.shape[-1] % 4 == 0 and y_coff % 4 == 0))
    b3 = split_on('c3b') and cin % 16 == 0 and cout in (32, 64) and bool(_lib.lib().atvs_conv3d_b_supported(int(cin), int(cout)))
    b16 = split_on('c16b') and cin in (8, 16) and cout == 16
    s2b = nsp == 3 and stride == 2 and dilation == 1 and ks == (3, 3, 3) and padding == 'SAME' and explicit_pad is None \
        and split_on('s2b') and cfg.conv_c16 and cfg.force_impl is None and residual is None and plane_bias is None \
        and bool(_lib.lib().atvs_conv3d_s2b_supported(int(cin), int(cout))) and outs[2] >= 8 \
        and 4.0 * M * (cout if y5 is None else y5.shape[-1]) < 2.0 ** 32 \
        and (y5 is None or (y5.shape[-1] % 4 == 0 and y_coff % 4 == 0))
    lazy_ok = cfg.sum_on_load and cfg.norm3d and ((c16_shape and ((b3 and in_sum is None) or (b16 and cin == 16))) or (s2b and in_sum is None))
    if (in_params is not None or in_sum is not None) and not lazy_ok:
        # no form of this shape forms its input on load: the passes the lazy input stands for, then the plain convolution
        if in_sum is not None:
            a = PendingBN(x, in_params, in_relu) if in_params is not None else x
            b = PendingBN(in_sum[0], in_sum[1], in_sum[2]) if in_sum[1] is not None else in_sum[0]
            x = PendingSum([a, b]).materialize()
        else:
            x = bn_apply(x, in_params, in_relu, out=_new(x, x.shape))
        x5, _ = _to5(x, groups, 'conv input')
        in_params, in_sum = None, None

    # ---- 3-D, 3x3x3, 1-2 input channels -> 8: the refinement stems, HBM-bound FMA kernel
    if nsp == 3 and stride == 1 and dilation == 1 and ks == (3, 3, 3) and tuple(pads) == (1, 1, 1) and cout == 8 \
            and cin <= 2 and bias is None and residual is None and cfg.stem and cfg.force_impl is None \
            and (y5 is None or (y5.shape[-1] % 4 == 0 and y_coff % 4 == 0)):
        import numpy as np
        ck = ('stem', key, str(x.device))
        wd = _pack_cache.get(ck)
        if wd is None:
            wd = _Packed()
            wd.key, wd.tab, wd.cin, wd.cout = key, None, cin, cout
            wd.wp = None if x.is_meta else torch.from_numpy(np.ascontiguousarray(w_host, dtype=np.float32)).to(x.device)
            _pack_cache[ck] = wd
        if y5 is None:
            y5 = _new(x, (G,) + tuple(outs) + (8,))
        st, sbuf = None, None
        if want_stats:
            rows = int(_lib.lib().atvs_conv_stem_rows(*outs))
            sbuf = _stats_buffer(x, rows, 16, groups=G)
            st = Stats()
            st.partial, st.blocks, st.cpad, st.count, st.groups = sbuf, rows, 16, M, G
        if _dev_ok(x5, y5, plane_bias):
            with _Timed(key, x5.shape[1:], 8, G):
                _call('atvs_conv_stem_f32', _p(x5), _p(wd.wp), _p(plane_bias), _p(y5),
                      ctypes.c_void_p(sbuf.data_ptr()) if sbuf is not None else ctypes.c_void_p(0), G, outs[0], outs[1],
                      outs[2], cin, int(y5.shape[-1]), int(y_coff), int(bool(relu)), _stream())
        y = out if out is not None else _from5(y5, nsp, groups)
        return (y, st) if want_stats else y

    # ---- 3-D, 3x3x3, 8 / 16 / 32 -> 16 and 16..64 -> 32 channels: one workgroup per CU, fully unrolled (the half- and
    # quarter-resolution U-Net layers, the AANet modules' shared | unique convolution)
    if c16_shape and ((cout == 16 and cin in (8, 16, 32)) or (cout == 32 and cin in (16, 32, 48, 64)) or b3):
        pk = pack_conv3d_b(key, w_host, x.device) if b3 else \
            pack_conv_c16b(key, w_host, x.device) if b16 else pack_conv_c16(key, w_host, x.device)
        if y5 is None:
            y5 = _new(x, (G,) + tuple(outs) + (cout,))
        st, sbuf = None, None
        if want_stats:
            rows = int(_lib.lib().atvs_conv_c16_grid(outs[0], outs[1], outs[2], G))
            sbuf = _stats_buffer(x, rows, cout, groups=G)
            st = Stats()
            st.partial, st.blocks, st.cpad, st.count, st.groups = sbuf, rows, cout, M, G
        if _dev_ok(x5, y5, bias):
            with _Timed(key, x5.shape[1:], cout, G):
                if b3 and in_params is not None:
                    _call('atvs_conv3d_b_norm_f32', _p(x5), _p(in_params), int(bool(in_relu)), _p(pk.wp), _p(bias), _p(y5), _p(sbuf),
                          G, outs[0], outs[1], outs[2], cin, cout, int(y5.shape[-1]), int(y_coff), int(bool(relu)), _stream())
                elif b3:
                    _call('atvs_conv3d_b_f32', _p(x5), _p(pk.wp), _p(bias), _p(y5), _p(sbuf), G, outs[0], outs[1], outs[2], cin,
                          cout, int(y5.shape[-1]), int(y_coff), int(bool(relu)), _stream())
                elif b16 and (in_params is not None or in_sum is not None):
                    x1, p1, r1 = in_sum if in_sum is not None else (None, None, False)
                    mask = (1 if in_relu else 0) | (2 if r1 else 0)
                    _call('atvs_conv_c16b_sum_f32', _p(x5), _p(in_params), _p(x1), _p(p1), int(mask), _p(pk.wp), _p(bias), _p(y5),
                          _p(sbuf), G, outs[0], outs[1], outs[2], cin, int(y5.shape[-1]), int(y_coff), int(bool(relu)), _stream())
                elif b16:
                    _call('atvs_conv_c16b_f32', _p(x5), _p(pk.wp), _p(bias), _p(y5), _p(sbuf), G, outs[0], outs[1], outs[2], cin,
                          int(y5.shape[-1]), int(y_coff), int(bool(relu)), _stream())
                else:
                    _call('atvs_conv_c16_f32', _p(x5), _p(pk.wp), _p(bias), _p(y5), _p(sbuf), G, outs[0], outs[1], outs[2], cin,
                          cout, int(y5.shape[-1]), int(y_coff), int(bool(relu)), _stream())
        y = out if out is not None else _from5(y5, nsp, groups)
        return (y, st) if want_stats else y

    # ---- 3-D, 3x3x3, stride 2, SAME, 16 k -> 32 / 64 channels: the U-Net encoders below half resolution on split-fp16 operands
    if s2b:
        pk = pack_conv3d_b(key, w_host, x.device, kind='s2b')
        if y5 is None:
            y5 = _new(x, (G,) + tuple(outs) + (cout,))
        st, sbuf = None, None
        if want_stats:
            rows = int(_lib.lib().atvs_conv3d_s2b_grid(outs[0], outs[1], outs[2], G))
            sbuf = _stats_buffer(x, rows, cout, groups=G)
            st = Stats()
            st.partial, st.blocks, st.cpad, st.count, st.groups = sbuf, rows, cout, M, G
        if _dev_ok(x5, y5, bias):
            with _Timed(key, x5.shape[1:], cout, G):
                if in_params is not None:
                    _call('atvs_conv3d_s2b_norm_f32', _p(x5), _p(in_params), int(bool(in_relu)), _p(pk.wp), _p(bias), _p(y5),
                          _p(sbuf), G, ins[0], ins[1], ins[2], cin, cout, int(y5.shape[-1]), int(y_coff), int(bool(relu)), _stream())
                else:
                    _call('atvs_conv3d_s2b_f32', _p(x5), _p(pk.wp), _p(bias), _p(y5), _p(sbuf), G, ins[0], ins[1], ins[2], cin,
                          cout, int(y5.shape[-1]), int(y_coff), int(bool(relu)), _stream())
        y = out if out is not None else _from5(y5, nsp, groups)
        return (y, st) if want_stats else y

    tile_y = 0
    if nsp == 3 and stride == 1 and dilation == 1 and ks == (3, 3, 3) and tuple(pads) == (1, 1, 1) \
            and tuple(outs) == ins:
        tile_y = tiled_tile_y(ins[1], ins[2], cout)
    xpair = bool(tile_y) and cfg.xpair and cout == 8 and (ins[2] >= 24 or cfg.force_impl == 'tiled') \
        and (y5 is None or (y5.shape[-1] % 4 == 0 and y_coff % 4 == 0))
    xp1w = xpair and cfg.xp1w and cin % 8 == 0 and residual is None
    if xp1w:
        pk = pack_conv_xp(key, w_host, x.device)
    elif xpair:
        tile_y = 4 if cin > 8 else tile_y
        pk = pack_conv_weights_tiled(key, _xpair_virtual_kernel(key, w_host), XPAIR_TAPS, False, x.device, tile_y, True)
        pk.cout = 8
    elif tile_y:
        pk = pack_conv_weights_tiled(key, w_host, taps, False, x.device, tile_y)
    else:
        pk = pack_conv_weights(key, w_host, taps, False, x.device)
    if y5 is None:
        y5 = _new(x, (G,) + tuple(outs) + (pk.cout,))
    if xp1w:
        blocks, tm = xp_blocks(outs[0], outs[1], outs[2], G), 0
    elif tile_y:
        blocks, tm = tiled_blocks(outs[0], outs[1], outs[2], tile_y, pk.cin, pk.cout, xpair, G), 0
    else:
        blocks, tm = conv_blocks(M, pk.ntiles, groups=G)
    st = None
    sbuf = None
    if want_stats:
        sbuf = _stats_buffer(x, blocks, pk.ntiles * 16, groups=G)
        st = Stats()
        st.partial, st.blocks, st.cpad, st.count, st.groups = sbuf, blocks, pk.ntiles * 16, M, G
    if xp1w:
        conv_xp_launch(x5, pk, y5, y_coff, bias, relu, sbuf, plane_bias)
    elif tile_y:
        fin = None
        if want_stats and cfg.fused_finalize and pk.cout <= 64 and not x.is_meta and G == 1:
            fin = Fin()
            fin.counter, fin.params, fin.stats = _fin_counter(x.device), _new(x, (3, pk.cout)), sbuf
            fin.rows, fin.arrivals, fin.channels, fin.fold, fin.count = blocks, blocks, pk.cout, 1, M
            st.params = fin.params
        conv_tiled_launch(x5, pk, y5, 1, (0, 0, 0), y_coff, tile_y, bias, res5, relu, sbuf, plane_bias, xpair=xpair,
                          fin=fin)
    else:
        conv_launch(x5, pk, y5, outs, stride, 1, (0, 0, 0), y_coff, bias, res5, relu, sbuf, tm, plane_bias, pads[0])
    y = out if out is not None else _from5(y5, nsp, groups)
    return (y, st) if want_stats else y


def conv3d_8to1(x, w_dev, groups=None):
    """3x3x3 SAME convolution (D,H,W,8) -> (D,H,W,1) (groups=G: (G,D,H,W,8) -> (G,D,H,W,1)); w_dev: device tensor
    of the TF kernel [3,3,3,8,1]."""
    x5, nsp = _to5(x, groups, 'conv3d_8to1 input')
    G, D, H, W, C = x5.shape
    if nsp != 3 or C != 8 or w_dev.numel() != 216:
        raise ValueError('conv3d_8to1: a volume with 8 input channels and a [3,3,3,8,1] kernel')
    y = _new(x, (D, H, W, 1) if groups is None else (G, D, H, W, 1))
    if _dev_ok(x, w_dev):
        _call('atvs_conv3d_8to1', _p(x), _p(w_dev), _p(y), G, D, H, W, _stream())
    return y


class SplitVolume(object):
    """A (B,D,h,w,C) network input whose channels are a concat of D-varying and D-constant parts.

    Stands for tf.concat([...tf.tile(x, [1,D,1,1,1])...], -1) of model.py:186-195, 329-336 without
    materialising the tiled parts.  var: (B,D,h,w,Cv); const: (B,h,w,Cc) (B = independent samples; a 4-D var /
    3-D const is one sample); chan_map: for each channel of the reference's concat, ('v', i) or ('c', i) --
    several channels may map to the same source (the 16 identical geo-view channels, quirk C7).
    planar=(D,h,w): var is stored chunk-planar, a (B, Cv/8, planar_stride(D,h,w)) buffer whose rows hold (D,h,w,8)
    (warp_planes(planar=True)): conv_split_siblings hands it to the x-pair kernel as it is, every other consumer gets the
    channel-last copy var_cl() makes."""

    def __init__(self, var, const, chan_map, planar=False, pieces=False):
        self.planar = tuple(int(v) for v in planar) if planar else False
        self.pieces = bool(pieces)               # (with planar) the buffer holds fp16 pieces: warp_planes(pieces=True)
        if self.pieces and not self.planar:
            raise ValueError('SplitVolume(pieces=True) needs planar=(D,h,w)')
        if var.dim() == (2 if self.planar else 4):
            var, const = var.unsqueeze(0), const.unsqueeze(0)
        self._var, self.const, self.chan_map = var, const, list(chan_map)
        self._cl = None
        self.device = var.device

    @property
    def var(self):
        """The D-varying part channel-last, (B,D,h,w,Cv)."""
        return self.var_cl()

    def var_cl(self):
        if not self.planar:
            return self._var
        if self._cl is None:
            B, K = self._var.shape[:2]
            D, h, w = self.planar
            pv = planar_pieces_decode(self._var, D, h, w) if self.pieces else planar_view(self._var, D, h, w)
            self._cl = pv.permute(0, 2, 3, 4, 1, 5).reshape(B, D, h, w, K * 8).contiguous()
        return self._cl

    @property
    def cv(self):
        """Channels of the D-varying part."""
        return self._var.shape[1] * 8 if self.planar else self._var.shape[-1]

    @property
    def shape(self):
        if self.planar:
            B, (D, h, w) = self._var.shape[0], self.planar
        else:
            B, D, h, w, _ = self._var.shape
        return (B, D, h, w, len(self.chan_map))

    def dim(self):
        return 5

    @property
    def is_meta(self):
        return self._var.is_meta

    def materialize(self):
        """The dense (B,D,h,w,C) tensor the reference would build."""
        B, D, h, w, _ = self.var.shape
        C = len(self.chan_map)
        out = _new(self.var, (B, D, h, w, C))
        for b in range(B):
            for ch, (kind, i) in enumerate(self.chan_map):
                if kind == 'v':
                    copy_channels(self.var[b], out[b], 1, i, ch)
                else:
                    src = _new(self.const, (h, w, 1))
                    copy_channels(self.const[b], src, 1, i, 0)
                    tile_planes(src, out[b], ch)
        return out


_fold_cache = {}


def _fold_split_weights(key, w_host, chan_map, cv, cc):
    """W (3,3,3,C,Cout) -> (W_var (3,3,3,Cv,Cout), W_planes (3,3,Cc,3*Cout)); cached per key."""
    import numpy as np
    ck = (key, tuple(chan_map))
    hit = _fold_cache.get(ck)
    if hit is not None:
        return hit
    w = np.asarray(w_host, np.float32)
    cout = w.shape[-1]
    wv = np.zeros((3, 3, 3, cv, cout), np.float32)
    wc = np.zeros((3, 3, 3, cc, cout), np.float32)
    for ch, (kind, i) in enumerate(chan_map):
        (wv if kind == 'v' else wc)[:, :, :, i, :] += w[:, :, :, ch, :]
    # the three sets of in-range kd taps: [kd=0 missing | all | kd=2 missing]
    planes = np.concatenate([wc[1] + wc[2], (wc[0] + wc[1]) + wc[2], wc[0] + wc[1]], axis=-1)
    hit = (wv, np.ascontiguousarray(planes))
    _fold_cache[ck] = hit
    return hit


def conv_split(sv, key, w_host, stride=1, want_stats=False, out=None, y_coff=0):
    """3x3x3 SAME convolution of a SplitVolume (B samples): conv3d over the D-varying channels plus the 2-D
    convolution of the D-constant channels (kd-summed kernel) added per depth plane in the epilogue.
    -> (B,D,h,w,Cout) [, Stats]."""
    B = sv.shape[0]
    cv, cc = sv.cv, sv.const.shape[-1]
    wv, planes = _fold_split_weights(key, w_host, sv.chan_map, cv, cc)
    pb = conv(sv.const, (key, 'planes'), planes, stride=stride, groups=B)            # (B, ho, wo, 3*Cout)
    return conv(sv.var, (key, 'var'), wv, stride=stride, want_stats=want_stats, plane_bias=pb, out=out, y_coff=y_coff,
                groups=B)






def planar_concat_ok(shape):
    """(D,h,w): should CostVolRefineNet's concat be chunk-planar?  Only when both its producer (the photo stem) and its
    consumer run on the split-fp16 x-pair kernel, which writes / reads planes."""
    D, h, w = (int(v) for v in shape)
    return (cfg.planar_concat and cfg.planar and _xkind() == 'xb' and cfg.prologue and cfg.force_impl is None
            and siblings_ok((D, h, w), 32, 8, 16) and 4.0 * 4 * planar_stride(D, h, w) < 2.0 ** 40)


def photo_pieces_ok(shape, chan):
    """(D,h,w), D-varying channels: should the refinement's photo volume be written as fp16 pieces?  When its one consumer is
    the photo stem on the split-operand x-pair kernel writing a plane of the chunk-planar concat (conv_split_into_plane)."""
    return cfg.pieces and chan in (16, 32, 64) and planar_concat_ok(shape)


def conv_split_into_plane(sv, key, w_host, buf, plane, planar):
    """conv_split (8 output channels) written into plane `plane` of the chunk-planar buffer buf (B, K, planar_stride):
    the photo stem of CostVolRefineNet as the producer of plane 0 of the concat.  -> Stats.
    sv: channel-last D-varying part, or chunk-planar fp16 pieces (SplitVolume(planar, pieces): warp_planes(mode=1, pieces))."""
    D, H, W = planar
    B, K, pstride = buf.shape
    if (sv.planar and (not sv.pieces or sv.planar != (D, H, W))) or sv.shape[0] != B or tuple(sv.shape[1:4]) != (D, H, W) or pstride != planar_stride(D, H, W) \
            or not buf.is_contiguous() or int(w_host.shape[-1]) != 8 or sv.cv % 8:
        raise ValueError('conv_split_into_plane: shapes')
    cv, cc = sv.cv, sv.const.shape[-1]
    wv, planes = _fold_split_weights(key, w_host, sv.chan_map, cv, cc)
    pb = conv(sv.const, (key, 'planes'), planes, stride=1, groups=B)                   # (B, h, w, 24)
    pk = pack_conv_xp((key, 'var'), wv, buf.device)
    if pk.kind != 'xb':
        raise ValueError('conv_split_into_plane: the split-fp16 x-pair kernel only')
    blocks = xp_blocks(D, H, W, B)
    sbuf = _stats_buffer(buf, blocks, 16, groups=B)
    st = Stats()
    st.partial, st.blocks, st.cpad, st.count, st.groups = sbuf, blocks, 16, D * H * W, B
    if sv.pieces:
        conv_xp_launch(sv._var, pk, buf, 0, None, False, sbuf, pb, ldy=8, y_gstride=K * pstride, y_off=int(plane) * pstride,
                       planar=sv.planar, pieces=True)
    else:
        conv_xp_launch(sv.var, pk, buf, 0, None, False, sbuf, pb, ldy=8, y_gstride=K * pstride, y_off=int(plane) * pstride)
    return st


def refine_stems(photo_raw, geo_var, geo_plane_bias, prob, hull, key, w_geo, w_prob, w_hull, planar_out=None):
    """The geo | prob | vishull stems of CostVolRefineNet in one pass, stored with the raw photo-stem output as whole
    rows of the 32-channel concat buffer (atvs_refine_stems_f32).  photo_raw (B,D,h,w,8), geo_var (B,D,h,w,2),
    geo_plane_bias (B,h,w,24), prob / hull (B,D,h,w,1); w_*: TF kernels [3,3,3,Cin,8] (numpy, D-varying channels only).
    -> (buffer (B,D,h,w,32) raw, Stats over the 24 computed channels).
    planar_out = chunk-planar buffer (B, 4, planar_stride(D,h,w)) whose plane 0 the photo stem has written itself
    (conv_split_into_plane): planes 1..3 are written here, photo_raw is None, the buffer is returned."""
    import numpy as np
    B, D, H, W, _ = geo_var.shape
    if planar_out is not None and (tuple(planar_out.shape) != (B, 4, planar_stride(D, H, W)) or not planar_out.is_contiguous()):
        raise ValueError('refine_stems: planar_out must be a contiguous (B, 4, planar_stride(D,h,w)) buffer')
    photo_raw = geo_var if planar_out is not None else photo_raw      # (device / meta reference below)
    ck = ('stems', key, str(photo_raw.device))
    pk = _pack_cache.get(ck)
    if pk is None:
        packed = np.empty(27 * 4 * 8, np.float32)
        args = [np.ascontiguousarray(a, dtype=np.float32) for a in (w_geo, w_prob, w_hull)]
        if args[0].shape != (3, 3, 3, 2, 8) or args[1].shape != (3, 3, 3, 1, 8) or args[2].shape != (3, 3, 3, 1, 8):
            raise ValueError('refine_stems: kernels [3,3,3,2,8], [3,3,3,1,8], [3,3,3,1,8]')
        rc = _lib.lib().atvs_refine_stems_pack(*[a.ctypes.data_as(ctypes.c_void_p) for a in args],
                                               packed.ctypes.data_as(ctypes.c_void_p))
        if rc:
            raise RuntimeError('atvs_refine_stems_pack failed (%d)' % rc)
        pk = _Packed()
        pk.key, pk.tab, pk.cin, pk.cout = key, None, 4, 24
        pk.wp = None if photo_raw.is_meta else torch.from_numpy(packed).to(photo_raw.device)
        _pack_cache[ck] = pk
    buf = _new(photo_raw, (B, D, H, W, 32)) if planar_out is None else planar_out
    rows = int(_lib.lib().atvs_conv_stem_rows(D, H, W))
    st = Stats()
    st.partial = torch.empty((B, rows, 2, 24), dtype=torch.float64, device=photo_raw.device)
    st.blocks, st.cpad, st.count, st.groups = rows, 24, D * H * W, B
    if _dev_ok(photo_raw, geo_var, geo_plane_bias, prob, hull, buf):
        with _Timed(key, (D, H, W, 4), 24, B):
            _call('atvs_refine_stems_f32', _p(photo_raw if planar_out is None else None), _p(geo_var), _p(geo_plane_bias),
                  _p(prob), _p(hull), _p(pk.wp), _p(buf), ctypes.c_void_p(st.partial.data_ptr()), B, D, H, W,
                  ctypes.c_long(planar_stride(D, H, W) if planar_out is not None else 0), _stream())
    return buf, st


def siblings_ok(shape, cin, cout, cout2):
    """Can conv(8 channels, stride 1) and conv(16 channels, stride 2) of one (D,H,W,cin) input share a launch?"""
    return (cfg.xp1w and cfg.xpair and cfg.siblings and cfg.force_impl != 'gather' and len(shape) == 3 and cout == 8
            and cout2 == 16 and cin % 8 == 0 and (shape[2] >= 24 or cfg.force_impl == 'tiled'))






def conv_siblings(x, key, w_host, key2, w2_host, plane_bias=None, plane_bias2=None, groups=None, planar=False, pieces=False):
    """The U-Net's two convolutions of one input in ONE launch: y = conv3x3x3(x, w) (8 channels, stride 1) and
    y2 = conv3x3x3(x, w2) (16 channels, stride 2, SAME), each with the partial moments of its output.
    x (D,H,W,Cin) (groups=G: (G,D,H,W,Cin)), Cin % 8 == 0.  Returns (y, Stats), (y2, Stats).
    x may be a PendingBN / PendingSum for which siblings_prologue_ok() holds: the batch norm (+ ReLU) of its producer(s)
    and the sum are then formed inside the launch, while the input is staged (no pass of their own)."""
    prologue = None
    if isinstance(x, LAZY):
        if not siblings_prologue_ok(x):
            raise ValueError('conv_siblings: this lazy input must be materialised first')
        if isinstance(x, PendingBN) and x.planar and x._final is None:
            planar = x.planar          # the refinement's concat: chunk-planar raw buffer, batch norm + ReLU pending
        x, prologue = x.prologue()
    if planar:             # x: (G, Cin/8, planar_stride(D,H,W)) chunk-planar buffer, planar = (D,H,W); not the direct fp32 kernel
        D, H, W = planar
        if (prologue is not None and _xkind() != 'xb') or groups is None or x.dim() != 3 \
                or not x.is_contiguous() or x.shape[2] != planar_stride(D, H, W):
            raise ValueError('conv_siblings(planar=(D,H,W)): a contiguous (G, Cin/8, planar_stride) buffer (a prologue only '
                             'on the split-fp16 kernel)')
        G, K = x.shape[:2]
        x5, nsp, cin = x, 3, K * 8
    else:
        x5, nsp = _to5(x, groups, 'conv_siblings input')
        G, D, H, W, cin = x5.shape
    if nsp != 3 or not siblings_ok((D, H, W), cin, int(w_host.shape[-1]), int(w2_host.shape[-1])):
        raise ValueError('conv_siblings: unsupported shapes')
    pk = pack_conv_xp(key, w_host, x.device)
    pk2 = pack_conv_xp_sibling(key2, w2_host, x.device)
    if pk.cin != cin or pk2.cin != cin:
        raise ValueError('conv_siblings %s: input has %d channels' % (key, cin))
    D2, H2, W2 = (D + 1) // 2, (H + 1) // 2, (W + 1) // 2
    lead = () if groups is None else (G,)
    y, y2 = _new(x, lead + (D, H, W, 8)), _new(x, lead + (D2, H2, W2, 16))
    blocks = xp_blocks(D, H, W, G)
    sbuf, sbuf2 = _stats_buffer(x, blocks, 16, groups=G), _stats_buffer(x, blocks, 16, groups=G)
    st, st2 = Stats(), Stats()
    st.partial, st.blocks, st.cpad, st.count, st.groups = sbuf, blocks, 16, D * H * W, G
    st2.partial, st2.blocks, st2.cpad, st2.count, st2.groups = sbuf2, blocks, 16, D2 * H2 * W2, G
    if plane_bias is not None and tuple(plane_bias.shape) != lead + (H, W, 24):
        raise ValueError('conv_siblings %s: plane_bias %s' % (key, tuple(plane_bias.shape)))
    if plane_bias2 is not None and tuple(plane_bias2.shape) != lead + (H2, W2, 48):
        raise ValueError('conv_siblings %s: plane_bias2 %s' % (key2, tuple(plane_bias2.shape)))
    if prologue is not None and prologue[0] is not None:
        prologue = (_to5(prologue[0], groups, 'conv_siblings second source')[0],) + tuple(prologue[1:])
    if pieces and (not planar or prologue is not None or _xkind() != 'xb'):
        raise ValueError('conv_siblings(pieces=True): a chunk-planar input without a prologue, on the split-operand kernel')
    conv_xp_launch(x5, pk, y, 0, None, False, sbuf, plane_bias, sibling=(pk2, y2, 0, sbuf2, plane_bias2),
                   prologue=prologue, planar=planar, pieces=pieces)
    return (y, st), (y2, st2)


def conv_split_siblings(sv, key, w_host, key2, w2_host):
    """conv_siblings over a SplitVolume (B samples): the D-constant channels enter both outputs as depth-plane biases."""
    B = sv.shape[0]
    cv, cc = sv.cv, sv.const.shape[-1]
    wv, planes = _fold_split_weights(key, w_host, sv.chan_map, cv, cc)
    wv2, planes2 = _fold_split_weights(key2, w2_host, sv.chan_map, cv, cc)
    pb = conv(sv.const, (key, 'planes'), planes, stride=1, groups=B)
    pb2 = conv(sv.const, (key2, 'planes'), planes2, stride=2, groups=B)
    if sv.planar and (not sv.pieces or _xkind() == 'xb'):
        return conv_siblings(sv._var, (key, 'var'), wv, (key2, 'var'), wv2, plane_bias=pb, plane_bias2=pb2, groups=B,
                             planar=sv.planar, pieces=sv.pieces)
    return conv_siblings(sv.var, (key, 'var'), wv, (key2, 'var'), wv2, plane_bias=pb, plane_bias2=pb2, groups=B)


_DECONV_OFFSETS = [(a, b, c) for a in (0, -1) for b in (0, -1) for c in (0, -1)]
_virt_cache = {}


def _deconv_virtual_kernel(key, w_host):
    """Dense virtual kernel of the fused transposed convolution: (8 offsets, Cin, 8 classes * Cout).
    Per axis: even outputs 2j take (k=0, i=j) and (k=2, i=j-1); odd outputs 2j+1 take (k=1, i=j)."""
    import numpy as np
    hit = _virt_cache.get(key)
    if hit is not None:
        return hit
    w = np.asarray(w_host, np.float32)               # [3,3,3,Cout,Cin]
    cout, cin = w.shape[3], w.shape[4]
    wv = np.zeros((8, cin, 8, cout), np.float32)
    kof = {(0, 0): 0, (0, -1): 2, (1, 0): 1}          # (parity, offset) -> k
    for oi, off in enumerate(_DECONV_OFFSETS):
        for cls in range(8):
            par = ((cls >> 2) & 1, (cls >> 1) & 1, cls & 1)
            ks = [kof.get((par[a], off[a])) for a in range(3)]
            if None in ks:
                continue
            wv[oi, :, cls, :] = w[ks[0], ks[1], ks[2]].T
    hit = wv.reshape(8, cin, 8 * cout)
    _virt_cache[key] = hit
    return hit


def deconv_sum_ok(src, cout, groups=None):
    """Can conv3d_transpose_s2 take this PendingSum as it is (atvs_deconv_up_b_sum_f32 forms it while staging)?"""
    if not (cfg.prologue and cfg.sum_on_load and cfg.force_impl is None and cfg.deconv_up and split_on('upb')) or src._final is not None:
        return False
    shape = tuple(src.shape)
    if len(shape) != (5 if groups is not None else 4):
        return False
    cin = shape[-1]
    if not (deconv_up_ok(cin, cout) and _lib.lib().atvs_deconv_up_b_sum_supported(int(cin), int(cout))):
        return False
    gs = set()
    for t in src.items:
        raw = t.raw if isinstance(t, PendingBN) else t
        if isinstance(t, PendingBN) and (t.planar or (t._final is None and t.params.numel() != (groups or 1) * 3 * cin)):
            return False
        if not raw.is_contiguous():
            return False
    return True


def conv3d_transpose_s2(x, key, w_host, relu=False, want_stats=False, groups=None):
    """tf.layers.conv3d_transpose(3, stride 2, SAME): (D,H,W,Cin) -> (2D,2H,2W,Cout) (groups=G: G samples).

    w_host: TF layout [3,3,3,Cout,Cin].  LDS-tiled path: all 8 output parity classes from one staged
    input tile per workgroup (N axis = class x channel); fallback: one gather launch per class.
    """
    terms = None
    if isinstance(x, PendingSum):
        # the skip sum formed inside the launch where the kernel is built for it (deconv_sum_ok), else formed first
        if x._final is None and deconv_sum_ok(x, int(w_host.shape[-2]), groups):
            terms = [(t.raw, t.params, t.relu) if (isinstance(t, PendingBN) and t._final is None)
                     else ((t.materialize() if isinstance(t, PendingBN) else t), None, False) for t in x.items]
            x = terms[0][0]
        else:
            x = x.materialize()
    x5, nsp = _to5(x, groups, 'conv3d_transpose input')
    G, D, H, W, Cin = x5.shape
    cout = int(w_host.shape[-2])
    lead = () if groups is None else (G,)
    y = _new(x, lead + (2 * D, 2 * H, 2 * W, cout))
    y5 = y.reshape((G, 2 * D, 2 * H, 2 * W, cout))
    M = D * H * W
    if deconv_up_ok(Cin, cout) and 32.0 * M * cout < 2.0 ** 32:
        # all 8 parity classes from one staged input tile, one workgroup per CU (csrc/deconv_up.hip)
        split = split_on('upb') and bool(_lib.lib().atvs_deconv_up_b_supported(int(Cin), int(cout)))     # deconv_up_b.hip
        pk = pack_deconv_up(key, w_host, x.device, '_b' if split else '')
        grid_fn = _lib.lib().atvs_deconv_up_b_grid if split else _lib.lib().atvs_deconv_up_grid
        blocks = int(grid_fn(int(D), int(H), int(W), int(cout), int(G)))
        st, sbuf = None, None
        if want_stats:
            sbuf = _stats_buffer(x, blocks, 16, groups=G)
            st = Stats()
            st.partial, st.blocks, st.cpad, st.count, st.groups = sbuf, blocks, 16, 8 * M, G
        if terms is not None and not split:
            raise RuntimeError('deconv_sum_ok admitted a sum the split-operand kernel does not take')
        if _dev_ok(x5, y5, *[t for tr in (terms or []) for t in tr[:2]]):
            with _Timed(key, x5.shape[1:], cout, G):
                if terms is not None:
                    (xa, pa, ra), (xb, pb, rb) = terms[0], terms[1]
                    xc, pc, rc = terms[2] if len(terms) > 2 else (None, None, False)
                    _call('atvs_deconv_up_b_sum_f32', _p(xa), _p(pa), _p(xb), _p(pb), _p(xc), _p(pc),
                          int(bool(ra)) | (int(bool(rb)) << 1) | (int(bool(rc)) << 2), _p(pk.wp), _p(y5), _p(sbuf), G, D, H, W,
                          Cin, cout, cout, 0, int(bool(relu)), 16, 0, _stream())
                elif split:
                    _call('atvs_deconv_up_b_f32', _p(x5), _p(pk.wp), _p(y5), _p(sbuf), G, D, H, W, Cin, cout, cout, 0,
                          int(bool(relu)), 16, 0, _stream())
                else:
                    _call('atvs_deconv_up_f32', _p(x5), _p(pk.wp), _p(y5), _p(sbuf), G, D, H, W, Cin, cout, cout, 0,
                          int(bool(relu)), _stream())
        return (y, st) if want_stats else y
    if cout == 32 and cfg.deconv_up and split_on('upb') and cfg.force_impl is None and 32.0 * M * cout < 2.0 ** 32 \
            and bool(_lib.lib().atvs_deconv_up_b_supported(int(Cin), 16)):
        # the 64 -> 32 layer (conv_b*_4_0) as two 16-channel launches of the split-fp16 kernel into the halves of y
        import numpy as np
        blocks = int(_lib.lib().atvs_deconv_up_b_grid(int(D), int(H), int(W), 16, int(G)))
        st, sbuf = None, None
        if want_stats:
            sbuf = _stats_buffer(x, blocks, 32, groups=G)
            st = Stats()
            st.partial, st.blocks, st.cpad, st.count, st.groups = sbuf, blocks, 32, 8 * M, G
        w = np.asarray(w_host)
        for h in range(2):
            pk = pack_deconv_up((key, 'half', h), np.ascontiguousarray(w[:, :, :, 16 * h:16 * h + 16, :]), x.device, '_b')
            if _dev_ok(x5, y5):
                with _Timed(key, x5.shape[1:], 16, G):
                    _call('atvs_deconv_up_b_f32', _p(x5), _p(pk.wp), _p(y5), _p(sbuf), G, D, H, W, Cin, 16, 32, 16 * h,
                          int(bool(relu)), 32, 16 * h, _stream())
        return (y, st) if want_stats else y
    classes = [(a, b, c) for a in (0, 1) for b in (0, 1) for c in (0, 1)]
    fused = cfg.force_impl != 'gather' and cout % 4 == 0 and cout <= 64 and (W >= 12 or cfg.force_impl == 'tiled')
    if fused:
        per = min(8, 128 // cout)                     # classes per launch (N <= 128 virtual channels)
        wv = _deconv_virtual_kernel(key, w_host)
        taps = tuple((i,) + off for i, off in enumerate(_DECONV_OFFSETS))
        nt = 1
        while nt * 16 < per * cout:
            nt *= 2
        tile_y = 8 if (nt <= 2 and H >= 16) else 4
        blocks = tiled_blocks(D, H, W, tile_y, Cin, per * cout, groups=G)
        nl = 8 // per
        in_kernel = bool(_lib.lib().atvs_conv_tiled_has_stats(D, H, W, tile_y, Cin, per * cout, G))
        st, sbufs = None, None
        if want_stats and in_kernel:
            # one statistics buffer per launch (class group), each (G, blocks, 2, cpad); bn_finalize folds the
            # launches' columns through `fold` on a buffer laid out (G, nl * blocks, 2, cpad)
            sall = torch.empty((G, nl * blocks, 2, nt * 16), dtype=torch.float64, device=x.device)
            st = Stats()
            st.partial, st.blocks, st.cpad, st.count, st.fold, st.groups = sall, blocks * nl, nt * 16, 8 * M, per, G
            sbufs = [_stats_buffer(x, blocks, nt * 16, groups=G) for _ in range(nl)] if (nl > 1 and G > 1) else None
        fin = None
        if st is not None and cfg.fused_finalize and cout <= 64 and not x.is_meta and G == 1:
            fin = Fin()
            fin.counter, fin.params, fin.stats = _fin_counter(x.device), _new(x, (3, cout)), st.partial
            fin.rows, fin.arrivals, fin.channels, fin.fold, fin.count = blocks * nl, blocks * nl, cout, per, 8 * M
            st.params = fin.params
        # the class groups are independent launches that each fill only part of the chip at the resolutions this path serves
        # (eighth resolution: 384 tiles): launches after the first go to side streams (parallel branches of a captured graph)
        main = torch.cuda.current_stream() if (nl > 1 and cfg.side_streams and x.is_cuda) else None
        sides = []
        for i in range(nl):
            wpart = wv[:, :, i * per * cout:(i + 1) * per * cout]
            pk = pack_conv_weights_tiled((key, 'cls', i), wpart, taps, False, x.device, tile_y)
            sb = None
            if st is not None:
                if sbufs is not None:
                    sb = sbufs[i]
                else:          # G == 1 or a single launch: the launch's rows are a contiguous slice
                    sb = st.partial.reshape(-1, 2, nt * 16)[i * blocks:(i + 1) * blocks] if G == 1 else st.partial
            if main is not None and i > 0:
                side = _side_stream(x.device, i - 1)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    conv_tiled_launch(x5, pk, y5, 2, (0, 0, 0), 0, tile_y, None, None, relu, sb, None, cout, i * per, fin=fin)
                sides.append(side)
            else:
                conv_tiled_launch(x5, pk, y5, 2, (0, 0, 0), 0, tile_y, None, None, relu, sb, None, cout, i * per, fin=fin)
        for side in sides:
            main.wait_stream(side)
        if sbufs is not None:
            for i in range(nl):
                st.partial[:, i * blocks:(i + 1) * blocks].copy_(sbufs[i])
        if want_stats and not in_kernel:
            st = channel_stats(y, groups)
        return (y, st) if want_stats else y
    pks = [pack_conv_weights(key, w_host, deconv_s2_class_taps(par), True, x.device) for par in classes]
    blocks, tm = conv_blocks(M, pks[0].ntiles, groups=G)
    cpad = pks[0].ntiles * 16
    st = None
    sbufs = None
    if want_stats:
        st = Stats()
        st.partial = torch.empty((G, blocks * 8, 2, cpad), dtype=torch.float64, device=x.device)
        st.blocks, st.cpad, st.count, st.groups = blocks * 8, cpad, 8 * M, G
        sbufs = [_stats_buffer(x, blocks, cpad, groups=G) for _ in range(8)] if G > 1 else None
    for i, (par, pk) in enumerate(zip(classes, pks)):
        sb = None
        if st is not None:
            sb = sbufs[i] if sbufs is not None else st.partial.reshape(-1, 2, cpad)[i * blocks:(i + 1) * blocks]
        conv_launch(x5, pk, y5, (D, H, W), 1, 2, par, 0, None, None, relu, sb, tm)
    if sbufs is not None:
        for i in range(8):
            st.partial[:, i * blocks:(i + 1) * blocks].copy_(sbufs[i])
    return (y, st) if want_stats else y


# --------------------------------------------------------------------------- batch norm / glue

def channel_stats(x, groups=None):
    """Partial sums of x viewed as (rows, C); groups=G: x is G independent samples stacked on its leading axis."""
    C = x.shape[-1]
    G = 1 if groups is None else int(groups)
    rows = x.numel() // C // G
    blocks = int(_lib.lib().atvs_channel_stats_num_blocks(ctypes.c_long(rows)))
    st = Stats()
    st.partial = torch.empty((G, blocks, 2, C), dtype=torch.float64, device=x.device)
    st.blocks, st.cpad, st.count, st.groups = blocks, C, rows, G
    if _dev_ok(x):
        _call('atvs_channel_stats', _p(x), G, ctypes.c_long(rows), C, ctypes.c_void_p(st.partial.data_ptr()), _stream())
    return st


_flag_pool = {}


def nonfinite_flag(device):
    """The device word atvs_bn_finalize ORs with 1 when a batch-norm moment is not finite (one per device, sticky)."""
    key = str(torch.device(device))
    f = _flag_pool.get(key)
    if f is None:
        f = _flag_pool[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return f


def nonfinite_seen(device, reset=True):
    """Did any batch norm since the last reset see a non-finite moment?  (Synchronises with the device.)"""
    f = _flag_pool.get(str(torch.device(device)))
    if f is None:
        return False
    seen = bool(int(f.item()))
    if seen and reset:
        f.zero_()
    return seen


def bn_params(st, C, ref, beta=None, eps=1e-3):
    """Stats -> params (3,C) = (mean, rstd, beta); (G,3,C) for the G independent samples of a grouped tensor."""
    if st.params is not None and beta is None and abs(eps - 1e-3) < 1e-12:
        return st.params
    G = st.groups
    params = _new(ref, (3, C) if G == 1 else (G, 3, C))
    if _dev_ok(ref, beta):
        _call('atvs_bn_finalize', ctypes.c_void_p(st.partial.data_ptr()), int(G), ctypes.c_long(st.blocks), st.cpad,
              int(st.fold), ctypes.c_long(st.count), _p(beta), ctypes.c_float(eps), _p(params), C,
              ctypes.c_void_p(nonfinite_flag(ref.device).data_ptr()), _stream())
    return params


def _param_groups(params):
    return 1 if params.dim() == 2 else int(params.shape[0])


def bn_apply(x, params, relu=False, out=None, C=None, c_off=0):
    """y = relu((x - mean) * rstd + beta).  With C / c_off: only that channel slice of the rows of x (in place).
    params (G,3,C): x is G independent samples stacked on its leading axis, each with its own parameters."""
    ld = x.shape[-1]
    C = ld if C is None else C
    y = x if out is None else out
    G = _param_groups(params)
    if _dev_ok(x, params, y):
        _call('atvs_bn_apply', _p(x), _p(params), _p(y), G, ctypes.c_long(x.numel() // ld // G), int(C), int(ld),
              int(c_off), int(bool(relu)), _stream())
    return y


def batch_norm(x, st=None, beta=None, relu=False, inplace=False, eps=1e-3, C=None, c_off=0, groups=None):
    """Training-mode BN of x with its own batch statistics (st = Stats from the producer, else computed).
    C / c_off select a channel slice of a wider buffer (statistics must then come from the producer)."""
    if st is None:
        if C is not None:
            raise ValueError('batch_norm on a channel slice needs the producer\'s statistics')
        st = channel_stats(x, groups)
    params = bn_params(st, x.shape[-1] if C is None else C, x, beta, eps)
    if C is not None:
        return bn_apply(x, params, relu, C=C, c_off=c_off)
    return bn_apply(x, params, relu, out=(x if inplace else _new(x, x.shape)))


class PendingBN(object):
    """A raw convolution output whose training-mode batch norm (+ ReLU) has not been applied yet.

    Layers whose only consumers are `add`s never need the normalised tensor on its own: the add kernel
    normalises on the fly (ops.bn_add).  Any other consumer calls materialize().  raw is batch-first
    (B, ..., C); params (3,C), or (B,3,C) for B independent samples."""

    def __init__(self, raw, params, relu, planar=None):
        # planar=(D,h,w): raw is a chunk-planar (B, C/8, planar_stride(D,h,w)) buffer (the refinement's concat); its one
        # consumer that reads planes is conv_siblings, everything else gets the channel-last tensor from materialize()
        self.raw, self.params, self.relu = raw, params, bool(relu)
        self.planar = tuple(int(v) for v in planar) if planar else None
        self._final = None
        self.device = raw.device

    @property
    def shape(self):
        if self.planar:
            return (self.raw.shape[0],) + self.planar + (self.raw.shape[1] * 8,)
        return tuple(self.raw.shape)

    def dim(self):
        return 5 if self.planar else self.raw.dim()

    @property
    def is_meta(self):
        return self.raw.is_meta

    def materialize(self):
        """The normalised channel-last tensor (computed once; in place on the raw buffer, or on a channel-last copy of it when
        the raw buffer is chunk-planar)."""
        if self._final is None:
            raw = self.raw
            if self.planar:
                B, K = raw.shape[:2]
                D, h, w = self.planar
                raw = planar_view(raw, D, h, w).permute(0, 2, 3, 4, 1, 5).reshape(B, D, h, w, K * 8).contiguous()
            self._final = bn_apply(raw, self.params, self.relu)
        return self._final

    def prologue(self):
        """(tensor, (None, params, None, relu, False)): this layer as the normalise-on-load input of a convolution."""
        if self._final is not None:
            return self._final, None
        return self.raw, (None, self.params, None, self.relu, False)


class PendingSum(object):
    """tf.add_n of two or three items (dense tensors or PendingBNs) that has not been formed yet: a consumer that can add on
    load (conv_siblings: two items; conv3d_transpose_s2: two or three) takes the items, any other consumer calls
    materialize() (= ops.bn_add / add_n)."""

    def __init__(self, items):
        if len(items) not in (2, 3) or any(tuple(t.shape) != tuple(items[0].shape) for t in items):
            raise ValueError('PendingSum: two or three items of one shape')
        # a chunk-planar raw buffer is not a channel-last operand: such an item enters the sum materialised
        self.items = [t.materialize() if isinstance(t, PendingBN) and t.planar else t for t in items]
        self._final = None
        self.device = items[0].device

    @property
    def shape(self):
        return tuple(self.items[0].shape)

    def dim(self):
        return len(self.shape)

    @property
    def is_meta(self):
        return any(getattr(t, 'is_meta', False) for t in self.items)

    def materialize(self):
        if self._final is None:
            if any(isinstance(t, PendingBN) and t._final is None for t in self.items) and self.shape[-1] % 4 == 0:
                self._final = bn_add(self.items)
            else:
                self._final = add_n([t.materialize() if isinstance(t, PendingBN) else t for t in self.items])
        return self._final

    def prologue(self):
        if self._final is not None:
            return self._final, None
        if len(self.items) != 2:
            raise ValueError('PendingSum.prologue: the x-pair kernels add two items on load')
        (a, pa), (b, pb) = (t.prologue() if isinstance(t, PendingBN) else (t, None) for t in self.items)
        return a, (b, pa[1] if pa else None, pb[1] if pb else None, bool(pa and pa[3]), bool(pb and pb[3]))


class LazySlice(object):
    """Channels [lo, hi) of a lazy layer (the stems inside the refinement's pending concat)."""

    def __init__(self, parent, lo, hi):
        self.parent, self.lo, self.hi = parent, int(lo), int(hi)
        self.device = parent.device

    @property
    def shape(self):
        return tuple(self.parent.shape[:-1]) + (self.hi - self.lo,)

    def dim(self):
        return len(self.shape)

    @property
    def is_meta(self):
        return self.parent.is_meta

    def materialize(self):
        return self.parent.materialize()[..., self.lo:self.hi]


LAZY = (PendingBN, PendingSum, LazySlice)





def siblings_prologue_ok(src):
    """Can conv_siblings take this lazy input as it is (the kernel forms it while staging)?  Built forms: one pending
    batch norm with Cin % 16 == 0 (the refinement's concat); a sum of two with Cin % 16 == 8 (the U-Net's stack inputs;
    Cin == 8 on the split-operand kernel)."""
    if not cfg.prologue or cfg.force_impl is not None or not cfg.xp1w:
        return False
    if isinstance(src, PendingBN):
        return src._final is None and src.shape[-1] % 16 == 0 and src.raw.is_contiguous()
    if isinstance(src, PendingSum):
        if src._final is not None or src.shape[-1] % 16 != 8 or len(src.items) != 2:
            return False
        if _xkind() == 'xb' and src.shape[-1] != 8:        # conv_xb's two-source form: one 8-channel chunk
            return False
        gs = set()
        for t in src.items:
            raw = t.raw if isinstance(t, PendingBN) else t
            if isinstance(t, PendingBN) and t._final is None:
                gs.add(_param_groups(t.params))
            if not raw.is_contiguous():
                return False
        return len(gs) <= 1
    return False


def bn_add(items, plus=None):
    """Sum of 2 or 3 items, each a dense tensor or a PendingBN (normalised on the fly); dims without batch.
    plus: ONE sample (the items' shape without the leading axis) -> (sum, plus + sum) from the same pass."""
    xs, ps, mask = [], [], 0
    for i, it in enumerate(items):
        if isinstance(it, PendingBN) and it._final is None and not it.planar:
            xs.append(it.raw)
            ps.append(it.params)
            mask |= (1 << i) if it.relu else 0
        else:
            t = it.materialize() if isinstance(it, PendingBN) else it
            xs.append(t)
            ps.append(None)
    C = xs[0].shape[-1]
    out = _new(xs[0], xs[0].shape)
    gs = set(_param_groups(p) for p in ps if p is not None)
    if len(gs) != 1:
        raise ValueError('bn_add: the pending batch norms disagree on the number of independent samples')
    G = gs.pop()
    x2, p2 = (xs[2], ps[2]) if len(xs) > 2 else (None, None)
    if plus is not None:
        if tuple(plus.shape) != tuple(out.shape[1:]) or out.shape[0] != G or not plus.is_contiguous():
            raise ValueError('bn_add: plus must be one contiguous sample of the items')
        out2 = _new(out, out.shape)
        if _dev_ok(*(xs + [p for p in ps if p is not None] + [plus])):
            _call('atvs_bn_add_plus', _p(xs[0]), _p(ps[0]), _p(xs[1]), _p(ps[1]), _p(x2), _p(p2), _p(out), _p(plus), _p(out2), G,
                  ctypes.c_long(out.numel() // C // G), C, int(mask), _stream())
        return out, out2
    if _dev_ok(*(xs + [p for p in ps if p is not None])):
        _call('atvs_bn_add', _p(xs[0]), _p(ps[0]), _p(xs[1]), _p(ps[1]), _p(x2), _p(p2), _p(out), G,
              ctypes.c_long(out.numel() // C // G), C, int(mask), _stream())
    return out


def add_n(tensors, out=None):
    """tf.add_n: ((a + b) + c) + ...; out: optional destination of the final sum (same shape, may not alias)."""
    acc = tensors[0]
    i = 1
    first = True
    while i < len(tensors):
        b = tensors[i]
        c = tensors[i + 1] if (first and i + 1 < len(tensors)) else None
        step = 2 if c is not None else 1
        dst = out if (out is not None and i + step >= len(tensors)) else _new(acc, acc.shape)
        if _dev_ok(acc, b, c, dst):
            _call('atvs_add_n', _p(acc), _p(b), _p(c), _p(dst), ctypes.c_long(acc.numel()), _stream())
        acc = dst
        i += step
        first = False
    return acc


def avg_pool_same(x, pool, stride, groups=None):
    """tf.layers.average_pooling2d(SAME) of (H,W,C) (groups=G: (G,H,W,C))."""
    G = 1 if groups is None else int(groups)
    H, W, C = x.shape[-3:]
    Ho, Wo = -(-H // stride), -(-W // stride)
    lead = () if groups is None else (G,)
    y = _new(x, lead + (Ho, Wo, C))
    ws = _new(x, (G, int(_lib.lib().atvs_avg_pool_ws_floats(int(H), int(W), int(C), int(stride)))))
    if _dev_ok(x):
        _call('atvs_avg_pool_same', _p(x), _p(y), _p(ws), G, H, W, C, int(pool), int(stride), _stream())
    return y


def resize_bilinear(x, size, out=None, c_off=0, groups=None):
    """align_corners bilinear resize of (H,W,C) (groups=G: (G,H,W,C)) into out[..., c_off:c_off+C]."""
    G = 1 if groups is None else int(groups)
    H, W, C = x.shape[-3:]
    Ho, Wo = int(size[0]), int(size[1])
    lead = () if groups is None else (G,)
    y = _new(x, lead + (Ho, Wo, C)) if out is None else out
    if _dev_ok(x, y):
        _call('atvs_resize_bilinear', _p(x), _p(y), G, H, W, C, Ho, Wo, y.shape[-1], int(c_off), _stream())
    return y


def copy_channels(src, dst, C, src_off=0, dst_off=0):
    rows = src.numel() // src.shape[-1]
    if _dev_ok(src, dst):
        _call('atvs_copy_channels', _p(src), _p(dst), ctypes.c_long(rows), int(C), src.shape[-1], int(src_off),
              dst.shape[-1], int(dst_off), _stream())
    return dst


def stack(tensors, dim=0):
    """tf.stack / torch.stack of up to 16 same-shaped tensors along a new axis `dim`, where every axis in front of `dim` has
    extent 1 (so that the result is the tensors laid end to end): one launch of the library's own copy kernel."""
    shape = tuple(tensors[0].shape)
    if any(tuple(t.shape) != shape for t in tensors) or any(int(v) != 1 for v in shape[:dim]):
        raise ValueError('ops.stack: same shapes, and only unit axes in front of the new one')
    out = _new(tensors[0], shape[:dim] + (len(tensors),) + shape[dim:])
    n = tensors[0].numel()
    if len(tensors) > 16 or n % 4:
        for i, t in enumerate(tensors):
            copy_channels(t.reshape(1, -1), out.reshape(len(tensors), -1)[i:i + 1], n)
        return out
    if _dev_ok(out, *tensors):
        arr = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
        _call('atvs_stack', arr, len(tensors), ctypes.c_long(n), _p(out), _stream())
    return out


def concat_channels(tensors):
    """tf.concat(axis=-1) of tensors sharing their leading dims."""
    total = sum(t.shape[-1] for t in tensors)
    out = _new(tensors[0], tuple(tensors[0].shape[:-1]) + (total,))
    off = 0
    for t in tensors:
        copy_channels(t, out, t.shape[-1], 0, off)
        off += t.shape[-1]
    return out


# --------------------------------------------------------------------------- AANet

def _ptr_array(ts):
    arr = (ctypes.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        arr[i] = t.data_ptr()
    return arr


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


# --------------------------------------------------------------------------- depth-map fusion (after the hot path)

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
