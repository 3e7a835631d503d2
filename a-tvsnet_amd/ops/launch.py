"""Dispatch policy (which kernel family takes a layer: the `*_ok` predicates) and the single-kernel launch wrappers of the 2-D
tower layers and the generic / tiled / x-pair 3-D kernels.
"""

import ctypes

import torch

from .. import _lib
from .base import Stats, _Timed, _call, _dev_ok, _new, _p, _stream, cfg
from .packing import _Packed, _pack_cache, _xp_cache, planar_stride, split_on
from .norm import PendingBN, PendingSum, _param_groups


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


def _pick_tile_m(M, ntiles):
    """Largest voxel-tile count per wavefront that still leaves >= 1024 workgroups (4 per CU)."""
    for tm in (8, 4, 2):
        if tm * ntiles <= 16 and -(-M // (64 * tm)) >= 1024:
            return tm
    return 1


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
