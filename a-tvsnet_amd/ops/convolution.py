"""`conv()` -- the convolution dispatcher behind cnn_wrapper/network.py -- and the composite forms: the lazily tiled cost volume
(`SplitVolume`), sibling launches, the refinement stems, the stride-2 transposed convolution.
"""

import ctypes

import torch

from .. import _lib
from .base import Stats, _Timed, _call, _dev_ok, _new, _p, _side_stream, _stream, cfg
from .packing import (_Packed, _fold_cache, _pack_cache, _virt_cache, _xkind, conv_taps, deconv_s2_class_taps,
    deconv_up_ok, pack_conv3d_b, pack_conv_c16, pack_conv_c16b, pack_conv_weights, pack_conv_weights_tiled, pack_conv_xp,
    pack_conv_xp_sibling, pack_deconv_up, planar_pieces_decode, planar_stride, planar_view, same_pad, split_on)
from .geometry import tile_planes
from .norm import LAZY, PendingBN, PendingSum, bn_apply, channel_stats, copy_channels, siblings_prologue_ok
from .launch import (Fin, XPAIR_TAPS, _fin_counter, _from5, _stats_buffer, _to5, _xpair_virtual_kernel, conv1x1,
    conv1x1_ok, conv2d_lds, conv2d_lds_ok, conv_blocks, conv_launch, conv_tiled_launch, conv_xp_launch, pack_conv2d_lds,
    tiled_blocks, tiled_tile_y, xp_blocks)


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
