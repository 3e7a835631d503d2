"""Training-mode batch norm and the element-wise glue; the DEFERRED batch-norm algebra: a layer consumed only by adds / by a
convolution that normalises on load is never normalised on its own (`PendingBN`, `PendingSum`, `LazySlice`).
"""

import ctypes

import torch

from .. import _lib
from .base import Stats, _call, _dev_ok, _new, _p, _stream, cfg
from .packing import _xkind, planar_view


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
