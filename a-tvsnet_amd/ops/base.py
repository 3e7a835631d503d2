"""ctypes plumbing of the C-ABI (include/atvsnet_hip.h) and every dispatch switch.

torch is plumbing here: it owns device memory (caching allocator) and the current HIP stream; every computation happens in the
gfx950 kernels reached through ctypes.  Tensors must be float32, contiguous and on a ``cuda`` device; ``meta`` tensors run the
same host code without launching (shape / memory planning and CPU plumbing tests).  CPU tensors are refused: there is no CPU
fallback on the product path.  Here: the switches (`cfg`, `configure`), pointer / stream helpers, `_call` (status -> exception),
HIP-event timing of watched launches (`watch`), the statistics handle `Stats`.
"""

import ctypes
import os

import torch

from .. import _lib


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
            from .packing import clear_pack_cache
            clear_pack_cache()
        return cfg

    def __exit__(self, *exc):
        for k, v in self._old.items():
            object.__setattr__(cfg, k, v)
        if self._clear:
            from .packing import clear_pack_cache
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


class Stats(object):
    """Per-workgroup partial sums of a tensor: feeds bn_finalize."""
    __slots__ = ('partial', 'blocks', 'cpad', 'count', 'fold', 'params', 'groups')

    def __init__(self):
        self.fold = 1
        self.params = None      # (3,C) moments already finished inside the producing launch
        self.groups = 1         # independent samples: partial is (groups, blocks, 2, cpad), count per sample


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


def _ptr_array(ts):
    arr = (ctypes.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        arr[i] = t.data_ptr()
    return arr
