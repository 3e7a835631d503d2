"""Per-phase tick counts of the SUMMING full-resolution decoder (deconv_up_b_sum_kernel: two roles, the skip sum formed on load) at
the bench's shape, beside the plain form on the materialised sum:
bash tools_dev/build_variant.sh ubdbg deconv_up_b -DATVS_UB_DEBUG; ATVS_LIB=tools_dev/_dbg/lib_ubdbg.so python tools_dev/phase_ub_sum.py [nterms]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import atvsnet_amd  # noqa: F401
from atvsnet_amd import ops, _lib
nterms = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device('cuda:0')
G, (D, H, W), cin, cout = 8, (96, 64, 80), 16, 8            # conv_b*_6_0 of cfg3: half-resolution input, 8 volumes per launch
g = torch.Generator().manual_seed(1)
w = (np.random.default_rng(0).standard_normal((3, 3, 3, cout, cin)) * 0.1).astype(np.float32)


def terms():
    ts = []
    for k in range(nterms):
        raw = torch.randn((G, D, H, W, cin), generator=g).to(dev)
        par = torch.stack([torch.randn((G, cin), generator=g) * 0.1, torch.rand((G, cin), generator=g) + 0.5,
                           torch.randn((G, cin), generator=g) * 0.1], 1).to(dev).contiguous()
        ts.append(ops.PendingBN(raw, par, relu=(k != 2)))
    return ts


def read(label):
    if not hasattr(_lib.lib(), 'atvs_debug_read_ub'):
        return
    buf = np.zeros(4096 * 8, np.uint64)
    assert _lib.lib().atvs_debug_read_ub(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    raw = buf.reshape(-1, 8)
    if label.startswith('summing'):          # two roles: rows (block, wavefront 0..7), wavefronts 4..7 stage
        rows = raw[:4096 // 8 * 8].reshape(-1, 8, 8)
        for role, sl, names in (('multiply', slice(0, 4), ['K loop', 'epilogue (stores, moments)', 'waiting at the barrier']),
                                ('staging', slice(4, 8), ['wait for the loads, batch norms, sum, split, LDS writes', 'the next stage\'s requests',
                                                          'waiting at the barrier'])):
            t = rows[:, sl, :].reshape(-1, 8)
            t = t[t[:, 7] > 0].astype(np.float64)
            ns = (t[:, 7].astype(np.uint64) & np.uint64(0xffff)).astype(np.float64).mean()
            tot = t[:, :3].sum(1).mean()
            print('%s (%s role): %d wavefronts, %.1f stages each, %.0f ticks per stage' % (label, role, len(t), ns, tot / ns))
            for i, n in enumerate(names):
                print('   %-75s %8.0f per stage (%.1f%%)' % (n, t[:, i].mean() / ns, 100 * t[:, i].mean() / tot))
        return
    raw = raw[raw[:, 7] > 0]
    t = raw.astype(np.float64)
    ns = (raw[:, 7] & np.uint64(0xffff)).astype(np.float64).mean()
    names = ['loop top (acc zero)', 'barrier A (images free)', 'split + pieces -> LDS', 'barrier B', 'K loop (+ next halo requests)', 'epilogue stores']
    tot = t[:, 1:6].sum(1).mean()
    print('%s: %d wavefronts, %.1f stages each, %.0f ticks per stage' % (label, len(t), ns, tot / ns))
    for i, n in enumerate(names):
        if i:
            print('   %-75s %8.0f per stage (%.1f%%)' % (n, t[:, i].mean() / ns, 100 * t[:, i].mean() / tot))


def timed(fn, label):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print('%s: %.1f us per launch' % (label, 200. * e0.elapsed_time(e1)))


s_on = ops.PendingSum(terms())
assert ops.deconv_sum_ok(s_on, cout, G)
timed(lambda: ops.conv3d_transpose_s2(s_on, ('ph', 'sum'), w, want_stats=True, groups=G), 'summing form (%d terms)' % nterms)
assert s_on._final is None
read('summing form')
x = ops.PendingSum(terms()).materialize()
timed(lambda: ops.conv3d_transpose_s2(x, ('ph', 'sum'), w, want_stats=True, groups=G), 'plain form on the materialised sum')
read('plain form')
