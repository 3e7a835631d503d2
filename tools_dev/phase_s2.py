"""Per-phase tick counts of conv3d_s2b.hip (development build: bash tools_dev/build_variant.sh s2dbg conv3d_s2b -DATVS_S2_DEBUG;
ATVS_LIB=tools_dev/_dbg/lib_s2dbg.so python tools_dev/phase_s2.py [2_0|3_0])"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import atvsnet_amd  # noqa: F401
from atvsnet_amd import ops, _lib
which = sys.argv[1] if len(sys.argv) > 1 else '2_0'
dev = torch.device('cuda:0')
G = 8
shape, cin, cout = {'2_0': ((96, 64, 80), 16, 32), '3_0': ((48, 32, 40), 32, 64)}[which]
x = torch.randn((G,) + shape + (cin,), device=dev)
w = (np.random.default_rng(0).standard_normal((3, 3, 3, cin, cout)) * 0.05).astype(np.float32)
for _ in range(3):
    ops.conv(x, ('ph', which), w, stride=2, want_stats=True, groups=G)
torch.cuda.synchronize()
buf = np.zeros(4096 * 8, np.uint64)
assert _lib.lib().atvs_debug_read_s2(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(-1, 8).astype(np.float64)
t = t[t[:, 7] > 0]
ns = t[:, 7].mean()
names = ['loop top (acc zero, first weight fragments)', 'barrier A (images free)', 'halo of this stage arrives (vmcnt 0)',
         'split + LDS write', 'barrier B', 'K loop (+ next halo requests)', 'epilogue stores']
tot = t[:, :7].sum(1).mean()
print('%s: %d wavefronts, %.1f stages each, %.0f ticks per stage' % (which, len(t), ns, tot / ns))
for i, n in enumerate(names):
    print('   %-50s %8.0f per stage (%.1f%%)' % (n, t[:, i].mean() / ns, 100 * t[:, i].mean() / tot))
