"""Per-phase tick counts of bottleneck_b.hip (development build: bash tools_dev/build_variant.sh btdbg bottleneck_b -DATVS_BT_DEBUG;
ATVS_LIB=tools_dev/_dbg/lib_btdbg.so python tools_dev/phase_bt.py [64|32])"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import atvsnet_amd  # noqa: F401
from atvsnet_amd import ops, _lib
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H, W = (128, 160) if C == 64 else (256, 320)
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
G = 5
x = torch.randn(G, H, W, C, device=dev)
params = ops.bn_params(ops.channel_stats(x, groups=G), C, x, torch.zeros(C, device=dev))
w1, w3 = [(rng.standard_normal((1, 1, C, C)) * (1.0 / C) ** 0.5).astype(np.float32) for _ in range(2)]
w2 = (rng.standard_normal((3, 3, C, C)) * (1.0 / (9 * C)) ** 0.5).astype(np.float32)
b = [torch.randn(C, device=dev) * 0.1 for _ in range(3)]
keys = tuple(('bench', C, n) for n in range(3))
for _ in range(3):
    ops.bottleneck(x, params, keys, w1, b[0], w2, b[1], w3, b[2])
torch.cuda.synchronize()
buf = np.zeros(4096 * 8, np.uint64)
assert _lib.lib().atvs_debug_read_bt(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(-1, 8).astype(np.float64)
t = t[t[:, 7] > 0]
names = ['P1 x fragments arrive (global latency)', 'P1 conv1 + r1 -> LDS', 'barrier 1', 'P2 conv2', 'barrier 2 + P3 r2 -> LDS + barrier 3',
         'P4 conv3 MFMAs', 'epilogue + moments']
tot = t[:, :7].sum(1).mean()
print('C=%d: %d wavefronts, %.0f ticks per tile (100 MHz ticks: %.1f us)' % (C, len(t), tot, tot / 100.0))
for i, n in enumerate(names):
    print('   %-45s %8.0f (%.1f%%)' % (n, t[:, i].mean(), 100 * t[:, i].mean() / tot))
