"""Per-phase cycle counts of the tiled conv kernel (development build with -DATVS_EXP=16)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import atvsnet_amd
from atvsnet_amd import ops, _lib
# usage: phase_times.py [--tiled|--deconv] D H W cin cout [G]    (--deconv: the fused transposed convolution, tiled kernel)
mode = 'xp'
if sys.argv[1].startswith('--'):
    mode = sys.argv.pop(1)[2:]
D, H, W, cin, cout = [int(v) for v in sys.argv[1:6]]
G = int(sys.argv[6]) if len(sys.argv) > 6 else 1
dev = torch.device('cuda:0')
x = torch.randn(G, D, H, W, cin, device=dev)
w = (np.random.default_rng(0).standard_normal((3, 3, 3, cin, cout)) * 0.1).astype(np.float32)
for _ in range(3):
    if mode == 'deconv':
        y, st = ops.conv3d_transpose_s2(x, 'bench_t', np.ascontiguousarray(np.swapaxes(w, -1, -2)), want_stats=True, groups=G)
    else:
        y, st = ops.conv(x, 'bench', w, want_stats=True, groups=G)
torch.cuda.synchronize()
buf = np.zeros(4096 * 8, np.uint64)
rc = getattr(_lib.lib(), 'atvs_debug_read' if mode == 'xp' else 'atvs_debug_read_tiled')(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0
b = buf.reshape(-1, 8).astype(np.float64)
b = b[b.sum(1) > 0]
names = ['barrier1 (wait others done K loop)', 'LDS write (+wait prefetch)', 'barrier2', 'prefetch issue', 'K loop', 'epilogue', 'loop top']
tot = b.sum(1).mean()
print('waves %d  mean total cycles %.0f' % (len(b), tot))
for i, n in enumerate(names):
    col = b[:, [0, 1, 2, 3, 4, 6, 5][i]] if False else b[:, i]
for i, n in zip([7, 6, 0, 1, 2, 3, 4, 5], ['epilogue of a tile (conv_xp only)', 'loop top + acc zero', 'barrier1 (others finish K loop)', 'LDS write (+wait prefetch)', 'barrier2', 'prefetch issue', 'K loop', 'epilogue (incl. continue)']):
    print('%-36s mean %9.0f  (%.1f%%)  min %9.0f max %9.0f' % (n, b[:, i].mean(), 100 * b[:, i].mean() / tot, b[:, i].min(), b[:, i].max()))
