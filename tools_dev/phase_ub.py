"""Per-phase tick counts of deconv_up_b.hip (development build: bash tools_dev/build_variant.sh ubdbg deconv_up_b -DATVS_UB_DEBUG;
ATVS_LIB=tools_dev/_dbg/lib_ubdbg.so python tools_dev/phase_ub.py [4_0|5_0|6_0])"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import atvsnet_amd  # noqa: F401
from atvsnet_amd import ops, _lib
which = sys.argv[1] if len(sys.argv) > 1 else '4_0'
dev = torch.device('cuda:0')
G = 8
D, H, W, cin, cout = {'4_0': (24, 16, 20, 64, 32), '5_0': (48, 32, 40, 32, 16), '6_0': (96, 64, 80, 16, 8)}[which]
x = torch.randn(G, D, H, W, cin, device=dev)
w = (np.random.default_rng(0).standard_normal((3, 3, 3, cout, cin)) * 0.1).astype(np.float32)
for _ in range(3):
    ops.conv3d_transpose_s2(x, ('ph', which), w, want_stats=True, groups=G)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.conv3d_transpose_s2(x, ('ph', which), w, want_stats=True, groups=G)
e1.record()
torch.cuda.synchronize()
print('%s: %.1f us for the layer (eager, one call)' % (which, 1e3 * e0.elapsed_time(e1)))
buf = np.zeros(4096 * 8, np.uint64)
assert _lib.lib().atvs_debug_read_ub(buf.ctypes.data_as(ctypes.c_void_p)) == 0
raw = buf.reshape(-1, 8)
raw = raw[raw[:, 7] > 0]
t = raw.astype(np.float64)
ns = (raw[:, 7] & np.uint64(0xffff)).astype(np.float64).mean()
wall = (raw[:, 7] >> np.uint64(16)).astype(np.float64) / 100.0          # us inside the kernel, per wavefront
print('wall clock inside the kernel per wavefront: mean %.1f us, max %.1f us; setup before the first stage %.0f ticks' % (wall.mean(), wall.max(), t[:, 6].mean()))
start = raw[:, 0].astype(np.float64) / 100.0
end = start + wall
print('workgroup starts spread over %.1f us; first start -> last end %.1f us' % (start.max() - start.min(), end.max() - start.min()))
hist, _ = np.histogram(start - start.min(), bins=8)
print('start-time histogram (8 bins over the spread):', hist.tolist())
print('ticks per us of wall clock: %.0f' % ((t[:, :7].sum(1) / np.maximum(wall, 1e-3)).mean()))
names = ['loop top (acc zero)', 'barrier A (images free)', 'weights -> LDS (streamed form), split + LDS write', 'barrier B', 'K loop (+ next halo requests)', 'epilogue stores (last chunk)']
tot = t[:, :6].sum(1).mean()
print('%d wavefronts, %.1f stages each, %.0f ticks per stage' % (len(t), ns, tot / ns))
for i, n in enumerate(names):
    print('   %-55s %8.0f per stage (%.1f%%)' % (n, t[:, i].mean() / ns, 100 * t[:, i].mean() / tot))
