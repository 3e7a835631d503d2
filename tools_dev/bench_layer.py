"""Micro-benchmark of one convolution layer shape (for rocprofv3 --pmc and A/B timing), batched form.
usage: bench_layer.py <2d|3d|deconv> G D H W cin cout k dilation stride [reps]      (2d: D is ignored;
deconv: the fused stride-2 transposed convolution, k / dilation / stride ignored)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import atvsnet_amd
from atvsnet_amd import ops
kind = sys.argv[1]
G, D, H, W, cin, cout, k, dil, stride = [int(v) for v in sys.argv[2:11]]
reps = int(sys.argv[11]) if len(sys.argv) > 11 else 5
dev = torch.device('cuda:0')
nsp = 2 if kind == '2d' else 3
if kind == 'deconv':
    k, stride = 3, 1
x = torch.randn((G,) + ((H, W) if nsp == 2 else (D, H, W)) + (cin,), device=dev)
w = (np.random.default_rng(0).standard_normal((k,) * nsp + (cin, cout)) * 0.1).astype(np.float32)
run = lambda: ops.conv(x, 'bench', w, stride=stride, dilation=dil, want_stats=True, groups=G)     # noqa: E731
if kind == 'deconv':
    wt = np.ascontiguousarray(np.swapaxes(w, -1, -2))          # [3,3,3,Cout,Cin]
    run = lambda: ops.conv3d_transpose_s2(x, 'bench_t', wt, want_stats=True, groups=G)     # noqa: E731
for _ in range(2):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
vox = G * (H // stride) * (W // stride) * (1 if nsp == 2 else D // stride)
gf = 2.0 * (k ** nsp) * cin * cout * vox / 1e9
print('conv%s G=%d %s %d->%d k%d d%d s%d: %.3f ms  %.1f TF/s' % (kind, G, (D, H, W), cin, cout, k, dil, stride, ms, gf / ms))
