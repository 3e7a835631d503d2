"""Socket power and shader clock while ONE x-pair launch repeats for a few seconds (amdgpu hwmon sampled from a second thread):
is the launch running into the package's power management?
    [ATVS_LIB=<variant .so>] python tools_dev/power_probe.py [dominantp|stack|refine|stem] [seconds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import atvsnet_amd                                     # noqa: F401
from atvsnet_amd import ops

which = sys.argv[1] if len(sys.argv) > 1 else 'dominantp'
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
dev = torch.device('cuda:0')
D, H, W = 192, 128, 160
rng = np.random.default_rng(0)
wt = lambda cin, cout: (rng.standard_normal((3, 3, 3, cin, cout)) * 0.1).astype(np.float32)   # noqa: E731
if which in ('dominantp', 'dominantpc'):          # dominantpc: the input as fp16 pieces (staged by LDS-DMA)
    G = 8
    x = torch.randn(G, 4, ops.planar_stride(D, H, W), device=dev)
    pb, pb2 = torch.randn(G, H, W, 24, device=dev), torch.randn(G, H // 2, W // 2, 48, device=dev)
    w8, w16 = wt(32, 8), wt(32, 16)
    run = lambda: ops.conv_siblings(x, 'a8', w8, 'a16', w16, plane_bias=pb, plane_bias2=pb2, groups=G, planar=(D, H, W), pieces=(which == 'dominantpc'))   # noqa: E731
elif which == 'stack':
    G = 8
    xa, xb = torch.randn(G, D, H, W, 8, device=dev), torch.randn(G, D, H, W, 8, device=dev)
    par = torch.stack([torch.randn(G, 8) * 0.1, torch.rand(G, 8) + 0.5, torch.randn(G, 8) * 0.1], 1).to(dev).contiguous()
    v8, v16 = wt(8, 8), wt(8, 16)
    run = lambda: ops.conv_siblings(ops.PendingSum([ops.PendingBN(xa, par, True), ops.PendingBN(xb, par, True)]), 'b8', v8, 'b16', v16, groups=G)   # noqa: E731
else:
    G = 4
    x = torch.randn(G, D, H, W, 16, device=dev)
    pb = torch.randn(G, H, W, 24, device=dev)
    u8 = wt(16, 8)
    run = lambda: ops.conv(x, 'd8', u8, want_stats=True, plane_bias=pb, groups=G)   # noqa: E731

import bench                                           # noqa: E402  (hwmon sampling: bench.power_probe)

for _ in range(20):
    run()
torch.cuda.synchronize()
r = bench.power_probe(run, secs)
print(which, r if r is None else {k: v for k, v in r.items() if k != 'note'})
if r is not None:       # (eager launches incl. the moments' finalize: ms_per_step is an upper bound of the launch's own duration)
    print('%s: %.2f J per launch (%.0f W x %.3f ms)' % (which, r['socket_W'] * r['ms_per_step'] * 1e-3, r['socket_W'], r['ms_per_step']))
