"""The 8 -> 1 probability heads at full size: G volumes of 192x128x160x8."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import atvsnet_amd
from atvsnet_amd import ops
dev = torch.device('cuda:0')
w = torch.randn(3, 3, 3, 8, 1, device=dev) * 0.1
for G in (8, 4, 1):
    x = torch.randn(G, 192, 128, 160, 8, device=dev)
    run = lambda: ops.conv3d_8to1(x, w, groups=G)      # noqa: E731
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print('G=%d  %.3f ms  %.2f TB/s (36 B per voxel)' % (G, ms, 36.0 * x.numel() / 8 / 1e9 / ms), flush=True)
