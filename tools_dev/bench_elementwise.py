"""Times of the element-wise batch-norm passes (norm.hip) at the sizes the cfg3 pipeline launches them:
python tools_dev/bench_elementwise.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import atvsnet_amd                                     # noqa: F401
from atvsnet_amd import ops

dev = torch.device('cuda:0')


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for G, D, H, W, C in [(4, 192, 128, 160, 8), (4, 96, 64, 80, 16), (4, 48, 32, 40, 32), (1, 192, 128, 160, 8), (4, 192, 128, 160, 32)]:
    x0, x1 = torch.randn(G, D, H, W, C, device=dev), torch.randn(G, D, H, W, C, device=dev)
    par = torch.stack([torch.randn(G, C) * 0.1, torch.rand(G, C) + 0.5, torch.randn(G, C) * 0.1], 1).to(dev).contiguous()
    nbytes = x0.numel() * 4
    t = timed(lambda: ops.bn_add([ops.PendingBN(x0, par, True), ops.PendingBN(x1, par, False)]))
    print('bn_add   G=%d %dx%dx%dx%d: %.3f ms, %.2f TB/s (2 reads + 1 write)' % (G, D, H, W, C, t, 3 * nbytes / t / 1e9))
    out = torch.empty_like(x0)
    t = timed(lambda: ops.bn_apply(x0, par, relu=True, out=out))
    print('bn_apply G=%d %dx%dx%dx%d: %.3f ms, %.2f TB/s (1 read + 1 write)' % (G, D, H, W, C, t, 2 * nbytes / t / 1e9))
    t = timed(lambda: ops.add_n([x0, x1], out=out))
    print('add_n    G=%d %dx%dx%dx%d: %.3f ms, %.2f TB/s (2 reads + 1 write)' % (G, D, H, W, C, t, 3 * nbytes / t / 1e9))
