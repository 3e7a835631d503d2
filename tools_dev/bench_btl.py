"""Fused residual unit (bottleneck_b.hip) against its three launches, at the tower's shapes (5 images per launch).
   python tools_dev/bench_btl.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import atvsnet_amd  # noqa: F401
from atvsnet_amd import ops
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
for C, H, W in ((64, 128, 160), (32, 256, 320)):
    G = 5
    x = torch.randn(G, H, W, C, device=dev)
    beta = torch.zeros(C, device=dev)
    params = ops.bn_params(ops.channel_stats(x, groups=G), C, x, beta)
    w1, w3 = [(rng.standard_normal((1, 1, C, C)) * (1.0 / C) ** 0.5).astype(np.float32) for _ in range(2)]
    w2 = (rng.standard_normal((3, 3, C, C)) * (1.0 / (9 * C)) ** 0.5).astype(np.float32)
    b = [torch.randn(C, device=dev) * 0.1 for _ in range(3)]
    keys = tuple(('bench', C, n) for n in range(3))

    def fused():
        return ops.bottleneck(x, params, keys, w1, b[0], w2, b[1], w3, b[2])

    def three():
        r = ops.conv(x, keys[0], w1, bias=b[0], relu=True, groups=G, in_params=params, in_relu=True)
        r = ops.conv(r, keys[1], w2, bias=b[1], relu=True, groups=G)
        return ops.conv(r, keys[2], w3, bias=b[2], residual=x, want_stats=True, groups=G)

    for name, fn in (('three launches', three), ('fused unit', fused)):
        for _ in range(3):
            y, _ = fn()
        torch.cuda.synchronize()
        # 20 units back to back in one HIP graph: the GPU's time, not Python's launch rate
        gr = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            with torch.cuda.graph(gr, stream=side):
                for _ in range(20):
                    fn()
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 100
        gf = 2.0 * 11 * C * C * G * H * W / 1e9
        print('C=%3d %dx%d x%d  %-15s %.1f us  %.1f TF/s algorithmic' % (C, H, W, G, name, ms * 1e3, gf / ms), flush=True)
