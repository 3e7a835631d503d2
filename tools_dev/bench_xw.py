"""The four x-pair launches of a depth map as bench.py issues them (batched), timed one by one with HIP events:
usage: [ATVS_LIB=tools_dev/_dbg/lib_X.so] python tools_dev/bench_xw.py [reps] [xb|xw]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import atvsnet_amd
from atvsnet_amd import ops
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
kind = sys.argv[2] if len(sys.argv) > 2 else 'xb'
ops.cfg.xb = kind == 'xb'
print('x-pair kernel:', kind)
dev = torch.device('cuda:0')
D, H, W = 192, 128, 160
rng = np.random.default_rng(0)


def wt(cin, cout):
    return (rng.standard_normal((3, 3, 3, cin, cout)) * 0.1).astype(np.float32)


def timed(name, fn, gf):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print('%-46s %.3f ms  %.1f TF/s' % (name, ms, gf / ms), flush=True)


V, V2 = D * H * W, (D // 2) * (H // 2) * (W // 2)
# 1. dominant: 32 warped channels -> 8 | 16 (stride 2), plane biases, 8 volumes
G = 8
planar = True                  # as the pipeline launches it: the warped half of the cost volume chunk-planar
x = torch.randn(G, 4, ops.planar_stride(D, H, W), device=dev) if planar else torch.randn(G, D, H, W, 32, device=dev)
pb, pb2 = torch.randn(G, H, W, 24, device=dev), torch.randn(G, H // 2, W // 2, 48, device=dev)
w8, w16 = wt(32, 8), wt(32, 16)
timed('conv_b0_0_1|1_0  32->8|16  G=8 (dominant%s)' % (', planar' if planar else ''),
      lambda: ops.conv_siblings(x, 'a8', w8, 'a16', w16, plane_bias=pb, plane_bias2=pb2, groups=G, planar=(D, H, W) if planar else False),
      G * (2.0 * 27 * 32 * 8 * V + 2.0 * 27 * 32 * 16 * V2) / 1e9)
del x, pb, pb2
# 2. stack inputs: two 8-channel sources, normalise + add on load, 8 volumes
xa, xb = torch.randn(G, D, H, W, 8, device=dev), torch.randn(G, D, H, W, 8, device=dev)
par = torch.stack([torch.randn(G, 8) * 0.1, torch.rand(G, 8) + 0.5, torch.randn(G, 8) * 0.1], 1).to(dev).contiguous()
v8, v16 = wt(8, 8), wt(8, 16)


def stack():
    lazy = ops.PendingSum([ops.PendingBN(xa, par, True), ops.PendingBN(xb, par, True)])
    return ops.conv_siblings(lazy, 'b8', v8, 'b16', v16, groups=G)


timed('conv_b1_0_1|1_0  8+8->8|16 add-on-load G=8', stack, G * (2.0 * 27 * 8 * 8 * V + 2.0 * 27 * 8 * 16 * V2) / 1e9)
del xa, xb
# 3. refinement: 32-channel concat, normalise on load, 4 volumes
G = 4
planar = kind == 'xb'
x = torch.randn(G, 4, ops.planar_stride(D, H, W), device=dev) if planar else torch.randn(G, D, H, W, 32, device=dev)
par = torch.stack([torch.randn(G, 32) * 0.1, torch.rand(G, 32) + 0.5, torch.randn(G, 32) * 0.1], 1).to(dev).contiguous()
timed('3dconv0_1|1_0  32->8|16 normalise-on-load G=4%s' % (', planar' if planar else ''),
      lambda: ops.conv_siblings(ops.PendingBN(x, par, True, planar=(D, H, W) if planar else None), 'c8', w8, 'c16', w16, groups=G),
      G * (2.0 * 27 * 32 * 8 * V + 2.0 * 27 * 32 * 16 * V2) / 1e9)
del x
# 4. photo stem: 16 -> 8, plane bias, 4 volumes
x = torch.randn(G, D, H, W, 16, device=dev)
pb = torch.randn(G, H, W, 24, device=dev)
u8 = wt(16, 8)
timed('photo stem  16->8  G=4', lambda: ops.conv(x, 'd8', u8, want_stats=True, plane_bias=pb, groups=G), G * 2.0 * 27 * 16 * 8 * V / 1e9)
