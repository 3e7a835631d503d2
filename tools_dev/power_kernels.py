"""Socket power / shader clock of single launches repeated for seconds (bench.power_probe): which kernels of the step sit at the
package's cap?   python tools_dev/power_kernels.py [seconds]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import atvsnet_amd                                     # noqa: F401
from atvsnet_amd import ops
import bench

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
wt = lambda *shape: (rng.standard_normal(shape) * 0.1).astype(np.float32)   # noqa: E731
D, H, W = 192, 128, 160
cases = {}
x8 = torch.randn(4, D, H, W, 8, device=dev)
w_aan = wt(3, 3, 3, 8, 16)
cases['conv_c16b 8 -> 16, 4 full-resolution volumes (AANet scores)'] = lambda: ops.conv(x8, 'pk_aan', w_aan, want_stats=False, groups=4)
x16 = torch.randn(8, D // 2, H // 2, W // 2, 16, device=dev)
w16 = wt(3, 3, 3, 16, 16)
cases['conv_c16b 16 -> 16, 8 half-resolution volumes'] = lambda: ops.conv(x16, 'pk_c16', w16, want_stats=True, groups=8)
xd = torch.randn(8, D // 2, H // 2, W // 2, 16, device=dev)
wd = wt(3, 3, 3, 8, 16)
cases['deconv_up_b 16 -> 8 to full resolution, 8 volumes'] = lambda: ops.conv3d_transpose_s2(xd, 'pk_up', wd, want_stats=True, groups=8)
a8, b8 = torch.randn(8, D, H, W, 8, device=dev), torch.randn(8, D, H, W, 8, device=dev)
par = torch.stack([torch.randn(8, 8) * 0.1, torch.rand(8, 8) + 0.5, torch.randn(8, 8) * 0.1], 1).to(dev).contiguous()
cases['bn_add, 8 full-resolution 8-channel volumes'] = lambda: ops.bn_add([ops.PendingBN(a8, par, True), ops.PendingBN(b8, par, False)])
img = torch.randn(5, 128, 160, 128, device=dev)
w2 = wt(3, 3, 128, 128)
cases['conv2d_b 128 -> 128 3x3, five images'] = lambda: ops.conv(img, 'pk_2d', w2, want_stats=True, groups=5)
for name, run in cases.items():
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    r = bench.power_probe(run, secs)
    print('%-62s %s' % (name, None if r is None else '%.3f ms  %6.1f W  %4d MHz' % (r['ms_per_step'], r['socket_W'], r['sclk_MHz'])))
