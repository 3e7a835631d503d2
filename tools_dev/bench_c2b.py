"""3x3 tower convolutions, 5 images of 128x160 per launch: fp32 MFMA (conv2d_lds.hip) vs split-fp16 (conv2d_b.hip)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import atvsnet_amd
from atvsnet_amd import ops
dev = torch.device('cuda:0')
G, H, W = 5, 128, 160
for cin, cout, dil in ((128, 128, 2), (128, 128, 4), (320, 128, 1), (64, 64, 1), (32, 32, 1)):
    x = torch.randn(G, H * (2 if cout == 32 else 1), W * (2 if cout == 32 else 1), cin, device=dev)
    w = (np.random.default_rng(0).standard_normal((3, 3, cin, cout)) * 0.05).astype(np.float32)
    for name, flag in (('fp32', False), ('split-fp16', True)):
        ops.cfg.split16 = flag
        ops.clear_pack_cache()
        run = lambda: ops.conv2d_lds(x, ('b', cin, cout, dil), w, dil, want_stats=True)      # noqa: E731
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
        gf = 2.0 * 9 * cin * cout * x.shape[0] * x.shape[1] * x.shape[2] / 1e9
        print('%3d -> %3d dil %d  %-10s %.3f ms  %.1f TF/s' % (cin, cout, dil, name, ms, gf / ms), flush=True)
ops.cfg.split16 = True
