"""1x1 tower convolutions, 5 images of 128x160 per launch: fp32 MFMA (conv1x1.hip) vs split-fp16 (conv1x1_b.hip)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import atvsnet_amd
from atvsnet_amd import ops
dev = torch.device('cuda:0')
G, H, W = 5, 128, 160
for cin, cout in ((128, 128), (128, 32), (32, 128), (64, 128), (64, 64), (32, 64)):
    x = torch.randn(G, H, W, cin, device=dev)
    w = (np.random.default_rng(0).standard_normal((1, 1, cin, cout)) * 0.05).astype(np.float32)
    for name, flag in (('fp32', False), ('split-fp16', True)):
        ops.cfg.split16 = flag
        ops.clear_pack_cache()
        run = lambda: ops.conv1x1(x, ('b', cin, cout), w, want_stats=True)      # noqa: E731
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        gb = 4.0 * (cin + cout) * G * H * W / 1e9
        print('%3d -> %3d  %-10s %.4f ms  %.2f TB/s' % (cin, cout, name, ms, gb / ms), flush=True)
ops.cfg.split16 = True
