"""Quarter- / eighth-resolution 3x3x3 layers, 8 volumes per launch: fp32 MFMA (conv_c16.hip / conv_mfma.hip) vs split-fp16 (conv3d_b.hip)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import atvsnet_amd
from atvsnet_amd import ops
dev = torch.device('cuda:0')
for G, D, H, W, cin, cout in ((8, 48, 32, 40, 32, 32), (4, 48, 32, 40, 32, 32), (8, 24, 16, 20, 64, 64), (4, 24, 16, 20, 64, 64)):
    x = torch.randn(G, D, H, W, cin, device=dev)
    w = (np.random.default_rng(0).standard_normal((3, 3, 3, cin, cout)) * 0.05).astype(np.float32)
    for name, flag in (('fp32', False), ('split-fp16', True)):
        ops.cfg.split16 = flag
        ops.clear_pack_cache()
        run = lambda: ops.conv(x, ('b', cin, cout), w, want_stats=True, groups=G)      # noqa: E731
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
        gf = 2.0 * 27 * cin * cout * G * D * H * W / 1e9
        print('G=%d %dx%dx%d %3d -> %3d  %-10s %.4f ms  %.1f TF/s' % (G, D, H, W, cin, cout, name, ms, gf / ms), flush=True)
for G, D, H, W, cin, cout in ((8, 96, 64, 80, 16, 32), (4, 96, 64, 80, 16, 32), (8, 48, 32, 40, 32, 64), (4, 48, 32, 40, 32, 64)):
    x = torch.randn(G, D, H, W, cin, device=dev)
    w = (np.random.default_rng(0).standard_normal((3, 3, 3, cin, cout)) * 0.05).astype(np.float32)
    for name, flag in (('fp32', False), ('split-fp16', True)):
        ops.cfg.split16 = flag
        ops.clear_pack_cache()
        run = lambda: ops.conv(x, ('s', cin, cout), w, stride=2, want_stats=True, groups=G)      # noqa: E731
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
        gf = 2.0 * 27 * cin * cout * G * D * H * W / 8 / 1e9
        print('stride 2  G=%d %dx%dx%d %3d -> %3d  %-10s %.4f ms  %.1f TF/s' % (G, D, H, W, cin, cout, name, ms, gf / ms), flush=True)
ops.cfg.split16 = True
