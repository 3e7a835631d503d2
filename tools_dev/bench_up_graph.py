import sys, os
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd())
import numpy as np, torch
import atvsnet_amd
from atvsnet_amd import ops
dev = torch.device('cuda:0')
for G, D, H, W, cin, cout in ((8, 48, 32, 40, 32, 16), (4, 48, 32, 40, 32, 16), (8, 24, 16, 20, 64, 32), (8, 96, 64, 80, 16, 8)):
    x = torch.randn(G, D, H, W, cin, device=dev)
    w = (np.random.default_rng(0).standard_normal((3, 3, 3, cout, cin)) * 0.05).astype(np.float32)
    run = lambda: ops.conv3d_transpose_s2(x, ('u', cin, cout), w, want_stats=True, groups=G)
    for _ in range(3): run()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph(); side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(gr, stream=side):
            for _ in range(10): run()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): gr.replay()
    e1.record(); torch.cuda.synchronize()
    print('G=%d %dx%dx%d %3d -> %3d  %.1f us per layer' % (G, D, H, W, cin, cout, e0.elapsed_time(e1) / 50 * 1e3), flush=True)
