"""The quarter- / eighth-resolution layers of a U-Net batch (8 volumes), graph-timed:  python tools_dev/bench_small_res.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import atvsnet_amd  # noqa: F401
from atvsnet_amd import ops
dev = torch.device('cuda:0')
G = 8
cases = [('2_0  s2 16->32 from 1/2', (96, 64, 80), 16, 32, 2), ('3_0  s2 32->64 from 1/4', (48, 32, 40), 32, 64, 2),
         ('2_1  32->32 at 1/4', (48, 32, 40), 32, 32, 1), ('3_1  64->64 at 1/8', (24, 16, 20), 64, 64, 1),
         ('1_1  16->16 at 1/2', (96, 64, 80), 16, 16, 1)]
for name, shape, cin, cout, stride in cases:
    x = torch.randn((G,) + shape + (cin,), device=dev)
    w = (np.random.default_rng(0).standard_normal((3, 3, 3, cin, cout)) * 0.05).astype(np.float32)
    run = lambda: ops.conv(x, ('sr', name), w, stride=stride, want_stats=True, groups=G)   # noqa: E731
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph(); side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(gr, stream=side):
            for _ in range(10):
                run()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    print('%-26s %.1f us' % (name, e0.elapsed_time(e1) / 50 * 1e3), flush=True)
