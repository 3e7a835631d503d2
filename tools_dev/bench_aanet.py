"""One AANet module (aanet_b.hip: score convolutions + cross-view softmax in one launch) against the two-launch form, at configs[2]
(4 sources, 192x128x160) and configs[3] (8 sources, 256x120x232): graph-timed, bitwise comparison.   python tools_dev/bench_aanet.py [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import atvsnet_amd  # noqa: F401
from atvsnet_amd import ops, variables
from atvsnet_amd.cnn_wrapper.atvsnet import AttAggregation_keepchannel

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda:0')
variables.default_store().init_synthetic(1234)
for nv, shape in ((4, (192, 128, 160)), (5, (192, 128, 160)), (8, (256, 120, 232)), (2, (192, 128, 160))):
    x = torch.randn((nv,) + shape + (8,), device=dev)
    outs = {}
    for fused in (True, False):
        with ops.configure(aanet_fused=fused):
            run = lambda: AttAggregation_keepchannel({'data': x}, is_training=True).get_output()      # noqa: E731
            for _ in range(2):
                y = run()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                y = run()
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            outs[fused] = (y.clone(), e0.elapsed_time(e1) / reps)
    V = float(np.prod(shape))
    gf = nv * 2.0 * 27 * 8 * 16 * V / 1e9
    print('%d views %s: one launch %.3f ms (%.0f TFLOP/s algorithmic), two launches %.3f ms, bitwise equal: %s, max |diff| / max %.1e'
          % (nv, shape, outs[True][1], gf / outs[True][1], outs[False][1], torch.equal(outs[True][0], outs[False][0]),
             float((outs[True][0] - outs[False][0]).abs().max() / outs[False][0].abs().max())), flush=True)
