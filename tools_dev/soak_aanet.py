"""aanet_b.hip alone, repeated: every launch of the two-role kernel (LDS hand-off between wavefronts, one LDS-only barrier per stage,
halo requests in flight across it) must produce the bits of the first -- a race shows up as a rare mismatch.
   python tools_dev/soak_aanet.py [repetitions per form]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import atvsnet_amd  # noqa: F401
from atvsnet_amd import variables
from atvsnet_amd.cnn_wrapper.atvsnet import AttAggregation_keepchannel

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device('cuda:0')
variables.default_store().init_synthetic(1234)
bad_total = 0
for nv, shape in ((4, (192, 128, 160)), (8, (128, 120, 232)), (1, (64, 48, 64)), (3, (37, 29, 53)), (5, (64, 64, 80))):
    x = torch.randn((nv,) + shape + (8,), device=dev)
    g = torch.cuda.CUDAGraph()
    AttAggregation_keepchannel({'data': x}, is_training=True).get_output()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        y = AttAggregation_keepchannel({'data': x}, is_training=True).get_output()
    g.replay()
    first = y.clone()
    bad = 0
    for i in range(reps):
        g.replay()
        if not torch.equal(y, first):
            bad += 1
    torch.cuda.synchronize()
    print('%d views %s: %d launches, %d differ from the first' % (nv, shape, reps, bad), flush=True)
    bad_total += bad
sys.exit(1 if bad_total else 0)
