"""Replay the captured depth-map pipeline N times and compare EVERY output with the first, bit for bit (a race between wavefronts or
an LDS-DMA that lands late shows up as a rare mismatch):   python tools_dev/soak_graph.py [cfg3|cfg2|cfg4] [replays]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import atvsnet_amd  # noqa: F401
from atvsnet_amd import synthetic, variables
from atvsnet_amd.atvsnet import example as ex
which = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
views, W, H, D = {'cfg3': (5, 640, 512, 192), 'cfg2': (2, 640, 512, 192), 'cfg4': (9, 928, 480, 256)}[which]
dev = torch.device('cuda:0')
variables.default_store().init_synthetic(1234)
imgs, cams = synthetic.make_inputs(views, H, W, D)
imgs, cams = torch.from_numpy(imgs).to(dev), torch.from_numpy(cams).to(dev)
gr = ex.GraphedInference(imgs, cams, D)
first = gr().clone()
bad = 0
t0 = time.time()
for i in range(n):
    out = gr()
    if not torch.equal(out, first):
        bad += 1
        print('replay %d differs: max |d| %.3e at %d elements' % (i, float((out - first).abs().max()), int((out != first).sum())), flush=True)
torch.cuda.synchronize()
print('%s: %d replays in %.1f s, %d differ from the first' % (which, n, time.time() - t0, bad))
sys.exit(1 if bad else 0)
