"""Per-layer HIP-event times of the two 2-D towers on the five 640x512 views of configs[2] (batched as the pipeline runs them).
   python tools_dev/profile_towers.py"""
import collections, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import atvsnet_amd  # noqa: F401
from atvsnet_amd import ops, synthetic, variables
from atvsnet_amd.atvsnet import model
dev = torch.device('cuda:0')
variables.default_store().init_synthetic(1234)
imgs, cams = synthetic.make_inputs(5, 512, 640, 192)
imgs = torch.from_numpy(imgs).to(dev)
for name, fn in (('deep tower (feature_extraction_batch)', lambda: model.feature_extraction_batch(imgs)),
                 ('shallow tower (shallow_feature_batch)', lambda: model.shallow_feature_batch(imgs))):
    fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print('%s: %.3f ms per call (graph replay)' % (name, e0.elapsed_time(e1) / 20))
    ops.watch('*')
    fn()
    ev = ops.watch(None)
    tot = 0.0
    for key, shp, cout, ms in ev:
        tot += ms
        print('   %-58s %-22s -> %3d  %7.1f us' % (str(key)[:58], str(tuple(shp)), cout, ms * 1e3))
    print('   watched launches: %d, %.3f ms (eager, includes launch gaps)' % (len(ev), tot))
