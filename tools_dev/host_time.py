import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import atvsnet_amd
from atvsnet_amd import ops, synthetic, variables
from atvsnet_amd.atvsnet import example as ex
dev = torch.device('cuda:0')
variables.default_store().init_synthetic(1234)
imgs, cams = synthetic.make_inputs(5, 512, 640, 192)
imgs, cams = torch.from_numpy(imgs).to(dev), torch.from_numpy(cams).to(dev)
for streams in (True, False):
    for _ in range(2):
        ex.infer_multiview(imgs, cams, 192, view_streams=streams)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = ex.infer_multiview(imgs, cams, 192, view_streams=streams)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('streams', streams, 'host issue ms/step %.1f  total ms/step %.1f' % ((t1 - t0) / 3 * 1e3, (t2 - t0) / 3 * 1e3))
