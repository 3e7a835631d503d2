"""Throughput with K depth maps in flight (one captured graph + static buffers per slot, one stream per slot)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import atvsnet_amd
from atvsnet_amd import synthetic, variables
from atvsnet_amd.atvsnet import example as ex
dev = torch.device('cuda:0')
variables.default_store().init_synthetic(1234)
imgs, cams = synthetic.make_inputs(5, 512, 640, 192)
imgs, cams = torch.from_numpy(imgs).to(dev), torch.from_numpy(cams).to(dev)
for k in (1, 2, 3):
    gs = [ex.GraphedInference(imgs, cams, 192) for _ in range(k)]
    ss = [torch.cuda.Stream(dev) for _ in range(k)]
    def run(n):
        for i in range(n):
            with torch.cuda.stream(ss[i % k]):
                gs[i % k]()
    run(2 * k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 12
    run(n)
    torch.cuda.synchronize()
    print('in flight %d: %.2f ms per depth map' % (k, (time.perf_counter() - t0) / n * 1e3))
    del gs
