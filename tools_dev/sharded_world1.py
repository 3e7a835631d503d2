"""The sharded path with a one-rank RCCL group at the bench workload: graph segments vs eager issue (what
bench.py runs per rank for --gpus N > 1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29612")
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import atvsnet_amd
from atvsnet_amd import synthetic, variables, parallel
variables.default_store().init_synthetic(1234)
imgs, cams = synthetic.make_inputs(5, 512, 640, 192)
imgs, cams = torch.from_numpy(imgs).to(dev), torch.from_numpy(cams).to(dev)
g = parallel.ShardedGraphedInference(imgs, cams, 192)
for mode, fn in (("graph segments", g), ("eager sharded", lambda: parallel.infer_multiview_sharded(imgs, cams, 192))):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    print(mode, "%.2f ms per depth map" % ((time.perf_counter() - t) / 4 * 1e3))
dist.destroy_process_group()
