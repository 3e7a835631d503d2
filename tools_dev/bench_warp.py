"""The plane-sweep warp of one view (32 channels, 192 planes of 128x160) into the chunk-planar cost-volume half."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import atvsnet_amd
from atvsnet_amd import ops, synthetic
dev = torch.device('cuda:0')
D, h, w, C = 192, 128, 160, 32
src = torch.randn(h, w, C, device=dev)
imgs, cams = synthetic.make_inputs(2, 512, 640, D)
cams = torch.from_numpy(cams).to(dev)[0]
cams4 = cams.clone()
cams4[:, 1, :2, :3] /= 4.0          # intrinsics of the quarter-resolution feature maps
ds, di = cams[0, 1, 3, 0:1].contiguous(), cams[0, 1, 3, 1:2].contiguous()
Hm = ops.get_homographies(cams4[0], cams4[1], ds, di, D)
out = torch.empty(C // 8, ops.planar_stride(D, h, w), device=dev)
for pieces in (False, True, False, True):
    run = lambda: ops.warp_planes(src, Hm, out=out, planar=True, pieces=pieces)      # noqa: E731
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 100
    print('warp (%s) %.4f ms  %.2f TB/s written' % ('fp16 pieces' if pieces else 'fp32 planar', ms, 4.0 * D * h * w * C / 1e9 / ms))
