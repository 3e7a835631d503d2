"""Micro-benchmark of one convolution configuration (for rocprofv3 --pmc and A/B timing)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import atvsnet_amd
from atvsnet_amd import ops
D, H, W, cin, cout = [int(v) for v in sys.argv[1:6]]
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
mode = sys.argv[7] if len(sys.argv) > 7 else ''
with_pb = mode in ('pb', 'sib')      # depth-plane bias (H, W, 3*cout), as the cost-volume layers have
sib = mode == 'sib'                  # + the stride-2 16-channel sibling in the same launch (conv_b0_0_1 | conv_b0_1_0)
dev = torch.device('cuda:0')
x = torch.randn(D, H, W, cin, device=dev)
w = (np.random.default_rng(0).standard_normal((3, 3, 3, cin, cout)) * 0.1).astype(np.float32)
pb = torch.randn(H, W, 3 * cout, device=dev) if with_pb else None
if sib:
    w2 = (np.random.default_rng(1).standard_normal((3, 3, 3, cin, 16)) * 0.1).astype(np.float32)
    pb2 = torch.randn((H + 1) // 2, (W + 1) // 2, 48, device=dev)
    run = lambda: ops.conv_siblings(x, 'bench', w, 'bench2', w2, plane_bias=pb, plane_bias2=pb2)     # noqa: E731
else:
    run = lambda: ops.conv(x, 'bench', w, want_stats=True, plane_bias=pb)                            # noqa: E731
for _ in range(2):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
gf = 2.0 * 27 * cin * cout * D * H * W / 1e9
if sib:
    gf += 2.0 * 27 * cin * 16 * ((D + 1) // 2) * ((H + 1) // 2) * ((W + 1) // 2) / 1e9
print('conv %dx%dx%d %d->%d%s: %.3f ms  %.1f TF/s useful' % (D, H, W, cin, cout, (' +plane bias' if with_pb else '') + (' +sibling' if sib else ''), ms, gf / ms))
