"""refine_stems at full size (4 volumes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import numpy as np
import torch
import atvsnet_amd
from atvsnet_amd import ops, _lib
dev = torch.device('cuda:0')
torch.manual_seed(0)
G, D, H, W = 4, 192, 128, 160
photo = torch.randn(G, D, H, W, 8, device=dev)
geo = torch.randn(G, D, H, W, 2, device=dev)
prob = torch.randn(G, D, H, W, 1, device=dev)
hull = torch.randn(G, D, H, W, 1, device=dev)
pb = torch.randn(G, H, W, 24, device=dev)
w = torch.randn(27 * 32, device=dev) * 0.1
y = torch.empty(G, D, H, W, 32, device=dev)
rows = int(_lib.lib().atvs_conv_stem_rows(D, H, W))
st = torch.empty(G * rows, 2, 24, dtype=torch.float64, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
run = lambda: ops._call('atvs_refine_stems_f32', P(photo), P(geo), P(pb), P(prob), P(hull), P(w), P(y), P(st), G, D, H, W, ops._stream())
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record()
torch.cuda.synchronize()
print('refine_stems %.3f ms  checksum %.6e' % (e0.elapsed_time(e1) / 10, float(y.double().sum())))
if len(sys.argv) > 1:          # dump for a bitwise A/B of two builds: python tools_dev/bench_stems.py out.pt
    torch.save({'y': y[:, ::16].cpu(), 'st': st.cpu(), 'ysum': y.double().sum(dim=(1, 2, 3)).cpu()}, sys.argv[1])
