"""Socket power and shader clock while ONE x-pair launch repeats for a few seconds (rocm-smi sampled from a second thread):
is the launch running into the package's power management?
    [ATVS_LIB=<variant .so>] python tools_dev/power_probe.py [dominantp|stack|refine|stem] [seconds]"""
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import atvsnet_amd                                     # noqa: F401
from atvsnet_amd import ops

which = sys.argv[1] if len(sys.argv) > 1 else 'dominantp'
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
dev = torch.device('cuda:0')
D, H, W = 192, 128, 160
rng = np.random.default_rng(0)
wt = lambda cin, cout: (rng.standard_normal((3, 3, 3, cin, cout)) * 0.1).astype(np.float32)   # noqa: E731
if which == 'dominantp':
    G = 8
    x = torch.randn(G, 4, ops.planar_stride(D, H, W), device=dev)
    pb, pb2 = torch.randn(G, H, W, 24, device=dev), torch.randn(G, H // 2, W // 2, 48, device=dev)
    w8, w16 = wt(32, 8), wt(32, 16)
    run = lambda: ops.conv_siblings(x, 'a8', w8, 'a16', w16, plane_bias=pb, plane_bias2=pb2, groups=G, planar=(D, H, W))   # noqa: E731
elif which == 'stack':
    G = 8
    xa, xb = torch.randn(G, D, H, W, 8, device=dev), torch.randn(G, D, H, W, 8, device=dev)
    par = torch.stack([torch.randn(G, 8) * 0.1, torch.rand(G, 8) + 0.5, torch.randn(G, 8) * 0.1], 1).to(dev).contiguous()
    v8, v16 = wt(8, 8), wt(8, 16)
    run = lambda: ops.conv_siblings(ops.PendingSum([ops.PendingBN(xa, par, True), ops.PendingBN(xb, par, True)]), 'b8', v8, 'b16', v16, groups=G)   # noqa: E731
else:
    G = 4
    x = torch.randn(G, D, H, W, 16, device=dev)
    pb = torch.randn(G, H, W, 24, device=dev)
    u8 = wt(16, 8)
    run = lambda: ops.conv(x, 'd8', u8, want_stats=True, plane_bias=pb, groups=G)   # noqa: E731

samples, stop = [], False


def sampler():
    while not stop:
        try:
            out = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--json'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                 timeout=5).stdout.decode()
            card = list(json.loads(out).values())[0]
            pw = [float(v) for k, v in card.items() if 'Power' in k and 'W' in k]
            sclk = [v for k, v in card.items() if k.startswith('sclk')]
            samples.append((time.time(), pw[0] if pw else float('nan'), sclk[0] if sclk else '?'))
        except Exception as e:          # noqa: BLE001
            samples.append((time.time(), float('nan'), repr(e)))
        time.sleep(0.2)


for _ in range(20):
    run()
torch.cuda.synchronize()
th = threading.Thread(target=sampler)
th.start()
t0 = time.time()
n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < secs:
    for _ in range(50):
        run()
    n += 50
    torch.cuda.synchronize()
e1.record()
torch.cuda.synchronize()
stop = True
th.join()
print('%s: %.3f ms per launch over %d launches (%.1f s, host loop included)' % (which, e0.elapsed_time(e1) / n, n, time.time() - t0))
for t, pw, clk in samples:
    print('   t = %4.1f s   %7.1f W   sclk %s' % (t - t0, pw, clk))
