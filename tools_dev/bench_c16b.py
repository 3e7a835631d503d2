"""16 -> 16 channel 3x3x3 convolution (conv_b*_1_1: 8 volumes of 96x64x80): fp32 MFMA (conv_c16.hip) vs split-fp16 (conv_c16b.hip),
time and error against a float64 reference on a sub-volume."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import atvsnet_amd
from atvsnet_amd import ops
from oracle import tf_ops as T
dev = torch.device('cuda:0')
for cin, G, D, H, W in ((16, 8, 96, 64, 80), (8, 4, 192, 128, 160)):
  print('%d -> 16 channels, %d volumes of %dx%dx%d' % (cin, G, D, H, W))
  x = torch.randn(G, D, H, W, cin, device=dev)
  w = (np.random.default_rng(0).standard_normal((3, 3, 3, cin, 16)) * 0.1).astype(np.float32)
  ref = T.conv(x[:1, :12].cpu().double(), torch.from_numpy(w).double(), 1, 'SAME')[0, 1:-1]
  for name, flag in (('fp32 MFMA (conv_c16)', False), ('split fp16 x2 (conv_c16b)', True)):
    ops.cfg.split16 = flag
    ops.clear_pack_cache()
    run = lambda: ops.conv(x, ('b', cin), w, want_stats=True, groups=G, relu=(cin == 8))      # noqa: E731
    for _ in range(3):
        y, st = run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    gf = 2.0 * 27 * cin * 16 * G * D * H * W / 1e9
    r = ref.clamp(min=0) if cin == 8 else ref
    err = float((y[0, 1:11].cpu().double() - r).abs().max() / r.abs().max())
    print('%-28s %.3f ms  %.1f TF/s (fp32-equivalent)  max err / max vs float64: %.2e' % (name, ms, gf / ms, err), flush=True)
ops.cfg.split16 = True
