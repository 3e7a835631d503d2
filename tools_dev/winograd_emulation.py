"""CPU emulation of the Winograd F(2,3)-along-y form of the 8-output-channel 3x3x3 convolutions (the x-pair layers), patched
into the oracle: how far does the final depth map move?  (numerics gate for csrc/conv_xw.hip)

    python tools_dev/winograd_emulation.py cfg1 | cfg2

Transforms: t = B^T d = [d0-d2, d1+d2, d2-d1, d1-d3]; U = G g = [g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2] (float64 -> float32);
y0 = m0+m1+m2, y1 = m1-m2-m3, every product / sum in float32."""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import atvsnet_amd                                     # noqa: E402,F401
from atvsnet_amd import synthetic, variables           # noqa: E402
from oracle import model as OM                         # noqa: E402
from oracle import tf_ops as T                         # noqa: E402

_plain = T.conv
COUNT = {'wino': 0, 'plain': 0}


def wino_y(x, w):
    """x (B,D,H,W,C), w [3,3,3,C,8] -> SAME stride-1 convolution via F(2,3) along H."""
    B, D, H, W, C = x.shape
    He = H + (H & 1)
    xc = x.permute(0, 4, 1, 2, 3)
    xp = F.pad(xc, (1, 1, 1, 1 + He - H, 1, 1))                    # (B,C,D+2,He+2,W+2)
    d0, d1, d2, d3 = xp[:, :, :, 0:He:2], xp[:, :, :, 1:He + 1:2], xp[:, :, :, 2:He + 2:2], xp[:, :, :, 3:He + 3:2]
    t = [d0 - d2, d1 + d2, d2 - d1, d1 - d3]
    g = w.double()
    U = [g[:, 0], (g[:, 0] + g[:, 1] + g[:, 2]) / 2, (g[:, 0] - g[:, 1] + g[:, 2]) / 2, g[:, 2]]    # [3(z),3(x),C,8]
    m = []
    for tp, u in zip(t, U):
        k = u.float().permute(3, 2, 0, 1).unsqueeze(3).contiguous()       # [8,C,3,1,3]
        m.append(F.conv3d(tp.contiguous(), k))
    y0 = (m[0] + m[1]) + m[2]
    y1 = (m[1] - m[2]) - m[3]
    y = torch.stack([y0, y1], 4).reshape(B, 8, D, He, W)[:, :, :, :H]
    return y.permute(0, 2, 3, 4, 1).contiguous()


def conv(x, w, stride=1, padding='SAME', dilation=1, bias=None, explicit_pad=None):
    if (x.dim() == 5 and tuple(w.shape[:3]) == (3, 3, 3) and w.shape[4] == 8 and stride == 1 and padding == 'SAME'
            and dilation == 1 and explicit_pad is None and w.shape[3] >= 8):
        COUNT['wino'] += 1
        y = wino_y(x, w)
        return y + bias if bias is not None else y
    COUNT['plain'] += 1
    return _plain(x, w, stride, padding, dilation, bias, explicit_pad)


def rel(a, b):
    return float(((a - b).abs() / b.abs()).mean())


def main(which):
    torch.set_num_threads(int(os.environ.get('ORACLE_THREADS', 6)))
    store = variables.VariableStore().init_synthetic(1234)
    W = {k: torch.from_numpy(v) for k, v in store.host.items()}
    # single-layer error against float64
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 12, 16, 20, 32, generator=g)
    w = torch.randn(3, 3, 3, 32, 8, generator=g) * 0.05
    ref = _plain(x.double(), w.double())
    e_plain = float((_plain(x, w).double() - ref).abs().max() / ref.abs().max())
    e_wino = float((wino_y(x, w).double() - ref).abs().max() / ref.abs().max())
    print('single layer 32->8, max-abs / max: direct fp32 %.2e, winograd-y fp32 %.2e' % (e_plain, e_wino), flush=True)
    if which == 'cfg1':
        cases = [(2, 128, 160, 32), (3, 128, 160, 32)]
    else:
        cases = [(2, 512, 640, 192)]
    for n, H, Wd, D in cases:
        imgs, cams = synthetic.make_inputs(n, H, Wd, D, seed=0)
        imgs, cams = torch.from_numpy(imgs), torch.from_numpy(cams)
        run = OM.run_twoview if n == 2 else OM.run_multiview
        with torch.no_grad():
            if which == 'cfg2':
                fx = np.load(os.path.join(ROOT, 'tests', 'golden', 'fullsize_cfg2.npz'))
                base = torch.from_numpy(fx['depth'])[None, ..., None]
            else:
                base = run(imgs, cams, W, D)
            T.conv = conv
            t0 = time.time()
            try:
                got = run(imgs, cams, W, D)
            finally:
                T.conv = _plain
        print('%d views %dx%d D=%d: rel-L1 winograd-y vs direct fp32 oracle %.3e  (%d winograd convs, %d plain, %.0f s)'
              % (n, Wd, H, D, rel(got, base), COUNT['wino'], COUNT['plain'], time.time() - t0), flush=True)
        if which == 'cfg2':
            d64 = torch.from_numpy(np.load(os.path.join(ROOT, 'tests', 'golden', 'fullsize_cfg2f64.npz'))['depth64'])[None, ..., None]
            print('   vs float64 networks: winograd-y %.3e, direct fp32 oracle %.3e' % (rel(got, d64), rel(base, d64)), flush=True)


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'cfg1')
