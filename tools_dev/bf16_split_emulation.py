"""CPU emulation of bf16-operand matrix-core convolutions with fp32 accumulation, patched into the oracle (BASELINE
configs[1] names "bf16 conv3d MFMA"): every convolution's input and kernel are split into bf16 pieces
x = x0 + x1 (+ x2), w = w0 + w1 (+ w2) (each piece = the bf16 rounding of the remainder), the kept products are convolved
in fp32 (a product of two bf16 values is exact in fp32) and summed.

    python tools_dev/bf16_split_emulation.py cfg1|cfg2 [forms...]        forms: 1 3 6  (number of products kept)

    1 product : x0*w0                                   (plain bf16)
    3 products: x0*w0 + x0*w1 + x1*w0                   (error ~2^-16 per product)
    6 products: all xi*wj with i + j <= 2               (error ~2^-24 per product: fp32-class; 6/16 of the fp32 MFMA time)
   16         : fp16 pieces, x = h0 + h1 / 2048 with h0 = f16(x), h1 = f16((x - h0) * 2048) (the residual scaled into the
                normal range), w likewise; THREE products: h0*g0 + (h0*g1 + h1*g0) / 2048 (error ~2^-22 per product; the
                cross terms accumulate apart and are scaled once)
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import atvsnet_amd                                     # noqa: E402,F401
from atvsnet_amd import synthetic, variables           # noqa: E402
from oracle import model as OM                         # noqa: E402
from oracle import tf_ops as T                         # noqa: E402

_plain = T.conv
_plain_t = T.conv3d_transpose_same
FORM = {'n': 6}


def pieces(t, k):
    out, r = [], t
    for _ in range(k):
        p = r.to(torch.bfloat16).to(torch.float32)
        out.append(p)
        r = r - p
    return out


def pairs(n):
    return {1: [(0, 0)], 3: [(0, 0), (0, 1), (1, 0)], 6: [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]}[n]


def f16_pieces(t):
    h0 = t.to(torch.float16).to(torch.float32)
    h1 = ((t - h0) * 2048.0).to(torch.float16).to(torch.float32)
    return h0, h1


def conv(x, w, stride=1, padding='SAME', dilation=1, bias=None, explicit_pad=None):
    n = FORM['n']
    if n == 16:
        (h0, h1), (g0, g1) = f16_pieces(x), f16_pieces(w)
        cross = _plain(h0, g1, stride, padding, dilation, None, explicit_pad) + _plain(h1, g0, stride, padding, dilation, None, explicit_pad)
        y = _plain(h0, g0, stride, padding, dilation, None, explicit_pad) + cross * (1.0 / 2048.0)
        return y + bias if bias is not None else y
    k = {1: 1, 3: 2, 6: 3}[n]
    xs, ws = pieces(x, k), pieces(w, k)
    y = None
    for i, j in reversed(pairs(n)):                    # small terms first
        t = _plain(xs[i], ws[j], stride, padding, dilation, None, explicit_pad)
        y = t if y is None else y + t
    return y + bias if bias is not None else y


def conv_t(x, w, stride=2):
    n = FORM['n']
    if n == 16:
        (h0, h1), (g0, g1) = f16_pieces(x), f16_pieces(w)
        return _plain_t(h0, g0, stride) + (_plain_t(h0, g1, stride) + _plain_t(h1, g0, stride)) * (1.0 / 2048.0)
    k = {1: 1, 3: 2, 6: 3}[n]
    xs, ws = pieces(x, k), pieces(w, k)
    y = None
    for i, j in reversed(pairs(n)):
        t = _plain_t(xs[i], ws[j], stride)
        y = t if y is None else y + t
    return y


def rel(a, b):
    return float(((a - b).abs() / b.abs()).mean())


def main(which, forms):
    torch.set_num_threads(int(os.environ.get('ORACLE_THREADS', 6)))
    store = variables.VariableStore().init_synthetic(1234)
    W = {k: torch.from_numpy(v) for k, v in store.host.items()}
    cases = [(2, 128, 160, 32), (3, 128, 160, 32)] if which == 'cfg1' else [(2, 512, 640, 192)]
    for n, H, Wd, D in cases:
        imgs, cams = synthetic.make_inputs(n, H, Wd, D, seed=0)
        imgs, cams = torch.from_numpy(imgs), torch.from_numpy(cams)
        run = OM.run_twoview if n == 2 else OM.run_multiview
        d64 = None
        with torch.no_grad():
            if which == 'cfg2':
                base = torch.from_numpy(np.load(os.path.join(ROOT, 'tests', 'golden', 'fullsize_cfg2.npz'))['depth'])[None, ..., None]
                d64 = torch.from_numpy(np.load(os.path.join(ROOT, 'tests', 'golden', 'fullsize_cfg2f64.npz'))['depth64'])[None, ..., None]
            else:
                base = run(imgs, cams, W, D)
            for f in forms:
                FORM['n'] = f
                T.conv, T.conv3d_transpose_same = conv, conv_t
                t0 = time.time()
                try:
                    got = run(imgs, cams, W, D)
                finally:
                    T.conv, T.conv3d_transpose_same = _plain, _plain_t
                msg = '%d views %dx%d D=%d, %d product(s): rel-L1 vs fp32 oracle %.3e' % (n, Wd, H, D, f, rel(got, base))
                if d64 is not None:
                    msg += '; vs float64 networks %.3e (fp32 oracle: %.3e)' % (rel(got, d64), rel(base, d64))
                print(msg + '  (%.0f s)' % (time.time() - t0), flush=True)


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'cfg1', [int(v) for v in sys.argv[2:]] or [1, 3, 6])
