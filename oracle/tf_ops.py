"""TF-1.5 op semantics restated on float32 PyTorch-CPU tensors (channel-last).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Each function names the TF op
and the reference call site it stands in for.  Semantics: SURVEY.md Appendix B.
"""
import math

import torch
import torch.nn.functional as F


def same_pad(in_size, k, s, d=1):
    """TF ``padding='SAME'`` split (Appendix B.1): end-heavy when odd."""
    out = -(-in_size // s)
    k_eff = (k - 1) * d + 1
    total = max((out - 1) * s + k_eff - in_size, 0)
    before = total // 2
    return before, total - before, out


def _to_cf(x):
    """(B, *sp, C) -> (B, C, *sp)."""
    nd = x.dim()
    return x.permute(0, nd - 1, *range(1, nd - 1)).contiguous()


def _to_cl(x):
    """(B, C, *sp) -> (B, *sp, C)."""
    nd = x.dim()
    return x.permute(0, *range(2, nd), 1).contiguous()


def conv(x, w, stride=1, padding='SAME', dilation=1, bias=None, explicit_pad=None):
    """tf.layers.conv2d / conv3d / tf.nn.conv3d / slim.conv2d.

    reference: cnn_wrapper/network.py:165-167,198-200,304,331,578-599.
    x: (B, [D,] H, W, Cin); w: TF layout [k.., Cin, Cout].
    ``explicit_pad`` = list of (before, after) per spatial dim, applied before a
    VALID convolution (the strided bottleneck conv2, network.py:589-595).
    """
    nsp = x.dim() - 2
    ks = w.shape[:nsp]
    strides = (stride,) * nsp if isinstance(stride, int) else tuple(stride)
    dil = (dilation,) * nsp if isinstance(dilation, int) else tuple(dilation)
    if explicit_pad is not None:
        pads = list(explicit_pad)
    elif padding == 'SAME':
        pads = [same_pad(x.shape[1 + i], ks[i], strides[i], dil[i])[:2] for i in range(nsp)]
    else:
        pads = [(0, 0)] * nsp
    xc = _to_cf(x)
    flat = []
    for p in reversed(pads):
        flat += [p[0], p[1]]
    if any(flat):
        xc = F.pad(xc, flat)
    # TF [k.., Cin, Cout] -> torch [Cout, Cin, k..]
    wt = w.permute(nsp + 1, nsp, *range(nsp)).contiguous()
    fn = F.conv2d if nsp == 2 else F.conv3d
    y = fn(xc, wt, bias=bias, stride=strides, dilation=dil)
    return _to_cl(y)


def conv3d_transpose_same(x, w, stride=2):
    """tf.layers.conv3d_transpose(k, stride, padding='SAME') (Appendix B.2).

    reference: cnn_wrapper/network.py:534-536.  w: TF layout
    [kd, kh, kw, Cout, Cin].  Output size = stride * in: the full transposed
    convolution (out[s*i + k] += in[i] * W[k]) cropped at the END.  For general
    k the TF crop starts at pad_before of the *forward* SAME conv, which is 0
    for k=3, s=2 on even sizes (the only configuration the path uses).
    """
    assert x.dim() == 5
    k = w.shape[0]
    xc = _to_cf(x)
    # torch conv_transpose weight layout [Cin, Cout, k..]
    wt = w.permute(4, 3, 0, 1, 2).contiguous()
    y = F.conv_transpose3d(xc, wt, stride=stride, padding=0)
    out = [x.shape[1 + i] * stride for i in range(3)]
    starts = []
    for i in range(3):
        # forward SAME conv on the (stride*in)-sized output: pad_before
        pb, _, _ = same_pad(out[i], k, stride)
        starts.append(pb)
    y = y[:, :, starts[0]:starts[0] + out[0], starts[1]:starts[1] + out[1], starts[2]:starts[2] + out[2]]
    return _to_cl(y)


def batch_norm_train(x, beta=None, eps=1e-3):
    """Training-mode batch norm with batch statistics, no gamma.

    tf.layers.batch_normalization(center=False, scale=False, training=True)
    (network.py:206-212,541-547) and slim.batch_norm defaults (center=True ->
    beta, scale=False; network.py:570-571).  Biased variance, eps=1e-3
    (Appendix B.3/B.4).  Statistics over every axis but the last.
    """
    axes = tuple(range(x.dim() - 1))
    mean = x.mean(dim=axes, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=axes, keepdim=True)
    y = (x - mean) * torch.rsqrt(var + eps)
    if beta is not None:
        y = y + beta
    return y


def avg_pool2d_same(x, k, s):
    """tf.layers.average_pooling2d(padding='SAME') (network.py:667-671).

    Window mean over VALID elements only (Appendix B.5).
    """
    pads = [same_pad(x.shape[1 + i], k, s)[:2] for i in range(2)]
    xc = _to_cf(x)
    flat = [pads[1][0], pads[1][1], pads[0][0], pads[0][1]]
    ones = torch.ones_like(xc[:, :1])
    xs = F.avg_pool2d(F.pad(xc, flat), k, s, divisor_override=1)
    cnt = F.avg_pool2d(F.pad(ones, flat), k, s, divisor_override=1)
    return _to_cl(xs / cnt)


def resize_bilinear_align_corners(x, size):
    """tf.image.resize_images(BILINEAR, align_corners=True) (Appendix B.6).

    reference: network.py:655, model.py:72-74.  x: (B, H, W, C).
    """
    B, H, W, C = x.shape
    oh, ow = int(size[0]), int(size[1])

    def axis(n_in, n_out):
        scale = (n_in - 1) / (n_out - 1) if n_out > 1 else 0.0
        src = torch.arange(n_out, dtype=torch.float32) * torch.tensor(scale, dtype=torch.float32)
        lo = torch.floor(src).to(torch.int64)
        hi = torch.minimum(torch.ceil(src).to(torch.int64), torch.tensor(n_in - 1))
        return lo, hi, (src - lo.to(torch.float32))

    ylo, yhi, yl = axis(H, oh)
    xlo, xhi, xl = axis(W, ow)
    top = x[:, ylo]
    bot = x[:, yhi]
    xl_ = xl.view(1, 1, ow, 1)
    yl_ = yl.view(1, oh, 1, 1)
    t = top[:, :, xlo] + (top[:, :, xhi] - top[:, :, xlo]) * xl_
    b = bot[:, :, xlo] + (bot[:, :, xhi] - bot[:, :, xlo]) * xl_
    return t + (b - t) * yl_


def softmax(x, axis):
    """tf.nn.softmax (max-subtracted, fp32; Appendix B.8)."""
    return torch.softmax(x, dim=axis)


def tf_round(x):
    """tf.round: round-half-to-even (Appendix B.7)."""
    return torch.round(x)


def linspace(start, stop, num):
    """tf.linspace: start + i * ((stop-start)/(num-1)) in fp32 (Appendix B.7)."""
    start = torch.as_tensor(start, dtype=torch.float32)
    stop = torch.as_tensor(stop, dtype=torch.float32)
    if num == 1:
        return start.reshape(1)
    step = (stop - start) / torch.tensor(float(num - 1), dtype=torch.float32)
    return start + step * torch.arange(num, dtype=torch.float32)
