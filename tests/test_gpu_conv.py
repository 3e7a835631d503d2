"""-m gpu: MFMA convolution / batch-norm / pooling kernels (through the C-ABI) vs the CPU oracle.

Floating point: fp32 MFMA products are exact fp32 FMAs, only the summation order
differs from the oracle's (oneDNN) -> tolerance 2e-5 * max|ref| (stated per test).
"""
import numpy as np
import os

import pytest
import torch

from oracle import tf_ops as T
from oracle import nets

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return scale * torch.randn(*shape, generator=g)


def _close(got, want, tol=2e-5):
    scale = float(want.abs().max()) + 1e-30
    err = float((got - want).abs().max())
    assert err <= tol * scale, 'max err %.3e vs scale %.3e' % (err, scale)


CONV3D = [
    # D, H, W, Cin, Cout, stride
    (8, 8, 16, 64, 8, 1),
    (8, 8, 16, 64, 16, 2),
    (6, 10, 12, 8, 8, 1),
    (4, 6, 10, 16, 32, 2),
    (4, 4, 5, 64, 64, 1),
    (5, 7, 9, 48, 8, 1),
    (5, 7, 9, 19, 8, 1),
    (5, 7, 9, 1, 8, 1),
    (6, 6, 6, 8, 1, 1),
    (3, 5, 33, 32, 16, 2),
]


@pytest.fixture(params=['gather', 'tiled', None])
def impl(request):
    from atvsnet_amd import ops
    with ops.configure(force_impl=request.param, clear_pack_cache=True):
        yield request.param


@pytest.mark.parametrize('D,H,W,Cin,Cout,stride', CONV3D + [(9, 17, 35, 64, 8, 1), (5, 20, 40, 48, 8, 1),
                                                              (4, 16, 32, 32, 32, 1), (7, 33, 18, 8, 16, 1)])
def test_conv3d_same(cuda, impl, D, H, W, Cin, Cout, stride):
    from atvsnet_amd import ops
    x = _rand((1, D, H, W, Cin), 1)
    w = _rand((3, 3, 3, Cin, Cout), 2, 0.2)
    want = T.conv(x, w, stride, 'SAME')[0]
    got, st = ops.conv(x[0].to(cuda), ('t', D, H, W, Cin, Cout, stride), w.numpy(), stride=stride, want_stats=True)
    assert tuple(got.shape) == tuple(want.shape)
    _close(got.cpu(), want)
    # statistics epilogue: per-channel sum / sum of squares of what was written
    s = st.partial.reshape(-1, 2, st.cpad).sum(0).cpu()
    _close(s[0, :Cout].float(), want.reshape(-1, Cout).double().sum(0).float(), 1e-5)
    _close(s[1, :Cout].float(), (want.reshape(-1, Cout).double() ** 2).sum(0).float(), 1e-5)


@pytest.mark.parametrize('tile_m', [1, 2, 4, 8])
def test_conv3d_tile_variants(cuda, tile_m):
    from atvsnet_amd import ops
    x = _rand((1, 6, 9, 21, 16), 3)
    w = _rand((3, 3, 3, 16, 16), 4, 0.2)
    want = T.conv(x, w, 1, 'SAME')[0]
    taps = ops.conv_taps((3, 3, 3), 1, (1, 1, 1))
    pk = ops.pack_conv_weights(('tv', tile_m), w.numpy(), taps, False, cuda)
    y = torch.empty(1, 6, 9, 21, 16, device=cuda)
    ops.conv_launch(x.to(cuda), pk, y, (6, 9, 21), 1, 1, (0, 0, 0), 0, tile_m=tile_m)      # (G,D,H,W,C) operands
    _close(y[0].cpu(), want)


CONV2D = [
    # H, W, Cin, Cout, k, stride, rate
    (32, 40, 3, 32, 3, 2, 1),
    (16, 20, 32, 32, 3, 1, 1),
    (16, 20, 64, 64, 3, 1, 2),
    (16, 20, 128, 128, 3, 1, 4),
    (16, 20, 32, 64, 1, 2, 1),
    (9, 13, 320, 128, 3, 1, 1),
    (9, 13, 128, 32, 1, 1, 1),
    (33, 41, 3, 16, 1, 4, 1),
    (2, 3, 128, 32, 3, 1, 1),
    (1, 1, 128, 32, 3, 1, 1),
]


@pytest.mark.parametrize('H,W,Cin,Cout,k,stride,rate', CONV2D)
def test_conv2d_same_bias_relu(cuda, H, W, Cin, Cout, k, stride, rate):
    from atvsnet_amd import ops
    x = _rand((1, H, W, Cin), 5)
    w = _rand((k, k, Cin, Cout), 6, 0.2)
    b = _rand((Cout,), 7)
    want = torch.clamp(T.conv(x, w, stride, 'SAME', rate, bias=b), min=0)[0]
    got = ops.conv(x[0].to(cuda), ('c2', H, W, Cin, Cout, k, stride, rate), w.numpy(), stride=stride, dilation=rate,
                   bias=b.to(cuda), relu=True)
    assert tuple(got.shape) == tuple(want.shape)
    _close(got.cpu(), want)


@pytest.mark.parametrize('stride', [2, 4])
def test_conv2d_explicit_pad_valid_with_residual(cuda, stride):
    """bottleneck conv2 with stride (network.py:589-595) and the fused shortcut add of conv3."""
    from atvsnet_amd import ops
    x = _rand((1, 32, 40, 16), 8)
    w = _rand((3, 3, 16, 16), 9, 0.2)
    b = _rand((16,), 10)
    want = T.conv(x, w, stride, 'VALID', 1, bias=b, explicit_pad=[(1, 1), (1, 1)])[0]
    res = _rand(tuple(want.shape), 11)
    got = ops.conv(x[0].to(cuda), ('c2e', stride), w.numpy(), stride=stride, explicit_pad=[(1, 1), (1, 1)],
                   bias=b.to(cuda), residual=res.to(cuda))
    _close(got.cpu(), want + res)


@pytest.mark.parametrize('D,H,W,Cin,Cout', [(3, 4, 5, 64, 32), (4, 4, 4, 32, 16), (6, 5, 7, 16, 8), (5, 18, 20, 16, 8)])
def test_conv3d_transpose(cuda, impl, D, H, W, Cin, Cout):
    from atvsnet_amd import ops
    x = _rand((1, D, H, W, Cin), 12)
    w = _rand((3, 3, 3, Cout, Cin), 13, 0.2)
    want = T.conv3d_transpose_same(x, w, 2)[0]
    got, st = ops.conv3d_transpose_s2(x[0].to(cuda), ('dc', D, H, W, Cin, Cout), w.numpy(), want_stats=True)
    assert tuple(got.shape) == (2 * D, 2 * H, 2 * W, Cout)
    _close(got.cpu(), want)
    s = st.partial.reshape(-1, 2, st.cpad).sum(0).cpu()
    folded = s[:, :st.fold * Cout].reshape(2, st.fold, Cout).sum(1)     # (class, channel) columns fold onto the channel
    _close(folded[0].float(), want.reshape(-1, Cout).double().sum(0).float(), 1e-5)
    _close(folded[1].float(), (want.reshape(-1, Cout).double() ** 2).sum(0).float(), 1e-5)
    bn = ops.batch_norm(got, st, relu=True).cpu()
    assert float((bn - torch.clamp(T.batch_norm_train(want[None]), min=0)[0]).abs().max()) < 2e-5


def test_conv_writes_channel_slice(cuda):
    from atvsnet_amd import ops
    x = _rand((1, 4, 6, 7, 8), 14)
    w = _rand((3, 3, 3, 8, 8), 15, 0.2)
    want = T.conv(x, w, 1, 'SAME')[0]
    buf = torch.full((4, 6, 7, 32), -7.0, device=cuda)
    ops.conv(x[0].to(cuda), 'slice', w.numpy(), out=buf, y_coff=16)
    out = buf.cpu()
    _close(out[..., 16:24], want)
    assert torch.all(out[..., :16] == -7.0) and torch.all(out[..., 24:] == -7.0)


@pytest.mark.parametrize('shape,beta', [((6, 7, 9, 8), False), ((33, 41, 3), True), ((10, 12, 128), True), ((2, 3, 32), False)])
def test_batch_norm_train(cuda, shape, beta):
    """training-mode BN, eps 1e-3, biased variance; tolerance 1e-5 absolute (outputs are O(1))."""
    from atvsnet_amd import ops
    x = _rand((1,) + shape, 16, 3.0) + 5.0
    b = _rand((shape[-1],), 17) if beta else None
    want = torch.clamp(T.batch_norm_train(x, beta=b), min=0)[0]
    got = ops.batch_norm(x[0].to(cuda), beta=(b.to(cuda) if beta else None), relu=True)
    assert float((got.cpu() - want).abs().max()) < 1e-5
    # known answer: mean 0, variance s^2/(s^2+eps) before relu
    raw = ops.batch_norm(x[0].to(cuda), relu=False).cpu().reshape(-1, shape[-1]).double()
    v = x.reshape(-1, shape[-1]).double().var(0, unbiased=False)
    assert float(raw.mean(0).abs().max()) < 1e-5
    assert torch.allclose(raw.var(0, unbiased=False), v / (v + 1e-3), rtol=1e-4)


@pytest.mark.parametrize('G,rows,C', [(1, 5 * 6 * 7, 8), (3, 4099, 16), (2, 1031, 24), (1, 4 * 1024 + 3, 128), (4, 777, 12),
                                      (2, 9000, 32)])
def test_elementwise_bn_passes_match_their_formula(cuda, G, rows, C):
    """bn_apply / bn_add (norm.hip: a workgroup covers four spans of 1024 floats of ONE sample) on sizes that end inside a
    span, channel counts that are and are not powers of two, several independent samples with their own parameters, a
    channel slice of a wider buffer, two and three terms -- against the same expression in torch (exact up to the
    fused-multiply-add contraction: 1e-6 relative)."""
    from atvsnet_amd import ops
    x = [_rand((G, rows, C), 200 + i, 2.0) for i in range(3)]
    par = [torch.stack([_rand((G, C), 210 + i), torch.rand(G, C, generator=torch.Generator().manual_seed(220 + i)) + 0.5,
                        _rand((G, C), 230 + i)], 1).contiguous() for i in range(3)]

    def bn(i, relu):
        y = (x[i] - par[i][:, 0][:, None]) * par[i][:, 1][:, None] + par[i][:, 2][:, None]
        return torch.clamp(y, min=0) if relu else y
    xd, pd = [t.to(cuda) for t in x], [t.to(cuda) for t in par]
    for relu in (False, True):
        _close(ops.bn_apply(xd[0], pd[0], relu=relu, out=torch.empty_like(xd[0])).cpu(), bn(0, relu), 1e-6)
    # a channel slice of a buffer twice as wide, in place; the other channels untouched
    wide = torch.full((G, rows, 2 * C + 4), -3.0, device=cuda)
    wide[..., 4:4 + C] = xd[1]
    ops.bn_apply(wide, pd[1], relu=True, C=C, c_off=4)
    _close(wide[..., 4:4 + C].cpu(), bn(1, True), 1e-6)
    assert float((wide[..., :4] + 3.0).abs().max()) == 0.0 and float((wide[..., 4 + C:] + 3.0).abs().max()) == 0.0
    # two and three terms, pending and dense mixed
    got2 = ops.bn_add([ops.PendingBN(xd[0], pd[0], True), ops.PendingBN(xd[1], pd[1], False)])
    _close(got2.cpu(), bn(0, True) + bn(1, False), 1e-6)
    got3 = ops.bn_add([ops.PendingBN(xd[0], pd[0], False), xd[2], ops.PendingBN(xd[1], pd[1], True)])
    _close(got3.cpu(), (bn(0, False) + x[2]) + bn(1, True), 1e-6)


def test_conv_bn_matches_oracle_layer(cuda):
    from atvsnet_amd import ops
    x = _rand((1, 6, 8, 10, 16), 18)
    W = {'l/conv3d/kernel': _rand((3, 3, 3, 16, 32), 19, 0.1)}
    want = nets.conv_bn(x, W, 'l', 32, 2)[0]
    y, st = ops.conv(x[0].to(cuda), 'l', W['l/conv3d/kernel'].numpy(), stride=2, want_stats=True)
    got = ops.batch_norm(y, st, relu=True, inplace=True)
    assert float((got.cpu() - want).abs().max()) < 2e-5


@pytest.mark.parametrize('H,W,pool', [(32, 40, 64), (32, 40, 32), (32, 40, 16), (32, 40, 8), (30, 45, 8), (128, 160, 64)])
def test_avg_pool_same_and_resize(cuda, H, W, pool):
    from atvsnet_amd import ops
    x = _rand((1, H, W, 128), 20)
    want = T.avg_pool2d_same(x, pool, pool)[0]
    got = ops.avg_pool_same(x[0].to(cuda), pool, pool)
    assert tuple(got.shape) == tuple(want.shape)
    _close(got.cpu(), want, 1e-5)
    up = T.resize_bilinear_align_corners(want[None], (H, W))[0]
    got_up = ops.resize_bilinear(got, (H, W))
    _close(got_up.cpu(), up, 1e-5)


def test_add_n_and_concat(cuda):
    from atvsnet_amd import ops
    a, b, c = _rand((5, 6, 7, 8), 21), _rand((5, 6, 7, 8), 22), _rand((5, 6, 7, 8), 23)
    assert torch.equal(ops.add_n([a.to(cuda), b.to(cuda)]).cpu(), a + b)
    assert torch.equal(ops.add_n([a.to(cuda), b.to(cuda), c.to(cuda)]).cpu(), (a + b) + c)
    d = _rand((5, 6, 7, 3), 24)
    assert torch.equal(ops.concat_channels([a.to(cuda), d.to(cuda), b.to(cuda)]).cpu(), torch.cat([a, d, b], -1))


@pytest.mark.parametrize('nv', [1, 2, 4])
def test_aanet(cuda, nv):
    """AANet through the fused 8->16 conv + combine kernel vs the oracle; tolerance 1e-5 absolute."""
    from atvsnet_amd import ops
    D, h, w = 6, 8, 10
    X = _rand((1, D, h, w, 8, nv), 25)
    W = {'a/attention_activation/weight_unique': _rand((3, 3, 3, 8, 8), 26, 0.3),
         'a/attention_activation/weight_shared': _rand((3, 3, 3, 8, 8), 27, 0.3)}
    want = nets.attention_aggregation(X, W, 'a')[0]
    w16 = np.concatenate([W['a/attention_activation/weight_shared'].numpy(),
                          W['a/attention_activation/weight_unique'].numpy()], axis=-1)
    xs = [X[0, ..., n].contiguous().to(cuda) for n in range(nv)]
    srs = [ops.conv(x, ('aan', nv), w16, relu=True) for x in xs]
    got = ops.aanet_combine(srs, xs)
    assert float((got.cpu() - want).abs().max()) < 1e-5
    if nv == 1:
        assert float((got.cpu() - X[0, ..., 0]).abs().max()) < 1e-6
    # the view-sharded form (three reductions) gives the same answer
    ssum = ops.aanet_partial(srs, xs, 0)
    umax = ops.aanet_partial(srs, xs, 1, ssum=ssum)
    acc = ops.aanet_partial(srs, xs, 2, ssum=ssum, umax=umax)
    got2 = ops.divide(acc[1], acc[0])
    assert float((got2.cpu() - want).abs().max()) < 1e-5


@pytest.mark.parametrize('stride', [1, 2])
def test_split_volume_conv_matches_dense(cuda, impl, stride):
    """conv over concat([var, tile(const)]) == conv3d(var) + per-plane 2-D conv of the constant channels,
    including replicated channels (geo_group, quirk C7) and first/last depth planes."""
    from atvsnet_amd import ops
    D, h, w = 6, 16, 20
    var = _rand((D, h, w, 2), 30)
    const = _rand((h, w, 3), 31)
    cmap = [('v', 0)] + [('v', 1)] * 4 + [('c', 2), ('c', 0), ('c', 1)]
    dense = torch.cat([var[..., :1]] + [var[..., 1:2]] * 4 +
                      [const[None, ..., 2:3].expand(D, -1, -1, -1), const[None, ..., :2].expand(D, -1, -1, -1)], -1)
    wgt = _rand((3, 3, 3, 8, 8), 32, 0.3)
    want = T.conv(dense[None], wgt, stride, 'SAME')[0]
    sv = ops.SplitVolume(var.to(cuda), const.to(cuda), cmap)
    assert sv.shape == (1, D, h, w, 8)
    assert torch.equal(sv.materialize().cpu()[0], dense)
    got, st = ops.conv_split(sv, ('split', stride), wgt.numpy(), stride=stride, want_stats=True)
    _close(got.cpu(), want)
    s = st.partial.reshape(-1, 2, st.cpad).sum(0).cpu()
    _close(s[0, :8].float(), want.reshape(-1, 8).double().sum(0).float(), 1e-5)


@pytest.mark.parametrize('xp1w', ['xb', 'xw', False])
@pytest.mark.parametrize('D,H,W,Cin', [(9, 19, 70, 32), (4, 8, 32, 16), (6, 21, 45, 8), (13, 9, 33, 24), (2, 3, 24, 16)])
def test_conv_xpair_kernels(cuda, xp1w, D, H, W, Cin):
    """The x-pair kernels for the 8-output-channel layers (one workgroup per CU: split-operand form conv_xb.hip (default),
    Winograd F(2,3)-along-y fp32 form conv_xw.hip; the tiled kernel's x-pair form): ragged sizes (odd H: a row pair that
    straddles the end), 16- and 8-channel chunks, with depth-plane bias + bias + residual + ReLU, written into a
    channel slice of a wider (concat) buffer, statistics of what was written."""
    from atvsnet_amd import ops
    with ops.configure(xp1w=bool(xp1w), xb=xp1w == 'xb', clear_pack_cache=True):
        x = _rand((1, D, H, W, Cin), 50)
        w = _rand((3, 3, 3, Cin, 8), 51, 0.2)
        b = _rand((8,), 52)
        pb = _rand((H, W, 24), 53)
        want = T.conv(x, w, 1, 'SAME', bias=b)[0]
        var = torch.ones(D, dtype=torch.long)
        var[0], var[-1] = 0, 2
        for z in range(D):
            want[z] += pb[..., int(var[z]) * 8:int(var[z]) * 8 + 8]
        # 1. plane bias + bias + ReLU into channels 4..11 of a 16-channel buffer
        out = torch.full((D, H, W, 16), -7.0, device=cuda)
        got, st = ops.conv(x[0].to(cuda), ('xp', D, H, W, Cin), w.numpy(), bias=b.to(cuda), relu=True, want_stats=True,
                           out=out, y_coff=4, plane_bias=pb.to(cuda))
        w1 = torch.clamp(want, min=0)
        _close(out.cpu()[..., 4:12], w1)
        assert float((out.cpu()[..., :4] + 7.0).abs().max()) == 0.0 and float((out.cpu()[..., 12:] + 7.0).abs().max()) == 0.0
        s = st.partial.reshape(-1, 2, st.cpad).sum(0).cpu()
        _close(s[0, :8].float(), w1.reshape(-1, 8).double().sum(0).float(), 1e-5)
        _close(s[1, :8].float(), (w1.reshape(-1, 8).double() ** 2).sum(0).float(), 1e-5)
        # 2. residual, no ReLU, dense output
        res = _rand((D, H, W, 8), 54)
        got2 = ops.conv(x[0].to(cuda), ('xp', D, H, W, Cin), w.numpy(), bias=b.to(cuda), residual=res.to(cuda),
                        plane_bias=pb.to(cuda))
        _close(got2.cpu(), want + res)


@pytest.fixture(params=['xb', 'xw'])
def xkernel(request):
    """Run a test on both one-workgroup-per-CU x-pair kernels: conv_xb.hip (split fp16 operands, default) and conv_xw.hip
    (fp32 Winograd)."""
    from atvsnet_amd import ops
    with ops.configure(xb=request.param == 'xb', clear_pack_cache=True):
        yield request.param


@pytest.mark.parametrize('D,H,W,Cin', [(8, 16, 64, 32), (9, 19, 70, 16), (6, 21, 45, 8), (5, 8, 33, 24), (4, 7, 32, 8)])
def test_conv_siblings_one_launch(cuda, xkernel, D, H, W, Cin):
    """conv(8 channels, stride 1) and conv(16 channels, stride 2, SAME) of one input from ONE launch equal the two
    separate convolutions (even and odd sizes: the stride-2 SAME padding moves), with plane biases and statistics."""
    from atvsnet_amd import ops
    ops.clear_pack_cache()
    x = _rand((1, D, H, W, Cin), 60)
    w = _rand((3, 3, 3, Cin, 8), 61, 0.2)
    w2 = _rand((3, 3, 3, Cin, 16), 62, 0.2)
    D2, H2, W2 = (D + 1) // 2, (H + 1) // 2, (W + 1) // 2
    pb, pb2 = _rand((H, W, 24), 63), _rand((H2, W2, 48), 64)
    want = T.conv(x, w, 1, 'SAME')[0]
    want2 = T.conv(x, w2, 2, 'SAME')[0]
    assert tuple(want2.shape) == (D2, H2, W2, 16)
    for z in range(D):
        v = 0 if z == 0 else (2 if z == D - 1 else 1)
        want[z] += pb[..., v * 8:v * 8 + 8]
    pz = D % 2
    for z in range(D2):
        first = 2 * z - pz
        v = 0 if first < 0 else (2 if first + 2 >= D else 1)
        want2[z] += pb2[..., v * 16:v * 16 + 16]
    (y, st), (y2, st2) = ops.conv_siblings(x[0].to(cuda), ('sib', D, H, W, Cin), w.numpy(), ('sib2', D, H, W, Cin), w2.numpy(),
                                           plane_bias=pb.to(cuda), plane_bias2=pb2.to(cuda))
    _close(y.cpu(), want)
    _close(y2.cpu(), want2)
    for got_st, ref, C in ((st, want, 8), (st2, want2, 16)):
        s = got_st.partial.reshape(-1, 2, got_st.cpad).sum(0).cpu()
        _close(s[0, :C].float(), ref.reshape(-1, C).double().sum(0).float(), 1e-5)
        _close(s[1, :C].float(), (ref.reshape(-1, C).double() ** 2).sum(0).float(), 1e-5)
        assert got_st.count == ref.shape[0] * ref.shape[1] * ref.shape[2]
    # without plane biases
    (y, _), (y2, _) = ops.conv_siblings(x[0].to(cuda), ('sib', D, H, W, Cin), w.numpy(), ('sib2', D, H, W, Cin), w2.numpy())
    _close(y2.cpu(), T.conv(x, w2, 2, 'SAME')[0])
    _close(y.cpu(), T.conv(x, w, 1, 'SAME')[0])


def test_split_operand_kernels_fail_loudly_outside_the_fp16_range(cuda):
    """Two fp16 pieces carry 22 significand bits but only fp16's exponent range: a weight beyond 65504 is
    refused when it is packed, and an activation beyond it gives a NON-FINITE output where it is read -- never a finite wrong
    value -- while values up to the limit stay exact to the usual tolerance."""
    from atvsnet_amd import ops
    ops.clear_pack_cache()
    D, H, W, Cin = 4, 8, 32, 16
    x = _rand((1, D, H, W, Cin), 70)
    w = _rand((3, 3, 3, Cin, 8), 71, 0.2)
    # large but representable: activations up to ~5e4 against 0.2-scale weights
    big = x * (5.0e4 / float(x.abs().max()))
    got = ops.conv(big[0].to(cuda), ('rng', 1), w.numpy())
    _close(got.cpu(), T.conv(big, w, 1, 'SAME')[0])
    # one activation beyond the range: every output that reads it is non-finite, and every FINITE output is the value from
    # before (the x-pair form multiplies the neighbouring pair's window by structural zeros: inf * 0 widens the NaN region
    # by a voxel along x, it never narrows it)
    bad = big.clone()
    bad[0, 2, 4, 16, 3] = 1.0e5
    got2 = ops.conv(bad[0].to(cuda), ('rng', 1), w.numpy()).cpu()
    touched = torch.zeros(D, H, W, dtype=torch.bool)
    touched[1:4, 3:6, 15:18] = True
    assert not torch.isfinite(got2[touched]).any()
    fin = torch.isfinite(got2)
    assert torch.equal(got2[fin], got.cpu()[fin])
    assert int((~fin.all(-1)).sum()) <= 3 * 3 * 6
    # a weight beyond the range is refused by the packer
    wbad = w.clone()
    wbad[1, 1, 1, 0, 0] = 7.0e4
    with pytest.raises(RuntimeError):
        ops.conv(x[0].to(cuda), ('rng', 2), wbad.numpy())
    ops.clear_pack_cache()


@pytest.mark.parametrize('B,D,h,w', [(1, 6, 16, 40), (2, 9, 19, 70)])
def test_planar_cost_volume_is_bitwise_the_channel_last_one(cuda, B, D, h, w):
    """The warped half of the cost volume written chunk-planar (F/8, D, h, w, 8) (atvs_warp_planes planar) and read by the
    Winograd x-pair launch (atvs_conv_xw_f32 x_planar): same values as the channel-last pair, bit for bit -- the warp,
    both convolutions, their statistics -- and the SplitVolume still materialises the reference's dense concat."""
    from atvsnet_amd import ops
    from oracle import homography_warping as G
    from oracle import model as OM
    ops.clear_pack_cache()
    F = 32
    cams = torch.from_numpy(np.load(os.path.join(os.path.dirname(__file__), 'golden', 'example2_0_cam.npy')))[None, None]
    cam2 = torch.from_numpy(np.load(os.path.join(os.path.dirname(__file__), 'golden', 'example2_1_cam.npy')))[None, None]
    cams = torch.cat([cams, cam2], 1).float()
    ds, di = OM.depth_start_interval(cams)
    Hm = G.get_homographies(cams[:, 0], cams[:, 1], D, ds, di)[0].to(cuda)
    feats = _rand((B, h, w, F), 90).to(cuda)
    const = _rand((B, h, w, F), 91).to(cuda)
    cl = torch.stack([ops.warp_planes(feats[b], Hm) for b in range(B)])
    pl = torch.stack([ops.warp_planes(feats[b], Hm, planar=True) for b in range(B)])
    assert tuple(pl.shape) == (B, F // 8, ops.planar_stride(D, h, w))            # padded chunk planes
    assert torch.equal(ops.planar_view(pl, D, h, w).permute(0, 2, 3, 4, 1, 5).reshape(B, D, h, w, F), cl)
    cmap = [('c', i) for i in range(F)] + [('v', i) for i in range(F)]
    sv_cl, sv_pl = ops.SplitVolume(cl, const, cmap), ops.SplitVolume(pl, const, cmap, planar=(D, h, w))
    assert sv_pl.shape == sv_cl.shape and sv_pl.cv == F
    w8, w16 = (_rand((3, 3, 3, 2 * F, 8), 92) * 0.1).numpy(), (_rand((3, 3, 3, 2 * F, 16), 93) * 0.1).numpy()
    (y, st), (y2, st2) = ops.conv_split_siblings(sv_pl, 'pl8', w8, 'pl16', w16)
    (r, rt), (r2, rt2) = ops.conv_split_siblings(sv_cl, 'pl8', w8, 'pl16', w16)
    assert torch.equal(y, r) and torch.equal(y2, r2)
    assert torch.equal(st.partial, rt.partial) and torch.equal(st2.partial, rt2.partial)
    assert torch.equal(sv_pl.materialize(), sv_cl.materialize())
    # the fp32 Winograd kernel reads the planar form too
    with ops.configure(xb=False, clear_pack_cache=True):
        (z, _), (z2, _) = ops.conv_split_siblings(sv_pl, 'pl8', w8, 'pl16', w16)
        (q, _), (q2, _) = ops.conv_split_siblings(sv_cl, 'pl8', w8, 'pl16', w16)
        assert torch.equal(z, q) and torch.equal(z2, q2)


@pytest.mark.parametrize('B,D,h,w,F', [(1, 6, 16, 40, 32), (2, 9, 19, 70, 32), (1, 5, 9, 33, 16), (1, 4, 8, 32, 64)])
def test_cost_volume_in_pieces_is_bitwise_the_planar_one(cuda, B, D, h, w, F):
    """atvs_warp_planes(pieces): the warped half of the cost volume leaves the warp as the two fp16 pieces of every value --
    bit for bit the split the convolution's staging wavefronts perform (h0 = fp16(x), h1 = fp16((x - h0) * 2048)), laid out
    [chunk][piece][D][h][w][8] -- and conv_xb stages them by LDS-DMA (x_pieces): both convolutions and their statistics equal the
    planar-fp32 launch bit for bit (ragged sizes: rows that end inside a DMA instruction's 64 records, volume borders)."""
    from atvsnet_amd import ops
    from oracle import homography_warping as G
    from oracle import model as OM
    ops.clear_pack_cache()
    cams = torch.from_numpy(np.load(os.path.join(os.path.dirname(__file__), 'golden', 'example2_0_cam.npy')))[None, None]
    cam2 = torch.from_numpy(np.load(os.path.join(os.path.dirname(__file__), 'golden', 'example2_1_cam.npy')))[None, None]
    cams = torch.cat([cams, cam2], 1).float()
    ds, di = OM.depth_start_interval(cams)
    Hm = G.get_homographies(cams[:, 0], cams[:, 1], D, ds, di)[0].to(cuda)
    feats = (_rand((B, h, w, F), 95) * 3.0).to(cuda)
    const = _rand((B, h, w, F), 96).to(cuda)
    pl = torch.stack([ops.warp_planes(feats[b], Hm, planar=True) for b in range(B)])
    pc = torch.stack([ops.warp_planes(feats[b], Hm, planar=True, pieces=True) for b in range(B)])
    assert pc.shape == pl.shape and pc.dtype == torch.float32
    # the pieces, bit for bit (numpy's float32 -> float16 conversion rounds to nearest even as the hardware does)
    n = D * h * w * 8
    x = ops.planar_view(pl, D, h, w).cpu().numpy()                                   # (B, K, D, h, w, 8)
    got = pc[..., :n].contiguous().view(torch.float16).cpu().numpy().reshape(B, F // 8, 2, D, h, w, 8)
    h0 = x.astype(np.float16)
    h1 = ((x - h0.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    assert np.array_equal(got[:, :, 0].view(np.uint16), h0.view(np.uint16))
    assert np.array_equal(got[:, :, 1].view(np.uint16), h1.view(np.uint16))
    dec = ops.planar_pieces_decode(pc, D, h, w).cpu().numpy()
    assert np.abs(dec - x).max() <= np.abs(x).max() * 2.0 ** -21
    # the launches
    cmap = [('c', i) for i in range(F)] + [('v', i) for i in range(F)]
    sv_pl = ops.SplitVolume(pl, const, cmap, planar=(D, h, w))
    sv_pc = ops.SplitVolume(pc, const, cmap, planar=(D, h, w), pieces=True)
    w8, w16 = (_rand((3, 3, 3, 2 * F, 8), 97) * 0.1).numpy(), (_rand((3, 3, 3, 2 * F, 16), 98) * 0.1).numpy()
    (y, st), (y2, st2) = ops.conv_split_siblings(sv_pc, 'pc8', w8, 'pc16', w16)
    (r, rt), (r2, rt2) = ops.conv_split_siblings(sv_pl, 'pc8', w8, 'pc16', w16)
    assert torch.equal(y, r) and torch.equal(y2, r2)
    assert torch.equal(st.partial, rt.partial) and torch.equal(st2.partial, rt2.partial)
    # any other consumer gets the decoded channel-last values (fp32 kernels, tests): close to the planar ones, not bitwise
    assert float((sv_pc.var - sv_pl.var).abs().max()) <= float(sv_pl.var.abs().max()) * 2.0 ** -21
    with pytest.raises(ValueError):
        ops.warp_planes(feats[0], Hm, pieces=True)                                   # pieces need the planar layout
    ops.clear_pack_cache()


@pytest.mark.parametrize('cin', [16, 8])
@pytest.mark.parametrize('G,D,H,W', [(1, 8, 16, 32), (3, 9, 19, 37), (2, 5, 8, 12)])
def test_conv_c16b_split16_matches_oracle(cuda, G, D, H, W, cin):
    """Split-operand form of the 8 / 16 -> 16 channel convolution (conv_c16b.hip: two fp16 pieces per operand, three
    products, fp32 accumulation on v_mfma_f32_16x16x32_f16) against the oracle at the UNCHANGED fp32 bar (2e-5 of the
    maximum), and no further from a float64 evaluation than the fp32 MFMA kernel is (x 2): bias + ReLU into a channel
    slice, statistics, ragged sizes, per-sample groups equal to single launches bit for bit."""
    from atvsnet_amd import ops
    x = _rand((G, D, H, W, cin), 70)
    w = _rand((3, 3, 3, cin, 16), 71, 0.15)
    b = _rand((16,), 72)
    want = torch.clamp(T.conv(x, w, 1, 'SAME', bias=b), min=0)
    ref64 = torch.clamp(T.conv(x.double(), w.double(), 1, 'SAME', bias=b.double()), min=0)
    outs = {}
    for flag in (False, True):
        with ops.configure(split16=flag, clear_pack_cache=True):
            buf = torch.full((G, D, H, W, 24), -3.0, device=cuda)
            got, st = ops.conv(x.to(cuda), ('c16b', G, D, H, W, cin), w.numpy(), bias=b.to(cuda), relu=True, want_stats=True,
                               out=buf, y_coff=4, groups=G)
            y = buf.cpu()
            _close(y[..., 4:20], want)
            assert float((y[..., :4] + 3.0).abs().max()) == 0.0 and float((y[..., 20:] + 3.0).abs().max()) == 0.0
            s = st.partial.reshape(G, -1, 2, st.cpad).sum(1).cpu()
            for g in range(G):
                _close(s[g, 0, :16].float(), want[g].reshape(-1, 16).double().sum(0).float(), 1e-5)
                _close(s[g, 1, :16].float(), (want[g].reshape(-1, 16).double() ** 2).sum(0).float(), 1e-5)
            outs[flag] = y[..., 4:20].double()
            if flag:
                one = ops.conv(x[1 % G].to(cuda), ('c16b', G, D, H, W, cin), w.numpy(), bias=b.to(cuda), relu=True)
                assert torch.equal(one.cpu(), y[1 % G, ..., 4:20])
    e32 = float((outs[False] - ref64).abs().max())
    e16 = float((outs[True] - ref64).abs().max())
    print('max abs error vs float64: fp32 MFMA %.3e, split fp16 %.3e' % (e32, e16))
    assert e16 <= 2.0 * e32 + 1e-7 * float(ref64.abs().max())


@pytest.mark.parametrize('cin,cout', [(32, 32), (64, 64), (16, 32), (48, 64)])
@pytest.mark.parametrize('G,D,H,W', [(1, 8, 16, 32), (3, 9, 19, 21), (2, 5, 8, 12)])
def test_conv3d_b_split16_matches_oracle(cuda, G, D, H, W, cin, cout):
    """Split-bf16 form of the 16 k -> 32 / 64 channel 3x3x3 convolutions (conv3d_b.hip: several 16-channel chunks, two or
    four output tiles per wavefront, weights streamed) against the oracle at the fp32 bar, and no further from a float64
    evaluation than the fp32 kernel the layer used before (x 2): bias + ReLU into a channel slice, statistics, ragged
    sizes, per-sample groups equal to single launches bit for bit."""
    from atvsnet_amd import ops
    x = _rand((G, D, H, W, cin), 80)
    w = _rand((3, 3, 3, cin, cout), 81, (2.0 / (27 * cin)) ** 0.5)
    b = _rand((cout,), 82)
    want = torch.clamp(T.conv(x, w, 1, 'SAME', bias=b), min=0)
    ref64 = torch.clamp(T.conv(x.double(), w.double(), 1, 'SAME', bias=b.double()), min=0)
    outs = {}
    for flag in (False, True):
        with ops.configure(split16=flag, clear_pack_cache=True):
            buf = torch.full((G, D, H, W, cout + 8), -3.0, device=cuda)
            got, st = ops.conv(x.to(cuda), ('c3b', G, D, H, W, cin, cout), w.numpy(), bias=b.to(cuda), relu=True,
                               want_stats=True, out=buf, y_coff=4, groups=G)
            y = buf.cpu()
            _close(y[..., 4:4 + cout], want)
            assert float((y[..., :4] + 3.0).abs().max()) == 0.0 and float((y[..., 4 + cout:] + 3.0).abs().max()) == 0.0
            s = st.partial.reshape(G, -1, 2, st.cpad).sum(1).cpu()
            for g in range(G):
                _close(s[g, 0, :cout].float(), want[g].reshape(-1, cout).double().sum(0).float(), 1e-5)
                _close(s[g, 1, :cout].float(), (want[g].reshape(-1, cout).double() ** 2).sum(0).float(), 1e-5)
            outs[flag] = y[..., 4:4 + cout].double()
            if flag:
                one = ops.conv(x[1 % G].to(cuda), ('c3b', G, D, H, W, cin, cout), w.numpy(), bias=b.to(cuda), relu=True)
                assert torch.equal(one.cpu(), y[1 % G, ..., 4:4 + cout])
    e32 = float((outs[False] - ref64).abs().max())
    e16 = float((outs[True] - ref64).abs().max())
    print('max abs error vs float64: fp32 MFMA %.3e, split fp16 %.3e' % (e32, e16))
    assert e16 <= 2.0 * e32 + 1e-7 * float(ref64.abs().max())


@pytest.mark.parametrize('cin,cout', [(16, 32), (32, 64), (48, 32)])
@pytest.mark.parametrize('G,D,H,W', [(1, 8, 16, 32), (3, 9, 19, 21), (2, 6, 8, 17), (1, 5, 7, 33)])
def test_conv3d_s2b_stride2_split16_matches_oracle(cuda, G, D, H, W, cin, cout):
    """Split-bf16 stride-2 3x3x3 convolutions (conv3d_s2b.hip: even | odd de-interleaved halo rows, waves split z plane and
    output-channel half) against the oracle at the fp32 bar and against a float64 evaluation (no further than the gather kernel
    x 2): TF SAME padding for even and odd sizes on every axis, bias + ReLU into a channel slice, statistics, groups."""
    from atvsnet_amd import ops
    x = _rand((G, D, H, W, cin), 90)
    w = _rand((3, 3, 3, cin, cout), 91, (2.0 / (27 * cin)) ** 0.5)
    b = _rand((cout,), 92)
    want = torch.clamp(T.conv(x, w, 2, 'SAME', bias=b), min=0)
    ref64 = torch.clamp(T.conv(x.double(), w.double(), 2, 'SAME', bias=b.double()), min=0)
    Do, Ho, Wo = want.shape[1:4]
    outs = {}
    for flag in (False, True):
        with ops.configure(split16=flag, clear_pack_cache=True):
            buf = torch.full((G, Do, Ho, Wo, cout + 8), -3.0, device=cuda)
            got, st = ops.conv(x.to(cuda), ('s2b', G, D, H, W, cin, cout), w.numpy(), stride=2, bias=b.to(cuda), relu=True,
                               want_stats=True, out=buf, y_coff=4, groups=G)
            y = buf.cpu()
            _close(y[..., 4:4 + cout], want)
            assert float((y[..., :4] + 3.0).abs().max()) == 0.0 and float((y[..., 4 + cout:] + 3.0).abs().max()) == 0.0
            s = st.partial.reshape(G, -1, 2, st.cpad).sum(1).cpu()
            for g in range(G):
                _close(s[g, 0, :cout].float(), want[g].reshape(-1, cout).double().sum(0).float(), 1e-5)
                _close(s[g, 1, :cout].float(), (want[g].reshape(-1, cout).double() ** 2).sum(0).float(), 1e-5)
            outs[flag] = y[..., 4:4 + cout].double()
            if flag:
                one = ops.conv(x[1 % G].to(cuda), ('s2b', G, D, H, W, cin, cout), w.numpy(), stride=2, bias=b.to(cuda), relu=True)
                assert torch.equal(one.cpu(), y[1 % G, ..., 4:4 + cout])
    e32 = float((outs[False] - ref64).abs().max())
    e16 = float((outs[True] - ref64).abs().max())
    print('max abs error vs float64: fp32 MFMA %.3e, split fp16 %.3e' % (e32, e16))
    assert e16 <= 2.0 * e32 + 1e-7 * float(ref64.abs().max())


def test_conv_split_siblings_match_dense(cuda):
    from atvsnet_amd import ops
    ops.clear_pack_cache()
    D, h, w = 6, 16, 40
    var = _rand((D, h, w, 8), 70)
    const = _rand((h, w, 5), 71)
    cmap = [('c', i) for i in range(5)] + [('v', i) for i in range(8)] + [('v', 3), ('c', 1)]
    dense = torch.cat([const[None].expand(D, -1, -1, -1), var, var[..., 3:4], const[None, ..., 1:2].expand(D, -1, -1, -1)], -1)
    w1 = _rand((3, 3, 3, 15, 8), 72, 0.3)
    w2 = _rand((3, 3, 3, 15, 16), 73, 0.3)
    sv = ops.SplitVolume(var.to(cuda), const.to(cuda), cmap)
    (y, st), (y2, st2) = ops.conv_split_siblings(sv, 'ss1', w1.numpy(), 'ss2', w2.numpy())
    _close(y.cpu(), T.conv(dense[None], w1, 1, 'SAME')[0])
    _close(y2.cpu(), T.conv(dense[None], w2, 2, 'SAME')[0])


@pytest.mark.parametrize('D,H,W', [(6, 8, 16), (5, 9, 21), (2, 3, 4), (16, 32, 40)])
def test_conv3d_8to1_head(cuda, D, H, W):
    from atvsnet_amd import ops
    x = _rand((1, D, H, W, 8), 40)
    w = _rand((3, 3, 3, 8, 1), 41, 0.3)
    want = T.conv(x, w, 1, 'SAME')[0]
    got = ops.conv3d_8to1(x[0].to(cuda), w.to(cuda))
    assert tuple(got.shape) == (D, H, W, 1)
    _close(got.cpu(), want)


@pytest.mark.parametrize('shape,cin,cout', [((48, 64, 80), 16, 8), ((24, 32, 40), 32, 32), ((96, 128, 160), 8, 8)])
def test_in_launch_finalize_matches_separate_finalize(cuda, shape, cin, cout):
    """The batch-norm moments finished by the last-arriving workgroup of the convolution equal the ones the
    separate atvs_bn_finalize launch computes from the same partial sums -- on every one of many launches
    (inter-workgroup hand-off: stale data would show up as run-to-run differences)."""
    from atvsnet_amd import ops
    D, H, W = shape
    x = _rand((D, H, W, cin), 50).to(cuda)
    w = _rand((3, 3, 3, cin, cout), 51, 0.2).numpy()
    ref = None
    with ops.configure(xp1w=False, conv_c16=False):           # the in-launch finalize lives in the tiled kernel
        with ops.configure(fused_finalize=True):
            for it in range(12):
                y, st = ops.conv(x, ('fin', shape, cin, cout), w, want_stats=True)
                assert st.params is not None
                fused = st.params.clone()
                st.params = None
                sep = ops.bn_params(st, cout, y)
                assert torch.allclose(fused[:2], sep[:2], rtol=1e-6, atol=1e-7) and torch.all(fused[2] == 0)
                if ref is None:
                    ref = fused
                assert torch.equal(fused, ref)
        y2, st2 = ops.conv(x, ('fin', shape, cin, cout), w, want_stats=True)
    assert st2.params is None and torch.equal(y2, y)


def test_weight_reload_changes_the_result(cuda):
    """Run a convolution, replace its kernel in the variable store, run it again: the new kernel must be used (the
    arranged copies are cached by variable name -- ADVICE r1) and the other device must not be touched."""
    import numpy as np
    from atvsnet_amd import variables
    from atvsnet_amd.cnn_wrapper.network import Network

    class One(Network):
        def setup(self):
            self.feed('data').conv(3, 16, 1, relu=False, name='reload_probe')

    store = variables.default_store()
    g = torch.Generator().manual_seed(21)
    x = torch.randn(1, 8, 16, 24, 16, generator=g).to(cuda)
    w1 = torch.randn(3, 3, 3, 16, 16, generator=g).numpy()
    store.set('reload_probe/kernel', w1)
    y1 = One({'data': x}, is_training=True).get_output().clone()
    store.set('reload_probe/kernel', 2.0 * w1)
    y2 = One({'data': x}, is_training=True).get_output()
    assert float((y2 - 2.0 * y1).abs().max()) <= 1e-5 * float(y1.abs().max())
    assert float(y1.abs().max()) > 0
    del store.host['reload_probe/kernel']


def test_tensor_on_another_device_is_refused(cuda):
    """Kernels launch on the CURRENT device's stream: a tensor that lives elsewhere must raise, not fault."""
    if torch.cuda.device_count() < 2:
        pytest.skip('one device')
    from atvsnet_amd import ops
    x = torch.zeros(4, 8, 8, device='cuda:1')
    with pytest.raises(RuntimeError):
        ops.channel_stats(x)


@pytest.fixture(params=['split16', 'fp32'])
def towerkernel(request):
    """Run a test on both 3x3 tower kernels: conv2d_b.hip (split-fp16 operands, default) and conv2d_lds.hip (fp32 MFMA)."""
    from atvsnet_amd import ops
    with ops.configure(split16=request.param == 'split16', clear_pack_cache=True):
        yield request.param


@pytest.mark.parametrize('cin,cout,dil,H,W,G', [
    (128, 128, 2, 32, 48, 1), (128, 128, 4, 30, 44, 2), (64, 128, 1, 17, 33, 3), (320, 128, 1, 16, 32, 1),
    (64, 64, 1, 24, 40, 2), (32, 32, 1, 36, 52, 2), (32, 64, 1, 9, 16, 1)])
def test_conv2d_lds_matches_oracle(cuda, towerkernel, cin, cout, dil, H, W, G):
    """The LDS-tiled 2-D convolutions of the feature towers (conv2d_b.hip: split-fp16 operands; conv2d_lds.hip: fp32 MFMA):
    every instantiation, ragged edges, several independent images per launch, bias / residual / ReLU and the per-image
    moments, both at the same fp32 bar."""
    from atvsnet_amd import ops
    g = torch.Generator().manual_seed(cin + cout + dil)
    x = torch.randn(G, H, W, cin, generator=g)
    w = torch.randn(3, 3, cin, cout, generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, generator=g)
    res = torch.randn(G, H, W, cout, generator=g)
    assert ops.conv2d_lds_ok(cin, cout, dil, H, W)
    y, st = ops.conv2d_lds(x.to(cuda), ('t', cin, cout, dil), w.numpy(), dil, b.to(cuda), res.to(cuda), True, True)
    want = torch.clamp(T.conv(x, w, 1, 'SAME', dilation=dil, bias=b) + res, min=0)
    assert float((y.cpu() - want).abs().max()) <= 2e-5 * float(want.abs().max())
    p = ops.bn_params(st, cout, y).cpu().reshape(G, 3, cout)
    for i in range(G):
        flat = want[i].reshape(-1, cout).double()
        assert float((p[i, 0] - flat.mean(0)).abs().max()) <= 1e-5
        assert float((p[i, 1] - 1.0 / torch.sqrt(flat.var(0, unbiased=False) + 1e-3)).abs().max()) <= 1e-4
    # plain form through ops.conv (one image): same kernel, no epilogue operands
    y1 = ops.conv(x[0].to(cuda), ('t', cin, cout, dil), w.numpy(), dilation=dil)
    w1 = T.conv(x[:1], w, 1, 'SAME', dilation=dil)[0]
    assert float((y1.cpu() - w1).abs().max()) <= 2e-5 * float(w1.abs().max())


def test_conv2d_lds_normalise_on_load(cuda, towerkernel):
    """in_params: the producer's training-mode batch norm (+ ReLU) applied while the tile is staged equals
    normalising first; the SAME padding is zero AFTER the normalisation."""
    from atvsnet_amd import ops
    g = torch.Generator().manual_seed(77)
    G, H, W, cin, cout = 2, 20, 36, 64, 64
    x = torch.randn(G, H, W, cin, generator=g) * 3 + 1
    w = torch.randn(3, 3, cin, cout, generator=g) * 0.05
    beta = torch.randn(cin, generator=g) * 0.1
    xd = x.to(cuda)
    st = ops.channel_stats(xd, groups=G)
    params = ops.bn_params(st, cin, xd, beta.to(cuda))
    assert tuple(params.shape) == (G, 3, cin)
    y = ops.conv2d_lds(xd, 'nol', w.numpy(), in_params=params, in_relu=True)
    xn = torch.stack([torch.clamp(T.batch_norm_train(x[i:i + 1], beta)[0], min=0) for i in range(G)])
    want = T.conv(xn, w, 1, 'SAME')
    assert float((y.cpu() - want).abs().max()) <= 3e-5 * float(want.abs().max())
    # and the explicit two-pass form on the device agrees to rounding
    two = ops.conv2d_lds(ops.bn_apply(xd.clone(), params, True), 'nol', w.numpy())
    assert float((y - two).abs().max()) <= 1e-6 * float(two.abs().max())


@pytest.mark.parametrize('cin,cout,H,W,G', [(128, 128, 16, 20, 2), (64, 128, 9, 13, 3), (32, 64, 7, 11, 1), (128, 32, 16, 16, 2),
                                            (64, 64, 5, 50, 2), (320, 128, 12, 11, 1), (96, 32, 3, 5, 2)])
def test_conv1x1_matches_oracle(cuda, towerkernel, cin, cout, H, W, G):
    """The 1x1 GEMM kernels of the towers (conv1x1_b.hip: split-fp16 operands, default; conv1x1.hip: fp32 MFMA): bias / residual /
    ReLU / per-image moments, ragged pixel counts, and the bottleneck's pre-activation applied on load -- same bars for both."""
    from atvsnet_amd import ops
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(G, H, W, cin, generator=g) * 2 + 0.5
    w = torch.randn(1, 1, cin, cout, generator=g) * (1.0 / cin) ** 0.5
    b = torch.randn(cout, generator=g)
    res = torch.randn(G, H, W, cout, generator=g)
    if towerkernel == 'fp32' and cin > 128:
        pytest.skip('conv1x1.hip keeps all weights in LDS: Cin <= 128')
    assert ops.conv1x1_ok(cin, cout)
    y, st = ops.conv(x.to(cuda), ('p', cin, cout), w.numpy(), bias=b.to(cuda), residual=res.to(cuda), want_stats=True, groups=G)
    want = T.conv(x, w, 1, 'SAME', bias=b) + res
    assert float((y.cpu() - want).abs().max()) <= 2e-5 * float(want.abs().max())
    p = ops.bn_params(st, cout, y).cpu().reshape(G, 3, cout)
    for i in range(G):
        flat = want[i].reshape(-1, cout).double()
        assert float((p[i, 0] - flat.mean(0)).abs().max()) <= 1e-5
        assert float((p[i, 1] - 1.0 / torch.sqrt(flat.var(0, unbiased=False) + 1e-3)).abs().max()) <= 1e-4
    # normalise-on-load == normalising first
    beta = torch.randn(cin, generator=g) * 0.1
    xd = x.to(cuda)
    if cin <= 128:
        params = ops.bn_params(ops.channel_stats(xd, groups=G), cin, xd, beta.to(cuda))
    else:                                   # atvs_channel_stats stops at 128 channels: the moments from torch
        xf = xd.reshape(G, -1, cin).double()
        params = torch.stack([xf.mean(1), 1.0 / torch.sqrt(xf.var(1, unbiased=False) + 1e-3),
                              beta.to(cuda).double().expand(G, cin)], 1).float().contiguous()
    y2 = ops.conv(xd, ('p', cin, cout), w.numpy(), bias=b.to(cuda), relu=True, groups=G, in_params=params, in_relu=True)
    xn = torch.stack([torch.clamp(T.batch_norm_train(x[i:i + 1], beta)[0], min=0) for i in range(G)])
    want2 = torch.clamp(T.conv(xn, w, 1, 'SAME', bias=b), min=0)
    assert float((y2.cpu() - want2).abs().max()) <= 3e-5 * float(want2.abs().max())
    # one image, legacy (unbatched) call form
    y1 = ops.conv(x[0].to(cuda), ('p', cin, cout), w.numpy())
    assert float((y1.cpu() - T.conv(x[:1], w, 1, 'SAME')[0]).abs().max()) <= 2e-5 * float(want.abs().max())


@pytest.mark.parametrize('C,H,W,G', [(64, 128, 160, 5), (64, 13, 37, 2), (64, 8, 16, 1), (32, 256, 320, 5), (32, 21, 50, 3),
                                     (32, 9, 17, 1), (64, 40, 32, 2)])
def test_bottleneck_fused_is_bitwise_the_three_launches(cuda, C, H, W, G):
    """bottleneck_b.hip: an identity-shortcut pre-activation residual unit (reference cnn_wrapper/network.py:552-602) as ONE
    launch -- conv1 over the tile and its halo, r1 / r2 in LDS as fp16 pieces.  Same K order and packed weights as the three
    launches it replaces (conv1x1_b -> conv2d_b -> conv1x1_b): the output must be THEIR bits, on full tower sizes and ragged
    ones (rows / columns cut by the 8 x 16 tile, images smaller than a tile row); against the oracle's bottleneck at the
    convolution bar; the moments it hands to the next unit against the output's own."""
    from atvsnet_amd import ops
    from oracle import nets
    g = torch.Generator().manual_seed(100 + C + H)
    x = torch.randn(G, H, W, C, generator=g) * 1.5 + 0.3
    W_ = {'u/preact/beta': torch.randn(C, generator=g) * 0.1}
    for name, shape, sc in (('conv1', (1, 1, C, C), (1.0 / C) ** 0.5), ('conv2', (3, 3, C, C), (1.0 / (9 * C)) ** 0.5),
                            ('conv3', (1, 1, C, C), (1.0 / C) ** 0.5)):
        W_['u/%s/weights' % name] = torch.randn(*shape, generator=g) * sc
        W_['u/%s/biases' % name] = torch.randn(C, generator=g) * 0.1
    xd = x.to(cuda)
    assert ops.bottleneck_ok(C, 1, H, W)
    params = ops.bn_params(ops.channel_stats(xd, groups=G), C, xd, W_['u/preact/beta'].to(cuda))
    keys = tuple(('btl', C, H, W, n) for n in ('conv1', 'conv2', 'conv3'))
    wb = [(W_['u/%s/weights' % n].numpy(), W_['u/%s/biases' % n].to(cuda)) for n in ('conv1', 'conv2', 'conv3')]
    y, st = ops.bottleneck(xd, params, keys, wb[0][0], wb[0][1], wb[1][0], wb[1][1], wb[2][0], wb[2][1])
    # the three launches
    r = ops.conv(xd, keys[0], wb[0][0], bias=wb[0][1], relu=True, groups=G, in_params=params, in_relu=True)
    r = ops.conv(r, keys[1], wb[1][0], bias=wb[1][1], relu=True, groups=G)
    ref, st_ref = ops.conv(r, keys[2], wb[2][0], bias=wb[2][1], residual=xd, want_stats=True, groups=G)
    assert torch.equal(y, ref)
    # the oracle (per image: batch statistics per call)
    want = torch.cat([nets.bottleneck(x[i:i + 1], W_, 'u', 3, C) for i in range(G)])
    assert float((y.cpu() - want).abs().max()) <= 3e-5 * float(want.abs().max())
    # the moments of y for the next unit: the fused rows against the unfused rows (other grouping, same sums) and torch
    p, p_ref = ops.bn_params(st, C, y).reshape(G, 3, C), ops.bn_params(st_ref, C, ref).reshape(G, 3, C)
    assert float((p - p_ref).abs().max()) <= 1e-6 * float(p_ref.abs().max())
    for i in range(G):
        flat = y[i].reshape(-1, C).double()
        assert float((p[i, 0].double() - flat.mean(0)).abs().max()) <= 1e-5
        assert float((p[i, 1].double() - 1.0 / torch.sqrt(flat.var(0, unbiased=False) + 1e-3)).abs().max()) <= 1e-4


def test_residual_block_with_fused_units_equals_unfused(cuda, weights):
    """res_block conv1_x of ResNetDS2SPP (8 units: one strided projection unit, seven identity units) through the operator API with
    the fused unit (default) and without (`ops.configure(bottleneck=False)`): the first identity unit sees identical inputs and
    moments -> identical bits; behind it the moments come from differently grouped partial sums, so the whole block is compared
    at 1e-6 of its scale; and the fused units really ran (launch watch)."""
    from atvsnet_amd import ops
    from atvsnet_amd.cnn_wrapper.network import Network

    class Block(Network):
        def setup(self):
            (self.feed('data').res_block(3, 64, num_block=8, stride=2, name='conv1_x'))

    x = torch.randn(2, 64, 96, 32, generator=torch.Generator().manual_seed(3)).to(cuda)
    outs, launches = {}, {}
    for fused in (True, False):
        with ops.configure(bottleneck=fused):
            ops.watch('*')
            net = Block({'data': x}, is_training=True, independent_samples=True)
            launches[fused] = [str(k[0]) for k in ops.watch(None)]
            outs[fused] = net.get_output().clone()
    a, b = outs[True], outs[False]
    assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
    # one launch per identity unit instead of three; the strided projection unit keeps its four
    unit = lambda ls, scope: sum(1 for k in ls if k.startswith(scope + '/'))      # noqa: E731
    assert unit(launches[True], 'conv1_x_1') == 1 and unit(launches[False], 'conv1_x_1') == 3
    assert unit(launches[True], 'conv1_x') == 1 and unit(launches[True], 'conv1_x_0') == 4
    assert len(launches[False]) - len(launches[True]) == 2 * 7


@pytest.mark.parametrize('dil,H,W,G,proj', [(2, 128, 160, 5, False), (4, 128, 160, 5, False), (4, 13, 37, 2, True), (2, 9, 17, 1, False),
                                            (4, 40, 32, 2, True)])
def test_conv2_and_conv3_of_a_unit_in_one_launch_bitwise(cuda, dil, H, W, G, proj):
    """conv2d_b.hip's TAIL form: conv2 (3x3, dilation 2 / 4) and conv3 (1x1) + shortcut of a 128-channel residual unit (reference
    cnn_wrapper/network.py:585-601) in one launch -- r2 crosses the wavefronts through LDS as fp16 pieces.  Bit for bit the two
    launches it replaces, at the tower's size and ragged ones, with the identity shortcut and a separate (projection) tensor;
    against the oracle's convolutions; its moments against the output's own."""
    from atvsnet_amd import ops
    C = 128
    g = torch.Generator().manual_seed(7 * dil + H)
    r1 = torch.clamp(torch.randn(G, H, W, C, generator=g), min=0)
    sc = torch.randn(G, H, W, C, generator=g)
    w2 = torch.randn(3, 3, C, C, generator=g) * (1.0 / (9 * C)) ** 0.5
    w3 = torch.randn(1, 1, C, C, generator=g) * (1.0 / C) ** 0.5
    b2, b3 = torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    assert ops.conv2d_tail_ok(C, dil, H, W)
    keys = (('tail', dil, H, 2), ('tail', dil, H, 3))
    rd, scd = r1.to(cuda), (sc.to(cuda) if proj else r1.to(cuda))
    y, st = ops.conv2d_tail(rd, keys, w2.numpy(), b2.to(cuda), w3.numpy(), b3.to(cuda), residual=scd, dilation=dil)
    r2 = ops.conv(rd, keys[0], w2.numpy(), dilation=dil, bias=b2.to(cuda), relu=True, groups=G)
    ref, st_ref = ops.conv(r2, keys[1], w3.numpy(), bias=b3.to(cuda), residual=scd, want_stats=True, groups=G)
    assert torch.equal(y, ref)
    want = T.conv(torch.clamp(T.conv(r1, w2, 1, 'SAME', dil, bias=b2), min=0), w3, 1, 'SAME', bias=b3) + (sc if proj else r1)
    assert float((y.cpu() - want).abs().max()) <= 3e-5 * float(want.abs().max())
    # the moments: rows per 4 x 16 tile here, per 128 pixels in conv1x1_b -- the same sums in another grouping
    p, p_ref = ops.bn_params(st, C, y), ops.bn_params(st_ref, C, ref)
    assert float((p - p_ref).abs().max()) <= 1e-6 * float(p_ref.abs().max())
    flat = y.reshape(G, -1, C).double()
    assert float((p.reshape(G, 3, C)[:, 0].double() - flat.mean(1)).abs().max()) <= 1e-5


@pytest.mark.parametrize('G,H,W,cin', [(5, 256, 320, 64), (2, 18, 34, 64), (1, 16, 32, 32), (3, 42, 70, 64)])
def test_strided_unit_conv2_on_the_split_kernel(cuda, G, H, W, cin):
    """conv2d_b.hip's stride-2 form: the 3x3 stride-2 convolution behind explicit symmetric padding 1 of a residual unit's first
    block (reference cnn_wrapper/network.py:588-595, quirk C17: taps centred on pixel 2 i) -- conv1_x_0/conv2 of ResNetDS2SPP at
    its size and ragged ones: against the oracle's convolution, the generic kernel it replaces, and its moments."""
    from atvsnet_amd import ops
    cout = 64
    g = torch.Generator().manual_seed(H + cin)
    x = torch.clamp(torch.randn(G, H, W, cin, generator=g), min=0)
    w = torch.randn(3, 3, cin, cout, generator=g) * (1.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    pad = [(1, 1), (1, 1)]
    y, st = ops.conv(x.to(cuda), ('s2', H, cin), w.numpy(), stride=2, explicit_pad=pad, bias=b.to(cuda), relu=True,
                     want_stats=True, groups=G)
    want = torch.clamp(T.conv(x, w, 2, 'VALID', bias=b, explicit_pad=pad), min=0)
    assert tuple(y.shape) == tuple(want.shape) == (G, H // 2, W // 2, cout)
    assert float((y.cpu() - want).abs().max()) <= 2e-5 * float(want.abs().max())
    with ops.configure(split_off=('c2b',), clear_pack_cache=True):
        ref = ops.conv(x.to(cuda), ('s2', H, cin), w.numpy(), stride=2, explicit_pad=pad, bias=b.to(cuda), relu=True, groups=G)
    assert float((y - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    p = ops.bn_params(st, cout, y).reshape(G, 3, cout)
    flat = y.reshape(G, -1, cout).double()
    assert float((p[:, 0].double() - flat.mean(1)).abs().max()) <= 1e-5
    assert float((p[:, 1].double() - 1.0 / torch.sqrt(flat.var(1, unbiased=False) + 1e-3)).abs().max()) <= 1e-4


@pytest.mark.parametrize('G,H,W', [(5, 2, 3), (5, 4, 5), (2, 8, 10), (1, 3, 17), (3, 9, 2)])
def test_tower_3x3_on_tiny_maps(cuda, G, H, W):
    """The pyramid branches' 3x3 convolutions (128 -> 32) on their pooled maps of 2 x 3 ... 8 x 10 pixels (reference
    cnn_wrapper/atvsnet.py:271-286) run on conv2d_b.hip with one masked tile per image: against the oracle's convolution and
    the moments of what was written (as few as 6 samples per image, quirk C16)."""
    from atvsnet_amd import ops
    cin, cout = 128, 32
    g = torch.Generator().manual_seed(H * 31 + W)
    x = torch.randn(G, H, W, cin, generator=g)
    w = torch.randn(3, 3, cin, cout, generator=g) * (1.0 / (9 * cin)) ** 0.5
    assert ops.conv2d_lds_ok(cin, cout, 1, H, W)
    y, st = ops.conv(x.to(cuda), ('tiny', H, W), w.numpy(), want_stats=True, groups=G)
    want = T.conv(x, w, 1, 'SAME')
    assert float((y.cpu() - want).abs().max()) <= 2e-5 * float(want.abs().max())
    p = ops.bn_params(st, cout, y).reshape(G, 3, cout)
    flat = want.reshape(G, -1, cout).double()
    assert float((p[:, 0].cpu().double() - flat.mean(1)).abs().max()) <= 1e-5
    assert float((p[:, 1].cpu().double() - 1.0 / torch.sqrt(flat.var(1, unbiased=False) + 1e-3)).abs().max()) <= 1e-3
