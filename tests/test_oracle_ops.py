"""CPU: the oracle's TF-op restatements against independent re-derivations (direct loops in
float64 numpy) and analytic known answers."""
import numpy as np
import pytest
import torch

from oracle import nets
from oracle import model as OM
from oracle import tf_ops as T


def _rand(shape, seed, scale=1.0):
    return scale * torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def conv_loops(x, w, stride, pads, dil=1):
    """Direct convolution: x (sp.., Cin), w (k.., Cin, Cout), pads = before per axis, float64."""
    x = np.asarray(x, np.float64)
    w = np.asarray(w, np.float64)
    nsp = x.ndim - 1
    ins, ks = x.shape[:nsp], w.shape[:nsp]
    outs = [-(-ins[i] // stride) for i in range(nsp)]
    y = np.zeros(tuple(outs) + (w.shape[-1],))
    for o in np.ndindex(*outs):
        for k in np.ndindex(*ks):
            src = tuple(o[i] * stride + k[i] * dil - pads[i] for i in range(nsp))
            if all(0 <= src[i] < ins[i] for i in range(nsp)):
                y[o] += x[src] @ w[k]
    return y


@pytest.mark.parametrize('size,stride', [(8, 1), (8, 2), (7, 2), (9, 1)])
def test_same_padding_is_end_heavy(size, stride):
    pb, pe, out = T.same_pad(size, 3, stride)
    assert out == -(-size // stride)
    assert pb + pe == max((out - 1) * stride + 3 - size, 0) and pb <= pe
    if size % 2 == 0 and stride == 2:
        assert (pb, pe) == (0, 1)            # Appendix B.1: NOT torch's symmetric (1,1)


@pytest.mark.parametrize('stride', [1, 2])
def test_conv3d_same_vs_loops(stride):
    x, w = _rand((1, 4, 6, 5, 3), 1), _rand((3, 3, 3, 3, 4), 2)
    pads = [T.same_pad(s, 3, stride)[0] for s in (4, 6, 5)]
    want = conv_loops(x[0], w, stride, pads)
    got = T.conv(x, w, stride, 'SAME')[0].numpy()
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 1e-4


@pytest.mark.parametrize('stride,rate', [(1, 1), (2, 1), (1, 2), (1, 4)])
def test_conv2d_same_vs_loops(stride, rate):
    x, w = _rand((1, 9, 11, 3), 3), _rand((3, 3, 3, 5), 4)
    pads = [T.same_pad(s, 3, stride, rate)[0] for s in (9, 11)]
    want = conv_loops(x[0], w, stride, pads, rate)
    assert np.abs(T.conv(x, w, stride, 'SAME', rate)[0].numpy() - want).max() < 1e-4


def test_conv2d_explicit_pad_valid_vs_loops():
    x, w = _rand((1, 8, 12, 2), 5), _rand((3, 3, 2, 3), 6)
    want = conv_loops(x[0], w, 4, [1, 1])          # taps centred on 4*i (quirk C17)
    got = T.conv(x, w, 4, 'VALID', 1, explicit_pad=[(1, 1), (1, 1)])[0].numpy()
    assert got.shape == (2, 3, 3) and np.abs(got - want[:2, :3]).max() < 1e-4


def test_conv3d_transpose_vs_scatter_loops():
    """Appendix B.2: out[2i+k] += in[i] W[k], cropped to 2*in at the end; kernel [k,k,k,Cout,Cin]."""
    x, w = _rand((1, 3, 2, 4, 3), 7), _rand((3, 3, 3, 2, 3), 8)
    xin, wk = x[0].double().numpy(), w.double().numpy()
    full = np.zeros((7, 5, 9, 2))
    for i in np.ndindex(3, 2, 4):
        for k in np.ndindex(3, 3, 3):
            o = tuple(2 * i[a] + k[a] for a in range(3))
            full[o] += wk[k] @ xin[i]
    want = full[:6, :4, :8]
    got = T.conv3d_transpose_same(x, w, 2)[0].numpy()
    assert got.shape == want.shape and np.abs(got - want).max() < 1e-4


def test_conv3d_transpose_is_gradient_of_same_conv():
    """The transposed conv equals d/dx of the forward SAME stride-2 conv (how TF defines it)."""
    x = _rand((1, 4, 4, 6, 2), 9).double().requires_grad_(True)
    w = _rand((3, 3, 3, 2, 3), 10).double()
    y = T.conv(x, w, 2, 'SAME')
    g = _rand(tuple(y.shape), 11).double()
    (y * g).sum().backward()
    # the forward kernel [k,k,k,Cin=2,Cout=3] read as a transposed-conv kernel [k,k,k,Cout_T=2,Cin_T=3]
    assert torch.allclose(T.conv3d_transpose_same(g, w, 2), x.grad, atol=1e-10)


def test_batch_norm_train_properties():
    x = _rand((1, 5, 6, 7, 4), 12, 3.0) + 2.0
    y = T.batch_norm_train(x).reshape(-1, 4).double()
    v = x.reshape(-1, 4).double().var(0, unbiased=False)
    assert y.mean(0).abs().max() < 1e-5
    assert torch.allclose(y.var(0, unbiased=False), v / (v + 1e-3), rtol=1e-4)
    beta = torch.tensor([0.5, -1.0, 0.0, 2.0])
    assert torch.allclose(T.batch_norm_train(x, beta=beta), T.batch_norm_train(x) + beta)
    # a 1x1 map normalises to exactly zero (quirk C16: SPP branch with a single pooled pixel)
    assert torch.all(T.batch_norm_train(_rand((1, 1, 1, 8), 13)) == 0)


@pytest.mark.parametrize('H,W,k', [(32, 40, 64), (32, 40, 16), (30, 45, 8), (7, 9, 4)])
def test_avg_pool_same_counts_valid_only(H, W, k):
    x = _rand((1, H, W, 2), 14)
    got = T.avg_pool2d_same(x, k, k)[0].numpy()
    pt, _, Ho = T.same_pad(H, k, k)
    pl, _, Wo = T.same_pad(W, k, k)
    assert got.shape == (Ho, Wo, 2)
    xn = x[0].double().numpy()
    for oy in range(Ho):
        for ox in range(Wo):
            ys, xs = slice(max(oy * k - pt, 0), min(oy * k - pt + k, H)), slice(max(ox * k - pl, 0), min(ox * k - pl + k, W))
            assert np.abs(got[oy, ox] - xn[ys, xs].mean((0, 1))).max() < 1e-5


def test_resize_bilinear_align_corners():
    x = _rand((1, 3, 5, 2), 15)
    y = T.resize_bilinear_align_corners(x, (9, 17))
    assert torch.equal(y[0, 0, 0], x[0, 0, 0]) and torch.allclose(y[0, -1, -1], x[0, -1, -1])
    assert torch.allclose(y[0, 4, 8], x[0, 1, 2], atol=1e-6)          # (8/16)*(5-1) = 2, (4/8)*(3-1) = 1
    assert torch.allclose(y[0, 0, 2], 0.5 * (x[0, 0, 0] + x[0, 0, 1]), atol=1e-6)
    one = T.resize_bilinear_align_corners(x[:, :1, :1], (4, 6))       # 1x1 source: constant
    assert torch.equal(one, x[:, :1, :1].expand(1, 4, 6, 2))


def test_linspace_and_round():
    ls = T.linspace(0.05, 0.05 + 31 * 0.01, 32)
    assert ls.dtype == torch.float32 and ls.shape == (32,) and float(ls[0]) == pytest.approx(0.05)
    assert torch.equal(T.tf_round(torch.tensor([0.5, 1.5, 2.5, -0.5])), torch.tensor([0., 2., 2., -0.]))


def test_softargmin_known_answers():
    D, h, w = 16, 3, 4
    ds, di = torch.tensor([0.1]), torch.tensor([0.02])
    cost = torch.full((1, D, h, w), 50.0)
    cost[0, 5] = -50.0
    assert torch.allclose(OM.prob2depth(cost, D, ds, di), torch.full((1, h, w, 1), 0.1 + 5 * 0.02), rtol=1e-6)
    assert torch.allclose(OM.prob2depth(torch.zeros(1, D, h, w), D, ds, di), torch.full((1, h, w, 1), 0.1 + 7.5 * 0.02),
                          rtol=1e-6)
    lo, up = OM.prob2depth_upsample(cost, D, ds, di)
    assert up.shape == (1, 4 * h, 4 * w, 1) and torch.allclose(up, torch.full_like(up, 0.2), rtol=1e-6)


def test_aanet_known_answers(weights):
    X = _rand((1, 4, 5, 6, 8, 1), 16)
    assert torch.allclose(nets.attention_aggregation(X, weights, 'attention_aggregate'), X[..., 0], atol=1e-6)
    Y = X.expand(-1, -1, -1, -1, -1, 3).contiguous()            # identical views -> X
    assert torch.allclose(nets.attention_aggregation(Y, weights, 'attention_aggregate'), X[..., 0], atol=1e-5)


def test_oracle_matches_committed_fixture(weights):
    """oracle/ is frozen by tests/golden/oracle_cfg1.npz (made by tests/golden/make_oracle_golden.py).
    Tolerance 1e-4 relative to the map's range: oneDNN's summation order depends on the thread count."""
    import os
    from atvsnet_amd import synthetic
    gold = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'oracle_cfg1.npz'))
    imgs, cams = synthetic.make_inputs(2, 128, 160, 32)
    S = {}
    d = OM.run_twoview(torch.from_numpy(imgs), torch.from_numpy(cams), weights, 32, S)
    for name, got in (('twoview_depth', d[0, ..., 0]), ('twoview_depth_b2', S['depth_b2'][0, ..., 0]),
                      ('twoview_ref_feature_c0', S['ref_feature'][0, ..., 0]),
                      ('twoview_refined_prob_d7', S['refined_prob_vol'][0, 7])):
        want = gold[name]
        assert np.abs(got.numpy() - want).max() <= 2e-4 * np.abs(want).max(), name
