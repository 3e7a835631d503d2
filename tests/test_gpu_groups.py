"""-m gpu: several independent network calls in ONE launch (the `groups` axis of the C-ABI, DESIGN.md "batched
evaluation").  The reference evaluates every network with batch 1 and per-call batch statistics (quirk C1); stacking
the calls of a depth map (views, siamese directions) on a leading axis must give the values of the separate calls:
the convolution outputs bit for bit (a workgroup never spans samples, same accumulation order), the batch-norm
moments to double-precision rounding (the partial sums are grouped differently)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


@pytest.mark.parametrize('shape,cin,cout,stride,kind', [
    ((16, 32, 48), 16, 16, 1, 'tiled'), ((12, 16, 24), 32, 32, 1, 'tiled'), ((6, 8, 12), 64, 64, 1, 'tiled/small'),
    ((16, 32, 48), 8, 8, 1, 'x-pair'), ((16, 32, 48), 16, 32, 2, 'gather s2'), ((16, 32, 48), 8, 16, 1, 'aanet 8->16'),
    ((16, 32, 48), 1, 8, 1, 'stem')])
def test_conv3d_groups_equal_separate_calls(cuda, shape, cin, cout, stride, kind):
    from atvsnet_amd import ops
    G = 3
    x = _rand((G,) + shape + (cin,), 5).to(cuda)
    w = (_rand((3, 3, 3, cin, cout), 6) * 0.1).numpy()
    yb, stb = ops.conv(x, ('g', kind, cin, cout), w, stride=stride, want_stats=True, groups=G)
    pb = ops.bn_params(stb, cout, yb)
    assert tuple(pb.shape) == (G, 3, cout)
    for g in range(G):
        y1, st1 = ops.conv(x[g], ('g', kind, cin, cout), w, stride=stride, want_stats=True)
        assert torch.equal(yb[g], y1), kind
        p1 = ops.bn_params(st1, cout, y1)
        assert float((pb[g] - p1).abs().max()) <= 1e-6 * float(p1.abs().max()), kind
    # batch norm (+ReLU) of the stacked tensor with per-sample parameters
    zb = ops.bn_apply(yb.clone(), pb, True)
    for g in range(G):
        z1 = ops.bn_apply(yb[g].clone(), pb[g].contiguous(), True)
        assert torch.equal(zb[g], z1)


def test_siblings_deconv_head_groups(cuda):
    from atvsnet_amd import ops
    G = 2
    x = _rand((G, 16, 32, 48, 16), 9).to(cuda)
    w8 = (_rand((3, 3, 3, 16, 8), 1) * 0.1).numpy()
    w16 = (_rand((3, 3, 3, 16, 16), 2) * 0.1).numpy()
    pb = _rand((G, 32, 48, 24), 3).to(cuda)
    pb2 = _rand((G, 16, 24, 48), 4).to(cuda)
    (y, st), (y2, st2) = ops.conv_siblings(x, 'sa', w8, 'sb', w16, plane_bias=pb, plane_bias2=pb2, groups=G)
    for g in range(G):
        (a, sa), (b, sb) = ops.conv_siblings(x[g], 'sa', w8, 'sb', w16, plane_bias=pb[g].contiguous(),
                                             plane_bias2=pb2[g].contiguous())
        assert torch.equal(y[g], a) and torch.equal(y2[g], b)
        assert float((ops.bn_params(st, 8, y)[g] - ops.bn_params(sa, 8, a)).abs().max()) <= 1e-6
        assert float((ops.bn_params(st2, 16, y2)[g] - ops.bn_params(sb, 16, b)).abs().max()) <= 1e-6
    # fused transposed convolution: one launch (16 -> 8) and two launches (64 -> 32)
    for cin, cout, shp in ((16, 8, (8, 16, 24)), (64, 32, (4, 8, 12))):
        xd = _rand((G,) + shp + (cin,), 11).to(cuda)
        wt = (_rand((3, 3, 3, cout, cin), 12) * 0.1).numpy()
        yd, sd = ops.conv3d_transpose_s2(xd, ('dg', cin), wt, want_stats=True, groups=G)
        pd = ops.bn_params(sd, cout, yd)
        for g in range(G):
            y1, s1 = ops.conv3d_transpose_s2(xd[g], ('dg', cin), wt, want_stats=True)
            assert torch.equal(yd[g], y1)
            assert float((pd[g] - ops.bn_params(s1, cout, y1)).abs().max()) <= 1e-6
    # 8 -> 1 head, soft-argmin
    x8 = _rand((G, 12, 16, 24, 8), 13).to(cuda)
    wh = _rand((3, 3, 3, 8, 1), 14).to(cuda)
    h = ops.conv3d_8to1(x8, wh, groups=G)
    ds, di = torch.tensor([0.05], device=cuda), torch.tensor([0.01], device=cuda)
    d = ops.softargmin(h.squeeze(-1).contiguous(), ds, di, groups=G)
    for g in range(G):
        h1 = ops.conv3d_8to1(x8[g], wh)
        assert torch.equal(h[g], h1)
        assert torch.equal(d[g], ops.softargmin(h1.squeeze(-1).contiguous(), ds, di))


def test_tower_batch_equals_separate_towers(cuda, weights):
    """ResNetDS2SPP over all views at once == one call per view (per-image statistics)."""
    from atvsnet_amd import synthetic
    from atvsnet_amd.atvsnet import model
    imgs, _ = synthetic.make_inputs(3, 128, 160, 32)
    imgs = torch.from_numpy(imgs).to(cuda)
    fb = model.feature_extraction_batch(imgs)
    sb = model.shallow_feature_batch(imgs)
    for v in range(3):
        f1 = model.TVSNet_feature_extraction(imgs, v)[0]
        assert float((fb[v] - f1).abs().max()) <= 2e-5 * float(f1.abs().max())
        s1 = model.extract_feature_shallow(imgs, 0, v)[1][0]
        assert float((sb[v] - s1).abs().max()) <= 2e-5 * float(s1.abs().max())


@pytest.mark.parametrize('views', [2, 4])
def test_batched_pipeline_equals_per_view_pipeline(cuda, weights, views):
    """The whole depth-map pipeline with every network evaluated once over its per-view calls (the default) against the
    call-per-view order of the reference, config 1 size; also as a HIP graph."""
    from atvsnet_amd import synthetic
    from atvsnet_amd.atvsnet import example as ex
    imgs, cams = synthetic.make_inputs(views, 128, 160, 32)
    imgs, cams = torch.from_numpy(imgs).to(cuda), torch.from_numpy(cams).to(cuda)
    run = (lambda b: ex.infer_twoview(imgs, cams, 32, batched=b)) if views == 2 else \
        (lambda b: ex.infer_multiview(imgs, cams, 32, batched=b, view_streams=False))
    per_view, batched = run(False), run(True)
    rel = float(((batched - per_view).abs() / per_view.abs()).mean())
    print('%d views: batched vs per-view rel-L1 %.3e, max abs %.3e' % (views, rel, float((batched - per_view).abs().max())))
    assert rel <= 2e-5
    g = ex.GraphedInference(imgs, cams, 32, batched=True)
    assert torch.equal(g(), batched)


@pytest.mark.parametrize('cin,shape,G', [(1, (16, 24, 40), 2), (2, (9, 13, 35), 3), (1, (8, 8, 32), 1)])
def test_stem_kernel_matches_oracle(cuda, cin, shape, G):
    """conv_stem.hip (1-2 channels -> 8, FMA): values, depth-plane bias, ReLU, per-sample moments, a slice of a wider
    buffer; and the MFMA form of the same layer."""
    from atvsnet_amd import ops
    from oracle import tf_ops as T
    D, H, W = shape
    x = _rand((G, D, H, W, cin), 31)
    w = _rand((3, 3, 3, cin, 8), 32) * 0.3
    pb = _rand((G, H, W, 24), 33)
    buf = torch.zeros(G, D, H, W, 32, device=cuda)
    y, st = ops.conv(x.to(cuda), ('stem', cin, shape), w.numpy(), relu=True, want_stats=True, out=buf, y_coff=8,
                     plane_bias=pb.to(cuda), groups=G)
    want = T.conv(x, w, 1, 'SAME')
    for z in range(D):
        v = 0 if z == 0 else (2 if z == D - 1 else 1)
        want[:, z] += pb[..., v * 8:v * 8 + 8]
    want = torch.clamp(want, min=0)
    got = buf[..., 8:16].cpu()
    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())
    assert float(buf[..., :8].abs().max()) == 0 and float(buf[..., 16:].abs().max()) == 0
    p = ops.bn_params(st, 8, buf).cpu().reshape(G, 3, 8)
    for g in range(G):
        flat = want[g].reshape(-1, 8).double()
        assert float((p[g, 0] - flat.mean(0)).abs().max()) <= 1e-5
        assert float((p[g, 1] - 1.0 / torch.sqrt(flat.var(0, unbiased=False) + 1e-3)).abs().max()) <= 1e-4
    with ops.configure(stem=False):
        y2 = ops.conv(x.to(cuda), ('stem-mfma', cin, shape), w.numpy(), relu=True, plane_bias=pb.to(cuda), groups=G)
    assert float((y2.cpu() - want).abs().max()) <= 2e-5 * float(want.abs().max())


@pytest.mark.parametrize('shape,G,bias', [((9, 13, 35), 3, True), ((21, 50, 70), 5, True), ((4, 8, 32), 1, False),
                                          ((6, 20, 33), 2, False), ((2, 3, 5), 2, True), ((1, 9, 40), 1, False)])
def test_refine_stems_matches_oracle(cuda, shape, G, bias):
    """refine_stems_kernel (conv_stem.hip): the geo | prob | vishull stems of CostVolRefineNet (reference
    cnn_wrapper/atvsnet.py:300-313) written with the raw photo stem as whole rows of the 32-channel concat.  Ragged
    sizes (depth not a multiple of the 4-plane tile, rows / columns cut by the 8 x 32 tile), more tiles than
    persistent workgroups ((21, 50, 70) x 5 = 630 tiles against 512), with and without the depth-plane bias; values
    and the per-sample moments of the 24 computed channels."""
    from atvsnet_amd import ops
    from oracle import tf_ops as T
    D, H, W = shape
    photo = _rand((G, D, H, W, 8), 41)
    geo, prob, hull = _rand((G, D, H, W, 2), 42), _rand((G, D, H, W, 1), 43), _rand((G, D, H, W, 1), 44)
    pb = _rand((G, H, W, 24), 45) if bias else None
    wg, wp, wh = _rand((3, 3, 3, 2, 8), 46) * 0.3, _rand((3, 3, 3, 1, 8), 47) * 0.3, _rand((3, 3, 3, 1, 8), 48) * 0.3
    buf, st = ops.refine_stems(photo.to(cuda), geo.to(cuda), pb.to(cuda) if bias else None, prob.to(cuda), hull.to(cuda),
                               ('stems-test', shape, G, bias), wg.numpy(), wp.numpy(), wh.numpy())
    g = T.conv(geo, wg, 1, 'SAME')
    if bias:
        for z in range(D):
            v = 0 if z == 0 else (2 if z == D - 1 else 1)
            g[:, z] += pb[..., v * 8:v * 8 + 8]
    want = torch.cat([photo, g, T.conv(prob, wp, 1, 'SAME'), T.conv(hull, wh, 1, 'SAME')], dim=-1)
    got = buf.cpu()
    assert torch.equal(got[..., :8], photo)
    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())
    p = ops.bn_params(st, 24, buf).cpu().reshape(G, 3, 24)
    for k in range(G):
        flat = want[k, ..., 8:].reshape(-1, 24).double()
        assert float((p[k, 0] - flat.mean(0)).abs().max()) <= 1e-5
        assert float((p[k, 1] - 1.0 / torch.sqrt(flat.var(0, unbiased=False) + 1e-3)).abs().max()) <= 1e-4


def _pending(x, w, key, G, relu=True):
    """A raw convolution output with its pending training-mode batch norm (what conv_bn(defer_bn=True) hands on)."""
    from atvsnet_amd import ops
    y, st = ops.conv(x, key, w, want_stats=True, groups=G)
    return ops.PendingBN(y, ops.bn_params(st, y.shape[-1], y), relu)


@pytest.mark.parametrize('G,shape,C', [(1, (16, 32, 48), 8), (3, (12, 24, 40), 8), (2, (9, 21, 37), 8), (2, (8, 16, 40), 24)])
def test_siblings_add_on_load_is_bitwise_the_materialised_sum(cuda, G, shape, C):
    """The U-Net's stack input I_b = bn_relu(conv_6_0) + bn_relu(conv_0_1) (reference cnn_wrapper/atvsnet.py:38-39,
    network.py:695-697) formed inside the x-pair launch: same values as bn_add followed by the plain launch.  (C = 24: three
    8-channel chunks -- taken by the fp32 kernels' two-source form, refused by the split-operand kernel.)"""
    from atvsnet_amd import ops
    xa, xb = _rand((G,) + shape + (8,), 1).to(cuda), _rand((G,) + shape + (C,), 2).to(cuda)
    wa, wb = (_rand((3, 3, 3, 8, C), 3) * 0.2).numpy(), (_rand((3, 3, 3, C, C), 4) * 0.2).numpy()
    w8, w16 = (_rand((3, 3, 3, C, 8), 5) * 0.1).numpy(), (_rand((3, 3, 3, C, 16), 6) * 0.1).numpy()
    for dense_second in (False, True):
        a = _pending(xa, wa, ('pa', G, C), G)
        b = xb.clone() if dense_second else _pending(xb, wb, ('pb', G, C), G)
        lazy = ops.PendingSum([a, b])
        if C != 8 and ops._xkind() == 'xb':
            # conv_xb's two-source form is built for ONE 8-channel chunk (the stack inputs): wider sums are materialised by
            # the caller (Network.conv_bn_siblings), the kernel entry refuses them
            assert not ops.siblings_prologue_ok(lazy)
            continue
        assert ops.siblings_prologue_ok(lazy)
        (y, st), (y2, st2) = ops.conv_siblings(lazy, ('pl8', C), w8, ('pl16', C), w16, groups=G)
        dense = ops.PendingSum([a, b]).materialize()
        (r, rt), (r2, rt2) = ops.conv_siblings(dense, ('pl8', C), w8, ('pl16', C), w16, groups=G)
        assert torch.equal(y, r) and torch.equal(y2, r2)
        assert torch.equal(ops.bn_params(st, 8, y), ops.bn_params(rt, 8, r))
        assert torch.equal(ops.bn_params(st2, 16, y2), ops.bn_params(rt2, 16, r2))


@pytest.mark.parametrize('G,shape', [(1, (16, 32, 48)), (4, (10, 20, 36))])
def test_siblings_normalise_on_load_is_bitwise_the_materialised_input(cuda, G, shape):
    """The refinement's 32-channel concat with its pending batch norm + ReLU (reference cnn_wrapper/atvsnet.py:300-316)
    normalised while the x-pair launch stages it; out-of-volume taps must read post-activation zeros."""
    from atvsnet_amd import ops
    x = _rand((G,) + shape + (32,), 7).to(cuda) + 0.5
    params = torch.stack([_rand((G, 32), 8) * 0.3, _rand((G, 32), 9).abs() + 0.5, _rand((G, 32), 10) + 0.7], 1).to(cuda)
    params = params.contiguous() if G > 1 else params[0].contiguous()
    w8, w16 = (_rand((3, 3, 3, 32, 8), 5) * 0.1).numpy(), (_rand((3, 3, 3, 32, 16), 6) * 0.1).numpy()
    lazy = ops.PendingBN(x.clone(), params, True)
    assert ops.siblings_prologue_ok(lazy)
    (y, st), (y2, st2) = ops.conv_siblings(lazy, 'nl8', w8, 'nl16', w16, groups=G)
    assert lazy._final is None                       # consumed raw
    dense = ops.bn_apply(x.clone(), params, True)
    (r, rt), (r2, rt2) = ops.conv_siblings(dense, 'nl8', w8, 'nl16', w16, groups=G)
    assert torch.equal(y, r) and torch.equal(y2, r2)
    assert torch.equal(st.partial, rt.partial) and torch.equal(st2.partial, rt2.partial)
    # a lazy input of a shape the kernel has no prologue for is refused (callers materialise it)
    bad = ops.PendingBN(_rand((G,) + shape + (8,), 1).to(cuda), params[..., :8].contiguous(), True)
    assert not ops.siblings_prologue_ok(bad)
    with pytest.raises(ValueError):
        ops.conv_siblings(bad, 'pl8', w8[:, :, :, :8], 'pl16', w16[:, :, :, :8], groups=G)


def test_networks_with_prologue_equal_materialised_inputs(cuda, weights):
    """StackedUNet_prob and CostVolRefineNet with add- / normalise-on-load against the same networks with the inputs
    materialised first (ops.configure(prologue=False)): every output bit for bit."""
    from atvsnet_amd import ops
    from atvsnet_amd.cnn_wrapper.atvsnet import StackedUNet_prob, CostVolRefineNet
    G = 2
    cost = _rand((G, 16, 32, 48, 64), 3).to(cuda)
    chan = 16                                        # the shallow features (model.py:309-316)
    photo = ops.SplitVolume(_rand((G, 16, 32, 48, chan), 20).to(cuda), _rand((G, 32, 48, 2 * chan), 21).to(cuda),
                            [('v', i) for i in range(chan)] + [('c', i) for i in range(2 * chan)])
    geo = ops.SplitVolume(_rand((G, 16, 32, 48, 2), 30).to(cuda), _rand((G, 32, 48, 2), 31).to(cuda),
                          [('v', 0)] + [('v', 1)] * chan + [('c', 0), ('c', 1)])
    res = {}
    for flag in (True, False):
        with ops.configure(prologue=flag):
            net = StackedUNet_prob({'data': cost}, is_training=True, independent_samples=True)
            ref = CostVolRefineNet({'photo_group': photo, 'geo_group': geo,
                                    'prob_vol': _rand((G, 16, 32, 48, 1), 40).to(cuda),
                                    'vis_hull': _rand((G, 16, 32, 48, 1), 41).to(cuda)}, is_training=True,
                                   independent_samples=True)
            res[flag] = [net.get_output_by_name('conv_b2_6_1').clone(), net.get_output_by_name('conv_b2_6_2').clone(),
                         ref.get_output().clone(), ref.get_output_by_name('global_refine_geo_3dconv').clone(),
                         ref.get_output_by_name('global_refine_concat').clone()]
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)


@pytest.mark.parametrize('G,shape,cin', [(2, (6, 8, 20), 64), (1, (5, 9, 18), 32), (3, (3, 4, 17), 48)])
def test_deconv_32_channels_as_two_split_launches(cuda, G, shape, cin):
    """conv_b*_4_0 (64 -> 32 at eighth resolution) as two 16-channel launches of the split-fp16 transposed-convolution kernel
    (weights of one chunk re-read per stage, channel halves of y and of the statistics rows): against torch, its moments, its
    grouped form against single launches, and against the class-fused fp32 form."""
    from oracle import tf_ops as T
    from atvsnet_amd import ops
    cout = 32
    x = _rand((G,) + shape + (cin,), 31)
    w = _rand((3, 3, 3, cout, cin), 32) * 0.2
    ops.clear_pack_cache()
    got, st = ops.conv3d_transpose_s2(x.to(cuda), ('up32', cin), w.numpy(), want_stats=True, groups=G)
    assert st.cpad == 32 and st.groups == G and st.fold == 1
    params = ops.bn_params(st, cout, got)
    for g in range(G):
        want = T.conv3d_transpose_same(x[g:g + 1], w, 2)[0]
        assert float((got[g].cpu() - want).abs().max()) <= 2e-5 * float(want.abs().max()) + 1e-6
        one, st1 = ops.conv3d_transpose_s2(x[g].to(cuda), ('up32', cin), w.numpy(), want_stats=True)
        assert torch.equal(one, got[g])
        pg = params[g] if G > 1 else params
        mean = want.reshape(-1, cout).double().mean(0)
        rstd = 1.0 / torch.sqrt(want.reshape(-1, cout).double().var(0, unbiased=False) + 1e-3)
        assert float((pg[0].cpu().double() - mean).abs().max()) <= 1e-5
        assert float((pg[1].cpu().double() - rstd).abs().max()) <= 1e-4 * float(rstd.abs().max())
    with ops.configure(split16=False, clear_pack_cache=True):
        ref, _ = ops.conv3d_transpose_s2(x.to(cuda), ('up32', cin), w.numpy(), want_stats=True, groups=G)
    assert float((ref - got).abs().max()) <= 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize('G,shape,cin,cout', [(1, (9, 17, 33), 16, 8), (2, (4, 8, 16), 32, 8), (3, (5, 9, 18), 64, 8),
                                             (2, (12, 24, 40), 16, 8), (1, (9, 17, 33), 32, 16), (3, (5, 9, 18), 16, 16),
                                             (2, (8, 12, 40), 64, 16)])
@pytest.mark.parametrize('split', [True, False])
def test_deconv_up_against_torch_and_the_class_fused_form(cuda, G, shape, cin, cout, split):
    """The 8- / 16-channel transposed convolutions (conv_b*_6_0, conv_b*_5_0, reference cnn_wrapper/network.py:510-550) on their own
    kernels (csrc/deconv_up_b.hip: split-fp16 operands, default where its weights fit LDS; csrc/deconv_up.hip: fp32 MFMA): against tf.layers.conv3d_transpose restated with torch (tolerance 2e-5 of the output scale: a
    different accumulation order), and its statistics / grouped form against the separate calls."""
    from oracle import tf_ops as T
    from atvsnet_amd import ops
    x = _rand((G,) + shape + (cin,), 21)
    w = _rand((3, 3, 3, cout, cin), 22) * 0.2
    with ops.configure(split16=split, clear_pack_cache=True):
        _deconv_up_checks(cuda, ops, T, x, w, G, shape, cin, cout)


def _deconv_up_checks(cuda, ops, T, x, w, G, shape, cin, cout):
    got, st = ops.conv3d_transpose_s2(x.to(cuda), ('up', cin, cout), w.numpy(), want_stats=True, groups=G)
    assert st.cpad == 16 and st.groups == G and st.fold == 1
    D, H, W = shape
    assert tuple(got.shape) == (G, 2 * D, 2 * H, 2 * W, cout)
    params = ops.bn_params(st, cout, got)
    for g in range(G):
        want = T.conv3d_transpose_same(x[g:g + 1], w, 2)[0]
        err = float((got[g].cpu() - want).abs().max())
        assert err <= 2e-5 * float(want.abs().max()) + 1e-6, (g, err)
        one, st1 = ops.conv3d_transpose_s2(x[g].to(cuda), ('up', cin, cout), w.numpy(), want_stats=True)
        assert torch.equal(one, got[g])
        p1 = ops.bn_params(st1, cout, one)
        pg = params[g] if G > 1 else params
        assert float((pg - p1).abs().max()) <= 1e-6 * float(p1.abs().max())
        mean = want.reshape(-1, cout).double().mean(0)
        assert float((pg[0].cpu().double() - mean).abs().max()) <= 1e-5
    with ops.configure(deconv_up=False):
        ref, _ = ops.conv3d_transpose_s2(x.to(cuda), ('up', cin, cout), w.numpy(), want_stats=True, groups=G)
    assert float((ref - got).abs().max()) <= 2e-5 * float(ref.abs().max())
    relu = ops.conv3d_transpose_s2(x.to(cuda), ('up', cin, cout), w.numpy(), relu=True, groups=G)
    assert torch.equal(relu, torch.clamp(got, min=0))


@pytest.mark.parametrize('G,shape,cin,cout', [(1, (9, 17, 33), 16, 16), (2, (4, 8, 16), 32, 16), (3, (5, 9, 18), 16, 16),
                                             (2, (12, 24, 40), 32, 16), (1, (9, 17, 33), 8, 16), (3, (5, 9, 18), 8, 16),
                                             (2, (12, 24, 40), 8, 16), (1, (9, 17, 33), 32, 32), (3, (5, 9, 18), 16, 32),
                                             (2, (12, 24, 40), 64, 32)])
def test_conv_c16_against_torch_and_the_tiled_kernel(cuda, G, shape, cin, cout):
    """The 16 / 32 -> 16 channel 3x3x3 convolutions (conv_b*_1_1, reference cnn_wrapper/network.py:165-215) on their own
    kernel (csrc/conv_c16.hip): against tf.nn.conv3d restated with torch (2e-5 of the output scale: another accumulation
    order), grouped form == separate calls bit for bit, moments, bias + ReLU, channel-slice output."""
    from oracle import tf_ops as T
    from atvsnet_amd import ops
    x = _rand((G,) + shape + (cin,), 31)
    w = _rand((3, 3, 3, cin, cout), 32) * 0.2
    got, st = ops.conv(x.to(cuda), ('c16', cin, cout), w.numpy(), want_stats=True, groups=G)
    assert st.cpad == cout and st.groups == G
    params = ops.bn_params(st, cout, got)
    for g in range(G):
        want = T.conv(x[g:g + 1], w, 1, 'SAME')[0]
        err = float((got[g].cpu() - want).abs().max())
        assert err <= 2e-5 * float(want.abs().max()) + 1e-6, (g, err)
        one, st1 = ops.conv(x[g].to(cuda), ('c16', cin, cout), w.numpy(), want_stats=True)
        assert torch.equal(one, got[g])
        pg = params[g] if G > 1 else params
        assert float((pg - ops.bn_params(st1, cout, one)).abs().max()) <= 1e-6 * float(pg.abs().max())
        assert float((pg[0].cpu().double() - want.reshape(-1, cout).double().mean(0)).abs().max()) <= 1e-5
    with ops.configure(conv_c16=False):
        ref = ops.conv(x.to(cuda), ('c16', cin, cout), w.numpy(), groups=G)
    assert float((ref - got).abs().max()) <= 2e-5 * float(ref.abs().max())
    b = _rand((cout,), 33)
    buf = torch.full((G,) + shape + (16 + cout,), -7.0, device=cuda)
    ops.conv(x.to(cuda), ('c16', cin, cout), w.numpy(), bias=b.to(cuda), relu=True, out=buf, y_coff=16, groups=G)
    assert torch.equal(buf[..., 16:], torch.clamp(got + b.to(cuda), min=0)) or \
        float((buf[..., 16:] - torch.clamp(got + b.to(cuda), min=0)).abs().max()) <= 1e-6
    assert torch.all(buf[..., :16] == -7.0)


@pytest.mark.parametrize('shape,G,bias', [((9, 13, 35), 3, True), ((21, 50, 70), 5, True), ((4, 8, 32), 1, False),
                                          ((2, 3, 5), 2, True)])
def test_refine_stems_planar_is_bitwise_the_channel_last_form(cuda, shape, G, bias):
    """refine_stems_kernel<PLANAR>: planes 1..3 of the chunk-planar concat = channels 8..31 of the channel-last rows, bit for
    bit, with the same statistics; plane 0 (the photo stem's) and the padding between the planes are not touched."""
    from atvsnet_amd import ops
    D, H, W = shape
    photo = _rand((G, D, H, W, 8), 41).to(cuda)
    geo, prob, hull = (_rand((G, D, H, W, c), 42 + i).to(cuda) for i, c in enumerate((2, 1, 1)))
    pb = _rand((G, H, W, 24), 45).to(cuda) if bias else None
    wg, wp, wh = _rand((3, 3, 3, 2, 8), 46) * 0.3, _rand((3, 3, 3, 1, 8), 47) * 0.3, _rand((3, 3, 3, 1, 8), 48) * 0.3
    key = ('stems-planar-test', shape, G, bias)
    cl, st = ops.refine_stems(photo, geo, pb, prob, hull, key, wg.numpy(), wp.numpy(), wh.numpy())
    ps = ops.planar_stride(D, H, W)
    buf = torch.full((G, 4, ps), 7.25, device=cuda)
    out, st2 = ops.refine_stems(None, geo, pb, prob, hull, key, wg.numpy(), wp.numpy(), wh.numpy(), planar_out=buf)
    assert out is buf
    planes = ops.planar_view(buf, D, H, W)                       # (G,4,D,H,W,8)
    for c in range(3):
        assert torch.equal(planes[:, 1 + c], cl[..., 8 + 8 * c:16 + 8 * c])
    assert bool((planes[:, 0] == 7.25).all()) and bool((buf[..., D * H * W * 8:] == 7.25).all())
    assert torch.equal(st.partial, st2.partial)


@pytest.mark.parametrize('shape,G', [((8, 16, 40), 2), ((6, 9, 33), 1), ((5, 24, 70), 3)])
def test_photo_stem_into_plane_and_planar_concat_consumer_are_bitwise(cuda, shape, G):
    """The photo stem written straight into plane 0 of the chunk-planar concat (atvs_conv_xb_f32 with y_group_stride) equals
    conv_split's dense output, and the x-pair launch that normalises the concat on load gives the same bits from the four
    planes as from the channel-last rows (reference cnn_wrapper/atvsnet.py:300-316)."""
    from atvsnet_amd import ops
    if ops._xkind() != 'xb':
        pytest.skip('the planar concat belongs to the split-fp16 x-pair kernel')
    D, H, W = shape
    chan = 16
    photo = ops.SplitVolume(_rand((G, D, H, W, chan), 20).to(cuda), _rand((G, H, W, 2 * chan), 21).to(cuda),
                            [('v', i) for i in range(chan)] + [('c', i) for i in range(2 * chan)])
    w = (_rand((3, 3, 3, 3 * chan, 8), 22) * 0.1).numpy()
    y, st = ops.conv_split(photo, 'photo-plane-test', w, want_stats=True)
    ps = ops.planar_stride(D, H, W)
    buf = torch.full((G, 4, ps), -3.5, device=cuda)
    st2 = ops.conv_split_into_plane(photo, 'photo-plane-test', w, buf, 0, (D, H, W))
    planes = ops.planar_view(buf, D, H, W)
    assert torch.equal(planes[:, 0], y.reshape(G, D, H, W, 8))
    assert bool((planes[:, 1:] == -3.5).all()) and torch.equal(st.partial, st2.partial)
    # the consumer: 32-channel concat, pending batch norm + ReLU, from planes and from rows
    rest = _rand((G, D, H, W, 24), 23).to(cuda)
    for c in range(3):
        planes[:, 1 + c] = rest[..., 8 * c:8 * c + 8]
    rows = torch.cat([y.reshape(G, D, H, W, 8), rest], -1).contiguous()
    params = torch.stack([_rand((G, 32), 8) * 0.3, _rand((G, 32), 9).abs() + 0.5, _rand((G, 32), 10) + 0.7], 1).to(cuda)
    params = params.contiguous() if G > 1 else params[0].contiguous()
    w8, w16 = (_rand((3, 3, 3, 32, 8), 5) * 0.1).numpy(), (_rand((3, 3, 3, 32, 16), 6) * 0.1).numpy()
    lazy_p = ops.PendingBN(buf, params, True, planar=(D, H, W))
    assert lazy_p.shape == (G, D, H, W, 32) and lazy_p.dim() == 5
    if not ops.siblings_ok((D, H, W), 32, 8, 16):
        assert torch.equal(lazy_p.materialize(), ops.bn_apply(rows.clone(), params, True))
        return
    (a, sa), (a2, sa2) = ops.conv_siblings(lazy_p, 'pc8', w8, 'pc16', w16, groups=G)
    assert lazy_p._final is None
    (b, sb), (b2, sb2) = ops.conv_siblings(ops.PendingBN(rows.clone(), params, True), 'pc8', w8, 'pc16', w16, groups=G)
    assert torch.equal(a, b) and torch.equal(a2, b2)
    assert torch.equal(sa.partial, sb.partial) and torch.equal(sa2.partial, sb2.partial)
    assert torch.equal(lazy_p.materialize(), ops.bn_apply(rows.clone(), params, True))     # any other consumer: rows


@pytest.mark.parametrize('D,h,w', [(6, 16, 40), (5, 9, 33)])
def test_photo_volume_in_pieces_feeds_the_photo_stem_bitwise(cuda, D, h, w):
    """The refinement's photo volume |warp_d(view) - ref| * mask (reference model.py:270-280) written by the warp as chunk-planar
    fp16 pieces (atvs_warp_planes mode 1, planar, pieces) and staged by the photo stem's launch with LDS-DMA: the pieces are
    bit for bit the split of the channel-last volume, and the stem's plane of the concat and its statistics are the same bits."""
    from atvsnet_amd import ops
    from oracle import homography_warping as G_
    from oracle import model as OM
    if ops._xkind() != 'xb':
        pytest.skip('pieces belong to the split-operand x-pair kernel')
    import os
    import numpy as np
    ops.clear_pack_cache()
    chan = 16
    gd = os.path.join(os.path.dirname(__file__), 'golden')
    cams = torch.stack([torch.from_numpy(np.load(os.path.join(gd, 'example2_%d_cam.npy' % i))) for i in range(2)])[None].float()
    ds, di = OM.depth_start_interval(cams)
    Hm = G_.get_homographies(cams[:, 0], cams[:, 1], D, ds, di)[0].to(cuda)
    vf, rf = (_rand((h, w, chan), 40) * 2.0).to(cuda), (_rand((h, w, chan), 41) * 2.0).to(cuda)
    cl = ops.warp_planes(vf, Hm, mode=1, ref=rf)                                              # (D,h,w,16)
    pc = ops.warp_planes(vf, Hm, mode=1, ref=rf, planar=True, pieces=True)                    # (2, planar_stride)
    n = D * h * w * 8
    x = cl.cpu().numpy().reshape(D, h, w, 2, 8).transpose(3, 0, 1, 2, 4)                      # (chunk, D, h, w, 8)
    got = pc[:, :n].contiguous().view(torch.float16).cpu().numpy().reshape(2, 2, D, h, w, 8)
    h0 = x.astype(np.float16)
    h1 = ((x - h0.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    assert np.array_equal(got[:, 0].view(np.uint16), h0.view(np.uint16)) and np.array_equal(got[:, 1].view(np.uint16), h1.view(np.uint16))
    const = _rand((1, h, w, 2 * chan), 42).to(cuda)
    cmap = [('v', i) for i in range(chan)] + [('c', i) for i in range(2 * chan)]
    sv_cl = ops.SplitVolume(cl[None], const, cmap)
    sv_pc = ops.SplitVolume(pc[None], const, cmap, planar=(D, h, w), pieces=True)
    wgt = (_rand((3, 3, 3, 3 * chan, 8), 43) * 0.1).numpy()
    ps = ops.planar_stride(D, h, w)
    buf_a, buf_b = torch.full((1, 4, ps), -3.5, device=cuda), torch.full((1, 4, ps), -3.5, device=cuda)
    st_a = ops.conv_split_into_plane(sv_cl, 'photo-pieces-test', wgt, buf_a, 0, (D, h, w))
    st_b = ops.conv_split_into_plane(sv_pc, 'photo-pieces-test', wgt, buf_b, 0, (D, h, w))
    assert torch.equal(buf_a, buf_b) and torch.equal(st_a.partial, st_b.partial)
    ops.clear_pack_cache()


def test_refinement_net_planar_concat_equals_channel_last_concat(cuda, weights):
    """CostVolRefineNet with its concat as four dense planes (default) against the channel-last concat
    (ops.configure(planar_concat=False)): every output bit for bit, and the planar form really ran."""
    from atvsnet_amd import ops
    from atvsnet_amd.cnn_wrapper.atvsnet import CostVolRefineNet
    if ops._xkind() != 'xb':
        pytest.skip('the planar concat belongs to the split-fp16 x-pair kernel')
    G, chan, shape = 2, 16, (16, 32, 48)
    photo = ops.SplitVolume(_rand((G,) + shape + (chan,), 20).to(cuda), _rand((G, 32, 48, 2 * chan), 21).to(cuda),
                            [('v', i) for i in range(chan)] + [('c', i) for i in range(2 * chan)])
    geo = ops.SplitVolume(_rand((G,) + shape + (2,), 30).to(cuda), _rand((G, 32, 48, 2), 31).to(cuda),
                          [('v', 0)] + [('v', 1)] * chan + [('c', 0), ('c', 1)])
    prob, hull = _rand((G,) + shape + (1,), 40).to(cuda), _rand((G,) + shape + (1,), 41).to(cuda)
    res = {}
    for flag in (True, False):
        with ops.configure(planar_concat=flag):
            ref = CostVolRefineNet({'photo_group': photo, 'geo_group': geo, 'prob_vol': prob, 'vis_hull': hull},
                                   is_training=True, independent_samples=True)
            concat = ref.layers['global_refine_concat']
            assert isinstance(concat, ops.PendingBN) and bool(concat.planar) == flag
            res[flag] = [ref.get_output().clone(), ref.get_output_by_name('global_refine_3dconv6_1').clone(),
                         ref.get_output_by_name('global_refine_photo_3dconv').clone(),
                         ref.get_output_by_name('global_refine_concat').clone()]
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)


@pytest.mark.parametrize('G,shape,nterms', [(2, (6, 10, 20), 3), (1, (9, 17, 33), 2), (3, (4, 4, 16), 3), (2, (5, 7, 37), 2),
                                            # the two-role kernel's pipeline: three stages per workgroup, a single tile, idle workgroups
                                            (1, (32, 32, 160), 3), (1, (1, 2, 3), 2), (8, (8, 8, 32), 3)])
def test_deconv_sums_its_inputs_on_load_bitwise(cuda, G, shape, nterms):
    """The full-resolution decoder conv_b*_6_0 (16 -> 8, reference cnn_wrapper/atvsnet.py:156-158,186-188) with its skip sum formed
    while the halo is staged (atvs_deconv_up_b_sum_f32) against the same layer behind the bn_add pass it replaces: pending batch
    norms with and without ReLU, a finished tensor among the terms, two and three terms, ragged tiles -- every bit and the
    moments; and the sum really was not formed."""
    from atvsnet_amd import ops
    cin, cout = 16, 8
    w = _rand((3, 3, 3, cout, cin), 5) * 0.2

    def terms(seed):
        ts = []
        for k in range(nterms):
            raw = _rand((G,) + shape + (cin,), seed + k).to(cuda)
            if k == 1 and nterms == 3:
                ts.append(raw)                                        # a finished tensor
            else:
                par = torch.stack([_rand((G, cin), seed + 10 + k) * 0.1, _rand((G, cin), seed + 30 + k).abs() + 0.5,
                                   _rand((G, cin), seed + 20 + k) * 0.1], 1).to(cuda).contiguous()
                ts.append(ops.PendingBN(raw, par, relu=(k != 2)))
        return ts
    s_on = ops.PendingSum(terms(40))
    assert ops.deconv_sum_ok(s_on, cout, G)
    got, st = ops.conv3d_transpose_s2(s_on, ('up-sum', cin), w.numpy(), want_stats=True, groups=G)
    assert s_on._final is None                                        # never materialised
    with ops.configure(prologue=False):
        s_off = ops.PendingSum(terms(40))
        assert not ops.deconv_sum_ok(s_off, cout, G)
        ref, st_ref = ops.conv3d_transpose_s2(s_off, ('up-sum', cin), w.numpy(), want_stats=True, groups=G)
        assert s_off._final is not None
    assert torch.equal(got, ref)
    assert torch.equal(ops.bn_params(st, cout, got), ops.bn_params(st_ref, cout, ref))


@pytest.mark.parametrize('G,shape,nterms', [(2, (6, 10, 20), 3), (1, (9, 17, 33), 2), (2, (12, 16, 48), 3)])
def test_deconv_sum_on_load_against_the_oracle(cuda, G, shape, nterms):
    """The same fused launch against the CPU ORACLE, not against other HIP launches: tf.layers.conv3d_transpose (oracle/tf_ops.py,
    reference cnn_wrapper/network.py:510-550) of the skip sum relu(bn(a)) + c + bn(b) evaluated in float64 from the same parameters
    (reference cnn_wrapper/atvsnet.py:156-158: conv_b*_5_1 = add(conv_b*_5_0, conv_b*_1_1 [, conv_b0_1_1]) -> conv_b*_6_0).
    Tolerance 2e-5 of the output scale (the split-operand products, a different accumulation order); the moments against the
    float64 mean / variance of the oracle's output."""
    from oracle import tf_ops as T
    from atvsnet_amd import ops
    cin, cout = 16, 8
    w = _rand((3, 3, 3, cout, cin), 5) * 0.2
    raws = [_rand((G,) + shape + (cin,), 60 + k) for k in range(nterms)]
    pars = [torch.stack([_rand((G, cin), 70 + k) * 0.1, _rand((G, cin), 90 + k).abs() + 0.5, _rand((G, cin), 80 + k) * 0.1], 1).contiguous()
            for k in range(nterms)]
    finished = 1 if nterms == 3 else None                 # a finished tensor among the terms
    ts = [raws[k].to(cuda) if k == finished else ops.PendingBN(raws[k].to(cuda), pars[k].to(cuda), relu=(k != 2)) for k in range(nterms)]
    s_on = ops.PendingSum(ts)
    assert ops.deconv_sum_ok(s_on, cout, G)
    got, st = ops.conv3d_transpose_s2(s_on, ('up-sum-oracle', cin), w.numpy(), want_stats=True, groups=G)
    assert s_on._final is None
    params = ops.bn_params(st, cout, got)
    for g in range(G):
        x = torch.zeros(shape + (cin,), dtype=torch.float64)
        for k in range(nterms):
            v = raws[k][g].double()
            if k != finished:
                m, sc, be = pars[k][g, 0].double(), pars[k][g, 1].double(), pars[k][g, 2].double()
                v = (v - m) * sc + be
                if k != 2:
                    v = torch.clamp(v, min=0)
            x = x + v
        want = T.conv3d_transpose_same(x[None], w.double(), 2)[0]
        scale = float(want.abs().max())
        assert float((got[g].cpu().double() - want).abs().max()) <= 2e-5 * scale, g
        flat = want.reshape(-1, cout)
        pg = params[g] if G > 1 else params
        assert float((pg[0].cpu().double() - flat.mean(0)).abs().max()) <= 1e-5 * scale
        rstd = 1.0 / torch.sqrt(flat.var(0, unbiased=False) + 1e-3)
        assert float((pg[1].cpu().double() / rstd - 1).abs().max()) <= 1e-5


def _pending_seeded(G, shape, cin, seed, relu, cuda):
    from atvsnet_amd import ops
    raw = _rand((G,) + shape + (cin,), seed).to(cuda)
    par = torch.stack([_rand((G, cin), seed + 10) * 0.1, _rand((G, cin), seed + 30).abs() + 0.5,
                       _rand((G, cin), seed + 20) * 0.1], 1).to(cuda).contiguous()
    return ops.PendingBN(raw, par, relu=relu)


def _term64(t):
    """A term of a skip sum in float64 on the CPU: a finished tensor, or relu?((raw - mean) * rstd + beta) of a PendingBN."""
    from atvsnet_amd import ops
    if not isinstance(t, ops.PendingBN):
        return t.cpu().double()
    raw, par = t.raw.cpu().double(), t.params.cpu().double()
    G, C = raw.shape[0], raw.shape[-1]
    bc = (G,) + (1,) * (raw.dim() - 2) + (C,)
    v = (raw - par[:, 0].reshape(bc)) * par[:, 1].reshape(bc) + par[:, 2].reshape(bc)
    return torch.clamp(v, min=0) if t.relu else v


@pytest.mark.parametrize('G,shape,cin,cout,stride,relu', [
    (2, (8, 16, 32), 16, 16, 1, True), (1, (9, 11, 19), 16, 16, 1, False), (3, (5, 9, 13), 16, 16, 1, True),    # conv_c16b
    (2, (8, 16, 24), 32, 32, 1, True), (1, (6, 9, 17), 64, 64, 1, True), (2, (5, 8, 12), 16, 32, 1, False),     # conv3d_b
    (2, (16, 32, 48), 16, 32, 2, True), (1, (9, 17, 33), 32, 64, 2, True), (3, (7, 10, 18), 16, 32, 2, False)])  # conv3d_s2b
def test_conv3d_normalises_its_input_on_load_bitwise(cuda, G, shape, cin, cout, stride, relu):
    """The U-Nets' encoders conv_b*_{1,2,3}_0 hand their consumers (conv_b*_{2,3}_0 stride 2, conv_b*_{1,2,3}_1; reference
    cnn_wrapper/atvsnet.py:10-26) the RAW output + moments: conv_c16b / conv3d_b / conv3d_s2b apply the batch norm (+ ReLU) while
    they stage the halo (atvs_*_norm_f32, atvs_conv_c16b_sum_f32 with one term) -- against the same layer behind the bn_apply
    pass it replaces: every bit and the moments, several samples per launch, ragged tiles (the zero padding must stay zero)."""
    from atvsnet_amd import ops
    w = (_rand((3, 3, 3, cin, cout), 5) * 0.1).numpy()
    pend = _pending_seeded(G, shape, cin, 60, relu, cuda)
    assert ops.norm_on_load_3d_ok(pend, 3, cout, stride)
    x, pro = pend.prologue()
    got, st = ops.conv(x, ('norm3d', cin, cout, stride), w, stride=stride, want_stats=True, groups=G, in_params=pro[1],
                       in_relu=pro[3])
    assert pend._final is None                                        # never materialised
    with ops.configure(sum_on_load=False):
        assert not ops.norm_on_load_3d_ok(pend, 3, cout, stride)
        ref, st_ref = ops.conv(x, ('norm3d', cin, cout, stride), w, stride=stride, want_stats=True, groups=G, in_params=pro[1],
                               in_relu=pro[3])                         # ops.conv's own fallback: bn_apply, then the plain kernel
    want, st_w = ops.conv(pend.materialize(), ('norm3d', cin, cout, stride), w, stride=stride, want_stats=True, groups=G)
    assert torch.equal(ref, want)
    assert torch.equal(got, want)
    assert torch.equal(ops.bn_params(st, cout, got), ops.bn_params(st_w, cout, want))
    # ... and against the ORACLE's convolution of the float64 batch norm (not only against other HIP launches)
    from oracle import tf_ops as T
    want64 = T.conv(_term64(_pending_seeded(G, shape, cin, 60, relu, cuda)), torch.from_numpy(w).double(), stride, 'SAME')      # (materialize() normalised pend.raw in place)
    assert float((got.cpu().double() - want64).abs().max()) <= 3e-5 * float(want64.abs().max())


@pytest.mark.parametrize('G,shape,forms', [(2, (8, 16, 32), 'pp'), (1, (9, 11, 19), 'pd'), (3, (5, 9, 13), 'dp'), (2, (4, 8, 16), 'pp')])
def test_conv_c16b_sums_its_inputs_on_load_bitwise(cuda, G, shape, forms):
    """conv_b{1,2}_1_1 behind conv_b{1,2}_1_1_concat = conv_b*_1_0 + conv_b{0,1}_5_0 (reference cnn_wrapper/atvsnet.py:45-46,
    75-76): the 16 -> 16 kernel forms the sum of two pending batch norms (or a pending one and a finished tensor) while it
    stages the halo (atvs_conv_c16b_sum_f32) -- against bn_add followed by the plain kernel, every bit and the moments."""
    from atvsnet_amd import ops
    w = (_rand((3, 3, 3, 16, 16), 7) * 0.1).numpy()

    def items(seed):
        out = []
        for k, f in enumerate(forms):
            out.append(_pending_seeded(G, shape, 16, seed + 100 * k, k == 0, cuda) if f == 'p'
                       else _rand((G,) + shape + (16,), seed + 100 * k).to(cuda))
        return out
    s_on = ops.PendingSum(items(80))
    assert ops.norm_on_load_3d_ok(s_on, 3, 16, 1)
    x, (x1, pa, pb, ra, rb) = s_on.prologue()
    got, st = ops.conv(x, ('sum16',), w, want_stats=True, groups=G, in_params=pa, in_relu=ra, in_sum=(x1, pb, rb))
    assert s_on._final is None
    want, st_w = ops.conv(ops.PendingSum(items(80)).materialize(), ('sum16',), w, want_stats=True, groups=G)
    assert torch.equal(got, want)
    assert torch.equal(ops.bn_params(st, 16, got), ops.bn_params(st_w, 16, want))
    # ... and against the ORACLE's convolution of the float64 sum (not only against other HIP launches)
    from oracle import tf_ops as T
    its = items(80)
    want64 = T.conv(_term64(its[0]) + _term64(its[1]), torch.from_numpy(w).double(), 1, 'SAME')
    assert float((got.cpu().double() - want64).abs().max()) <= 3e-5 * float(want64.abs().max())
    with ops.configure(sum_on_load=False):                             # ops.conv's own fallback
        ref = ops.conv(x, ('sum16',), w, groups=G, in_params=pa, in_relu=ra, in_sum=(x1, pb, rb))
    assert torch.equal(ref, want)


@pytest.mark.parametrize('G,shape', [(4, (6, 10, 20)), (1, (5, 7, 9)), (3, (4, 4, 16))])
def test_skip_add_with_the_residual_base_in_one_pass_bitwise(cuda, G, shape):
    """global_refine_3dconv6_1 (two pending batch norms) and refined_cost = filtered_cost + cost_residual of every source view
    (reference atvsnet/model.py:438-439) from ONE pass (atvs_bn_add_plus) -- the bits of bn_add and of add_n per sample."""
    from atvsnet_amd import ops
    items = [_pending_seeded(G, shape, 8, 300 + 100 * k, True, cuda) for k in range(2)]
    base = _rand(shape + (8,), 9).to(cuda)
    y, y2 = ops.bn_add(items, plus=base)
    want = ops.bn_add([_pending_seeded(G, shape, 8, 300 + 100 * k, True, cuda) for k in range(2)])
    assert torch.equal(y, want)
    for g in range(G):
        assert torch.equal(y2[g], ops.add_n([base, want[g]]))


@pytest.mark.parametrize('nv,shape', [(4, (16, 24, 40)), (2, (9, 13, 35)), (3, (4, 8, 16)), (1, (6, 10, 20)), (4, (5, 9, 17)),
                                      (5, (12, 16, 48)), (8, (8, 24, 33)), (6, (5, 13, 16)), (7, (9, 12, 20)), (4, (64, 48, 64))])
def test_aanet_module_in_one_launch_bitwise(cuda, weights, nv, shape):
    """aanet_b.hip: the shared | unique score convolutions of every view and the cross-view softmax + weighted sum (reference
    cnn_wrapper/network.py:282-351,378-408) as ONE launch (multiplying + staging / combining wavefronts, [S|R] handed over in LDS)
    -- against the two-launch form (conv_c16b per view + aanet_combine) bit for bit up to 4 views (within 2e-6 of the maximum for the
    running softmax of 5 .. 8 views), and against the oracle's attention_aggregation; ragged tiles, 1 to 8 views (configs[3] has 8 sources), workgroups with two tiles (64 x 48 x 64 = 384 tiles on
    256 workgroups) and with fewer tiles than workgroups."""
    from atvsnet_amd import ops
    from atvsnet_amd.cnn_wrapper.atvsnet import AttAggregation_keepchannel
    from oracle import nets
    xs = [_rand(shape + (8,), 70 + n) for n in range(nv)]
    stacked = torch.stack(xs, 0).to(cuda)
    assert ops.aanet_fused_ok([stacked[n] for n in range(nv)])
    outs = {}
    for fused in (True, False):
        with ops.configure(aanet_fused=fused):
            net = AttAggregation_keepchannel({'data': stacked}, is_training=True)
            outs[fused] = net.get_output().clone()
    want = nets.attention_aggregation(torch.stack(xs, -1)[None], weights, 'attention_aggregate')
    scale = float(want.abs().max())
    if nv <= 4:
        assert torch.equal(outs[True], outs[False])
    else:
        # 5 .. 8 views: the running softmax over views without the (shift-invariant) S_sum term -- the reference's literal formula up
        # to the rounding of (R_n - S_n) + S_sum, not its bits (aanet_b.hip); a tenth of the oracle bar against the two-launch form
        d = float((outs[True] - outs[False]).abs().max())
        print('%d views: one launch vs two launches max |diff| %.2e of %.2e' % (nv, d, scale))
        assert d <= 2e-6 * scale
    assert float((outs[True].cpu() - want).abs().max()) <= 2e-5 * scale
    assert float((outs[False].cpu() - want).abs().max()) <= 2e-5 * scale
