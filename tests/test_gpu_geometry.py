"""-m gpu: geometry + soft-argmin HIP kernels (through the C-ABI) vs the CPU oracle.

Geometry is integer/index work plus fixed-order fp32 arithmetic -> BIT-EXACT bar
(built with -ffp-contract=off; oracle uses the same operation order).
Soft-argmin uses expf -> tolerance 2e-6 relative.
"""
import numpy as np
import pytest
import torch

from oracle import homography_warping as G
from oracle import model as OM

pytestmark = pytest.mark.gpu


def _example_cams(views=2):
    from atvsnet_amd import synthetic
    cams = synthetic.make_cams(views, 128, 160, 32)
    return torch.from_numpy(cams)[None]          # (1,N,2,4,4)


def _golden_cams():
    import os
    d = os.path.join(os.path.dirname(__file__), 'golden')
    c = np.stack([np.load(os.path.join(d, 'example2_%d_cam.npy' % i)) for i in range(2)]).astype(np.float32)
    return torch.from_numpy(c)[None]


@pytest.mark.parametrize('which', ['synthetic', 'example2'])
@pytest.mark.parametrize('D', [1, 32, 192])
def test_homographies_bit_exact(cuda, which, D):
    from atvsnet_amd import ops
    cams = _example_cams() if which == 'synthetic' else _golden_cams()
    ds, di = OM.depth_start_interval(cams)
    for a, b in ((0, 1), (1, 0)):
        want = G.get_homographies(cams[:, a], cams[:, b], D, ds, di)[0]
        got = ops.get_homographies(cams[0, a].to(cuda).contiguous(), cams[0, b].to(cuda).contiguous(),
                                   ds.to(cuda), di.to(cuda), D).cpu()
        assert torch.equal(got, want)


def _feat(h, w, C, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(1, h, w, C, generator=g)


@pytest.mark.parametrize('h,w,C,D', [(32, 40, 32, 32), (30, 36, 16, 8), (17, 23, 4, 5), (32, 40, 1, 16), (9, 11, 3, 4)])
def test_warp_planes_bit_exact(cuda, h, w, C, D):
    from atvsnet_amd import ops
    cams = _example_cams()
    # intrinsics are for 32x40; other sizes simply sample more out-of-range pixels (exercises the masks)
    ds, di = OM.depth_start_interval(cams)
    H = G.get_homographies(cams[:, 0], cams[:, 1], D, ds, di)
    src = _feat(h, w, C, 1)
    want = torch.stack([G.homography_warping(src, H[:, d])[0] for d in range(D)])
    wm = torch.stack([G.homography_warping(src, H[:, d], output_mask=True)[1][0, ..., 0] for d in range(D)])
    got, mask = ops.warp_planes(src[0].to(cuda), H[0].to(cuda), want_mask=True)
    assert torch.equal(got.cpu(), want)
    assert torch.equal(mask.cpu(), wm.to(torch.float32))
    assert 0.05 < wm.float().mean() < 1.0          # both valid and invalid samples are present


@pytest.mark.parametrize('h,w,C,D', [(32, 40, 32, 8), (17, 23, 4, 5), (32, 40, 1, 16), (9, 11, 3, 4)])
def test_homography_warping_nearest_bit_exact(cuda, h, w, C, D):
    """homography_warping(method='nearest') through the reference-named face (reference homography_warping.py:45-56,
    230-271): tf.round sampling, out-of-range pixels read pixel (0,0) and are NOT zeroed (quirk C4)."""
    from atvsnet_amd.atvsnet import homography_warping as HW
    cams = _example_cams()
    ds, di = OM.depth_start_interval(cams)
    H = G.get_homographies(cams[:, 0], cams[:, 1], D, ds, di)
    src = _feat(h, w, C, 7)
    want = torch.stack([G.homography_warping(src, H[:, d], method='nearest')[0] for d in range(D)])
    wm = torch.stack([G.homography_warping(src, H[:, d], method='nearest', output_mask=True)[1][0] for d in range(D)])
    got, mask = HW.homography_warping(src.to(cuda), H.to(cuda), method='nearest', output_mask=True)
    assert got.shape == (1, D, h, w, C) and mask.shape == (1, D, h, w, 1) and mask.dtype == torch.bool
    assert torch.equal(got[0].cpu(), want)
    assert torch.equal(mask[0].cpu(), wm)
    inv = ~wm[..., 0]
    assert inv.any() and torch.equal(want[inv], src[0, 0, 0].expand_as(want[inv]))      # un-masked pixel (0,0)
    # single homography (B,3,3), no mask: the reference's plain call form
    one = HW.homography_warping(src.to(cuda), H[:, 1].to(cuda), method='nearest')
    assert torch.equal(one[0].cpu(), want[1])
    with pytest.raises(ValueError):
        HW.homography_warping(src.to(cuda), H[:, 1].to(cuda), method='cubic')


def test_cost_volume_bit_exact(cuda):
    from atvsnet_amd import ops
    cams = _example_cams()
    ds, di = OM.depth_start_interval(cams)
    D, h, w, C = 32, 32, 40, 32
    rf, vf = _feat(h, w, C, 2), _feat(h, w, C, 3)
    want = OM.build_cost_volume(rf, vf, cams, D, ds, di, 0, 1)[0]
    H = ops.get_homographies(cams[0, 0].to(cuda), cams[0, 1].to(cuda), ds.to(cuda), di.to(cuda), D)
    got = ops.build_cost_volume(rf[0].to(cuda), vf[0].to(cuda), H)
    assert torch.equal(got.cpu(), want)
    # reverse direction (model.py:413, quirk C11)
    want = OM.build_cost_volume(vf, rf, cams, D, ds, di, 1, 0)[0]
    H = ops.get_homographies(cams[0, 1].to(cuda), cams[0, 0].to(cuda), ds.to(cuda), di.to(cuda), D)
    got = ops.build_cost_volume(vf[0].to(cuda), rf[0].to(cuda), H)
    assert torch.equal(got.cpu(), want)


def test_identity_camera_warp(cuda):
    """Known answer: identity pose -> identity away from the last row/col, which go to 0."""
    from atvsnet_amd import ops
    cams = _example_cams()
    ds, di = OM.depth_start_interval(cams)
    src = _feat(12, 20, 8, 5)[0]
    H = ops.get_homographies(cams[0, 0].to(cuda), cams[0, 0].to(cuda), ds.to(cuda), di.to(cuda), 3)
    out = ops.warp_planes(src.to(cuda), H).cpu()
    for d in range(3):
        assert torch.allclose(out[d, :-1, :-1], src[:-1, :-1], atol=1e-5)
        assert torch.all(out[d, -1] == 0) and torch.all(out[d, :, -1] == 0)


def test_refinement_volumes_bit_exact(cuda):
    """photo / geo / visual-hull volumes and the D-constant maps of model.py:270-336."""
    from atvsnet_amd import ops
    cams = _example_cams()
    ds, di = OM.depth_start_interval(cams)
    D, h, w = 16, 32, 40
    g = torch.Generator().manual_seed(7)
    ref_f, view_f = _feat(h, w, 16, 11), _feat(h, w, 16, 12)
    d_ref = 0.05 + 0.3 * torch.rand(1, h, w, 1, generator=g)
    d_view = 0.05 + 0.3 * torch.rand(1, h, w, 1, generator=g)
    d_view[0, :3, :5] = 0.0                       # invalid depths exercise the 1e-10 clip / mask
    init = torch.stack([d_ref, d_view], 1)
    prob = torch.randn(1, D, h, w, generator=g)
    want = OM.refinement_inputs(init, cams, D, ds, di, None, prob, None, 0, 1, shallow=(ref_f, view_f))

    c = lambda t: t.to(cuda).contiguous()
    ref_cam, view_cam = c(cams[0, 0]), c(cams[0, 1])
    dsg, dig = c(ds), c(di)
    H = ops.get_homographies(ref_cam, view_cam, dsg, dig, D)
    vtrans = ops.transform_depth(c(d_view[0, ..., 0]), view_cam, ref_cam)
    assert torch.equal(vtrans.cpu(), G.transform_depth(d_view, cams[:, 1], cams[:, 0])[0, ..., 0])

    photo = torch.empty(D, h, w, 48, device=cuda)
    ops.warp_planes(c(view_f[0]), H, out=photo, c_off=0, mode=1, ref=c(ref_f[0]))
    wf, mp = ops.warp_by_depth(c(view_f[0]), ref_cam, view_cam, c(d_ref[0, ..., 0]))
    from atvsnet_amd import ops as O
    perr = O.absdiff_mask(wf, c(ref_f[0]), mp)
    ops.tile_planes(perr, photo, 16)
    ops.tile_planes(c(ref_f[0]), photo, 32)
    assert torch.equal(photo.cpu(), want['photo_group'][0])

    geo = torch.empty(D, h, w, 19, device=cuda)
    ops.geo_ref_planes(c(d_ref[0, ..., 0]), dsg, dig, geo, 0)
    ops.warp_planes(vtrans.reshape(h, w, 1), H, out=geo, c_off=1, mode=2, depth_start=dsg, depth_interval=dig, rep=16)
    wd, mg = ops.warp_by_depth(vtrans.reshape(h, w, 1), ref_cam, view_cam, c(d_ref[0, ..., 0]), method='nearest')
    gerr = O.absdiff_mask(wd, c(d_ref[0]), mg)
    ops.tile_planes(gerr, geo, 17)
    ops.tile_planes(c(d_ref[0]), geo, 18)
    assert torch.equal(geo.cpu(), want['geo_group'][0])

    hull = ops.visual_hull(c(d_ref[0, ..., 0]), vtrans, H, dsg, dig)
    assert torch.equal(hull.cpu(), want['vis_hull'][0, ..., 0])


def test_refinement_glue_launches_equal_their_parts_bitwise(cuda):
    """The refinement's fused / batched geometry launches (round 5) against the launches they replace, bit for bit:
    atvs_geo_volume = geo_ref_planes + warp_planes(mode 2) (reference atvsnet/model.py:285-300), rep 1 (one 8-byte store per voxel)
    and rep 16 (the reference's replicated channel, quirk C7); atvs_transform_depth_batch = n x transform_depth (:289,321-324) for
    more maps than one launch holds (18 > 16), distinct cameras per map, invalid depths; atvs_warp_by_depth_err with copy_ref =
    warp_by_depth + absdiff_mask + the tiled reference (:309-316,329-334)."""
    from atvsnet_amd import ops
    cams = _example_cams(4)
    ds, di = OM.depth_start_interval(cams)
    D, h, w = 12, 24, 40
    g = torch.Generator().manual_seed(17)
    c = lambda t: t.to(cuda).contiguous()                       # noqa: E731
    cam = [c(cams[0, i]) for i in range(4)]
    dsg, dig = c(ds), c(di)
    d_ref = c(0.05 + 0.3 * torch.rand(h, w, generator=g))
    maps = []
    for k in range(18):
        m = 0.05 + 0.3 * torch.rand(h, w, generator=g)
        m[k % h, : 1 + k % 5] = 0.0                              # invalid depths: the 1e-10 clip / mask
        maps.append(c(m))
    jobs = [(maps[k], cam[1 + k % 3], cam[(k // 3) % 4]) for k in range(18)]
    got = ops.transform_depth_batch(jobs)
    for (m, lc, rc), o in zip(jobs, got):
        assert torch.equal(o, ops.transform_depth(m, lc, rc))
    H = ops.get_homographies(cam[0], cam[1], dsg, dig, D)
    for ld, c_off, rep in ((2, 0, 1), (19, 0, 16), (4, 1, 2)):
        a = torch.zeros(D, h, w, ld, device=cuda)
        b = torch.zeros(D, h, w, ld, device=cuda)
        ops.geo_volume(d_ref, got[0], H, dsg, dig, a, c_off, rep)
        ops.geo_ref_planes(d_ref, dsg, dig, b, c_off)
        ops.warp_planes(got[0].reshape(h, w, 1), H, out=b, c_off=c_off + 1, mode=2, depth_start=dsg, depth_interval=dig, rep=rep)
        assert torch.equal(a, b), (ld, c_off, rep)
    for C, method in ((16, 'bilinear'), (1, 'nearest')):
        src, ref = c(torch.randn(h, w, C, generator=g)), c(torch.randn(h, w, C, generator=g))
        out = torch.zeros(h, w, 2 * C + 3, device=cuda)
        ops.warp_by_depth_err(src, ref, cam[0], cam[1], d_ref, out, 1, method, True, copy_ref=True)
        wf, mask = ops.warp_by_depth(src, cam[0], cam[1], d_ref, method=method)
        assert torch.equal(out[..., 1:1 + C], ops.absdiff_mask(wf, ref, mask))
        assert torch.equal(out[..., 1 + C:1 + 2 * C], ref)
        assert float(out[..., 0].abs().max()) == 0.0 and float(out[..., 1 + 2 * C:].abs().max()) == 0.0


@pytest.mark.parametrize('D,h,w', [(32, 32, 40), (192, 16, 24), (1, 5, 7), (7, 3, 130)])
def test_softargmin(cuda, D, h, w):
    from atvsnet_amd import ops
    g = torch.Generator().manual_seed(D)
    cost = 6.0 * torch.randn(1, D, h, w, generator=g)
    ds, di = torch.tensor([0.05]), torch.tensor([0.31 / max(D, 1)])
    want = OM.prob2depth(cost, D, ds, di)[0, ..., 0]
    got = ops.softargmin(cost[0].to(cuda), ds.to(cuda), di.to(cuda)).cpu()
    assert torch.allclose(got, want, rtol=2e-6, atol=1e-8)
    _, want_up = OM.prob2depth_upsample(cost, D, ds, di)
    got_up = ops.upsample_softargmin(cost[0].to(cuda), ds.to(cuda), di.to(cuda)).cpu()
    assert got_up.shape == (4 * h, 4 * w)
    assert torch.allclose(got_up, want_up[0, ..., 0], rtol=2e-6, atol=1e-8)


def test_softargmin_known_answers(cuda):
    """one-hot minimum -> that plane's delta; constant cost -> mid-range."""
    from atvsnet_amd import ops
    D, h, w = 16, 4, 6
    ds, di = torch.tensor([0.1], device=cuda), torch.tensor([0.02], device=cuda)
    cost = torch.full((D, h, w), 50.0, device=cuda)
    cost[5] = -50.0
    assert torch.allclose(ops.softargmin(cost, ds, di).cpu(), torch.full((h, w), 0.1 + 5 * 0.02), rtol=1e-6)
    flat = torch.zeros(D, h, w, device=cuda)
    assert torch.allclose(ops.softargmin(flat, ds, di).cpu(), torch.full((h, w), 0.1 + 7.5 * 0.02), rtol=1e-6)


def _query_points(h, w, n, seed):
    """Texture coordinates around and beyond the image, with exact half-integers (tf.round ties), the borders of the
    valid range and non-finite values."""
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(n, generator=g) * (w + 6) - 3).float()
    y = (torch.rand(n, generator=g) * (h + 6) - 3).float()
    special = torch.tensor([0.5, 1.0, 1.5, 2.0, 2.5, w - 1.0, w - 0.5, w - 0.5 - 1e-4, 0.0, 0.49999, float('nan'),
                            float('inf'), -float('inf')])
    k = special.numel()
    x[:k] = special
    y[k:2 * k] = torch.where(special == w - 1.0, torch.tensor(h - 1.0), special)
    y[2 * k] = h - 0.5
    return x, y


@pytest.mark.parametrize('method', ['bilinear', 'nearest'])
@pytest.mark.parametrize('h,w,C', [(32, 40, 32), (17, 23, 3), (9, 11, 1)])
def test_interpolate_bit_exact(cuda, method, h, w, C):
    """interpolate with caller-supplied coordinates (reference homography_warping.py:31-104) against the oracle."""
    from atvsnet_amd.atvsnet import homography_warping as HW
    img = _feat(h, w, C, 5)
    x, y = _query_points(h, w, h * w, 7)
    want, wm = G.interpolate(img, x, y, output_mask=True, method=method)
    got, gm = HW.interpolate(img.to(cuda), x.to(cuda), y.to(cuda), output_mask=True, method=method)
    assert gm.dtype == torch.bool and torch.equal(gm.cpu(), wm)
    got, want = got.cpu(), want
    same = (got == want) | (torch.isnan(got) & torch.isnan(want))      # inf * 0 = NaN on both sides (tf.multiply)
    assert bool(same.all())
    only = HW.interpolate(img.to(cuda), x.to(cuda), y.to(cuda), method=method).cpu()
    assert bool(((only == want) | (torch.isnan(only) & torch.isnan(want))).all())


def test_interpolate_on_the_pixel_grid_is_the_identity_warp(cuda):
    """get_pixel_grids -> interpolate = the image away from the last row / column (SURVEY 8c known answer 2)."""
    from atvsnet_amd.atvsnet import homography_warping as HW
    h, w, C = 12, 20, 4
    img = _feat(h, w, C, 9)
    grid = HW.get_pixel_grids(h, w)
    gx, gy = G.get_pixel_grids(h, w)
    assert grid.shape == (3 * h * w,)
    assert torch.equal(grid.cpu(), torch.cat([gx, gy, torch.ones(h * w)]))
    out = HW.interpolate(img.to(cuda), grid[:h * w], grid[h * w:2 * h * w]).cpu().reshape(h, w, C)
    assert torch.equal(out[:-1, :-1], img[0, :-1, :-1])
    assert bool((out[-1] == 0).all()) and bool((out[:, -1] == 0).all())
    near = HW.interpolate(img.to(cuda), grid[:h * w], grid[h * w:2 * h * w], method='nearest').cpu().reshape(h, w, C)
    assert torch.equal(near[:-1, :-1], img[0, :-1, :-1])
    assert torch.equal(near[-1, 3], img[0, 0, 0])                      # invalid -> pixel (0,0), not masked


def test_interpolate_refuses_what_the_reference_cannot_mean(cuda):
    from atvsnet_amd.atvsnet import homography_warping as HW
    img = _feat(4, 5, 2, 1).to(cuda)
    z = torch.zeros(20, device=cuda)
    with pytest.raises(ValueError):
        HW.interpolate(img, z, z, method='bicubic')
    with pytest.raises(ValueError):
        HW.interpolate(img, z, z[:7])
    with pytest.raises(RuntimeError):
        HW.interpolate(img.cpu(), z.cpu(), z.cpu())
