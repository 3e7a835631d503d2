"""CPU: the oracle's plane-sweep geometry against 3-D re-projection and known answers
(SURVEY.md 8c items 1-3), on the reference's own example/2 cameras (tests/golden/*.npy)."""
import os

import numpy as np
import pytest
import torch

from oracle import homography_warping as G
from oracle import model as OM


def cams_example2():
    d = os.path.join(os.path.dirname(__file__), 'golden')
    c = np.stack([np.load(os.path.join(d, 'example2_%d_cam.npy' % i)) for i in range(2)])
    return torch.from_numpy(c.astype(np.float32))[None], c


def reproject(cams64, px, py, delta):
    """Back-project pixel (px,py) of camera 0 onto the fronto-parallel plane of inverse depth delta,
    project into camera 1 (float64, plain pinhole maths)."""
    R0, t0, K0 = cams64[0, 0, :3, :3], cams64[0, 0, :3, 3], cams64[0, 1, :3, :3]
    R1, t1, K1 = cams64[1, 0, :3, :3], cams64[1, 0, :3, 3], cams64[1, 1, :3, :3]
    Xc = np.linalg.inv(K0) @ np.array([px, py, 1.0]) / delta
    Xw = R0.T @ (Xc - t0)
    p = K1 @ (R1 @ Xw + t1)
    return p[0] / p[2], p[1] / p[2]


def test_homographies_match_3d_reprojection():
    cams, c64 = cams_example2()
    ds, di = OM.depth_start_interval(cams)
    H = G.get_homographies(cams[:, 0], cams[:, 1], 128, ds, di)[0].double().numpy()
    for d in (0, 64, 127):
        delta = float(ds) + d * float(di)
        for px, py in ((40.5, 30.5), (0.5, 0.5), (159.5, 119.5), (80.0, 60.0)):
            q = H[d] @ np.array([px, py, 1.0])
            wx, wy = reproject(c64, px, py, delta)
            assert abs(q[0] / q[2] - wx) < 2e-3 and abs(q[1] / q[2] - wy) < 2e-3
    q = H[0] @ np.array([40.5, 30.5, 1.0])      # value recorded in SURVEY.md 8c-1
    assert q[0] / q[2] == pytest.approx(73.0717, abs=2e-3) and q[1] / q[2] == pytest.approx(34.7608, abs=2e-3)


def test_inverse_and_matmul_helpers():
    g = torch.Generator().manual_seed(0)
    m = torch.randn(5, 3, 3, generator=g) + 3 * torch.eye(3)
    assert torch.allclose(G.mm3(m, G.inv3(m)), torch.eye(3).expand(5, 3, 3), atol=1e-5)
    a, b = torch.randn(2, 3, 3, generator=g), torch.randn(2, 3, 3, generator=g)
    assert torch.allclose(G.mm3(a, b), a @ b, atol=1e-6)


def test_identity_warp_and_border_semantics():
    img = torch.randn(1, 12, 20, 3, generator=torch.Generator().manual_seed(1))
    eye = torch.eye(3)[None]
    out, mask = G.homography_warping(img, eye, output_mask=True)
    assert torch.equal(out[0, :-1, :-1], img[0, :-1, :-1])
    assert torch.all(out[0, -1] == 0) and torch.all(out[0, :, -1] == 0)           # quirk C3: x < W-1 strictly
    assert bool(mask[0, :-1, :-1].all()) and not bool(mask[0, -1].any()) and not bool(mask[0, :, -1].any())
    near = G.homography_warping(img, eye, method='nearest')
    assert torch.equal(near[0, :-1, :-1], img[0, :-1, :-1])
    assert torch.equal(near[0, -1, 3], img[0, 0, 0])                               # quirk C4: invalid -> pixel (0,0)
    # a camera warped onto itself: K R R^T K^-1 is the identity up to fp32 rounding
    cams, _ = cams_example2()
    ds, di = OM.depth_start_interval(cams)
    H = G.get_homographies(cams[:, 0], cams[:, 0], 4, ds, di)
    assert torch.allclose(H[0, 2] / H[0, 2, 2, 2], torch.eye(3), atol=1e-4)
    out = G.homography_warping(img, H[:, 2])
    assert torch.allclose(out[0, 1:-1, 1:-1], img[0, 1:-1, 1:-1], atol=1e-3)


def test_interpolate_exact_values():
    img = torch.arange(12, dtype=torch.float32).reshape(1, 3, 4, 1)
    # one sample per pixel like the reference (B*H*W coordinates); the first five are the probes
    x = torch.tensor([1.0, 1.5, 2.25, -0.2, 3.0] + [1.0] * 7)        # texture coords (pixel centre + 0.5)
    y = torch.tensor([0.5, 1.0, 2.0, 1.0, 1.0] + [1.0] * 7)
    out, valid = G.interpolate(img, x, y, output_mask=True)
    # x-0.5 = 0.5 -> between px 0,1; y-0.5 = 0 -> row 0 : 0.5
    assert out[0, 0] == pytest.approx(0.5)
    assert out[1, 0] == pytest.approx(0.5 * (1 + 5))      # x=1.0, y=0.5 -> rows 0/1 at col 1
    assert out[2, 0] == pytest.approx(0.5 * (4 * 1 + 1.75 + 4 * 2 + 1.75))   # x=1.75, y=1.5
    assert valid[:5].tolist() == [True, True, True, False, True] and out[3, 0] == 0
    # NaN coordinates are invalid and yield NaN-free indices; value follows tf.multiply (NaN*0 = NaN)
    o2, v2 = G.interpolate(img, torch.tensor([float('nan')] + [1.0] * 11), torch.ones(12), output_mask=True)
    assert not bool(v2[0]) and torch.isnan(o2[0, 0])


def test_transform_depth_identity_and_mask():
    cams, _ = cams_example2()
    d = 0.05 + 0.3 * torch.rand(1, 6, 9, 1, generator=torch.Generator().manual_seed(2))
    d[0, 0, :3] = 0.0
    out = G.transform_depth(d, cams[:, 0], cams[:, 0])
    assert torch.allclose(out[d > 1e-10], d[d > 1e-10], rtol=1e-4)
    assert torch.all(out[0, 0, :3] == 0)
    # two different cameras: matches transforming the back-projected point, float64
    _, c64 = cams_example2()
    out = G.transform_depth(d, cams[:, 1], cams[:, 0])
    R1, t1, K1 = c64[1, 0, :3, :3], c64[1, 0, :3, 3], c64[1, 1, :3, :3]
    R0, t0 = c64[0, 0, :3, :3], c64[0, 0, :3, 3]
    for (yy, xx) in ((2, 4), (5, 8), (1, 0)):
        z = 1.0 / float(d[0, yy, xx, 0])
        Xw = R1.T @ (np.linalg.inv(K1) @ np.array([xx + 0.5, yy + 0.5, 1.0]) * z - t1)
        z0 = (R0 @ Xw + t0)[2]
        assert float(out[0, yy, xx, 0]) == pytest.approx(1.0 / z0, rel=1e-4)


def test_warp_by_depth_consistent_with_homography_on_a_plane():
    """A constant inverse-depth map is a fronto-parallel plane: the per-pixel warp equals the plane's homography."""
    cams, _ = cams_example2()
    ds, di = OM.depth_start_interval(cams)
    img = torch.randn(1, 120, 160, 4, generator=torch.Generator().manual_seed(3))   # the cameras' own resolution
    H = G.get_homographies(cams[:, 0], cams[:, 1], 8, ds, di)
    delta = float(ds) + 5 * float(di)
    a, ma = G.homography_warping(img, H[:, 5], output_mask=True)
    b, mb = G.homography_warping_by_depth(img, cams[:, 0], cams[:, 1], torch.full((1, 120, 160, 1), delta), output_mask=True)
    both = (ma & mb)[0, ..., 0]
    assert both.float().mean() > 0.2
    assert float((a - b)[0][both].abs().max()) < 1e-2


def test_visual_hull_values_and_quirk_c6():
    cams, _ = cams_example2()
    cams3 = torch.cat([cams, cams[:, 1:2]], 1)
    ds, di = OM.depth_start_interval(cams)
    g = torch.Generator().manual_seed(4)
    depths = 0.05 + 0.3 * torch.rand(1, 2, 10, 14, generator=g)
    hull = G.get_visual_hull(depths, cams3, 6, ds, di, ref_id=0, view_num=2)
    assert hull.shape == (1, 6, 10, 14, 1)
    assert set(np.unique(hull.numpy()).tolist()) <= {0.0, 0.5, 1.0}
    # first term: [ref > delta_d] -> monotone non-increasing in d for the reference part
    ref_part = torch.stack([(depths[:, 0] > (ds + di * float(d))).float() for d in range(6)], 1)
    assert torch.all(hull[..., 0] * 2 >= ref_part)
