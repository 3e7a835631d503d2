"""CPU: the oracle of the depth-map fusion (oracle/fusibile.py, restating reference fusibile/fusibile.cu:138-277) on a
synthetic scene with known answers, and the host side of atvsnet/depth_fusion.py (file formats, camera packing)."""
import os

import numpy as np
import pytest

from fusion_scene import make_scene


def test_fusion_of_exact_depth_maps_recovers_the_plane():
    from oracle import fusibile as F
    Ps, depths, normals, images, n, d0 = make_scene(4)
    pts, cols = F.fuse(Ps, depths, normals, images, disp_thresh=0.01, num_consistent=2)
    rows, cols_n = depths.shape[1:]
    assert len(pts) > 0.8 * 4 * rows * cols_n                 # nearly every pixel of every view is confirmed
    assert float(np.abs(pts.astype(np.float64) @ n + d0).max()) < 2e-3      # every point lies on the plane
    assert cols.dtype == np.uint8 and cols.shape == (len(pts), 3)
    # a view whose depths are wrong by 5 % is never confirmed and confirms nobody
    bad = depths.copy()
    bad[1] *= 1.05
    cams = F.pack_cameras(Ps)
    nd = np.concatenate([normals, bad[..., None]], -1).astype(np.float32)
    img4 = np.concatenate([images.astype(np.float32), np.zeros(images.shape[:3] + (1,), np.float32)], -1)
    _, _, _, created = F.fuse_reference(cams, nd, img4, 1, 0.01, 2 * np.pi, 2)
    assert not created.any()
    _, _, _, created0 = F.fuse_reference(cams, nd, img4, 0, 0.01, 2 * np.pi, 3)
    assert not created0.any()                                 # only 2 of the 3 others can agree now
    _, _, _, created0 = F.fuse_reference(cams, nd, img4, 0, 0.01, 2 * np.pi, 2)
    assert created0.mean() > 0.8
    # zero depth (probability-filtered pixel) creates no point
    hole = depths.copy()
    hole[0, 10:20, 10:20] = 0
    nd = np.concatenate([normals, hole[..., None]], -1).astype(np.float32)
    _, _, _, created = F.fuse_reference(cams, nd, img4, 0, 0.01, 2 * np.pi, 1)
    assert not created[10:20, 10:20].any() and created[30:, 30:].all()


def test_normal_threshold_and_averaging():
    from oracle import fusibile as F
    Ps, depths, normals, images, n, d0 = make_scene(3)
    cams = F.pack_cameras(Ps)
    nd = np.concatenate([normals, depths[..., None]], -1).astype(np.float32)
    img4 = np.concatenate([images.astype(np.float32), np.zeros(images.shape[:3] + (1,), np.float32)], -1)
    # camera-frame normals of a plane differ by the relative rotation (3 degrees per view): 0.04 rad passes 1-view steps only
    X, nrm, tex, created = F.fuse_reference(cams, nd, img4, 0, 0.01, 0.08, 1)
    _, _, _, created_tight = F.fuse_reference(cams, nd, img4, 0, 0.01, 0.001, 1)
    assert created.mean() > 0.8 and not created_tight.any()
    inner = created[8:-8, 8:-8]
    assert inner.all()
    # the averaged colour stays close to the reference image where every view agrees (same surface colour + noise)
    assert float(np.abs(tex[8:-8, 8:-8, :3] - images[0, 8:-8, 8:-8].astype(np.float32)).mean()) < 12.0
    assert float(np.abs(np.linalg.norm(nrm[8:-8, 8:-8], axis=-1) - 1.0).max()) < 1e-2


def test_camera_packing_matches_the_oracle_and_the_geometry():
    from atvsnet_amd.atvsnet import depth_fusion as DF
    from oracle import fusibile as F
    Ps = make_scene(4)[0]
    a, b = DF.pack_cameras(Ps), F.pack_cameras(Ps)
    assert a.shape == b.shape == (4, 28) and np.allclose(a, b, rtol=1e-6, atol=1e-6)
    for i, P in enumerate(Ps):
        C = a[i, 21:24].astype(np.float64)
        assert np.abs(P @ np.append(C, 1.0)).max() < 1e-3          # the centre projects to 0
        assert abs(a[i, 27] - 60.0) < 1e-3                          # focal length recovered from P
        assert np.allclose(a[i, 12:21].reshape(3, 3) @ P[:, :3], np.eye(3), atol=1e-4)


def test_gipuma_files_roundtrip(tmp_path):
    from atvsnet_amd.atvsnet import depth_fusion as DF
    from atvsnet_amd.tools import ply
    rng = np.random.default_rng(3)
    d = rng.uniform(0, 5, (7, 9)).astype(np.float32)
    nrm = rng.normal(size=(7, 9, 3)).astype(np.float32)
    DF.write_gipuma_dmb(str(tmp_path / 'd.dmb'), d)
    DF.write_gipuma_dmb(str(tmp_path / 'n.dmb'), nrm)
    assert np.array_equal(DF.read_gipuma_dmb(str(tmp_path / 'd.dmb')), d)
    assert np.array_equal(DF.read_gipuma_dmb(str(tmp_path / 'n.dmb')), nrm)
    raw = open(str(tmp_path / 'd.dmb'), 'rb').read()
    assert np.frombuffer(raw[:16], '<i4').tolist() == [1, 7, 9, 1] and len(raw) == 16 + 7 * 9 * 4
    DF.fake_colmap_normal(str(tmp_path / 'd.dmb'), str(tmp_path / 'fn.dmb'))
    fn = DF.read_gipuma_dmb(str(tmp_path / 'fn.dmb'))
    assert np.allclose(fn[d > 0], 1 / 1.732050808) and fn.shape == (7, 9, 3)
    pts = rng.normal(size=(11, 3)).astype(np.float32)
    pts[3, 1] = np.inf
    col = rng.integers(0, 255, (11, 3)).astype(np.uint8)
    ply.write_ply(str(tmp_path / 'm.ply'), pts, col)
    p2, c2 = ply.read_ply(str(tmp_path / 'm.ply'))
    assert np.array_equal(c2, col) and np.array_equal(p2[3], [0, 0, 0]) and np.array_equal(p2[0], pts[0])
    head = open(str(tmp_path / 'm.ply'), 'rb').read(200).decode('latin-1')
    assert head.startswith('ply\nformat binary_little_endian 1.0\nelement vertex 11\nproperty float x\n')
    assert 'property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n' in head


def write_dense_folder(root, n_views=3, prob_hole=True):
    """<root>/depths_atvsnet/{%08d.pfm, _prob.pfm, .jpg, .txt} as eval_pointcloud writes them."""
    from PIL import Image
    from atvsnet_amd.atvsnet.preprocess import write_cam, write_pfm
    Ps, depths, normals, images, n, d0 = make_scene(n_views)
    out = os.path.join(root, 'depths_atvsnet')
    os.makedirs(out)
    rows, cols = depths.shape[1:]
    K = np.array([[60.0, 0, cols / 2.0], [0, 60.0, rows / 2.0], [0, 0, 1]])
    for i in range(n_views):
        stem = os.path.join(out, '%08d' % i)
        write_pfm(stem + '.pfm', depths[i])
        prob = np.full((rows, cols), 0.95, np.float32)
        if prob_hole:
            prob[5:15, 5:15] = 0.2
        write_pfm(stem + '_prob.pfm', prob)
        Image.fromarray(np.ascontiguousarray(images[i][:, :, ::-1])).save(stem + '.png')      # lossless: exact colours
        Image.fromarray(np.ascontiguousarray(images[i][:, :, ::-1])).save(stem + '.jpg')
        cam = np.zeros((2, 4, 4))
        cam[0] = np.eye(4)
        cam[0, :3, :] = np.linalg.inv(K) @ Ps[i]
        cam[1, :3, :3] = K
        cam[1, 3] = (0.1, 0.01, 32, 0.42)
        write_cam(stem + '.txt', cam)
    return Ps, depths, normals, images, n, d0


def test_probability_filter_and_gipuma_layout(tmp_path):
    from atvsnet_amd.atvsnet import depth_fusion as DF
    from atvsnet_amd.atvsnet.preprocess import load_pfm
    root = str(tmp_path)
    Ps, depths = write_dense_folder(root)[:2]
    DF.probability_filter(root, 0.8)
    with open(os.path.join(root, 'depths_atvsnet', '00000001_prob_filtered.pfm'), 'rb') as f:
        filt = load_pfm(f)
    assert (filt[5:15, 5:15] == 0).all() and np.array_equal(filt[20:, 20:], depths[1][20:, 20:])
    pf = os.path.join(root, 'points_atvsnet')
    DF.atvsnet_to_gipuma(root, pf)
    assert sorted(os.listdir(os.path.join(pf, 'images'))) == ['%08d.jpg' % i for i in range(3)]
    assert os.path.isfile(os.path.join(pf, '2333__00000002', 'disp.dmb'))
    assert os.path.isfile(os.path.join(pf, '2333__00000002', 'normals.dmb'))
    P1 = DF.read_p_file(os.path.join(pf, 'cams', '00000001.jpg.P'))
    assert np.allclose(P1, Ps[1], rtol=1e-9, atol=1e-9)
    assert np.array_equal(DF.read_gipuma_dmb(os.path.join(pf, '2333__00000001', 'disp.dmb')), filt)
