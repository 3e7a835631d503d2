"""-m gpu: the depth-map fusion kernel (csrc/fusion.hip, reference fusibile/fusibile.cu:138-277) against its oracle,
bit for bit, and the depth_fusion driver end to end."""
import os

import numpy as np
import pytest
import torch

from fusion_scene import make_scene
from test_fusion import write_dense_folder

pytestmark = pytest.mark.gpu


def _operands(n_views, seed, noise):
    from atvsnet_amd.atvsnet import depth_fusion as DF
    Ps, depths, normals, images, _, _ = make_scene(n_views, rows=56, cols=72, seed=seed)
    rng = np.random.default_rng(seed)
    depths = depths * (1.0 + noise * rng.normal(size=depths.shape)).astype(np.float32)     # some views disagree
    depths[:, 3:9, 40:50] = 0                                                              # filtered pixels
    cams = DF.pack_cameras(Ps)
    nd = np.ascontiguousarray(np.concatenate([normals, depths[..., None]], -1).astype(np.float32))
    img4 = np.ascontiguousarray(np.concatenate([images.astype(np.float32), np.zeros(images.shape[:3] + (1,), np.float32)], -1))
    return Ps, depths, normals, images, cams, nd, img4


@pytest.mark.parametrize('n_views,noise,disp,nthr,ncons', [(4, 0.004, 0.01, 2 * np.pi, 2), (3, 0.0, 0.01, 0.08, 1),
                                                           (5, 0.01, 0.02, 2 * np.pi, 3)])
def test_fusion_kernel_bit_exact(cuda, n_views, noise, disp, nthr, ncons):
    from atvsnet_amd import ops
    from oracle import fusibile as F
    _, _, _, _, cams, nd, img4 = _operands(n_views, 7, noise)
    cd, ndd, imd = (torch.from_numpy(a).to(cuda) for a in (cams, nd, img4))
    some = 0
    for ref in range(n_views):
        X, nrm, tex, created = F.fuse_reference(cams, nd, img4, ref, disp, nthr, ncons)
        coord, normal, texture, cr = [t.cpu().numpy() for t in ops.fusibile(cd, ndd, imd, ref, disp, nthr, ncons)]
        assert np.array_equal(cr > 0, created), ref
        assert np.array_equal(coord[..., :3], X, equal_nan=True), ref
        assert np.array_equal(normal[..., :3], nrm, equal_nan=True), ref
        assert np.array_equal(texture[..., :3], tex[..., :3], equal_nan=True), ref
        some += int(created.sum())
    assert 0 < some < n_views * nd.shape[1] * nd.shape[2]          # the case decides something


def test_depth_fusion_driver_end_to_end(cuda, tmp_path):
    """probability filter -> gipuma files -> fusion on the GPU -> PLY, against the oracle on the same files."""
    from atvsnet_amd.atvsnet import depth_fusion as DF
    from atvsnet_amd.tools import ply
    from oracle import fusibile as F
    root = str(tmp_path)
    _, _, _, _, n, d0 = write_dense_folder(root, n_views=4)
    DF.main(['--dense_folder', root, '--prob_threshold', '0.8', '--disp_threshold', '0.01', '--num_consistent', '2'])
    pts, cols = ply.read_ply(os.path.join(root, 'final3d_model.ply'))
    assert len(pts) > 0.6 * 4 * 48 * 64
    assert float(np.abs(pts.astype(np.float64) @ n + d0).max()) < 5e-3
    # the oracle on exactly what the driver read back from disk
    pf = os.path.join(root, 'points_atvsnet')
    names = ['%08d' % i for i in range(4)]
    Ps = [DF.read_p_file(os.path.join(pf, 'cams', s + '.jpg.P')) for s in names]
    depths = np.stack([DF.read_gipuma_dmb(os.path.join(pf, '2333__' + s, 'disp.dmb')) for s in names])
    normals = np.stack([DF.read_gipuma_dmb(os.path.join(pf, '2333__' + s, 'normals.dmb')) for s in names])
    images = np.stack([DF._imread_bgr(os.path.join(pf, 'images', s + '.jpg')) for s in names])
    want_p, want_c = F.fuse(Ps, depths, normals, images, 0.01, 360 * np.pi / 180.0, 2)
    assert np.array_equal(pts, want_p) and np.array_equal(cols, want_c)
