"""-m gpu: probability maps (get_propability_map / prob2depth(out_prob_map=True), reference model.py:13-129) and the
ETH3D batch driver (eval_pointcloud.py) end to end on a small synthetic scene."""
import os

import numpy as np
import pytest
import torch

from oracle import model as OM

pytestmark = pytest.mark.gpu


def _vol(D, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    return 2.0 * torch.randn(1, D, h, w, generator=g)


@pytest.mark.parametrize('D,h,w', [(8, 5, 7), (32, 32, 40), (192, 16, 20)])
def test_probability_map_given_depth(cuda, D, h, w):
    """The 4-tap gather (with and without the on-the-fly soft-max) against the oracle for the SAME depth map:
    depths inside, outside and exactly on the hypotheses."""
    from atvsnet_amd import ops
    vol = _vol(D, h, w, 1)
    ds, di = torch.tensor([0.4]), torch.tensor([0.05])
    g = torch.Generator().manual_seed(2)
    depth = 0.4 + 0.05 * (D + 4) * torch.rand(h, w, generator=g) - 0.1
    depth[0, 0], depth[0, 1], depth[1, 0] = 0.4 + 0.05 * 3, 0.4, 0.4 + 0.05 * (D - 1)      # integral coordinates
    probs = torch.softmax(-vol, dim=1)
    want = OM.get_propability_map(probs, depth.reshape(1, h, w, 1), ds, di).reshape(h, w)
    got_plain = ops.probability_map(probs[0].contiguous().to(cuda), depth.to(cuda), ds.to(cuda), di.to(cuda), 1, False)
    assert float((got_plain.cpu() - want).abs().max()) <= 1e-6
    got_soft = ops.probability_map(vol[0].contiguous().to(cuda), depth.to(cuda), ds.to(cuda), di.to(cuda), 1, True)
    assert float((got_soft.cpu() - want).abs().max()) <= 2e-6          # soft-max terms: fp32 exp / sum order


def test_prob2depth_with_probability_maps(cuda):
    """model.prob2depth / prob2depth_upsample(out_prob_map=True) vs the oracle.  The plane indices come from
    floor/ceil of the regressed depth: pixels whose coordinate is within 1e-3 of an integer may pick another
    plane than the oracle's (depth differs by 1e-6 relative) and are compared through the oracle evaluated on the
    HIP depth instead."""
    from atvsnet_amd.atvsnet import model
    D, h, w = 48, 12, 20
    vol = _vol(D, h, w, 3)
    ds, di = torch.tensor([0.5]), torch.tensor([0.02])
    d, d_up, p, p_up = [t.cpu() for t in model.prob2depth_upsample(vol.to(cuda), D, ds.to(cuda), di.to(cuda), True)]
    wd, wd_up, wp, wp_up = OM.prob2depth_upsample(vol, D, ds, di, True)
    assert float((d - wd).abs().max()) <= 1e-5 * float(wd.abs().max())
    assert float((d_up - wd_up).abs().max()) <= 1e-5 * float(wd_up.abs().max())
    for got, depth, volume in ((p, d, vol), (p_up, d_up, OM.upsample_prob_vol(vol))):
        ref = OM.get_propability_map(torch.softmax(-volume, dim=1), depth, ds, di)
        assert float((got - ref).abs().max()) <= 5e-6
    frac = lambda x: ((x - ds) / di - torch.round((x - ds) / di)).abs()                         # noqa: E731
    safe = frac(wd) > 1e-3
    assert float((p - wp)[safe].abs().max()) <= 5e-6 and float(safe.float().mean()) > 0.9
    d1, p1 = model.prob2depth(vol.to(cuda), D, ds.to(cuda), di.to(cuda), True)
    assert torch.equal(d1.cpu(), d) and torch.equal(p1.cpu(), p)
    # the reference-signature entry point: a probability volume in, plain gather
    g = model.get_propability_map(torch.softmax(-vol, dim=1).to(cuda), d.to(cuda), ds.to(cuda), di.to(cuda))
    assert float((g.cpu() - OM.get_propability_map(torch.softmax(-vol, dim=1), d, ds, di)).abs().max()) <= 1e-6


def _write_scene(root, n_views, H, W, rng):
    from PIL import Image
    from atvsnet_amd import synthetic
    from atvsnet_amd.atvsnet import preprocess as P
    scene = os.path.join(root, 'eth3d', 'toy')
    os.makedirs(os.path.join(scene, 'images'))
    os.makedirs(os.path.join(scene, 'cams'))
    imgs, cams = synthetic.make_inputs(n_views, H, W, 16)
    for v in range(n_views):
        Image.fromarray(imgs[0, v].astype(np.uint8)[:, :, ::-1]).save(os.path.join(scene, 'images', '%08d.jpg' % v), quality=95)
        cam = cams[0, v].astype(np.float64).copy()
        # full-resolution intrinsics + a metric depth range (the driver converts it to inverse depth itself)
        cam[1, 0, 0] *= 4; cam[1, 1, 1] *= 4; cam[1, 0, 2] *= 4; cam[1, 1, 2] *= 4
        cam[1, 3] = (2.0, 0.05, 16, 0.0)
        P.write_cam(os.path.join(scene, 'cams', '%08d_cam.txt' % v), cam)
    with open(os.path.join(scene, 'pair.txt'), 'w') as f:
        f.write('2\n0\n2 1 1.0 2 0.5\n1\n2 0 1.0 2 0.5\n')
    return scene


def test_eth3d_driver_end_to_end(cuda, tmp_path, weights):
    """cli -> pair.txt / cams / JPEGs -> depth + probability PFMs, camera, images, runtime file; the written depth is
    the pipeline's output for the loaded data (graph replay == eager)."""
    from atvsnet_amd import FLAGS
    from atvsnet_amd.atvsnet import eval_pointcloud as E, example, preprocess as P
    FLAGS.reset()
    try:
        root = str(tmp_path)
        _write_scene(root, 3, 128, 160, np.random.default_rng(0))
        E.cli(['--data_root', root, '--savepath', os.path.join(root, 'out'), '--view_num', '3', '--max_d', '16',
               '--max_w', '160', '--max_h', '128', '--synthetic_weights', '--scenes', 'toy'])
        out = os.path.join(root, 'out', 'toy', 'depths_atvsnet')
        for idx in (0, 1):
            for suffix in ('.pfm', '_prob.pfm', '.jpg', '.txt', '.png'):
                assert os.path.exists(os.path.join(out, '%08d%s' % (idx, suffix)))
        assert open(os.path.join(root, 'out', 'toy', 'zz_runtime.txt')).read().startswith('runtime ')
        with open(os.path.join(out, '00000000.pfm'), 'rb') as f:
            depth = P.load_pfm(f)
        with open(os.path.join(out, '00000000_prob.pfm'), 'rb') as f:
            prob = P.load_pfm(f)
        assert depth.shape == (32, 40) and prob.shape == (32, 40)
        assert np.isfinite(depth).all() and (depth > 0).all() and (prob > 0).all() and (prob <= 4.0).all()      # up to 3 p0 + p1 where the depth clips to the first plane
        # the same entry run eagerly from the loaded data
        mvs = E.gen_data_list(os.path.join(root, 'eth3d', 'toy'))
        assert len(mvs) == 2 and len(mvs[0]) == 6
        raw, images, cams, _, index = E.load_data(mvs, 0)
        assert index == 0 and images.shape == (1, 3, 128, 160, 3) and raw.shape == (1, 3, 32, 40, 3) and cams.shape == (1, 3, 2, 4, 4)
        assert abs(float(images[0, 0].mean())) < 1e-3              # centred input, as the reference feeds it
        d, d_up, p, p_up = example.infer_multiview(torch.from_numpy(images).to(cuda), torch.from_numpy(cams.astype(np.float32)).to(cuda),
                                                   16, out_prob_map=True)
        want = 1.0 / d.cpu().numpy().reshape(32, 40)
        assert np.abs(depth - want).max() <= 1e-5 * np.abs(want).max()
        assert d_up.shape == (1, 128, 160, 1) and p_up.shape == (1, 128, 160, 1)
        assert np.abs(prob - p.cpu().numpy().reshape(32, 40)).max() <= 1e-5
        with open(os.path.join(out, '00000000.txt')) as f:
            cam = P.load_cam(f)
        assert np.allclose(cam[1, 3, 2], 16) and cam[1, 0, 0] > 0
        # the same scene with the two queued depth maps on disjoint halves of every XCD: the files' bytes are the serial run's
        E.cli(['--data_root', root, '--savepath', os.path.join(root, 'out2'), '--view_num', '3', '--max_d', '16', '--max_w', '160',
               '--max_h', '128', '--synthetic_weights', '--scenes', 'toy', '--maps_in_flight', 'cu_split'])
        assert E._Pipelines.CO_RESIDENT == 'cu_split'
        for idx in (0, 1):
            for suffix in ('.pfm', '_prob.pfm'):
                a = open(os.path.join(out, '%08d%s' % (idx, suffix)), 'rb').read()
                b = open(os.path.join(root, 'out2', 'toy', 'depths_atvsnet', '%08d%s' % (idx, suffix)), 'rb').read()
                assert a == b, (idx, suffix)
    finally:
        E._Pipelines.CO_RESIDENT = False
        FLAGS.reset()
