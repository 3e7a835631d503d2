"""CPU: the entry point's data loading (reference atvsnet/example.py:312-342) on the reference's own example/2 files
(tests/golden/example2/*: data) -- cv2.imread's contract (BGR uint8, H x W x 3) met by the PIL decode, cameras and ground truth as
stored, the views-found count main() falls back on, and the decoded pixels the oracle fixture was generated from."""
import hashlib
import os
import sys

import numpy as np

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
sys.path.insert(0, GOLD)


def test_load_example_on_the_reference_example2(tmp_path):
    from PIL import Image
    from make_example_golden import example_dir
    from atvsnet_amd.atvsnet import example as ex
    d = example_dir(str(tmp_path))
    images, cams, gt, valid = ex.load_example(d, 2)
    assert valid == 2
    assert images.shape == (2, 480, 640, 3) and images.dtype == np.uint8 and images.flags['C_CONTIGUOUS']
    assert cams.shape == (2, 2, 4, 4) and gt.shape == (480, 640, 1) and gt.dtype == np.float32
    for i in range(2):
        rgb = np.asarray(Image.open(os.path.join(d, '%d.jpg' % i)).convert('RGB'))
        assert np.array_equal(images[i][:, :, ::-1], rgb)                        # channel 0 is BLUE, as cv2.imread returns it
        assert np.array_equal(cams[i], np.load(os.path.join(GOLD, 'example2_%d_cam.npy' % i)))
    assert np.allclose(cams[:, 0, 3], [0, 0, 0, 1]) and float(cams[0, 1, 3, 0]) > 0 and float(cams[0, 1, 3, 1]) > 0
    gold = np.load(os.path.join(GOLD, 'example2_twoview.npz'))
    assert hashlib.sha1(images.tobytes()).hexdigest() == str(gold['images_sha1'])   # the pixels the oracle fixture was made from
    assert gold['inverse_depth'].shape == (480, 640) and gold['error'].shape == (14,)


def test_load_example_counts_the_views_it_finds(tmp_path, capsys):
    from make_example_golden import example_dir
    from atvsnet_amd.atvsnet import example as ex
    d = example_dir(str(tmp_path))
    images, cams, gt, valid = ex.load_example(d, 5)                               # the reference's default view_num on a 2-view folder
    out = capsys.readouterr().out
    assert valid == 2 and images.shape[0] == 2 and cams.shape[0] == 2
    assert out.count('not exist. check view_num') == 3 and '2.jpg' in out and '4_cam.npy' in out
    os.remove(os.path.join(d, '0_gt.npy'))
    assert ex.load_example(d, 2)[2] is None
