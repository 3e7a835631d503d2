"""Generates tests/golden/example2_twoview.npz: the CPU oracle's two-view answer on the REFERENCE's own example inputs
(tests/golden/example2/{0,1}.jpg, 0_gt.npy and tests/golden/example2_{0,1}_cam.npy are the data files of
/root/reference/example/2, copied as data) with the seeded synthetic weights -- the checkpoint is not distributed.

What the entry point does with them (reference example.py:304-345, 219-302): decode the JPEGs to BGR uint8 (cv2.imread there,
PIL here), stack, feed as float32 0..255 with the .npy cameras, max_d = FLAGS.max_d = 128, write 1 / inverse depth to pred.npy.
The fixture holds the oracle's inverse-depth map (pred.npy is 1 / it), calc_error(1 / it, gt) and a SHA-1 of the decoded
images (so that a different JPEG decoder on another box is noticed instead of showing up as a parity failure).

    python tests/golden/make_example_golden.py          (build container, CPU only; ~1 min on 8 cores)
"""
import hashlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import atvsnet_amd                                     # noqa: E402,F401
from atvsnet_amd import variables                      # noqa: E402
from atvsnet_amd.atvsnet import example as ex          # noqa: E402   (the product's loader: same decode as the entry point)
from atvsnet_amd.atvsnet.eval_errors import calc_error  # noqa: E402   (host code pinned by calc_error_golden.npz)
from oracle import model as OM                         # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
MAX_D = 128                                            # FLAGS.max_d default (reference example.py:36)


def example_dir(dst):
    """Lay tests/golden's example/2 data out the way the reference expects it under <dst>/2/."""
    import shutil
    d = os.path.join(dst, '2')
    os.makedirs(d, exist_ok=True)
    for n in ('0.jpg', '1.jpg', '0_gt.npy'):
        shutil.copy(os.path.join(HERE, 'example2', n), os.path.join(d, n))
    for i in range(2):
        shutil.copy(os.path.join(HERE, 'example2_%d_cam.npy' % i), os.path.join(d, '%d_cam.npy' % i))
    return d


def main():
    import tempfile
    torch.set_num_threads(int(os.environ.get('ORACLE_THREADS', min(os.cpu_count() or 1, 8))))
    t0 = time.time()
    with tempfile.TemporaryDirectory() as tmp:
        images, cams, gt, valid = ex.load_example(example_dir(tmp), 2)
    assert valid == 2 and images.dtype == np.uint8 and images.shape == (2, 480, 640, 3)
    W = {k: torch.from_numpy(v) for k, v in variables.VariableStore().init_synthetic(1234).host.items()}
    imgs = torch.from_numpy(images.astype(np.float32))[None]
    cm = torch.from_numpy(cams.astype(np.float32))[None]
    with torch.no_grad():
        inv = OM.run_twoview(imgs, cm, W, MAX_D)[0, ..., 0].numpy()
    depth = inv.copy()
    depth[depth <= 0] = float('inf')
    depth = 1.0 / depth
    err, _ = calc_error(depth, np.squeeze(gt))
    out = {'inverse_depth': inv, 'error': np.asarray(err, np.float64),
           'images_sha1': np.array(hashlib.sha1(images.tobytes()).hexdigest()),
           'seconds': np.float64(time.time() - t0)}
    path = os.path.join(HERE, 'example2_twoview.npz')
    np.savez_compressed(path, **out)
    print('example2: %.0f s, %.2f MB, inverse depth mean %.6f, sha1 %s' %
          (out['seconds'], os.path.getsize(path) / 1e6, float(inv.mean()), out['images_sha1']))


if __name__ == '__main__':
    main()
