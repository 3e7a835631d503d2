"""Generates tests/golden/calc_error_golden.npz by importing the REFERENCE's
atvsnet/eval_errors.py (pure numpy) from /root/reference -- run in the build container only.

Inputs are strided sub-samples of the reference's bundled example/{0,1,2} prediction / ground
truth pairs (kept small), plus edge cases (NaN / inf / non-positive pixels); the expected
outputs are what the reference function returns on exactly those arrays.  The full-size
known answers of example/*/result/error.xlsx are copied next to it as data files.
"""
import os
import shutil
import sys

import numpy as np

REF = '/root/reference'
sys.path.insert(0, os.path.join(REF, 'atvsnet'))
import eval_errors as ref          # noqa: E402

out = {}
here = os.path.dirname(os.path.abspath(__file__))
for i in (0, 1, 2):
    pred = np.load(os.path.join(REF, 'example/%d/result/pred.npy' % i))
    gt = np.squeeze(np.load(os.path.join(REF, 'example/%d/0_gt.npy' % i)))
    full, _ = ref.calc_error(pred, gt)
    out['full_%d' % i] = full                       # equals the xlsx column
    p, g = pred[3::8, 5::8].copy(), gt[3::8, 5::8].copy()
    e, info = ref.calc_error(p, g)
    out['pred_%d' % i], out['gt_%d' % i], out['err_%d' % i] = p, g, e
    out['info_%d' % i] = np.array(info[:4], dtype=np.float64)
    shutil.copy(os.path.join(REF, 'example/%d/result/error.xlsx' % i), os.path.join(here, 'example%d_error.xlsx' % i))
    os.chmod(os.path.join(here, 'example%d_error.xlsx' % i), 0o644)
# edge cases: NaN, inf, zero and negative pixels in either map
rng = np.random.default_rng(0)
g = rng.uniform(1.0, 9.0, size=(24, 31)).astype(np.float32)
p = (g * rng.uniform(0.9, 1.1, size=g.shape)).astype(np.float32)
g[0, :5] = np.nan
g[1, :5] = 0.0
g[2, :5] = -1.0
g[3, :5] = np.inf
p[4, :5] = np.nan
p[5, :5] = np.inf
p[6, :5] = 0.0
e, info = ref.calc_error(p, g, num_depths=64, inlier_threshold=[1, 2, 4])
out['pred_edge'], out['gt_edge'], out['err_edge'] = p, g, e
np.savez_compressed(os.path.join(here, 'calc_error_golden.npz'), **out)
print('wrote', {k: v.shape for k, v in out.items()})
