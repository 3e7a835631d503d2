"""CPU restatement of the reference's depth-error metrics (atvsnet/eval_errors.py:25-93).

TEST INFRASTRUCTURE (see oracle/__init__.py).  PINNED: checked against golden vectors
produced by the reference function itself (tests/golden/calc_error_golden.npz) and against
the reference's example/*/result/error.xlsx.  Written metric by metric in float64-free numpy
like the reference (input dtype arithmetic, float32 result vector).
"""
import numpy as np


def calc_error(pred_in, gt_in, num_depths=100, inlier_threshold=(1, 3, 5, 10)):
    pred = pred_in.copy()
    gt = gt_in.copy()
    gt[np.isnan(gt)] = 0.0
    pred[np.isnan(pred)] = 0.0
    rng = np.sort(gt[(gt < 1e10) & (gt > 0.0)])
    interval = float(rng[-1] - rng[0]) / float(num_depths)
    mask = (gt > 0.0) & (gt < 1e10) & (pred > 0.0) & (pred < 1e10)
    n = float(mask.sum())
    gt[~mask] = 1.0
    pred[~mask] = 1.0
    d = mask * np.abs(gt - pred)
    dinv = mask * np.abs(1.0 / gt - 1.0 / pred)
    dlog = mask * np.abs(np.log(gt) - np.log(pred))
    out = np.zeros(10 + len(inlier_threshold), np.float32)
    out[0] = d.sum() / n
    out[1] = np.sqrt(np.float32((d * d).sum() / n))
    out[2] = dinv.sum() / n
    out[3] = np.sqrt(np.float32((dinv * dinv).sum() / n))
    out[4] = dlog.sum() / n
    msl = (dlog * dlog).sum() / n
    out[5] = np.sqrt(msl)
    ls = (mask * (np.log(gt) - np.log(pred))).sum()
    out[6] = np.sqrt(msl - (ls * ls / (n * n)))
    out[7] = (d / gt).sum() / n
    out[8] = ((d * d) / (gt * gt)).sum() / n
    out[9] = d.sum() / interval / n
    rel = d[mask] / interval
    for i, th in enumerate(inlier_threshold):
        out[10 + i] = float((rel < th).sum()) / n
    return out, [num_depths, interval, rng[0], rng[-1], list(inlier_threshold)]
