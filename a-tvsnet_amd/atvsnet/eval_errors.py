"""Depth-map error metrics (reference: atvsnet/eval_errors.py:25-93), host-side numpy.

Ten error metrics plus the inlier ratios of the range-normalised absolute error, in the
reference's order and dtype (float32 result vector); invalid pixels (NaN, <= 0, >= 1e10 in
either map) are excluded.  Pinned by tests/golden/calc_error_golden.npz, generated from the
reference module, and by the reference's example/*/result/error.xlsx numbers.
"""
import numpy as np

inlier_thres = [1, 3, 5, 10]

err_metrics_namelist = ['mae', 'rmse', 'inverse_mae', 'inverse_rmse', 'log_mae', 'log_rmse', 'scale_invariant_log',
                        'abs_relative', 'squared_relative', 'mae_normalized']

acc_metrics_namelist = ['inlier_ratios_' + str(i) for i in inlier_thres]


def calc_error(depth_predict_in, depth_gt_in, num_depths=100, inlier_threshold=inlier_thres):
    """-> (errors float32 (10 + len(inlier_threshold),), infos list)."""
    assert depth_predict_in.shape == depth_gt_in.shape
    pred = np.where(np.isnan(depth_predict_in), 0.0, depth_predict_in).astype(depth_predict_in.dtype)
    gt = np.where(np.isnan(depth_gt_in), 0.0, depth_gt_in).astype(depth_gt_in.dtype)

    # depth range of the ground truth -> width of one of `num_depths` bins
    g = np.sort(gt[(gt > 0.0) & (gt < 1e10)].ravel())
    bin_width = float(g[-1] - g[0]) / float(num_depths)

    ok = (gt > 0.0) & (gt < 1e10) & (pred > 0.0) & (pred < 1e10)
    n = float(np.sum(ok))
    assert n > 0
    gt = np.where(ok, gt, 1.0).astype(gt.dtype)
    pred = np.where(ok, pred, 1.0).astype(pred.dtype)

    absd = ok * np.abs(gt - pred)
    absd_inv = ok * np.abs(1.0 / gt - 1.0 / pred)
    absd_log = ok * np.abs(np.log(gt) - np.log(pred))
    mean_sq_log = np.sum(absd_log * absd_log) / n
    signed_log = np.sum(ok * (np.log(gt) - np.log(pred)))

    e = np.zeros(10 + len(inlier_threshold), dtype=np.float32)
    e[0] = np.sum(absd) / n
    e[1] = np.sum(absd * absd) / n
    e[1] = np.sqrt(e[1])
    e[2] = np.sum(absd_inv) / n
    e[3] = np.sum(absd_inv * absd_inv) / n
    e[3] = np.sqrt(e[3])
    e[4] = np.sum(absd_log) / n
    e[5] = np.sqrt(mean_sq_log)
    e[6] = np.sqrt(mean_sq_log - (signed_log * signed_log / (n * n)))
    e[7] = np.sum(absd / gt) / n
    e[8] = np.sum((absd * absd) / (gt * gt)) / n
    e[9] = np.sum(absd) / bin_width / n
    scaled = absd[ok] / bin_width
    for i, th in enumerate(inlier_threshold):
        e[10 + i] = float(np.sum(scaled < th)) / n
    return e, [num_depths, bin_width, g[0], g[-1], inlier_threshold]
