"""CPU restatement of the reference's plane-sweep geometry.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows
/root/reference/atvsnet/homography_warping.py function by function; float32
torch-CPU tensors, channel-last, B=1 like every caller (quirk C8).

Small 3x3 products are written as explicit fixed-order sums of products
(no fused multiply-add, no BLAS) so that the HIP kernels, compiled with
-ffp-contract=off and the same operation order, reproduce the sampling
coordinates bit for bit; the reference's tf.matmul / tf.matrix_inverse order
is not observable (TF is not installed), so this is a restatement choice.
"""
import torch

from . import tf_ops

INVERSE_DEPTH = True   # FLAGS.inverse_depth (example.py:47), True everywhere on the path


def mm3(a, b):
    """(...,3,K)x(...,K,N) product, K=3, fixed order ((a0*b0 + a1*b1) + a2*b2)."""
    return (a[..., :, 0:1] * b[..., 0:1, :] + a[..., :, 1:2] * b[..., 1:2, :]) + a[..., :, 2:3] * b[..., 2:3, :]


def inv3(m):
    """3x3 inverse by adjugate / determinant (stands in for tf.matrix_inverse,
    homography_warping.py:123,199,290)."""
    a, b, c = m[..., 0, 0], m[..., 0, 1], m[..., 0, 2]
    d, e, f = m[..., 1, 0], m[..., 1, 1], m[..., 1, 2]
    g, h, i = m[..., 2, 0], m[..., 2, 1], m[..., 2, 2]
    c00 = e * i - f * h
    c01 = c * h - b * i
    c02 = b * f - c * e
    c10 = f * g - d * i
    c11 = a * i - c * g
    c12 = c * d - a * f
    c20 = d * h - e * g
    c21 = b * g - a * h
    c22 = a * e - b * d
    det = (a * c00 + b * c10) + c * c20
    adj = torch.stack([torch.stack([c00, c01, c02], -1),
                       torch.stack([c10, c11, c12], -1),
                       torch.stack([c20, c21, c22], -1)], -2)
    return adj / det[..., None, None]


def get_pixel_grids(height, width):
    """homography_warping.py:8-17 -> (x, y) pixel-centre coordinates, each (H*W,)."""
    xs = tf_ops.linspace(0.5, float(width) - 0.5, width)
    ys = tf_ops.linspace(0.5, float(height) - 0.5, height)
    yy, xx = torch.meshgrid(ys, xs, indexing='ij')
    return xx.reshape(-1), yy.reshape(-1)


def interpolate(image, x, y, output_mask=False, method='bilinear'):
    """homography_warping.py:31-104.  image (B,H,W,C); x,y flat (B*H*W,)."""
    B, H, W, C = image.shape
    x = x - 0.5
    y = y - 0.5
    valid = (x >= 0) & (y >= 0) & (x < float(W - 1)) & (y < float(H - 1))
    valid = valid & ~torch.isnan(x) & ~torch.isnan(y)
    b = torch.arange(B).repeat_interleave(H * W)
    vi = valid.to(torch.int64)
    if method == 'nearest':
        # tf.round half-to-even, int cast, invalid -> index 0 (value NOT masked: quirk C4)
        x0 = tf_ops.tf_round(x).to(torch.int64) * vi
        y0 = tf_ops.tf_round(y).to(torch.int64) * vi
        out = image[b, y0, x0]
        return (out, valid) if output_mask else out
    vm = valid.to(x.dtype)
    # floor and int-cast BEFORE masking, multiply by the mask, then clip (reference order :59-75).
    # A non-finite coordinate times 0.0 stays NaN, exactly as tf.multiply does (:64-65).
    x0 = torch.floor(x).to(torch.int64)
    y0 = torch.floor(y).to(torch.int64)
    x1 = (x0 + 1) * vi
    y1 = (y0 + 1) * vi
    x0 = x0 * vi
    y0 = y0 * vi
    x = x * vm
    y = y * vm
    x0 = x0.clamp(0, W - 1)
    x1 = x1.clamp(0, W - 1)
    y0 = y0.clamp(0, H - 1)
    y1 = y1.clamp(0, H - 1)
    pa = image[b, y0, x0]
    pb = image[b, y0, x1]
    pc = image[b, y1, x0]
    pd = image[b, y1, x1]
    x0f, x1f, y0f, y1f = x0.to(x.dtype), x1.to(x.dtype), y0.to(x.dtype), y1.to(x.dtype)
    wa = ((y1f - y) * (x1f - x)).unsqueeze(1)
    wb = ((y1f - y) * (x - x0f)).unsqueeze(1)
    wc = ((y - y0f) * (x1f - x)).unsqueeze(1)
    wd = ((y - y0f) * (x - x0f)).unsqueeze(1)
    out = ((wa * pa + wb * pb) + wc * pc) + wd * pd
    return (out, valid) if output_mask else out


def _split_cam(cam):
    """cam (B,2,4,4) -> R (B,3,3), t (B,3,1), K (B,3,3)."""
    return cam[:, 0, :3, :3], cam[:, 0, :3, 3:4], cam[:, 1, :3, :3]


def get_homographies(left_cam, right_cam, depth_num, depth_start, depth_interval):
    """homography_warping.py:179-227 -> (B, D, 3, 3).

    H_d = K_r R_r (I - (c_r - c_l) n_l^T * delta_d) R_l^T K_l^-1,
    delta_d = depth_start + d*interval (inverse depth: multiply, :215-216).
    """
    R_l, t_l, K_l = _split_cam(left_cam)
    R_r, t_r, K_r = _split_cam(right_cam)
    B = R_l.shape[0]
    d = torch.arange(depth_num, dtype=torch.float32)
    depth = depth_start.reshape(B, 1) + d.reshape(1, -1) * depth_interval.reshape(B, 1)   # (B,D)
    K_l_inv = inv3(K_l)
    R_l_T = R_l.transpose(1, 2)
    R_r_T = R_r.transpose(1, 2)
    fronto = R_l[:, 2:3, :]                       # (B,1,3)
    c_l = -mm3(R_l_T, t_l)
    c_r = -mm3(R_r_T, t_r)
    c_rel = c_r - c_l                             # (B,3,1)
    temp = c_rel * fronto                         # outer product (B,3,3)
    eye = torch.eye(3, dtype=torch.float32).reshape(1, 1, 3, 3)
    dm = depth.reshape(B, depth_num, 1, 1)
    if INVERSE_DEPTH:
        mid0 = eye - temp[:, None] * dm
    else:
        mid0 = eye - temp[:, None] / dm
    mid1 = mm3(R_l_T, K_l_inv)[:, None]           # (B,1,3,3)
    mid2 = mm3(mid0, mid1)
    return mm3(K_r[:, None], mm3(R_r[:, None], mid2))


def warp_coords(homography, height, width):
    """The projective part of homography_warping (:237-257): -> x,y flat (B*H*W,)."""
    px, py = get_pixel_grids(height, width)
    h = homography
    B = h.shape[0]

    def row(i):
        return (h[:, i, 0:1] * px[None] + h[:, i, 1:2] * py[None]) + h[:, i, 2:3]
    xa, ya, dv = row(0), row(1), row(2)
    dv = dv + (dv == 0.0).to(torch.float32) * 1e-7
    return (xa / dv).reshape(-1), (ya / dv).reshape(-1)


def homography_warping(image, homography, method='bilinear', output_mask=False):
    """homography_warping.py:230-271.  image (B,H,W,C), homography (B,3,3)."""
    B, H, W, C = image.shape
    x, y = warp_coords(homography, H, W)
    res = interpolate(image, x, y, output_mask=output_mask, method=method)
    if output_mask:
        return res[0].reshape(B, H, W, C), res[1].reshape(B, H, W, 1)
    return res.reshape(B, H, W, C)


def _relative_pose(left_cam, right_cam):
    """mat = K_r R_r R_l^T K_l^-1 ; vec = K_r R_r c_l + K_r t_r  (:123-146, :290-313)."""
    R_l, t_l, K_l = _split_cam(left_cam)
    R_r, t_r, K_r = _split_cam(right_cam)
    K_l_inv = inv3(K_l)
    R_l_T = R_l.transpose(1, 2)
    c_l = -mm3(R_l_T, t_l)
    mat = mm3(K_r, mm3(R_r, mm3(R_l_T, K_l_inv)))
    vec = mm3(K_r, mm3(R_r, c_l)) + mm3(K_r, t_r)
    return mat, vec


def homography_warping_by_depth(image, left_cam, right_cam, depth_image, output_mask=False, method='bilinear'):
    """homography_warping.py:108-176: p' ~ M p + v * delta(p) (inverse depth: multiply :149-150)."""
    B, H, W, C = image.shape
    mat, vec = _relative_pose(left_cam, right_cam)
    px, py = get_pixel_grids(H, W)
    dflat = depth_image.reshape(B, 1, H * W)
    v = vec * dflat if INVERSE_DEPTH else vec / dflat      # (B,3,HW)

    def row(i):
        return ((mat[:, i, 0:1] * px[None] + mat[:, i, 1:2] * py[None]) + mat[:, i, 2:3]) + v[:, i]
    xa, ya, dv = row(0), row(1), row(2)
    x = (xa / dv).reshape(-1)
    y = (ya / dv).reshape(-1)
    res = interpolate(image, x, y, output_mask=output_mask, method=method)
    if output_mask:
        return res[0].reshape(B, H, W, C), res[1].reshape(B, H, W, 1)
    return res.reshape(B, H, W, C)


def transform_depth(left_depth, left_cam, right_cam):
    """homography_warping.py:275-326 (quirk C15: clip with own max, re-mask with INPUT validity)."""
    shape = left_depth.shape
    B, H, W = shape[0], shape[1], shape[2]
    mat, vec = _relative_pose(left_cam, right_cam)
    px, py = get_pixel_grids(H, W)
    d = left_depth
    if INVERSE_DEPTH:
        valid = d > 1e-10
        d = torch.clamp(d, min=1e-10)
        d = torch.minimum(d, left_depth.max())
        d = 1.0 / d
        d = d * valid.to(d.dtype)
    dflat = d.reshape(B, 1, H * W)
    gx, gy, gz = px[None] * dflat[:, 0], py[None] * dflat[:, 0], dflat[:, 0]
    dz = ((mat[:, 2, 0:1] * gx + mat[:, 2, 1:2] * gy) + mat[:, 2, 2:3] * gz) + vec[:, 2]
    out = dz.reshape(shape)
    if INVERSE_DEPTH:
        out = torch.minimum(torch.clamp(out, min=1e-10), out.max())
        out = 1.0 / out
        out = out * valid.to(out.dtype)
    return out


def get_visual_hull(depth_images, cams, depth_num, depth_start, depth_interval, ref_id=0, view_num=2):
    """homography_warping.py:329-387.  depth_images (B,N,H,W) -> (B,D,H,W,1).

    Quirk C6: the non-reference views are taken as id_reorder[1:] of
    range(view_num) with 0 and ref_id swapped, whatever the current source is.
    """
    B, N, H, W = depth_images.shape
    ids = list(range(view_num))
    ids[0] = ref_id
    ids[ref_id] = 0
    ref_cam = cams[:, ref_id]
    ref_depth = depth_images[:, ref_id]
    homos, trans = [], []
    for vi in ids[1:]:
        view_cam = cams[:, vi]
        homos.append(get_homographies(ref_cam, view_cam, depth_num, depth_start, depth_interval))
        trans.append(transform_depth(depth_images[:, vi], view_cam, ref_cam))
    hull = []
    for di in range(depth_num):
        cur = depth_start + depth_interval * float(di)          # (B,)
        sl = cur.reshape(B, 1, 1) * torch.ones(B, H, W)
        vm = (ref_depth > 0).to(torch.float32)
        if INVERSE_DEPTH:
            s = vm * (ref_depth > sl).to(torch.float32)
        else:
            s = vm * (sl > ref_depth).to(torch.float32)
        for k in range(view_num - 1):
            wd = homography_warping(trans[k].unsqueeze(-1), homos[k][:, di], method='nearest').squeeze(-1)
            vm2 = (wd > 0).to(torch.float32)
            if INVERSE_DEPTH:
                s = s + vm2 * (wd > sl).to(torch.float32)
            else:
                s = s + vm2 * (sl > wd).to(torch.float32)
        hull.append(s)
    hull = torch.stack(hull, 1) / float(view_num)
    return hull.unsqueeze(-1)
