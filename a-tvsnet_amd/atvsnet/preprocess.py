"""Data formats either side of the path: camera text files, PFM maps, pair lists, image scaling / cropping.

Same names, arguments and results as the helpers of /root/reference/atvsnet/preprocess.py that the ETH3D
driver (eval_pointcloud.py) uses -- center_image :20-25, scale_camera :27-37, scale_mvs_camera :39-43,
scale_image :45-50, scale_mvs_input :52-62, crop_mvs_input :64-94, mask_depth_image :96-103, load_cam
:105-142, write_cam :144-163, load_pfm :165-197, write_pfm :200-229, gen_pipeline_mvs_list :233-259 -- written
on numpy only (no OpenCV, no TensorFlow file_io).  FLAGS.view_num / max_h / max_w / max_d are read where the
reference reads them.

cv2.resize is re-implemented (`scale_image`): half-pixel-centre sampling; for 8-bit images the 11-bit
fixed-point weights and the two-pass rounding of OpenCV's vectorised resizer.  OpenCV is not installed here,
so that restatement is UNPINNED: expect at most one grey level of difference per pixel.
"""
from __future__ import print_function

import math
import os
import re
import sys

import numpy as np

from ..flags import FLAGS

_COEF_ONE = 1 << 11        # INTER_RESIZE_COEF_SCALE


def center_image(img):
    """Zero-mean / unit-variance per channel over the image plane -> float32."""
    x = np.asarray(img, dtype=np.float32)
    mu = x.mean(axis=(0, 1), keepdims=True)
    sd = np.sqrt(x.var(axis=(0, 1), keepdims=True))
    return (x - mu) / (sd + 0.00000001)


def scale_camera(cam, scale=1):
    """Intrinsics of an image resized by `scale`: fx, fy, cx, cy scale; everything else is copied."""
    out = np.array(cam, copy=True)
    for r, c in ((0, 0), (1, 1), (0, 2), (1, 2)):
        out[1][r][c] = cam[1][r][c] * scale
    return out


def scale_mvs_camera(cams, scale=1):
    for v in range(FLAGS.view_num):
        cams[v] = scale_camera(cams[v], scale=scale)
    return cams


def _resize_taps(n_dst, n_src, step):
    """INTER_LINEAR: left source index and right-hand weight of every destination sample."""
    pos = (np.arange(n_dst, dtype=np.float64) + 0.5) * step - 0.5
    left = np.floor(pos).astype(np.int64)
    frac = (pos - left).astype(np.float32)
    before, after = left < 0, left >= n_src - 1
    frac[before | after] = 0.0
    left[before] = 0
    left[after] = n_src - 1
    return left, np.minimum(left + 1, n_src - 1), frac


def scale_image(image, scale=1, interpolation='linear'):
    """cv2.resize(image, None, fx=scale, fy=scale, INTER_LINEAR | INTER_NEAREST)."""
    if interpolation not in ('linear', 'nearest'):
        return None
    src = np.asarray(image)
    h, w = src.shape[:2]
    H, W = int(np.rint(h * scale)), int(np.rint(w * scale))
    if min(H, W) < 1:
        raise ValueError('scale_image: scale %r empties a %dx%d image' % (scale, w, h))
    step = 1.0 / scale
    if interpolation == 'nearest':
        rows = np.minimum((np.arange(H) * step).astype(np.int64), h - 1)
        cols = np.minimum((np.arange(W) * step).astype(np.int64), w - 1)
        return src[rows][:, cols]
    y0, y1, wy = _resize_taps(H, h, step)
    x0, x1, wx = _resize_taps(W, w, step)
    px = src.reshape(h, w, -1)
    if src.dtype == np.uint8:
        q = lambda t: np.rint(t * _COEF_ONE).astype(np.int64)                     # noqa: E731
        ax0, ax1, by0, by1 = q(1.0 - wx), q(wx), q(1.0 - wy), q(wy)
        p = px.astype(np.int64)
        horiz = p[:, x0] * ax0[None, :, None] + p[:, x1] * ax1[None, :, None]     # 8.11 fixed point
        top, bot = horiz[y0] >> 4, horiz[y1] >> 4
        acc = ((by0[:, None, None] * top) >> 16) + ((by1[:, None, None] * bot) >> 16)
        out = np.clip((acc + 2) >> 2, 0, 255).astype(np.uint8)
    else:
        p = px.astype(np.float32)
        cx, cy = wx[None, :, None], wy[:, None, None]
        horiz = p[:, x0] * (1.0 - cx) + p[:, x1] * cx
        out = (horiz[y0] * (1.0 - cy) + horiz[y1] * cy).astype(src.dtype)
    return out.reshape((H, W) + src.shape[2:])


def scale_mvs_input(images, cams, depth_image=None, scale=1):
    """Every view's image and camera by `scale`; a depth image (nearest) when given."""
    for v in range(FLAGS.view_num):
        images[v] = scale_image(images[v], scale=scale)
        cams[v] = scale_camera(cams[v], scale=scale)
    if depth_image is None:
        return images, cams
    return images, cams, scale_image(depth_image, scale=scale, interpolation='nearest')


def _fit(size, limit, base):
    """Network-compatible extent of one axis: capped at `limit`, otherwise rounded up to a multiple of `base`."""
    return limit if size > limit else int(math.ceil(size / base) * base)


def crop_mvs_input(images, cams, depth_image=None, base_image_size=32):
    """Centre-crop every view to (<= max_h, <= max_w, multiples of base_image_size) and move the principal point."""
    y0 = x0 = y1 = x1 = 0
    for v in range(FLAGS.view_num):
        h, w = images[v].shape[0:2]
        nh, nw = _fit(h, FLAGS.max_h, base_image_size), _fit(w, FLAGS.max_w, base_image_size)
        y0, x0 = int(math.ceil((h - nh) / 2)), int(math.ceil((w - nw) / 2))
        y1, x1 = y0 + nh, x0 + nw
        images[v] = images[v][y0:y1, x0:x1]
        cams[v][1][0][2] -= x0
        cams[v][1][1][2] -= y0
    if depth_image is None:
        return images, cams
    return images, cams, depth_image[y0:y1, x0:x1]          # the last view's window, as in the reference


def mask_depth_image(depth_image, min_depth, max_depth):
    """Depths outside (min_depth, max_depth] become 0 (THRESH_TOZERO then THRESH_TOZERO_INV) -> (H,W,1)."""
    d = np.array(depth_image, copy=True)
    d[np.logical_or(~(d > min_depth), d > max_depth)] = 0
    return d[:, :, None]


def load_cam(file, interval_scale=1):
    """Open text file 'extrinsic <16> intrinsic <9> [depth_min interval [planes [depth_max]]]' -> (2,4,4) float64:
    [0] = extrinsic, [1][:3,:3] = intrinsic, [1][3] = (depth_min, interval * interval_scale, planes, depth_max)."""
    tok = file.read().split()
    cam = np.zeros((2, 4, 4))
    cam[0] = np.array(tok[1:17], dtype=np.float64).reshape(4, 4)
    cam[1, :3, :3] = np.array(tok[18:27], dtype=np.float64).reshape(3, 3)
    n = len(tok)
    if n in (29, 30, 31):
        dmin, step = float(tok[27]), float(tok[28]) * interval_scale
        planes = float(tok[29]) if n >= 30 else FLAGS.max_d
        dmax = float(tok[30]) if n == 31 else dmin + step * planes
        cam[1, 3] = (dmin, step, planes, dmax)
    return cam


def write_cam(file, cam):
    """The text form load_cam reads (31 words)."""
    row = lambda values: ''.join(str(v) + ' ' for v in values) + '\n'           # noqa: E731
    text = 'extrinsic\n' + ''.join(row(cam[0][i][:4]) for i in range(4)) + '\n'
    text += 'intrinsic\n' + ''.join(row(cam[1][i][:3]) for i in range(3))
    text += '\n' + ' '.join(str(cam[1][3][k]) for k in range(4)) + '\n'
    with open(file, 'w') as f:
        f.write(text)


def load_pfm(file):
    """Open BINARY file -> float32 (H,W) ('Pf') or (H,W,3) ('PF'), first row = top of the image."""
    line = lambda: file.readline().decode('latin-1')                            # noqa: E731
    kind = line().rstrip()
    if kind not in ('PF', 'Pf'):
        raise Exception('Not a PFM file.')
    dims = re.match(r'^(\d+)\s(\d+)\s$', line())
    if not dims:
        raise Exception('Malformed PFM header.')
    width, height = int(dims.group(1)), int(dims.group(2))
    little = float(line().rstrip()) < 0          # the sign of the scale is the byte order
    data = np.frombuffer(file.read(), '<f4' if little else '>f4')
    data = data.reshape((height, width, 3) if kind == 'PF' else (height, width))
    return data[::-1].copy()                     # stored bottom row first


def write_pfm(file, image, scale=1):
    """float32 (H,W) | (H,W,1) -> 'Pf', (H,W,3) -> 'PF'; rows bottom-up, scale sign = byte order."""
    if image.dtype.name != 'float32':
        raise Exception('Image dtype must be float32.')
    if image.ndim == 3 and image.shape[2] == 3:
        magic = b'PF\n'
    elif image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1):
        magic = b'Pf\n'
    else:
        raise Exception('Image must have H x W x 3, H x W x 1 or H x W dimensions.')
    order = image.dtype.byteorder
    if order == '<' or (order == '=' and sys.byteorder == 'little'):
        scale = -scale
    header = magic + ('%d %d\n%f\n' % (image.shape[1], image.shape[0], scale)).encode('latin-1')
    with open(file, 'wb') as f:
        f.write(header + np.ascontiguousarray(image[::-1]).tobytes())


def gen_pipeline_mvs_list(dense_folder):
    """<dense_folder>/pair.txt -> per reference image [ref.jpg, ref_cam.txt, src1.jpg, src1_cam.txt, ...] with at
    most FLAGS.view_num - 1 sources.  pair.txt: count, then per entry `ref_id  n_src  (src_id score) x n_src`."""
    with open(os.path.join(dense_folder, 'pair.txt')) as f:
        tok = iter(f.read().split())
    pair = lambda idx: [os.path.join(dense_folder, 'images', '%08d.jpg' % idx),       # noqa: E731
                        os.path.join(dense_folder, 'cams', '%08d_cam.txt' % idx)]
    out = []
    for _ in range(int(next(tok))):
        paths = pair(int(next(tok)))
        n_src = int(next(tok))
        srcs = [(int(next(tok)), next(tok))[0] for _ in range(n_src)]
        for idx in srcs[:min(FLAGS.view_num - 1, n_src)]:
            paths += pair(idx)
        out.append(paths)
    return out
