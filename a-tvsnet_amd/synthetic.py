"""Seeded synthetic inputs of BASELINE.json's shapes.

Images: float32 BGR-like in [0, 255], smooth random texture (8 random-phase
2-D sinusoids per channel + U(0, 8) noise) so that warps are non-degenerate.
Cameras: quarter-scale intrinsics like the reference's examples
(example/*/i_cam.npy), reference extrinsic = identity, sources rotated about y
by +-2 deg * ceil(i/2) and shifted along x; cam[1,3,0:2] = (inverse-depth
start, interval) with the range 0.05 .. 0.36 split into D hypotheses.
"""
import numpy as np


def make_images(num_views, height, width, seed=0):
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(height, dtype=np.float64), np.arange(width, dtype=np.float64), indexing='ij')
    imgs = np.zeros((num_views, height, width, 3), np.float32)
    # one shared scene texture, shifted per view, so that views correlate
    comps = []
    for c in range(3):
        comps.append([(rng.uniform(0.01, 0.12), rng.uniform(0.01, 0.12), rng.uniform(0, 2 * np.pi),
                       rng.uniform(0.3, 1.0)) for _ in range(8)])
    for v in range(num_views):
        shift = 3.0 * ((v + 1) // 2) * (-1.0) ** v
        for c in range(3):
            acc = np.zeros((height, width))
            norm = 0.0
            for fx, fy, ph, amp in comps[c]:
                acc += amp * np.sin(fx * (xx + shift) + fy * yy + ph)
                norm += amp
            img = 127.5 + 119.0 * acc / norm + rng.uniform(0.0, 8.0, size=(height, width))
            imgs[v, :, :, c] = np.clip(img, 0.0, 255.0)
    return imgs


def make_cams(num_views, height, width, depth_num, d_min=0.05, d_max=0.36):
    """cams (N,2,4,4) float32 for feature size (height/4, width/4)."""
    h, w = height // 4, width // 4
    cams = np.zeros((num_views, 2, 4, 4), np.float64)
    for i in range(num_views):
        k = (i + 1) // 2
        sgn = (-1.0) ** i
        ang = np.deg2rad(2.0 * k) * sgn
        R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
        t = np.array([-0.25 * k * sgn, 0.0, 0.0])
        E = np.eye(4)
        E[:3, :3] = R
        E[:3, 3] = t
        cams[i, 0] = E
        cams[i, 1, :3, :3] = np.array([[0.89 * w, 0, w / 2.0], [0, 0.89 * w, h / 2.0], [0, 0, 1]])
        cams[i, 1, 3, 0] = d_min
        cams[i, 1, 3, 1] = (d_max - d_min) / depth_num
    return cams.astype(np.float32)


def make_inputs(num_views, height, width, depth_num, seed=0):
    """-> images (1,N,H,W,3) float32, cams (1,N,2,4,4) float32 (numpy)."""
    return make_images(num_views, height, width, seed)[None], make_cams(num_views, height, width, depth_num)[None]
