"""A small synthetic multi-view scene for the fusion tests: a tilted plane seen by N pinhole cameras, with exact depth
maps (ray / plane intersection), smooth colour images and unit normals."""
import numpy as np


def make_scene(n_views=4, rows=48, cols=64, seed=0):
    rng = np.random.default_rng(seed)
    K = np.array([[60.0, 0, cols / 2.0], [0, 60.0, rows / 2.0], [0, 0, 1]])
    n = np.array([0.1, -0.05, -1.0])
    n /= np.linalg.norm(n)
    d0 = 5.0                                   # plane: n . X + d0 = 0  (in front of the cameras, z ~ 5)
    Ps, depths, normals, images = [], [], [], []
    ys, xs = np.meshgrid(np.arange(rows, dtype=np.float64), np.arange(cols, dtype=np.float64), indexing='ij')
    for i in range(n_views):
        ang = np.deg2rad(3.0 * (i - (n_views - 1) / 2.0))
        R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
        C = np.array([0.4 * (i - (n_views - 1) / 2.0), 0.05 * i, 0.0])
        t = -R @ C
        P = K @ np.concatenate([R, t[:, None]], 1)
        # ray of pixel (x, y): X = C + s * R^T K^-1 (x, y, 1); depth (z in the camera frame) = s
        rays = np.stack([xs, ys, np.ones_like(xs)], -1) @ np.linalg.inv(K).T @ R
        s = -(n @ C + d0) / (rays @ n)
        Ps.append(P)
        depths.append(s.astype(np.float32))
        normals.append(np.broadcast_to((R @ n).astype(np.float32), (rows, cols, 3)).copy())
        X = C + s[..., None] * rays
        img = 127.0 + 100.0 * np.stack([np.sin(X[..., 0] * 2.0), np.cos(X[..., 1] * 3.0), np.sin(X[..., 0] + X[..., 1])], -1)
        images.append(np.clip(img + rng.uniform(0, 4, img.shape), 0, 255).astype(np.uint8))
    return (np.stack(Ps), np.stack(depths), np.stack(normals), np.stack(images), n, d0)
