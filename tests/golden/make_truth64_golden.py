"""Generates tests/golden/truth64_mid.npz: the two-view pipeline at 320x256, D=64 evaluated with the
CPU oracle's NETWORKS in float64 (geometry -- the sampling coordinates -- stays float32 so that the same
pixels are sampled), i.e. the value the float32 pipelines approximate, plus the float32 oracle's own
result.  The -m gpu test checks that the HIP path is as close to the float64 value as the float32 oracle
is (both differ from it only by float32 rounding / summation order, amplified by 31 batch-normalised
layers and a peaked soft-argmin).  Run from the repository root (takes ~1 min on 8 cores).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import atvsnet_amd                                     # noqa: E402,F401
from atvsnet_amd import synthetic, variables           # noqa: E402
from oracle import homography_warping as G             # noqa: E402
from oracle import model as OM                         # noqa: E402
from oracle import tf_ops as T                         # noqa: E402

torch.set_num_threads(8)
store = variables.VariableStore().init_synthetic(1234)
W32 = {k: torch.from_numpy(v) for k, v in store.host.items()}
W64 = {k: v.double() for k, v in W32.items()}
imgs, cams = synthetic.make_inputs(2, 256, 320, 64)
imgs, cams = torch.from_numpy(imgs), torch.from_numpy(cams)
d32 = OM.run_twoview(imgs, cams, W32, 64)


def f32_boundary(fn):
    """Run a geometry function in float32 on float64 tensors (coordinates must not change)."""
    def wrapped(*args, **kw):
        args = [a.float() if isinstance(a, torch.Tensor) and a.dtype == torch.float64 else a for a in args]
        out = fn(*args, **kw)
        if isinstance(out, tuple):
            return tuple(o.double() if o.dtype == torch.float32 else o for o in out)
        return out.double() if out.dtype == torch.float32 else out
    return wrapped


saved = {}
for name in ('homography_warping', 'homography_warping_by_depth', 'transform_depth', 'get_visual_hull'):
    saved[name] = getattr(G, name)
    setattr(G, name, f32_boundary(saved[name]))
orig_linspace = T.linspace
T.linspace = lambda a, b, n: orig_linspace(float(a), float(b), n).double()
try:
    d64 = OM.run_twoview(imgs.double(), cams, W64, 64)
finally:
    for name, fn in saved.items():
        setattr(G, name, fn)
    T.linspace = orig_linspace
rel = float(((d32.double() - d64).abs() / d64.abs()).mean())
print('float32 oracle vs float64 networks: rel-L1 %.3e' % rel)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'truth64_mid.npz'),
                    depth64=d64[0, ..., 0].numpy().astype(np.float32), depth32=d32[0, ..., 0].numpy(),
                    oracle32_rel_l1=np.float64(rel))
