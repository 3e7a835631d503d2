"""Generates tests/golden/oracle_cfg1.npz: outputs of the CPU oracle (oracle/) on the seeded
synthetic inputs and weights at BASELINE.json config 1 (160x128 image, D=32): the two-view
pipeline and a 3-view multi-view pipeline.  The fixture freezes the oracle (a regression in
oracle/ shows up on CPU) and is what the -m gpu pipeline tests also compare with.
Run from the repository root:  python tests/golden/make_oracle_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import atvsnet_amd                                     # noqa: E402,F401
from atvsnet_amd import synthetic, variables           # noqa: E402
from oracle import model as OM                         # noqa: E402

torch.set_num_threads(8)
store = variables.VariableStore().init_synthetic(1234)
W = {k: torch.from_numpy(v) for k, v in store.host.items()}
out = {}
imgs, cams = synthetic.make_inputs(2, 128, 160, 32)
S = {}
d = OM.run_twoview(torch.from_numpy(imgs), torch.from_numpy(cams), W, 32, S)
out['twoview_depth'] = d[0, ..., 0].numpy()
out['twoview_depth_b2'] = S['depth_b2'][0, ..., 0].numpy()
out['twoview_depth_view'] = S['depth_view'][0, ..., 0].numpy()
out['twoview_ref_feature_c0'] = S['ref_feature'][0, ..., 0].numpy()
out['twoview_refined_prob_d7'] = S['refined_prob_vol'][0, 7].numpy()
imgs, cams = synthetic.make_inputs(3, 128, 160, 32)
S = {}
d = OM.run_multiview(torch.from_numpy(imgs), torch.from_numpy(cams), W, 32, S)
out['multiview3_depth'] = d[0, ..., 0].numpy()
out['multiview3_depth_agg_init'] = S['depth_agg_init'][0, ..., 0].numpy()
out['multiview3_cost_agg_d5'] = S['cost_volume_agg'][0, 5].numpy()
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'oracle_cfg1.npz'), **out)
print({k: (v.shape, float(np.abs(v).mean())) for k, v in out.items()})
