"""Stage-by-stage GPU vs oracle comparison of the two-view path at the smoke configuration."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import atvsnet_amd
from atvsnet_amd import variables, synthetic, ops
from atvsnet_amd.atvsnet import model as M, example as ex
from oracle import model as OM
D = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda:0')
store = variables.default_store().init_synthetic(1234)
W = {k: torch.from_numpy(v) for k, v in store.host.items()}
imgs, cams = synthetic.make_inputs(2, 128, 160, D)
imgs, cams = torch.from_numpy(imgs), torch.from_numpy(cams)
S = {}
want = OM.run_twoview(imgs, cams, W, D, S)
gi, gc = imgs.to(dev), cams.to(dev)
ds, di = ex.depth_range(gc)
def rel(a, b):
    a = a.materialize() if hasattr(a, 'materialize') else a
    return float((a.cpu() - b).abs().max()) / (float(b.abs().max()) + 1e-30)
rf, vf = M.TVSNet_feature_extraction(gi, 0), M.TVSNet_feature_extraction(gi, 1)
print('ref_feature', rel(rf, S['ref_feature']), 'view_feature', rel(vf, S['view_feature']))
cv = M.build_cost_volume(rf, vf, gc, D, ds, di, ref_id=0, view_id=1, lazy=False)
print('cost_volume', rel(cv, S['cost_volume']))
cvl = M.build_cost_volume(rf, vf, gc, D, ds, di, ref_id=0, view_id=1, lazy=True)
pv, filt = M.cost_volume_reasoning(cvl, output_filtered_cost=True)
print('prob_vol_b2', rel(pv, S['prob_vol_b2']), 'filtered', rel(filt, S['filtered_cost_volume']))
d = M.prob2depth(pv, D, ds, di)
print('depth_b2', rel(d, S['depth_b2']))
cvv = M.build_cost_volume(vf, rf, gc, D, ds, di, ref_id=1, view_id=0, lazy=True)
pvv = M.cost_volume_reasoning(cvv, output_filtered_cost=False)
dv = M.prob2depth(pvv, D, ds, di)
print('depth_view', rel(dv, S['depth_view']))
init = torch.stack([d, dv], dim=1)
_, pres = M.refinement(init, gc, D, ds, di, gi, pv, ref_id=0, view_id=1, view_homographies=None, num_depths=2, depth_ref_id=0, depth_view_id=1)
print('prob_residual', rel(pres, S['prob_residual']))
oinit = torch.stack([S['depth_b2'], S['depth_view']], 1)
_, pres2 = M.refinement(oinit.to(dev), gc, D, ds, di, gi, S['prob_vol_b2'].to(dev), ref_id=0, view_id=1, view_homographies=None, num_depths=2, depth_ref_id=0, depth_view_id=1)
print('prob_residual from oracle inputs', rel(pres2, S['prob_residual']))
refined = S['prob_vol_b2'] + S['prob_residual']
_, up = M.prob2depth_upsample(refined.to(dev), D, ds, di)
print('final from oracle refined volume', rel(up, want), 'rel-L1', float(((up.cpu() - want).abs() / want.abs()).mean()))
