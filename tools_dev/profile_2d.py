import collections, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import atvsnet_amd
from atvsnet_amd import ops, synthetic, variables
from atvsnet_amd.atvsnet import model
dev = torch.device('cuda:0')
variables.default_store().init_synthetic(1234)
imgs, cams = synthetic.make_inputs(2, 512, 640, 192)
imgs = torch.from_numpy(imgs).to(dev)
model.TVSNet_feature_extraction(imgs, 0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); model.TVSNet_feature_extraction(imgs, 0); e1.record(); torch.cuda.synchronize()
print('tower total ms', e0.elapsed_time(e1))
ops.watch('*')
model.TVSNet_feature_extraction(imgs, 0)
ev = ops.watch(None)
agg = collections.OrderedDict()
for key, shp, cout, ms in ev:
    k = (shp, cout, 'k3' if 'conv2' in str(key) or 'conv0' in str(key) or 'fusion0' in str(key) or 'branch' in str(key) else 'k1')
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += ms
print('conv ms', sum(v[1] for v in agg.values()))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(k, 'n', v[0], 'ms %.3f avg %.3f' % (v[1], v[1] / v[0]))
