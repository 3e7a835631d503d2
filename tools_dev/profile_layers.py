"""Per-layer convolution timing (HIP events around each launch) at the bench workload."""
import collections, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import atvsnet_amd
from atvsnet_amd import ops, synthetic, variables
from atvsnet_amd.atvsnet import example as ex
views = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device('cuda:0')
variables.default_store().init_synthetic(1234)
imgs, cams = synthetic.make_inputs(views, 512, 640, 192)
imgs, cams = torch.from_numpy(imgs).to(dev), torch.from_numpy(cams).to(dev)
ex.infer_multiview(imgs, cams, 192)
torch.cuda.synchronize()
ops.watch('*')
ex.infer_multiview(imgs, cams, 192)
ev = ops.watch(None)
agg = collections.OrderedDict()
for key, shp, cout, ms in ev:
    k = (str(key), shp, cout)
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += ms
tot = sum(v[1] for v in agg.values())
print('total conv ms', tot)
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    key, shp, cout = k
    vox = shp[0] * shp[1] * shp[2]
    print('%-52s in %-20s cout %3d  n %3d  ms %7.3f  avg %7.3f' % (key[:52], shp, cout, v[0], v[1], v[1] / v[0]))
