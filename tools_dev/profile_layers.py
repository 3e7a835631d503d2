"""Per-layer convolution timing (HIP events around each launch, eager, one stream) at a bench workload.
usage: profile_layers.py [views] [H W D]   ->  every convolution launch aggregated by (weight key, input shape)."""
import collections, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import atvsnet_amd
from atvsnet_amd import ops, synthetic, variables
from atvsnet_amd.atvsnet import example as ex
views = int(sys.argv[1]) if len(sys.argv) > 1 else 5
H, W, D = [int(v) for v in sys.argv[2:5]] if len(sys.argv) > 4 else (512, 640, 192)
dev = torch.device('cuda:0')
variables.default_store().init_synthetic(1234)
imgs, cams = synthetic.make_inputs(views, H, W, D)
imgs, cams = torch.from_numpy(imgs).to(dev), torch.from_numpy(cams).to(dev)
run = (lambda: ex.infer_twoview(imgs, cams, D)) if views == 2 else (lambda: ex.infer_multiview(imgs, cams, D, view_streams=False))
run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); run(); e1.record(); torch.cuda.synchronize()
print('eager single-stream step ms', e0.elapsed_time(e1))
ops.watch('*')
run()
ev = ops.watch(None)
agg = collections.OrderedDict()
for key, shp, cout, ms in ev:
    k = (str(key), shp, cout)
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += ms
tot = sum(v[1] for v in agg.values())
print('total conv ms', tot, 'launches', sum(v[0] for v in agg.values()))
fam = collections.OrderedDict()
for k, v in agg.items():
    key, shp, cout = k
    kind = '2d' if shp[0] == 1 else '3d'
    f = fam.setdefault((kind, shp[1:3] if kind == '2d' else shp[:3]), [0, 0.0])
    f[0] += v[0]; f[1] += v[1]
for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print('FAMILY %-28s n %4d ms %7.3f' % (k, v[0], v[1]))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    key, shp, cout = k
    print('%-60s in %-20s cout %3d  n %3d  ms %7.3f  avg %7.3f' % (key[:60], shp, cout, v[0], v[1], v[1] / v[0]))
