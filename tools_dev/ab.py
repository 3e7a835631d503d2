"""A/B of pipeline variants in ONE process (interleaved rounds): usage: ab.py "<setup A>" "<setup B>" ...
Each setup is python code run before that variant's graph is captured; `kw` (dict) in it sets GraphedInference kwargs,
e.g.  ab.py "kw=dict(batched=False)" "kw=dict(batched=True)"."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import atvsnet_amd
from atvsnet_amd import ops, synthetic, variables
from atvsnet_amd.atvsnet import example as ex, model
dev = torch.device('cuda:0')
variables.default_store().init_synthetic(1234)
views, H, W, D = [int(v) for v in os.environ.get('ATVS_AB_CFG', '5,512,640,192').split(',')]      # views,H,W,D
imgs, cams = synthetic.make_inputs(views, H, W, D)
imgs, cams = torch.from_numpy(imgs).to(dev), torch.from_numpy(cams).to(dev)
variants = []
for code in sys.argv[1:]:
    ns = {'ops': ops, 'ex': ex, 'model': model, 'kw': {}}
    exec(code, ns)
    variants.append((code, ex.GraphedInference(imgs, cams, D, **ns['kw'])))
res = {c: [] for c, _ in variants}
for rnd in range(6):
    for code, g in variants:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            g()
        torch.cuda.synchronize()
        res[code].append((time.perf_counter() - t0) / 4 * 1e3)
for code, v in res.items():
    v = sorted(v)
    print('%-60s median %.2f ms  min %.2f' % (code[:60], v[len(v) // 2], v[0]))
