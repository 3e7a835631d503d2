"""N pairs of depth maps in flight on disjoint halves of every XCD (example.PipelinedInference(co_resident='cu_split')), two different
input sets alternating over the slots, EVERY result compared bit for bit with its one-at-a-time value:
python tools_dev/soak_cu_split.py [cfg3|cfg2] [pairs] [slots]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import atvsnet_amd  # noqa: F401
from atvsnet_amd import synthetic, variables
from atvsnet_amd.atvsnet import example as ex
which = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
slots = int(sys.argv[3]) if len(sys.argv) > 3 else 2
views, W, H, D = {'cfg3': (5, 640, 512, 192), 'cfg2': (2, 640, 512, 192)}[which]
dev = torch.device('cuda:0')
variables.default_store().init_synthetic(1234)
sets = []
for seed in (0, 7):
    i, c = synthetic.make_inputs(views, H, W, D, seed=seed)
    sets.append((torch.from_numpy(i).pin_memory(), torch.from_numpy(c).pin_memory()))       # host tensors: the slots' streams copy them
p = ex.PipelinedInference(sets[0][0].to(dev), sets[0][1].to(dev), D, slots=slots, co_resident='cu_split')
want = []
for im, cm in sets:
    want.append(p.result(p.submit(im, cm), host=True).clone())
    torch.cuda.synchronize()
assert not torch.equal(want[0], want[1])
bad, t0 = 0, time.time()
for r in range(n):
    ts = [(p.submit(*sets[(r + k) % 2]), (r + k) % 2) for k in range(slots)]
    for t, w in ts:
        if not torch.equal(p.result(t, host=True), want[w]):
            bad += 1
            print('round %d slot %d differs' % (r, t), flush=True)
print('%s: %d rounds of %d maps in flight (cu_split) in %.1f s, %d results differ from the one-at-a-time value' % (which, n, slots, time.time() - t0, bad))
sys.exit(1 if bad else 0)
