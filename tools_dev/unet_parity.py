"""Layer-by-layer GPU vs oracle comparison of the stacked U-Net at a small depth count."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import atvsnet_amd
from atvsnet_amd import variables, ops
from atvsnet_amd.cnn_wrapper.atvsnet import StackedUNet_prob
from oracle import nets
D = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda:0')
store = variables.default_store().init_synthetic(1234)
W = {k: torch.from_numpy(v) for k, v in store.host.items()}
g = torch.Generator().manual_seed(3)
data = torch.randn(1, D, 32, 40, 64, generator=g)
L = {}
want_p, want_f = nets.stacked_unet_prob(data, W, L)
net = StackedUNet_prob({'data': data.to(dev)}, is_training=True)
for name, w in L.items():
    if name == 'data':
        continue
    gt = net.get_output_by_name(name)
    if hasattr(gt, 'materialize'):
        gt = gt.materialize()
    gt = gt.cpu()
    print('%-22s %-24s max rel err %.3e' % (name, tuple(w.shape), float((gt - w).abs().max()) / (float(w.abs().max()) + 1e-30)))
