"""Per-phase tick counts of aanet_b.hip's two roles (development build: bash tools_dev/build_variant.sh abdbg aanet_b -DATVS_AB_DEBUG;
ATVS_LIB=tools_dev/_dbg/lib_abdbg.so python tools_dev/phase_ab.py [views])"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import atvsnet_amd  # noqa: F401
from atvsnet_amd import ops, _lib, variables
from atvsnet_amd.cnn_wrapper.atvsnet import AttAggregation_keepchannel
nv = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device('cuda:0')
variables.default_store().init_synthetic(1234)
x = torch.randn((nv, 192, 128, 160, 8), device=dev)
for _ in range(3):
    AttAggregation_keepchannel({'data': x}, is_training=True).get_output()
torch.cuda.synchronize()
buf = np.zeros(2048 * 8, np.uint64)
assert _lib.lib().atvs_debug_read_ab(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf.reshape(-1, 8, 8).astype(np.float64)                # [workgroup][wavefront][counter]
t = t[t[:, 0, 7] > 0]
ns = t[:, 0, 7].mean()
for role, sl, names in (('multiplying wavefronts', slice(0, 4), ['loop top (zero, first fragments)', 'K loop', 'epilogue + hand-off write', 'barrier']),
                        ('staging / combining wavefronts', slice(4, 8), ['split + LDS write', 'next halo requests', 'hand-off read', 'combine (per stage average)', 'barrier', 'wait for the halo (vmcnt 0)'])):
    r = t[:, sl, :].reshape(-1, 8)
    tot = r[:, :len(names)].sum(1).mean()
    print('%s: %.1f stages each, %.0f ticks per stage' % (role, ns, tot / ns))
    for i, n in enumerate(names):
        print('   %-42s %8.0f per stage (%.1f%%)' % (n, r[:, i].mean() / ns, 100 * r[:, i].mean() / tot))
