"""Per-phase cycle counts of conv_xw.hip (development build: bash tools_dev/build_variant.sh dbg conv_xw -DATVS_XW_DEBUG;
ATVS_LIB=tools_dev/_dbg/lib_dbg.so python tools_dev/phase_xw.py [dominant|stack|refine|stem])"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import atvsnet_amd
from atvsnet_amd import ops, _lib
which = sys.argv[1] if len(sys.argv) > 1 else 'dominant'
dev = torch.device('cuda:0')
D, H, W = 192, 128, 160
rng = np.random.default_rng(0)
wt = lambda cin, cout: (rng.standard_normal((3, 3, 3, cin, cout)) * 0.1).astype(np.float32)   # noqa: E731
if which == 'dominant':
    G = 8
    x = torch.randn(G, D, H, W, 32, device=dev)
    pb, pb2 = torch.randn(G, H, W, 24, device=dev), torch.randn(G, H // 2, W // 2, 48, device=dev)
    w8, w16 = wt(32, 8), wt(32, 16)
    run = lambda: ops.conv_siblings(x, 'a8', w8, 'a16', w16, plane_bias=pb, plane_bias2=pb2, groups=G)   # noqa: E731
    mf = (384, 112)
elif which in ('dominantp', 'dominantpc'):           # as the pipeline launches it: the warped half chunk-planar (dominantpc: as fp16 pieces)
    G = 8
    x = torch.randn(G, 4, ops.planar_stride(D, H, W), device=dev)
    pb, pb2 = torch.randn(G, H, W, 24, device=dev), torch.randn(G, H // 2, W // 2, 48, device=dev)
    w8, w16 = wt(32, 8), wt(32, 16)
    run = lambda: ops.conv_siblings(x, 'a8', w8, 'a16', w16, plane_bias=pb, plane_bias2=pb2, groups=G, planar=(D, H, W), pieces=(which == 'dominantpc'))   # noqa: E731
    mf = (384, 112)
elif which == 'stack':
    G = 8
    xa, xb = torch.randn(G, D, H, W, 8, device=dev), torch.randn(G, D, H, W, 8, device=dev)
    par = torch.stack([torch.randn(G, 8) * 0.1, torch.rand(G, 8) + 0.5, torch.randn(G, 8) * 0.1], 1).to(dev).contiguous()
    v8, v16 = wt(8, 8), wt(8, 16)
    run = lambda: ops.conv_siblings(ops.PendingSum([ops.PendingBN(xa, par, True), ops.PendingBN(xb, par, True)]), 'b8', v8, 'b16', v16, groups=G)   # noqa: E731
    mf = (384, 112)
elif which == 'refine':
    G = 4
    x = torch.randn(G, D, H, W, 32, device=dev)
    par = torch.stack([torch.randn(G, 32) * 0.1, torch.rand(G, 32) + 0.5, torch.randn(G, 32) * 0.1], 1).to(dev).contiguous()
    w8, w16 = wt(32, 8), wt(32, 16)
    run = lambda: ops.conv_siblings(ops.PendingBN(x, par, True), 'c8', w8, 'c16', w16, groups=G)   # noqa: E731
    mf = (384, 112)
elif which == 'stemp':                                # the refinement's photo stem as the pipeline launches it: 16 channels as fp16 pieces
    G = 4
    ps = ops.planar_stride(D, H, W)
    xv = torch.zeros(G, 2, ps, device=dev)
    xv[..., :D * H * W * 8] = (torch.randn(G, 2, D * H * W * 16, device=dev) * 0.5).half().view(torch.float32)
    cst = torch.randn(G, H, W, 16, device=dev)
    sv = ops.SplitVolume(xv, cst, [('v', i) for i in range(16)] + [('c', i) for i in range(16)], planar=(D, H, W), pieces=True)
    u8 = wt(32, 8)
    buf = torch.zeros(G, 4, ps, device=dev)
    run = lambda: ops.conv_split_into_plane(sv, 'e8', u8, buf, 0, (D, H, W))   # noqa: E731
    mf = (384, 0)
else:
    G = 4
    x = torch.randn(G, D, H, W, 16, device=dev)
    pb = torch.randn(G, H, W, 24, device=dev)
    u8 = wt(16, 8)
    run = lambda: ops.conv(x, 'd8', u8, want_stats=True, plane_bias=pb, groups=G)   # noqa: E731
    mf = (384, 0)
for _ in range(int(os.environ.get('ITERS', 3))):
    run()
torch.cuda.synchronize()
buf = np.zeros(8192 * 8, np.uint64)
rc = _lib.lib().atvs_debug_read_xw(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0
b = buf.reshape(-1, 8).astype(np.float64)
if os.environ.get('XB'):
    # conv_xb.hip: 8 wavefronts per workgroup, rows (workgroup * 8 + wave); waves 0..3 multiply, 4..7 stage
    b = b[:512 * 8].reshape(512, 8, 8)
    b = b[b[:, 0, 7] > 0]
    ns = b[:, 0, 7].mean()
    cons, prod = b[:, :4].reshape(-1, 8), b[:, 4:].reshape(-1, 8)
    raw = buf.reshape(-1, 8)[:512 * 8].reshape(512, 8, 8)[:, 0, 6]
    raw = raw[raw > 0]
    wall, cyc = (raw >> np.uint64(32)).astype(np.float64), (raw & np.uint64(0xffffffff)).astype(np.float64) * 256
    print('%s: workgroups %d, stages per workgroup %.0f; shader clock during the launch %.2f GHz (%.3f ms)' % (which, len(b), ns, (cyc / (wall * 10.0)).mean(), wall.mean() / 1e5))
    for nm, rows, names in (('consumer', cons, ['setup + first fragments', '-', 'main K loop (108 MFMAs x 16 cyc = 1728)', 'sibling K loop (21 MFMAs x 16 cyc = 336)', 'epilogue', 'barrier wait']),
                            ('producer', prod, ['weights -> LDS, next weights / parameters requested', 'rest', 'next tile, mask, descriptors', 'slots: prologue + split + LDS write + next load', '-', 'barrier wait'])):
        tot = rows[:, :6].sum(1).mean()
        print('%s wavefronts: %.0f cycles per stage' % (nm, tot / ns))
        for i, n in enumerate(names):
            if n != '-':
                print('   %-52s %8.0f per stage (%.1f%%)' % (n, rows[:, i].mean() / ns, 100 * rows[:, i].mean() / tot))
    sys.exit(0)
b = b[b[:, 7] > 0]
ns = b[:, 7].mean()
NP = 6
tot = b[:, :NP].sum(1).mean()
print('%s: waves %d, stages per wave %.0f, mean cycles per stage %.0f (s_memtime ticks = 100 MHz x ? -- ratios matter)' % (which, len(b), ns, tot / ns))
names = ['loop top (+acc zero)', 'stage setup (pf_tile, addresses)', 'main K loop (%d MFMAs)' % mf[0], 'sibling K loop (%d MFMAs)' % mf[1], 'epilogue (every nchunk-th stage)', 'barrier']
for i, n in enumerate(names):
    print('%-40s %9.0f per stage (%.1f%%)' % (n, b[:, i].mean() / ns, 100 * b[:, i].mean() / tot))
