"""Two depth maps at once on DISJOINT halves of the chip (hipExtStreamCreateWithCUMask): no SIMD is shared by two kernels (the
co-residency fault of DESIGN.md appendix B cannot occur), and an MFMA-bound, power-capped launch of one map runs beside whatever the
other map is doing.  Prints depth maps / s of: one map at a time | two maps co-resident on plain streams | two maps on masked halves
(three ways of halving), and whether every output is bitwise the single-map one.   python tools_dev/cu_mask_probe.py [cfg3|cfg2] [reps]"""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import atvsnet_amd  # noqa: F401
from atvsnet_amd import synthetic, variables
from atvsnet_amd.atvsnet import example as ex
which = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
views, W, H, D = {'cfg3': (5, 640, 512, 192), 'cfg2': (2, 640, 512, 192)}[which]
dev = torch.device('cuda:0')
variables.default_store().init_synthetic(1234)
imgs, cams = synthetic.make_inputs(views, H, W, D)
imgs, cams = torch.from_numpy(imgs).to(dev), torch.from_numpy(cams).to(dev)
hip = ctypes.CDLL('libamdhip64.so')


def masked_stream(bits):
    """bits: iterable of CU indices (0..255) the stream may use"""
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= (1 << (b % 32))
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, 'hipExtStreamCreateWithCUMask -> %d' % rc
    return torch.cuda.ExternalStream(st.value, device=dev)


def run_pair(streams, graphs, n):
    """n maps per graph, the two graphs replayed concurrently on their streams -> (maps / s, outputs of the last round)"""
    evs = [torch.cuda.Event() for _ in graphs]
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        for g, st, e in zip(graphs, streams, evs):
            st.wait_event(e)
            with torch.cuda.stream(st):
                g()
                e.record(st)
    torch.cuda.synchronize()
    dt = time.time() - t0
    return len(graphs) * n / dt


g0 = ex.GraphedInference(imgs, cams, D)
first = g0().clone()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(reps):
    g0()
torch.cuda.synchronize()
one = reps / (time.time() - t0)
print('%s one map at a time: %.2f maps/s (%.2f ms)' % (which, one, 1e3 / one), flush=True)
extra = [ex.GraphedInference(imgs, cams, D) for _ in range(3)]
for g in extra:
    g()
g1 = extra[0]
torch.cuda.synchronize()
cases = [('plain streams (co-resident)', lambda: [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]),
         ('masked: CUs 0-127 | 128-255', lambda: [masked_stream(range(0, 128)), masked_stream(range(128, 256))]),
         ('masked: even | odd CUs', lambda: [masked_stream(range(0, 256, 2)), masked_stream(range(1, 256, 2))]),
         ('masked: CUs with (i / 8) even | odd', lambda: [masked_stream([i for i in range(256) if (i // 8) % 2 == 0]),
                                                          masked_stream([i for i in range(256) if (i // 8) % 2 == 1])]),
         ('masked: one map on 0-127 only (half the chip alone)', lambda: [masked_stream(range(0, 128))]),
         ('masked: one map on the even CUs only', lambda: [masked_stream(range(0, 256, 2))]),
         ('masked: CUs i % 4 (four maps)', lambda: [masked_stream(range(k, 256, 4)) for k in range(4)]),
         ('masked: CUs i % 3 (three maps)', lambda: [masked_stream(range(k, 256, 3)) for k in range(3)]),
         ('masked: pairs (i / 2) even | odd', lambda: [masked_stream([i for i in range(256) if (i // 2) % 2 == 0]),
                                                       masked_stream([i for i in range(256) if (i // 2) % 2 == 1])]),
         ('masked: even | odd again', lambda: [masked_stream(range(0, 256, 2)), masked_stream(range(1, 256, 2))])]
for name, mk in cases:
    try:
        sts = mk()
        gs = ([g0] + extra)[:len(sts)]
        run_pair(sts, gs, 3)
        r = run_pair(sts, gs, reps)
        same = all(torch.equal(g.out if hasattr(g, 'out') else g(), first) for g in gs)
        print('%-55s %.2f maps/s (x %.3f), outputs bitwise the single-map one: %s' % (name, r, r / one, same), flush=True)
    except Exception as e:
        print('%-55s failed: %r' % (name, e), flush=True)
