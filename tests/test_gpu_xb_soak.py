"""-m gpu: conv_xb.hip's staging wavefronts wait for their loads by hand-written counts.  A wrong count would not
fail deterministically -- it would read a register a little too early, sometimes.  These tests repeat each form of the launch a
few hundred times at full size (every tile of a 192 x 128 x 160 volume, 256 workgroups busy, memory latency as in the product)
and demand that EVERY repetition is bitwise the first one, which itself is checked against the fp32 kernel of the same layer
(different arithmetic: tolerance 2e-5 of the maximum)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

D, H, W = 192, 128, 160
REPS = int(os.environ.get('ATVS_SOAK_REPS', 200))


def _wt(rng, cin, cout):
    return (rng.standard_normal((3, 3, 3, cin, cout)) * 0.1).astype(np.float32)


def _params(G, C, dev):
    return torch.stack([torch.randn(G, C) * 0.1, torch.rand(G, C) + 0.5, torch.randn(G, C) * 0.1], 1).to(dev).contiguous()


def _repeat_bitwise(run):
    first = [t.clone() for t in run()]
    bad = 0
    for _ in range(REPS):
        out = run()
        bad += int(not all(torch.equal(a, b) for a, b in zip(out, first)))
    assert bad == 0, '%d of %d repetitions differ from the first' % (bad, REPS)
    return first


def _against_fp32(ops, run, first):
    with ops.configure(split16=False, clear_pack_cache=True):
        ref = run()
    for a, b in zip(first, ref):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())


@pytest.mark.parametrize('form', ['streamed weights, plane biases (dominant launch)', 'input in pieces, staged by LDS-DMA (dominant launch)',
                                  'normalise on load (refinement)',
                                  'two sources, add on load (stack inputs)', 'no sibling (photo stem)'])
def test_every_repetition_of_a_full_size_launch_is_bitwise_the_first(form):
    from atvsnet_amd import ops
    dev = torch.device('cuda:0')
    torch.manual_seed(5)
    rng = np.random.default_rng(5)
    ops.clear_pack_cache()
    if form.startswith('streamed') or form.startswith('input in pieces'):
        G = 8
        pieces = form.startswith('input in pieces')
        x = torch.randn(G, 4, ops.planar_stride(D, H, W), device=dev)
        if pieces:                       # any fp16 bit patterns are valid pieces: finite ones, as a warp would write them
            n = D * H * W * 8
            half = (torch.randn(G, 4, 2 * n, device=dev) * 2.0).half()
            x[..., :n] = half.view(torch.float32)
        pb, pb2 = torch.randn(G, H, W, 24, device=dev), torch.randn(G, H // 2, W // 2, 48, device=dev)
        w8, w16 = _wt(rng, 32, 8), _wt(rng, 32, 16)

        def run():
            (y, st), (y2, st2) = ops.conv_siblings(x, 'sk8', w8, 'sk16', w16, plane_bias=pb, plane_bias2=pb2, groups=G, planar=(D, H, W),
                                                   pieces=pieces)
            return y, y2, st.partial, st2.partial
    elif form.startswith('normalise'):
        G = 4
        x = torch.randn(G, D, H, W, 32, device=dev)
        par = _params(G, 32, dev)
        w8, w16 = _wt(rng, 32, 8), _wt(rng, 32, 16)

        def run():
            (y, st), (y2, st2) = ops.conv_siblings(ops.PendingBN(x, par, True), 'sn8', w8, 'sn16', w16, groups=G)
            return y, y2, st.partial, st2.partial
    elif form.startswith('two'):
        G = 8
        xa, xb = torch.randn(G, D, H, W, 8, device=dev), torch.randn(G, D, H, W, 8, device=dev)
        par = _params(G, 8, dev)
        w8, w16 = _wt(rng, 8, 8), _wt(rng, 8, 16)

        def run():
            src = ops.PendingSum([ops.PendingBN(xa, par, True), ops.PendingBN(xb, par, True)])
            (y, st), (y2, st2) = ops.conv_siblings(src, 'st8', w8, 'st16', w16, groups=G)
            return y, y2, st.partial, st2.partial
    else:
        G = 4
        x = torch.randn(G, D, H, W, 16, device=dev)
        pb = torch.randn(G, H, W, 24, device=dev)
        w8 = _wt(rng, 16, 8)

        def run():
            y, st = ops.conv(x, 'sp8', w8, want_stats=True, plane_bias=pb, groups=G)
            return y, st.partial
    first = _repeat_bitwise(run)
    if not form.startswith('input in pieces'):           # (the fp32 kernel does not read pieces; test_gpu_conv.py ties them to the planar form)
        _against_fp32(ops, lambda: run()[:2 if len(first) == 4 else 1], first[:2 if len(first) == 4 else 1])
    ops.clear_pack_cache()
