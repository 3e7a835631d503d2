"""The fp16 range of the split-operand tower kernels: margin with the seeded weights, and what happens without one.

The 2-D towers' residual stream is not normalised (reference cnn_wrapper/network.py:570-601: slim convolutions with biases
between a unit's pre-activation batch norm and its output), so with a real checkpoint an activation CAN exceed 65,504, where a
value's first fp16 piece becomes infinity.  These tests pin: (a) how far the seeded weights are from that at the metric's size,
(b) that crossing it is reported -- by the sticky device flag the batch norms set and by the host's finite check, with the
documented message -- and never returned as a depth map, (c) that ATVS_SPLIT16=0 / ops.configure(split16=False) computes the
finite fp32 answer of the same weights."""
import numpy as np
import pytest
import torch

import atvsnet_amd  # noqa: F401
from oracle import nets

pytestmark = pytest.mark.gpu
FP16_MAX = 65504.0


def _tower_layer_maxima(net):
    out = {}
    for name, t in net.layers.items():
        if isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32:
            out[name] = float(t.abs().max())
    return out


def test_margin_of_the_seeded_towers_at_the_metric_size(cuda, weights):
    """Largest |activation| of every tower layer on configs[2]'s five 640x512 views with the seeded weights: what the split
    kernels' inputs reach, against fp16's 65,504."""
    from atvsnet_amd import synthetic
    from atvsnet_amd.cnn_wrapper.atvsnet import ResNetDS2SPP, ResNetDS2SPP_shallow_f16
    imgs, _ = synthetic.make_inputs(5, 512, 640, 192)
    x = torch.from_numpy(imgs)[0].to(cuda)                               # (5, 512, 640, 3): five calls of the tower
    worst = ('', 0.0)
    for cls in (ResNetDS2SPP, ResNetDS2SPP_shallow_f16):
        net = cls({'data': x}, is_training=True, independent_samples=True)
        for name, m in _tower_layer_maxima(net).items():
            if name != 'data' and m > worst[1]:
                worst = (cls.__name__ + '/' + name, m)
    margin = FP16_MAX / worst[1]
    print('largest tower activation %.1f at %s: %.0f x below the fp16 range' % (worst[1], worst[0], margin))
    assert np.isfinite(worst[1]) and margin >= 16.0


def test_an_activation_beyond_the_fp16_range_is_reported_and_fp32_kernels_compute_it(cuda, weights):
    """Scale one residual unit's conv1 / conv2 kernels (legal weights, each far inside fp16's range) until its r2 exceeds 65,504:
    the split-operand tower turns it into inf / NaN -- the device flag set by the next batch norm makes the host's check raise
    FloatingPointError naming ATVS_SPLIT16=0 (the VALUES alone need not: a later ReLU can swallow the NaN); with split16 off the
    same weights give the finite answer the oracle computes."""
    from atvsnet_amd import ops, synthetic, variables
    from atvsnet_amd.atvsnet import example as ex
    from atvsnet_amd.cnn_wrapper.atvsnet import ResNetDS2SPP
    store = variables.default_store()
    imgs, _ = synthetic.make_inputs(2, 128, 160, 32)
    x = torch.from_numpy(imgs)[:, 0]                                      # (1, 128, 160, 3)
    names = ['conv1_x_1/conv1/weights', 'conv1_x_1/conv2/weights']
    saved = {n: weights[n].numpy().copy() for n in names}
    scaled = dict(weights)
    try:
        for n in names:
            store.set(n, saved[n] * 2000.0)
            scaled[n] = torch.from_numpy(saved[n] * 2000.0)
        assert max(float(np.abs(saved[n] * 2000.0).max()) for n in names) < 0.1 * FP16_MAX      # the WEIGHTS are legal
        ops.clear_pack_cache()
        want = nets.resnet_ds2_spp(x, scaled)
        assert bool(torch.isfinite(want).all())
        ops.nonfinite_seen(cuda)                                          # clear
        net = ResNetDS2SPP({'data': x.to(cuda)}, is_training=True)
        got = net.get_output().cpu().numpy()
        with pytest.raises(FloatingPointError, match='ATVS_SPLIT16=0'):
            ex.check_device(cuda)                                         # the sticky flag of atvs_bn_finalize
        assert not ops.nonfinite_seen(cuda)                               # ... cleared by the check
        # the values themselves may NOT show it: a ReLU behind the overflow (fmaxf(NaN, 0) = 0) can swallow the NaN, and the
        # features then come out finite and wrong -- which is why the flag exists and every driver consults it
        if np.isfinite(got).all():
            assert float(np.abs(got - want.numpy()).max()) > 1e-2 * float(want.abs().max())
        else:
            with pytest.raises(FloatingPointError, match='non-finite.*ATVS_SPLIT16=0'):
                ex.check_finite(got, 'tower features')
        with ops.configure(split16=False, clear_pack_cache=True):
            net32 = ResNetDS2SPP({'data': x.to(cuda)}, is_training=True)
            got32 = net32.get_output().cpu()
            ex.check_device(cuda)                                         # nothing to report
        assert bool(torch.isfinite(got32).all())
        assert float((got32 - want).abs().max()) <= 1e-3 * float(want.abs().max())
    finally:
        for n in names:
            store.set(n, saved[n])
        ops.clear_pack_cache()


class _overflowing_tower(object):
    """The seeded weights with one residual unit's conv1 / conv2 kernels scaled by 2000 (legal fp16 weights): r2 of that unit leaves
    the fp16 range on the split-operand kernels; fp32 kernels and the oracle compute it."""
    NAMES = ['conv1_x_1/conv1/weights', 'conv1_x_1/conv2/weights']

    def __init__(self, weights):
        self.weights = weights

    def __enter__(self):
        from atvsnet_amd import ops, variables
        self.store = variables.default_store()
        self.saved = {n: self.weights[n].numpy().copy() for n in self.NAMES}
        scaled = dict(self.weights)
        for n in self.NAMES:
            self.store.set(n, self.saved[n] * 2000.0)
            scaled[n] = torch.from_numpy(self.saved[n] * 2000.0)
        ops.clear_pack_cache()
        return scaled

    def __exit__(self, *exc):
        from atvsnet_amd import ops
        for n in self.NAMES:
            self.store.set(n, self.saved[n])
        ops.clear_pack_cache()
        return False


def test_the_drivers_return_the_fp32_map_when_the_split_kernels_overflow(cuda, weights, capsys):
    """VERDICT r5 #4 / reference cnn_wrapper/network.py:165-167,570-601 (fp32 end to end): an fp16-range overflow is not the user's
    problem.  The eager drivers' guard (example.infer_checked), GraphedInference.checked() and PipelinedInference.result() each
    recompute THAT map on the fp32 matrix cores in-process and return it: finite, bit for bit the map ops.configure(split16=False)
    computes, inside the bar against the oracle evaluated with the same weights.  The plain split path on these weights does set
    the flag (else this test would test nothing)."""
    from atvsnet_amd import ops, synthetic
    from atvsnet_amd.atvsnet import example as ex
    from oracle import model as OM
    imgs, cams = synthetic.make_inputs(2, 128, 160, 32)
    imgs_t, cams_t = torch.from_numpy(imgs), torch.from_numpy(cams)
    di, dc = imgs_t.to(cuda), cams_t.to(cuda)
    with _overflowing_tower(weights) as scaled:
        want = OM.run_twoview(imgs_t, cams_t, scaled, 32)
        assert bool(torch.isfinite(want).all())
        ops.nonfinite_seen(cuda)
        ex.infer_twoview(di, dc, 32)
        assert ops.nonfinite_seen(cuda), 'the scaled weights no longer overflow the split-operand towers'
        with ops.configure(split16=False):
            fp32 = ex.infer_twoview(di, dc, 32).cpu()
        assert not ops.nonfinite_seen(cuda) and bool(torch.isfinite(fp32).all())
        err = float(((fp32 - want).abs() / want.abs()).mean())
        print('fp32 kernels vs the oracle on the overflowing weights: rel-L1 %.2e' % err)
        assert err <= 1e-3
        # (1) the eager drivers' guard
        ex._fallback_logged[0] = False
        got = ex.infer_checked(lambda: ex.infer_twoview(di, dc, 32), cuda).cpu()
        assert torch.equal(got, fp32)
        assert 'recomputed on the fp32 matrix cores' in capsys.readouterr().out
        assert ops.cfg.split16 == (ops.Config._DEFAULTS['split16'])            # the switch is restored
        # (2) a captured graph
        g = ex.GraphedInference(di, dc, 32)
        ops.nonfinite_seen(cuda)
        assert torch.equal(g.checked().cpu(), fp32)
        assert g._fp32 is not None and not g._fp32.split16
        assert torch.equal(g.checked(di, dc).cpu(), fp32)                       # second time: the fallback graph is replayed
        # (3) the queue
        p = ex.PipelinedInference(di, dc, 32, slots=2)
        ops.nonfinite_seen(cuda)
        t0 = p.submit(di, dc)
        t1 = p.submit(di, dc)
        assert torch.equal(p.result(t0).cpu(), fp32)
        assert torch.equal(p.result(t1).cpu(), fp32)
        # (4) what the fp32 kernels cannot compute either is still an error, not a map
        bad = di.clone()
        bad[0, 0, 5, 7, 1] = float('nan')
        with pytest.raises(FloatingPointError, match='fp32 kernels'):
            ex.infer_checked(lambda: ex.infer_twoview(bad, dc, 32), cuda)
    # with the seeded weights nothing falls back
    ops.nonfinite_seen(cuda)
    g2 = ex.GraphedInference(di, dc, 32)
    g2.checked()
    assert g2._fp32 is None
