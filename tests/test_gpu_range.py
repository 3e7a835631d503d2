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
