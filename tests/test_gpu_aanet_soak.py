"""-m gpu: aanet_b.hip's two roles meet through LDS (hand-off of a view's [S|R], image buffers written a stage ahead) behind ONE
LDS-only barrier per stage, with halo requests in flight across it.  A wrong buffer parity or a missing wait would not
fail deterministically.  Each form of the launch -- the exact arithmetic (1..4 views, state in two register sets), the running
softmax (5..8 views), workgroups with one tile, with many tiles, with ragged tiles -- is repeated a few hundred times at sizes that
keep all 256 workgroups busy, and EVERY repetition must be bitwise the first one (itself checked against the two-launch form by
tests/test_gpu_groups.py).  tools_dev/soak_aanet.py is the long form (round 6: 5 x 2,000 launches, 0 differences)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
REPS = int(os.environ.get('ATVS_SOAK_REPS', 200))


@pytest.mark.parametrize('nv,shape', [(4, (192, 128, 160)), (8, (64, 120, 232)), (1, (64, 48, 64)), (3, (37, 29, 53)), (5, (64, 64, 80))])
def test_aanet_module_repeats_bitwise(cuda, weights, nv, shape):
    from atvsnet_amd import ops
    from atvsnet_amd.cnn_wrapper.atvsnet import AttAggregation_keepchannel
    g = torch.Generator().manual_seed(7 * nv)
    x = torch.randn((nv,) + shape + (8,), generator=g).to(cuda)
    assert ops.aanet_fused_ok([x[n] for n in range(nv)])
    run = lambda: AttAggregation_keepchannel({'data': x}, is_training=True).get_output()      # noqa: E731
    first = run().clone()
    assert bool(torch.isfinite(first).all())
    bad = 0
    for _ in range(REPS):
        bad += int(not torch.equal(run(), first))
    assert bad == 0, '%d of %d repetitions differ from the first' % (bad, REPS)
