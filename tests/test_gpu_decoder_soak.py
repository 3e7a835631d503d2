"""-m gpu: the summing decoder's two roles (csrc/deconv_up_b.hip, deconv_up_b_sum_kernel) meet through LDS behind ONE LDS-only barrier
per stage: the staging wavefronts write the image pair the multiplying wavefronts read a stage later, with the stage after next's loads
and the previous stage's stores in flight across the barrier.  A wrong pair parity or a missing wait would not fail deterministically.
Each form -- three and two terms, many stages per workgroup, one tile, ragged tiles, idle workgroups -- is repeated a few hundred
times and EVERY repetition must be bitwise the first one (itself checked against bn_add + the plain decoder by
tests/test_gpu_groups.py::test_deconv_sums_its_inputs_on_load_bitwise), the moments included."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
REPS = int(os.environ.get('ATVS_SOAK_REPS', 200))


@pytest.mark.parametrize('G,shape,nterms', [(8, (96, 64, 80), 3), (1, (32, 32, 160), 2), (2, (5, 7, 37), 3), (1, (1, 2, 3), 2)])
def test_summing_decoder_repeats_bitwise(cuda, G, shape, nterms):
    from atvsnet_amd import ops
    cin, cout = 16, 8
    g = torch.Generator().manual_seed(11 * G + nterms)
    w = (torch.randn((3, 3, 3, cout, cin), generator=g) * 0.2).numpy()
    ts = []
    for k in range(nterms):
        raw = torch.randn((G,) + shape + (cin,), generator=g).to(cuda)
        par = torch.stack([torch.randn((G, cin), generator=g) * 0.1, torch.rand((G, cin), generator=g) + 0.5,
                           torch.randn((G, cin), generator=g) * 0.1], 1).to(cuda).contiguous()
        ts.append(ops.PendingBN(raw, par, relu=(k != 2)))
    s = ops.PendingSum(ts)
    assert ops.deconv_sum_ok(s, cout, G)

    def run():
        y, st = ops.conv3d_transpose_s2(s, ('soak-up-sum', G, shape, nterms), w, want_stats=True, groups=G)
        return y, ops.bn_params(st, cout, y)
    y0, p0 = run()
    y0, p0 = y0.clone(), p0.clone()
    assert s._final is None and bool(torch.isfinite(y0).all())
    bad = 0
    for _ in range(REPS):
        y, pr = run()
        bad += int(not (torch.equal(y, y0) and torch.equal(pr, p0)))
    assert bad == 0, '%d of %d repetitions differ from the first' % (bad, REPS)
