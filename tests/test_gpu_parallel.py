"""-m gpu: the view-sharded path (a-tvsnet_amd/parallel.py) on the HIP kernels with a one-rank RCCL
process group: same collectives, same kernels as the multi-GPU run, must equal the single-GPU pipeline."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_sharded_path_world1_equals_single_gpu(cuda, weights):
    import torch.distributed as dist
    from atvsnet_amd import parallel, synthetic
    from atvsnet_amd.atvsnet import example as ex
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29731')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=cuda)
    try:
        imgs, cams = synthetic.make_inputs(4, 128, 160, 32)
        imgs, cams = torch.from_numpy(imgs).to(cuda), torch.from_numpy(cams).to(cuda)
        want = ex.infer_multiview(imgs, cams, 32, view_streams=False)
        got = parallel.infer_multiview_sharded(imgs, cams, 32)
        # the sharded AANet computes num/den instead of sum(score*X): same value up to rounding
        assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
    finally:
        dist.destroy_process_group()
