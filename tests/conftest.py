import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    import torch
    # the CPU oracle (oneDNN) slows down badly with hundreds of threads on these small convolutions
    torch.set_num_threads(min(os.cpu_count() or 1, 32))


@pytest.fixture(scope='session')
def weights():
    """Seeded synthetic weights as {tf_name: torch CPU tensor} plus the product's store."""
    import torch
    import atvsnet_amd  # noqa: F401
    from atvsnet_amd import variables
    store = variables.default_store()
    store.init_synthetic(1234)
    return {k: torch.from_numpy(v) for k, v in store.host.items()}


@pytest.fixture(scope='session')
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from atvsnet_amd import _lib
    _lib.lib()      # fail loudly if the HIP library is missing
    return torch.device('cuda:0')
