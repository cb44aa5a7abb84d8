import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="module", autouse=True)
def _hbm_goes_back_to_the_device():
    """The full-size configurations (test_gpu_configs.py) build their databases in torch tensors -- a hundred GB and more -- and torch's caching
    allocator keeps what they free.  The library allocates with hipMalloc, beside that cache: by the time the file-pipeline tests ran, 1.5 of
    288 GiB were free and a lane's workspace no longer fitted (UTREE_E_NOMEM, twice in a row on the round's last boxes).  After every module,
    once its own fixtures are gone (an autouse fixture is set up first and torn down last), the cache is handed back."""
    yield
    # ... and the device handles the test modules keep per fixture (`_TREES`): a handle that has searched a file holds its lanes' buffers -- about
    # 3.5 GB of HBM per lane, four lanes -- until it is closed; a dozen of them across the modules filled the card
    for name, mod in list(sys.modules.items()):
        trees = getattr(mod, "_TREES", None) if name.startswith("test_gpu") else None
        if isinstance(trees, dict):
            for v in trees.values():
                try:
                    v[1].close()
                except Exception:
                    pass
            trees.clear()
    if "torch" in sys.modules:
        import gc
        import torch
        if torch.cuda.is_available():
            gc.collect()
            torch.cuda.empty_cache()
