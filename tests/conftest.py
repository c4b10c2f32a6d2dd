import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    return torch.device("cuda:0")


def rel_l2(a, b):
    a = a.float().cpu()
    b = b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture
def hooks_library():
    """The test / probe build of the library (merv_amd/lib/libmerv_hip_hooks.so: the same sources with -DMERV_TUNING_HOOKS) for the
    duration of one test: the product library reads no MERV_* tuning variable and its merv_debug_set_* are no-ops, so tests that force a
    tile configuration / kernel form (to cover every instantiation the product selects by shape) run on this build."""
    from merv_amd import _lib
    prev_lib, prev_env = _lib._lib, os.environ.get("MERV_TUNING_HOOKS")
    os.environ["MERV_TUNING_HOOKS"] = "1"
    _lib._lib = None
    try:
        lib = _lib.load()
        assert lib.merv_tuning_hooks() == 1, "libmerv_hip_hooks.so was built without -DMERV_TUNING_HOOKS"
        yield lib
    finally:
        _lib._lib = prev_lib
        if prev_env is None:
            os.environ.pop("MERV_TUNING_HOOKS", None)
        else:
            os.environ["MERV_TUNING_HOOKS"] = prev_env
