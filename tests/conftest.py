import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _gpu_available() -> bool:
    try:
        from slimt_amd import capi
        return capi.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not silently pass.
    pass


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def hip():
    """The product C ABI. GPU tests call the kernels only through this."""
    from slimt_amd import capi
    if capi.device_count() <= 0:
        pytest.fail("no HIP device visible: GPU tests need a real MI355X "
                    "(there is no CPU fallback in slimt_amd)")
    return capi


_MODEL_CACHE = {}


@pytest.fixture(scope="session")
def synth_models():
    """Seeded synthetic models, cached per (preset, eos_bias)."""
    from slimt_amd import synth

    def get(preset, eos_bias=-100.0, seed=1234):
        key = (preset, eos_bias, seed)
        if key not in _MODEL_CACHE:
            _MODEL_CACHE[key] = synth.make_model(preset, seed=seed, eos_bias=eos_bias)
        return _MODEL_CACHE[key]

    return get


def ulp_diff(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return np.abs(a - b)
