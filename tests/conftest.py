import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    import torch
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: (torch.from_numpy(z[k]) if z[k].ndim > 0 else z[k].item()) for k in z.files}


def rel_l1(x, ref):
    """mean|x-ref| / mean|ref|."""
    x = x.detach().double().cpu()
    ref = ref.detach().double().cpu()
    return float((x - ref).abs().mean() / ref.abs().mean().clamp_min(1e-30))


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture
def set_option():
    """`set_option(name, value)`: adamvs_set_option for the duration of one test (include/adamvs_hip.h "OPTIONS": which of two
    equivalent kernel forms a layer takes); every option touched is restored afterwards."""
    import ada_mvs_amd  # noqa: F401
    from ada_mvs_amd import _lib
    saved = {}

    def set_(name, value):
        if name not in saved:
            saved[name] = _lib.get_option(name)
        _lib.set_option(name, value)
    yield set_
    for name, value in saved.items():
        _lib.set_option(name, value)
