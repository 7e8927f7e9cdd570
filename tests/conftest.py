import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    p = importlib.import_module("libperseus-sdr_amd")
    plumbing = os.path.join(os.path.dirname(p.SDR_LIB), "perseus_plumbing")
    multi = os.path.join(os.path.dirname(p.SDR_LIB), "perseus_multi_bench")
    if not all(os.path.exists(f) for f in (p.DDC_LIB, p.SDR_LIB, plumbing, multi)):
        p.build()
    return p


@pytest.fixture(scope="session")
def O():
    from oracle import oracle
    oracle.build()
    return oracle


def load_taps(name):
    return np.fromfile(os.path.join(GOLD, f"taps_{name}.f32"), dtype=np.float32)


@pytest.fixture(scope="session")
def taps():
    return load_taps


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture
def tune(pkg):
    """set process-wide launcher knobs (pddc_set_tunable) for one test; restored afterwards"""
    saved = {}

    def _set(name, value):
        if name not in saved:
            saved[name] = pkg.get_tunable(name)
        pkg.set_tunable(name, value)

    yield _set
    for k, v in saved.items():
        pkg.set_tunable(k, v)
