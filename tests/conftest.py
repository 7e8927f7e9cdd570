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
    config.addinivalue_line("markers", "perf: asserts on TIME or throughput -- records its numbers (gpurun_out/perf_record.jsonl), "
                                       "fails only on gross breakage (>= 25 %), and runs after every parity test")


# `pytest -m gpu -x` is how the suite is run on the GPU box: one failure hides everything behind it.  So the order is
# by what a failure would mean -- bit-exact rows of SURVEY.md 8 first (unpack / pack / golden fixtures), then the
# oracle comparisons of each kernel family, then the API and the multi-GPU plumbing -- and every test that looks at
# a clock (marker `perf`) comes after ALL of them, whatever file it lives in.
_FILE_ORDER = ["test_gpu_parity", "test_gpu_i8x", "test_gpu_i8", "test_gpu_cascade", "test_gpu_fullsize", "test_gpu_gang",
               "test_gpu_api", "test_multi_gpu"]


def pytest_collection_modifyitems(config, items):
    def rank(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return (1 if item.get_closest_marker("perf") else 0,
                _FILE_ORDER.index(mod) if mod in _FILE_ORDER else -1)
    items.sort(key=rank)                      # stable: the order inside a file stays as written


@pytest.fixture
def perf_record(request):
    """record(name, value, unit=..., **more): one JSON line per reading in gpurun_out/perf_record.jsonl (merged back from
    the GPU box); what a `perf` test measured is kept whether or not it asserts on it"""
    import json
    import time
    path = os.path.join(ROOT, "gpurun_out", "perf_record.jsonl")

    def record(name, value, **more):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "a") as f:
            f.write(json.dumps(dict(test=request.node.name, name=name, value=value, t=round(time.time(), 1), **more)) + "\n")
        print(f"perf: {request.node.name}: {name} = {value} {more.get('unit', '')}")
    return record


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
