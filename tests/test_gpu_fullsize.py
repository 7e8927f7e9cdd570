"""GPU tests (-m gpu) at the size the bench runs: 2^28 samples per launch (BASELINE configs 2, 3, 5).
The tile schedule of k_fir8 depends on the tile count (static runs per block + dynamic chunks), so
the bench shape is checked itself, not by analogy: windows of the output around the first dynamic
chunk, block-range seams, the last tile and 20 random places are compared with the CPU oracle, which
only needs each window plus its halo.  Uses bench.py's own checker (what its `verified` object is)."""
import importlib
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
NS = 1 << 28


def _bench():
    sys.path.insert(0, ROOT)
    return importlib.import_module("bench")


def _run(pkg, O, dev, name, steps=2, corrupt=None):
    import torch
    b = _bench()
    shard = importlib.import_module("libperseus-sdr_amd.shard")
    wl = b.workload_def(name)
    d_in = pkg.synth_lcg(6 * NS, 12345, 0, dev)
    pipe = pkg.Pipeline(wl["stages"], mix=wl["mix"])
    if wl["mix"]:
        pipe.set_freg(wl["freg"])
    out = torch.empty((pipe.max_output(NS) + 8, 2), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    n = 0
    for _ in range(steps):                              # the second step starts from real history
        n = pipe.process_ptr(d_in.data_ptr(), NS, out.data_ptr(), out.shape[0], st)
    torch.cuda.synchronize()
    sched = pipe.schedule(NS)
    if corrupt:
        corrupt(out, sched, wl)

    def fetch(a0, b0):
        idx = torch.arange(6 * a0, 6 * b0, device=dev, dtype=torch.int64) % (6 * NS)
        return d_in[idx].cpu().numpy()

    v = b.verify_last_output(O, shard, fetch, lambda j0, j1: out[j0:j1].cpu().numpy(), NS, wl, (steps - 1) * NS, sched)
    pipe.close()
    return v, sched, n


@pytest.mark.parametrize("name", ["d8_127", "d8_255", "c320"])
def test_bench_shape_windows_vs_oracle(pkg, O, dev, name):
    v, sched, n = _run(pkg, O, dev, name)
    assert sched["ntiles"] == NS // sched["tile"] and sched["nblocks"] == 512
    if name != "c320":
        assert sched["S"] > 0 and sched["nblocks"] * sched["S"] < sched["ntiles"]      # static part + dynamic tail
    assert v["windows"] >= 24 and v["n_outputs"] == n
    assert v["ok"] and v["max_rel_err"] <= 1e-6, v


def test_first_batch_from_zero_history_at_full_size(pkg, O, dev):
    v, _, _ = _run(pkg, O, dev, "d8_127", steps=1)
    assert v["ok"], v


def test_window_check_catches_a_broken_seam(pkg, O, dev):
    """The checker must see what a broken schedule would do: wipe the first tile of the first
    dynamic chunk (what a block that missed its chunk would leave) -> the check fails."""
    def corrupt(out, sched, wl):
        o0 = sched["nblocks"] * sched["S"] * sched["tile"] // wl["decim"]
        out[o0:o0 + sched["tile"] // wl["decim"]] = 0.0

    v, _, _ = _run(pkg, O, dev, "d8_127", corrupt=corrupt)
    assert not v["ok"] and v["max_rel_err"] > 0.1

    def corrupt_last(out, sched, wl):
        out[NS // wl["decim"] - 8:NS // wl["decim"]] *= 1.001      # a subtle error in the last tile

    v, _, _ = _run(pkg, O, dev, "d8_127", corrupt=corrupt_last)
    assert not v["ok"]
