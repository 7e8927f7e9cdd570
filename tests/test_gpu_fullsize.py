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


def _rest_behind_a_large_allocation(dev):
    """During the first second or so behind an allocation of tens of GiB the chip sometimes runs every stream 4-5 % slower
    for some tenths of a second (profiles/r03/n_slow_state_investigation.txt); timing comparisons wait it out, as bench.py does"""
    import time
    import torch
    torch.cuda.synchronize(dev)
    time.sleep(3.0)


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
    assert sched["S"] > 0 and sched["nblocks"] * sched["S"] < sched["ntiles"]          # static part + dynamic tail
    assert v["windows"] >= 24 and v["n_outputs"] == n
    assert v["ok"] and v["max_rel_err"] <= 1e-6, v


@pytest.mark.parametrize("rate", [1600000, 2000000, 1000000])
def test_decimate_by_ten_plans_at_full_size(pkg, O, dev, rate):
    """The three plans that start with a decimate-by-10 stage (1.6 / 2 / 1 MS/s, perseus-sdr.c:776-892 picks the rate) at the
    bench's size: 2^28 samples are not a multiple of 10, so the second batch starts in another decimation phase than the
    first (the phase goes into the matrix kernel's taps as a delay), 26215 tiles of 10240 samples go round 256 blocks, and
    the last tile is ragged.  Windows at the tile seams a walk could get wrong (first and last tile of the first round, the
    wrap to the second, the middle, the last tiles) + 20 random ones against the oracle, on the second batch."""
    import torch
    b = _bench()
    shard = importlib.import_module("libperseus-sdr_amd.shard")
    stages = [(d, t) for d, t, _l in pkg.api_plan(rate)]
    assert stages[0][0] == 10
    dtot = int(np.prod([d for d, _ in stages]))
    wl = {"stages": stages, "mix": True, "freg": 381178347, "decim": dtot}
    d_in = pkg.synth_lcg(6 * NS, 12345, 0, dev)
    pipe = pkg.Pipeline(stages, mix=True)
    pipe.set_freg(wl["freg"])
    assert pipe.on_i8(NS) == 2
    out = torch.empty((pipe.max_output(NS) + 8, 2), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    n = 0
    for _ in range(2):
        n = pipe.process_ptr(d_in.data_ptr(), NS, out.data_ptr(), out.shape[0], st)
    torch.cuda.synchronize()
    ntiles = -(-(NS // 10) // 1024)
    sched = {"tile": 10240, "nblocks": 256, "S": 1, "K": 1, "ntiles": ntiles}      # k_fir_i8x's walk: tile t -> block t mod 256

    def fetch(a0, b0):
        idx = torch.arange(6 * a0, 6 * b0, device=dev, dtype=torch.int64) % (6 * NS)
        return d_in[idx].cpu().numpy()

    v = b.verify_last_output(O, shard, fetch, lambda j0, j1: out[j0:j1].cpu().numpy(), NS, wl, NS, sched)
    pipe.close()
    assert v["n_outputs"] == n and v["windows"] >= 24
    assert v["ok"] and v["max_rel_err"] <= 1e-6, v


@pytest.mark.parametrize("case", ["tuned127", "tuned48", "pair_2p26"])
def test_tuned_matrix_core_stages_at_full_size(pkg, O, dev, case):
    """k_fir_i8x's tuned forms at the bench's size (the bench workloads themselves are untuned or run the vector pair there):
    127 and 48 taps with the NCO at 2^28 samples (both tap sets in one operand, 32768 tiles round 256 blocks), and the x320
    plan at 2^26 -- the largest batch whose pair runs on k_fir_i8x (chunks of four tiles, a chunk's first tile making its own
    porch).  Windows at the walk's seams + 20 random ones against the oracle, on the second batch."""
    import torch
    from conftest import load_taps
    b = _bench()
    shard = importlib.import_module("libperseus-sdr_amd.shard")

    def lowpass(ntaps, cutoff):
        k = np.arange(ntaps) - (ntaps - 1) / 2.0
        h = np.sinc(2 * cutoff * k) * np.hamming(ntaps)
        return (h / h.sum()).astype(np.float32)

    ns = NS if case != "pair_2p26" else 1 << 26
    stages = {"tuned127": [(8, load_taps("d8_127"))], "tuned48": [(8, lowpass(48, 0.05))],
              "pair_2p26": [(d, t) for d, t, _l in pkg.api_plan(250000)]}[case]
    dtot = int(np.prod([d for d, _ in stages]))
    wl = {"stages": stages, "mix": True, "freg": 381178347, "decim": dtot}
    d_in = pkg.synth_lcg(6 * ns, 12345, 0, dev)
    pipe = pkg.Pipeline(stages, mix=True)
    pipe.set_freg(wl["freg"])
    assert pipe.on_i8(ns) == 2 and pipe.fused_pair(ns) == (2 if case == "pair_2p26" else 0)
    out = torch.empty((pipe.max_output(ns) + 8, 2), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    n = 0
    for _ in range(2):
        n = pipe.process_ptr(d_in.data_ptr(), ns, out.data_ptr(), out.shape[0], st)
    torch.cuda.synchronize()
    C = 4 if case == "pair_2p26" else 1
    sched = {"tile": 8192, "nblocks": 256, "S": C, "K": C, "ntiles": ns // 8192}

    def fetch(a0, b0):
        idx = torch.arange(6 * a0, 6 * b0, device=dev, dtype=torch.int64) % (6 * ns)
        return d_in[idx].cpu().numpy()

    v = b.verify_last_output(O, shard, fetch, lambda j0, j1: out[j0:j1].cpu().numpy(), ns, wl, ns, sched)
    pipe.close()
    assert v["n_outputs"] == n and v["windows"] >= 24
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


def test_arena_search_agrees_with_the_kernel(pkg, O, dev):
    """pddc_arena_search (include/perseus_ddc.h) ranks (input slot, output slot) pairs of one large allocation with a
    read+write probe stream.  The pair it returns must be a fast one for the real kernel too: k_fir8 (127 taps, 2^28
    samples) timed on the helper's best pair is within 3 % of the best of a dozen pairs timed directly, and when the
    kernel sees both speeds (>= 4 % apart) the helper's worst pair is one of the slow ones."""
    import ctypes as C
    import torch
    b = _bench()
    L = pkg.ddc_lib()
    wl = b.workload_def("d8_127")
    free_b, _ = torch.cuda.mem_get_info(dev)
    gib = min(176, (free_b - (16 << 30)) >> 30)
    if gib < 32:
        pytest.skip("less than 32 GiB free")
    slot, in_bytes, out_off = 8 << 30, 6 * NS, 2 << 30
    pipe = pkg.Pipeline(wl["stages"])
    rows = pipe.max_output(NS) + 8
    arena = torch.empty(gib << 30, dtype=torch.uint8, device=dev)
    _rest_behind_a_large_allocation(dev)
    nslot = (gib << 30) // slot
    i_sl, o_sl, best, worst = C.c_size_t(), C.c_size_t(), C.c_float(), C.c_float()
    table = (C.c_float * (2 * nslot))()
    pkg.check(L.pddc_arena_search(arena.data_ptr(), gib << 30, slot, in_bytes, out_off, rows * 8, 2, C.byref(i_sl),
                                  C.byref(o_sl), table, C.byref(best), C.byref(worst)))
    tab = np.array(table[:]).reshape(2, nslot)
    in_slots = [0, nslot // 2]
    assert i_sl.value in in_slots and o_sl.value < nslot and 0 < best.value <= worst.value
    assert abs(tab.min() - best.value) < 1e-6 and abs(tab.max() - worst.value) < 1e-6
    st = torch.cuda.current_stream(dev).cuda_stream
    for k in in_slots:                                   # search first, fill later
        pkg.check(L.pddc_synth_lcg(arena.data_ptr() + k * slot, in_bytes, 12345, 0, st))

    def kernel_ms(i, o):
        a, c = arena.data_ptr() + i * slot, arena.data_ptr() + o * slot + out_off
        for _ in range(30):
            pipe.process_ptr(a, NS, c, rows, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(24):
            pipe.process_ptr(a, NS, c, rows, st)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / 24

    for _ in range(150):
        pipe.process_ptr(arena.data_ptr(), NS, arena.data_ptr() + out_off, rows, st)
    direct = {(i, o): kernel_ms(i, o) for i in in_slots for o in range(0, nslot, max(1, nslot // 6))}
    # single readings can fall into a phase in which the chip runs EVERY launch 4-5 % slower for tens of milliseconds
    # (profiles/r03/n_per_step_20runs.txt): what is compared is read twice, the smaller reading counts
    t_best = min(kernel_ms(i_sl.value, o_sl.value) for _ in range(2))
    wi, wo = divmod(int(tab.argmax()), nslot)
    t_worst = min(kernel_ms(in_slots[wi], wo) for _ in range(2))
    slowest = max(direct, key=direct.get)
    direct[slowest] = min(direct[slowest], kernel_ms(*slowest))
    lo, hi = min(direct.values()), max(direct.values())
    print(f"probe best pair in{i_sl.value}/out{o_sl.value}: kernel {t_best:.4f} ms; probe worst pair: kernel {t_worst:.4f} ms; "
          f"direct {lo:.4f} .. {hi:.4f} ms over {len(direct)} pairs; probe {best.value:.3f} .. {worst.value:.3f} ms")
    assert t_best <= 1.05 * lo
    if hi > 1.06 * lo:
        assert t_worst > 1.02 * t_best
    # the result is still right at the chosen place
    n = pipe.process_ptr(arena.data_ptr() + i_sl.value * slot, NS, arena.data_ptr() + o_sl.value * slot + out_off, rows, st)
    y = np.empty((4096, 2), np.float32)
    pkg.check(L.pddc_memcpy_d2h(y.ctypes.data, arena.data_ptr() + o_sl.value * slot + out_off + 8 * (n - 4096), y.nbytes, st))
    pkg.check(L.pddc_stream_sync(st))
    assert np.isfinite(y).all() and np.abs(y).max() > 0
    pipe.close()
    del arena
    torch.cuda.empty_cache()


def test_the_placement_rule_finds_a_fast_pair_in_a_handful_of_probes(pkg, O, dev):
    """Input at the start of ONE 80 GiB allocation, the output side probed right behind it (first come) and at +32 / +48 /
    +64 GiB, every slot only if none of those gains.  Two forms: pddc_pipeline_arena_place probes with the pipeline's own
    kernel, pddc_arena_place with a read+write stream that models the vector kernels.  Each is judged by the real kernel
    it is meant for (127 taps, 2^28 samples: the matrix-core kernel by default, the vector kernel under PDDC_NO_I8): the
    slot it returns is within 4 % of the best of ALL slots."""
    import ctypes as C
    import torch
    b = _bench()
    L = pkg.ddc_lib()
    wl = b.workload_def("d8_127")
    free_b, _ = torch.cuda.mem_get_info(dev)
    gib = min(80, (free_b - (16 << 30)) >> 30)
    if gib < 72:
        pytest.skip("less than 72 GiB free")
    slot, in_bytes, out_off = 8 << 30, 6 * NS, 2 << 30
    arena = torch.empty(gib << 30, dtype=torch.uint8, device=dev)
    _rest_behind_a_large_allocation(dev)
    nslot = (gib << 30) // slot
    st = torch.cuda.current_stream(dev).cuda_stream
    pkg.check(L.pddc_synth_lcg(arena.data_ptr(), in_bytes, 12345, 0, st))
    for vector in (False, True):
        if vector:
            os.environ["PDDC_NO_I8"] = "1"
        try:
            pipe = pkg.Pipeline(wl["stages"])
            assert bool(pipe.on_i8(NS)) == (not vector)
            rows = pipe.max_output(NS) + 8
            o_sl, fc, best, npr = C.c_size_t(), C.c_float(), C.c_float(), C.c_int()
            if vector:
                pkg.check(L.pddc_arena_place(arena.data_ptr(), gib << 30, slot, in_bytes, out_off, rows * 8, C.byref(o_sl),
                                             C.byref(fc), C.byref(best), C.byref(npr), st))
            else:
                pkg.check(L.pddc_pipeline_arena_place(pipe._h, arena.data_ptr(), gib << 30, slot, NS, out_off, C.byref(o_sl),
                                                      C.byref(fc), C.byref(best), C.byref(npr), st))
            assert 1 <= o_sl.value < nslot and 0 < best.value <= fc.value and 4 <= npr.value <= nslot

            def kernel_ms(o):
                c = arena.data_ptr() + o * slot + out_off
                for _ in range(30):
                    pipe.process_ptr(arena.data_ptr(), NS, c, rows, st)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(24):
                    pipe.process_ptr(arena.data_ptr(), NS, c, rows, st)
                e1.record()
                e1.synchronize()
                return e0.elapsed_time(e1) / 24

            for _ in range(150):
                pipe.process_ptr(arena.data_ptr(), NS, arena.data_ptr() + out_off, rows, st)
            direct = {o: kernel_ms(o) for o in range(1, nslot)}
            lo, hi = min(direct.values()), max(direct.values())
            print(f"{'stream probe, vector kernel' if vector else 'kernel probe, matrix-core kernel'}: slot {o_sl.value} after "
                  f"{npr.value} probes (probe {best.value:.4f} ms, first come {fc.value:.4f}); kernel there "
                  f"{direct[o_sl.value]:.4f} ms, first come {direct[1]:.4f}, all slots {lo:.4f} .. {hi:.4f}")
            chosen_ms = direct[o_sl.value]
            if chosen_ms > 1.04 * lo:
                # the chip has phases in which EVERY launch is 4-5 % slower for some tens of milliseconds
                # (profiles/r03/n_per_step_20runs.txt); `lo` is a minimum over nine readings, this was one: read it again
                chosen_ms = min(chosen_ms, kernel_ms(o_sl.value), kernel_ms(o_sl.value))
            # Four probes of nine slots can miss a LONE fast slot (one box in a dozen shows such a map: one slot at 0.339 ms,
            # eight at 0.365): then the rule must at least not have done worse than the first-come buffers it started from.
            lone = sum(1 for v in direct.values() if v <= 1.04 * lo) == 1
            assert chosen_ms <= 1.04 * lo or (lone and chosen_ms <= 1.01 * direct[1]), (chosen_ms, lo, direct)
            pipe.close()
        finally:
            os.environ.pop("PDDC_NO_I8", None)
    del arena
    torch.cuda.empty_cache()
