"""Quick timing of k_fir_i8x against the vector kernels (GPU box): tuned /8 first stages and the x320 cascade at 2^28
samples, first-come placement (no arena search), 30 steps each, three alternating rounds.  Usage: python tools/i8x_time.py"""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("libperseus-sdr_amd")
dev = torch.device("cuda:0")
GOLD = os.path.join(ROOT, "tests", "golden")


def taps(n):
    return np.fromfile(os.path.join(GOLD, f"taps_{n}.f32"), dtype=np.float32)


def lowpass(ntaps, cutoff):
    k = np.arange(ntaps) - (ntaps - 1) / 2.0
    h = np.sinc(2 * cutoff * k) * np.hamming(ntaps)
    return (h / h.sum()).astype(np.float32)


def timeit(stages, opts, ns, steps=30, mix=True):
    pipe = pkg.Pipeline(stages, mix=mix)
    for k, v in opts.items():
        pipe.set_option(k, v)
    if mix:
        pipe.set_center_freq(7.1e6)
    d_in = pkg.synth_lcg(6 * ns, 12345, 0, dev)
    out = torch.empty((pipe.max_output(ns) + 8, 2), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    for _ in range(10):
        pipe.process_ptr(d_in.data_ptr(), ns, out.data_ptr(), out.shape[0], st)
    pipe.fence(st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        pipe.process_ptr(d_in.data_ptr(), ns, out.data_ptr(), out.shape[0], st)
    pipe.fence(st)
    e1.record()
    torch.cuda.synchronize()
    kind = (pipe.on_i8(ns), pipe.fused_pair(ns))
    pipe.close()
    del d_in, out
    return e0.elapsed_time(e1) / steps, kind


if __name__ == "__main__":
    ns = 1 << 28
    cases = {
        "d8_127+nco": [(8, taps("d8_127"))],
        "d8_255+nco": [(8, taps("d8_255"))],
        "d8_48+nco": [(8, lowpass(48, 0.05))],
        "c320 fixture 32/64/161": [(8, taps("c320_s1_d8_32")), (8, taps("c320_s2_d8_64")), (5, taps("c320_s3_d5_161"))],
        "c320 api-like 48/56/144": [(8, lowpass(48, 0.05)), (8, lowpass(56, 0.05)), (5, lowpass(144, 0.08))],
        "2M api-like 48/144 (8*5)": [(8, lowpass(48, 0.05)), (5, lowpass(144, 0.08))],
    }
    for name, stages in cases.items():
        for rnd in range(2):
            for label, opts in (("i8x", {}), ("vector", {"i8x": 0})):
                ms, kind = timeit(stages, opts, ns)
                print(f"{name:28s} {label:7s} round {rnd}: {ms:.4f} ms  {ns / ms / 1e6:8.1f} GS/s  kernels {kind}", flush=True)
    # small batches (what the API's receivers push)
    for n2 in (1 << 22, 1 << 24):
        st = cases["c320 api-like 48/56/144"]
        for label, opts in (("i8x", {}), ("vector", {"i8x": 0})):
            ms, kind = timeit(st, opts, n2, steps=200)
            print(f"c320 api-like 2^{int(np.log2(n2))}            {label:7s}: {ms * 1e3:.1f} us  {n2 / ms / 1e6:8.1f} GS/s  kernels {kind}", flush=True)
