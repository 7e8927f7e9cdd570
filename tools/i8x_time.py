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


def timeit(stages, opts, ns, steps=30, mix=True, fp16=False):
    pipe = pkg.Pipeline(stages, mix=mix, taps_fp16=fp16)
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
    L = pkg.ddc_lib()
    if hasattr(L, "pddc_i8x_probe_dump") and os.environ.get("I8X_PROBE"):
        L.pddc_i8x_probe_dump.argtypes = [__import__("ctypes").c_int]
        L.pddc_i8x_probe_dump(1)                 # (probe builds, tools/ubench/i8x_probe_r05.patch: clears what the warm-up left)
        pipe.process_ptr(d_in.data_ptr(), ns, out.data_ptr(), out.shape[0], st)
        torch.cuda.synchronize()
        print("probe of one launch:", flush=True)
        L.pddc_i8x_probe_dump(0)
    kind = (pipe.on_i8(ns), pipe.fused_pair(ns))
    k0 = pipe.time_stage0(d_in.data_ptr(), ns, out.data_ptr(), steps, st)
    kind = kind + (round(k0, 4),)
    pipe.close()
    del d_in, out
    return e0.elapsed_time(e1) / steps, kind


if __name__ == "__main__":
    ns = 1 << 28
    if len(sys.argv) > 2 and sys.argv[1] == "one":
        # one case, for profilers: python3 tools/i8x_time.py one <case> [opt=value ...]
        allc = {
            "d8_127+nco": ([(8, taps("d8_127"))], True), "d8_255+nco": ([(8, taps("d8_255"))], True),
            "d8_48+nco": ([(8, lowpass(48, 0.05))], True), "d8_127": ([(8, taps("d8_127"))], False),
            "pair": ([(8, lowpass(48, 0.05)), (8, lowpass(56, 0.05))], True),
            "c320api": ([(8, lowpass(48, 0.05)), (8, lowpass(56, 0.05)), (5, lowpass(144, 0.08))], True),
        }
        stages, mix = allc[sys.argv[2]]
        opts = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in sys.argv[3:]}
        for rnd in range(2):
            ms, kind = timeit(stages, opts, ns, mix=mix, steps=20)
            print(f"{sys.argv[2]} {opts} round {rnd}: {ms:.4f} ms {ns / ms / 1e6:8.1f} GS/s kernels {kind}", flush=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "plainsizes":
        # untuned first stages of every length: the vector kernel (k_fir8), k_fir_i8 (65..256 taps) and k_fir_i8x's plain form
        for name, h in (("d8_32", taps("c320_s1_d8_32")), ("d8_48", lowpass(48, 0.05)), ("d8_127", taps("d8_127")), ("d8_255", taps("d8_255"))):
            for lg in (20, 22, 24, 26, 28):
                n = 1 << lg
                row = []
                for label, opts in (("vec", {"no_i8": 1}), ("i8", {"i8x_plain": 0}), ("x L0", {"i8x_plain": 1, "i8x_layout": 0}),
                                    ("x L1", {"i8x_plain": 1, "i8x_layout": 1})):
                    ms, kind = timeit([(8, h)], opts, n, steps=200 if lg <= 24 else 30, mix=False)
                    row.append(f"{label}({kind[0]}) {ms * 1e3:8.1f} us")
                print(f"{name:8s} 2^{lg}: " + " | ".join(row), flush=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "plain":
        # verdict item 5: does carrying the history inside LDS (chunks of C > 1 tiles) pay for the untuned long stages?
        for rnd in range(3):
            for name in ("d8_127", "d8_255"):
                ms, kind = timeit([(8, taps(name))], {}, ns, mix=False)
                print(f"{name} k_fir_i8            round {rnd}: {ms:.4f} ms  kernel {kind[2]}", flush=True)
                for lay in (0, 1):
                    for C in (1, 2, 4):
                        ms, kind = timeit([(8, taps(name))], {"i8x_plain": 1, "i8x_layout": lay, "i8x_chunk": C}, ns, mix=False)
                        print(f"{name} k_fir_i8x L{lay} C={C}    round {rnd}: {ms:.4f} ms  kernel {kind[2]}", flush=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "sizes":
        # where does the matrix-core path pay?  whole-step time by batch size, i8x layouts against the vector kernels
        pair = [(8, lowpass(48, 0.05)), (8, lowpass(56, 0.05))]
        casc = pair + [(5, lowpass(144, 0.08))]
        for name, stages in (("d8_48+nco", [(8, lowpass(48, 0.05))]), ("d8_127+nco", [(8, taps("d8_127"))]),
                             ("d8_255+nco", [(8, taps("d8_255"))]), ("pair 48/56", pair), ("x320 48/56/144", casc)):
            for lg in (18, 20, 22, 24, 26, 28):
                n = 1 << lg
                row = []
                for label, opts in (("vec", {"i8x": 0}), ("L0", {"i8x_layout": 0}), ("L1", {"i8x_layout": 1}), ("L2", {"i8x_layout": 2}),
                                    ("L2c8", {"i8x_layout": 2, "i8x_chunk": 8})):
                    if label == "L2c8" and len(stages) < 2:
                        continue
                    ms, kind = timeit(stages, opts, n, steps=200 if lg <= 24 else 30)
                    row.append(f"{label} {ms * 1e3:8.1f} us")
                print(f"{name:16s} 2^{lg}: " + " | ".join(row), flush=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "layouts":
        pair = [(8, lowpass(48, 0.05)), (8, lowpass(56, 0.05))]
        for rnd in range(2):
            for name, stages, mix, extra in (("d8_127 plain", [(8, taps("d8_127"))], False, {"i8x_plain": 1}),
                                             ("d8_127+nco", [(8, taps("d8_127"))], True, {}),
                                             ("d8_255+nco", [(8, taps("d8_255"))], True, {}),
                                             ("d8_48+nco", [(8, lowpass(48, 0.05))], True, {}),
                                             ("pair 48/56", pair, True, {}),
                                             ("pair 48/56 C=8", pair, True, {"i8x_chunk": 8}),
                                             ("pair 48/56 C=2", pair, True, {"i8x_chunk": 2})):
                for lay in (0, 1, 2):
                    opts = dict(extra)
                    opts["i8x_layout"] = lay
                    ms, kind = timeit(stages, opts, ns, mix=mix)
                    print(f"layout {lay} {name:16s} round {rnd}: {ms:.4f} ms {ns / ms / 1e6:8.1f} GS/s kernels {kind}", flush=True)
            for name, stages, mix in (("d8_127 k_fir_i8", [(8, taps("d8_127"))], False), ("pair 48/56 k_fir8", pair, True)):
                ms, kind = timeit(stages, {"i8x": 0}, ns, mix=mix)
                print(f"reference {name:16s} round {rnd}: {ms:.4f} ms {ns / ms / 1e6:8.1f} GS/s kernels {kind}", flush=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "chunks":
        pair = [(8, lowpass(48, 0.05)), (8, lowpass(56, 0.05)), (5, lowpass(144, 0.08))]
        for rnd in range(2):
            for C in (1, 2, 4, 8, 16, 100000):
                for name, stages, mix, extra in (("d8_127 plain", [(8, taps("d8_127"))], False, {"i8x_plain": 1}),
                                                 ("d8_127+nco", [(8, taps("d8_127"))], True, {}),
                                                 ("d8_48+nco", [(8, lowpass(48, 0.05))], True, {}),
                                                 ("c320 api-like", pair, True, {})):
                    opts = dict(extra)
                    opts["i8x_chunk"] = C
                    ms, kind = timeit(stages, opts, ns, mix=mix)
                    print(f"C={C:6d} {name:16s} round {rnd}: {ms:.4f} ms {ns / ms / 1e6:8.1f} GS/s kernels {kind}", flush=True)
        sys.exit(0)
    cases = {
        "d8_127+nco": [(8, taps("d8_127"))],
        "d8_255+nco": [(8, taps("d8_255"))],
        "d8_48+nco": [(8, lowpass(48, 0.05))],
        "c320 fixture 32/64/161": [(8, taps("c320_s1_d8_32")), (8, taps("c320_s2_d8_64")), (5, taps("c320_s3_d5_161"))],
        "c320 api-like 48/56/144": [(8, lowpass(48, 0.05)), (8, lowpass(56, 0.05)), (5, lowpass(144, 0.08))],
        "2M api-like 48/144 (8*5)": [(8, lowpass(48, 0.05)), (5, lowpass(144, 0.08))],
    }
    for name, h in (("d8_127 no nco", taps("d8_127")), ("d8_255 no nco", taps("d8_255")), ("d8_48 no nco", lowpass(48, 0.05))):
        for rnd in range(2):
            for label, opts in (("i8x", {"i8x_plain": 1}), ("default", {}), ("vector", {"no_i8": 1})):
                ms, kind = timeit([(8, h)], opts, ns, mix=False)
                print(f"{name:28s} {label:7s} round {rnd}: {ms:.4f} ms  {ns / ms / 1e6:8.1f} GS/s  kernels {kind}", flush=True)
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
