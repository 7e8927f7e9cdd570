#!/usr/bin/env python3
"""Sporadic faults need repetition: the same batch N times through one pipeline form, every output compared bit for bit
with the first run's (which is checked against the oracle).  usage: python tools/repeat_bits.py [reps]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_taps
pkg = importlib.import_module("libperseus-sdr_amd")
from oracle import oracle as O
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
FREG = 381178347
def lowpass(n, c):
    k = np.arange(n) - (n - 1) / 2.0
    h = np.sinc(2 * c * k) * np.hamming(n)
    return (h / h.sum()).astype(np.float32)
cases = {"d10 51 taps": ([(10, lowpass(51, 0.04))], 10240 * 300 + 24), "d10 + /5": ([(10, lowpass(51, 0.04)), (5, lowpass(117, 0.08))], 10240 * 300 + 24),
         "127 tuned": ([(8, load_taps("d8_127"))], 8192 * 600), "48 tuned": ([(8, lowpass(48, 0.05))], 8192 * 600),
         "pair 48/56": ([(8, lowpass(48, 0.05)), (8, lowpass(56, 0.05))], 8192 * 600),
         "pair 127/56": ([(8, load_taps("d8_127")), (8, lowpass(56, 0.05))], 8192 * 600)}
bad = 0
for name, (stages, n) in cases.items():
    packed = O.lcg_bytes(6 * n, 99)
    x = torch.from_numpy(packed).to(dev)
    for layout in (0, 1, 2):
        pipe = pkg.Pipeline(stages, mix=True)
        pipe.set_freg(FREG)
        pipe.set_option("i8x_layout", layout)
        first = None
        nbad = 0
        for r in range(reps):
            pipe.reset()
            y = pipe.process(x).cpu().numpy().reshape(-1)
            if first is None:
                first = y
                err = O.rel_err(y, O.ddc_chain(packed, stages, freg=FREG, mix=True))
                assert err <= 1e-6, (name, layout, err)
            elif not np.array_equal(first, y):
                nbad += 1
        print(f"{name:12s} layout {layout}: {reps} runs, {nbad} differ from the first (oracle err {err:.2e}, kernels {pipe.on_i8(n)}, {pipe.fused_pair(n)})", flush=True)
        bad += nbad
        pipe.close()
print("TOTAL differing runs:", bad)
sys.exit(1 if bad else 0)
