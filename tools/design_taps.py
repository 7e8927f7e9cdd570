#!/usr/bin/env python3
"""Design the FIR tap sets used by tests, bench and the default rate plans.

The reference holds no tap values (they live in FPGA bitstreams),
so these are authored.  scipy is only needed HERE (the build container); the
results are committed as raw float32 fixtures under tests/golden/, so the GPU box
does not need scipy.  (The drop-in API designs the taps of its rate plans itself,
Kaiser windows in perseus_api.c plan_build.)

    python tools/design_taps.py          # rewrites the fixtures
"""
import json
import os

import numpy as np
from scipy import signal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

# name, ntaps, fs, passband edge, stopband edge  (Hz); all low-pass, unity DC gain
SETS = [
    # config 2: 80 MS/s -> 10 MS/s, 127 taps; alias-free +-3.5 MHz
    ("d8_127", 127, 80e6, 3.5e6, 6.5e6),
    # config 5: 255 taps, alias-free +-4.25 MHz
    ("d8_255", 255, 80e6, 4.25e6, 5.75e6),
    # config 3 cascade 8*8*5 = 320 -> 250 kS/s, protected band +-100 kHz
    ("c320_s1_d8_32", 32, 80e6, 0.1e6, 9.875e6),
    ("c320_s2_d8_64", 64, 10e6, 0.1e6, 1.125e6),
    ("c320_s3_d5_161", 161, 1.25e6, 0.1e6, 0.15e6),
]


def design(ntaps, fs, fp, fst):
    h = signal.remez(ntaps, [0, fp, fst, fs / 2], [1, 0], weight=[1, 10], fs=fs, maxiter=200)
    h = h / np.sum(h)
    return h.astype(np.float32)


def atten_db(h, fs, fst):
    w, H = signal.freqz(h.astype(np.float64), worN=1 << 15, fs=fs)
    return float(-20 * np.log10(np.max(np.abs(H[w >= fst])) + 1e-300))


def main():
    os.makedirs(GOLD, exist_ok=True)
    manifest = {}
    for name, n, fs, fp, fst in SETS:
        h = design(n, fs, fp, fst)
        h.tofile(os.path.join(GOLD, f"taps_{name}.f32"))
        manifest[name] = {
            "ntaps": n, "fs_hz": fs, "pass_hz": fp, "stop_hz": fst,
            "stop_atten_db": round(atten_db(h, fs, fst), 1),
            "sum": float(np.sum(h.astype(np.float64))),
            "file": f"taps_{name}.f32",
        }
    with open(os.path.join(GOLD, "taps_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print(json.dumps(manifest, indent=1))


if __name__ == "__main__":
    main()
