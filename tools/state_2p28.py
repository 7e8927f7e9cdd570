"""Where every first-stage form stands on THIS box at 2^28 samples (first-come buffers, 30 steps, two rounds): the numbers a
kernel change is judged against.  usage: python tools/state_2p28.py [case-substring ...]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from i8x_time import timeit, taps, lowpass, pkg
ns = 1 << 28
api = [(d, t) for d, t, _l in pkg.api_plan(250000)]
cases = [
    ("plain 127", [(8, taps("d8_127"))], False, {}),
    ("plain 255", [(8, taps("d8_255"))], False, {}),
    ("plain 255 fp16-stored", [(8, taps("d8_255"))], False, {"taps_fp16": 1}),
    ("tuned 32", [api[0]], True, {}),
    ("tuned 127", [(8, taps("d8_127"))], True, {}),
    ("tuned 255", [(8, taps("d8_255"))], True, {}),
    ("pair api i8x L-1 C0", api[:2], True, {"i8x_pair_max_log2": 28}),
    ("pair api i8x L1 C4", api[:2], True, {"i8x_pair_max_log2": 28, "i8x_layout": 1, "i8x_chunk": 4}),
    ("pair api i8x L1 C8", api[:2], True, {"i8x_pair_max_log2": 28, "i8x_layout": 1, "i8x_chunk": 8}),
    ("pair api i8x L2 C8", api[:2], True, {"i8x_pair_max_log2": 28, "i8x_layout": 2, "i8x_chunk": 8}),
    ("pair api i8x L0 C4", api[:2], True, {"i8x_pair_max_log2": 28, "i8x_layout": 0, "i8x_chunk": 4}),
    ("pair api k_fir8", api[:2], True, {}),
    ("c320 api in line", api, True, {}),
    ("c320 api i8x pair", api, True, {"i8x_pair_max_log2": 28}),
]
sel = sys.argv[1:]
for rnd in range(2):
    for name, stages, mix, opts in cases:
        if sel and not any(s in name for s in sel):
            continue
        o = dict(opts)
        fp16 = bool(o.pop("taps_fp16", 0))
        try:
            ms, kind = timeit(stages, o, ns, mix=mix, fp16=fp16)
            print(f"round {rnd} {name:24s} {ms:.4f} ms {ns / ms / 1e6:7.1f} GS/s  kernels (i8, pair, stage0 ms) {kind}", flush=True)
        except Exception as e:
            print(f"round {rnd} {name:24s} FAILED {e}", flush=True)
