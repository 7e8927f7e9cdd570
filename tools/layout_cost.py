"""What does it cost to keep the loaders from finishing tiles (LAYOUT 1) where that is the default?  (Round 5 review 3b: packed
fp32 beside a matrix wave is avoided, guarded and not understood; layout 1 is the one arrangement in which a finishing wave
shares its SIMD with a matrix wave by construction.)  The two families that default to layout 1 -- the tuned first stage of
129..256 taps and the fused pair below 2^23 samples -- under layouts 0, 1, 2, same box, three rounds, kernel time by events.
usage (GPU box): python tools/layout_cost.py"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from i8x_time import timeit, taps, lowpass, pkg
api = [(d, t) for d, t, _l in pkg.api_plan(250000)][:2]
cases = [("tuned 255", [(8, taps("d8_255"))], (28, 24), [({"i8x_layout": l}, f"L{l}") for l in (0, 1, 2)]),
         ("tuned 160", [(8, lowpass(160, 0.05))], (28,), [({"i8x_layout": l}, f"L{l}") for l in (0, 1, 2)]),
         ("pair api", api, (18, 20, 22, 23), [({"i8x_layout": 0, "i8x_chunk": 4}, "L0C4"), ({"i8x_layout": 1, "i8x_chunk": 4}, "L1C4"),
                                              ({"i8x_layout": 2, "i8x_chunk": 4}, "L2C4"), ({"i8x_layout": 2, "i8x_chunk": 8}, "L2C8"),
                                              ({"i8x_layout": 2, "i8x_chunk": 2}, "L2C2"), ({}, "default")])]
for rnd in range(3):
    for name, stages, logs, variants in cases:
        for lg in logs:
            ns = 1 << lg
            steps = 30 if lg >= 26 else 400
            row = []
            for o, tag in variants:
                ms, kind = timeit(stages, dict(o), ns, steps=steps, mix=True)
                row.append(f"{tag} {kind[2] * 1e3:8.2f} us")
            print(f"round {rnd} {name:10s} 2^{lg}: " + "   ".join(row), flush=True)
