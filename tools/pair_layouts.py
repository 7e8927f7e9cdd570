"""the fused /64 pair (the API's 250 kS/s plan: 32 / 41 taps, tuned) on k_fir_i8x by batch size, layout and chunk, against
k_fir8's pair.  usage: python tools/pair_layouts.py"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from i8x_time import timeit, pkg
api = [(d, t) for d, t, _l in pkg.api_plan(250000)][:2]
for rnd in range(2):
    for lg in (22, 24, 26, 28):
        ns = 1 << lg
        steps = 400 if lg <= 24 else 60
        row = []
        for name, o in (("k_fir8", {"i8x_pair": 0, "i8x": 0}), ("L1C4", {"i8x_layout": 1, "i8x_chunk": 4}), ("L1C8", {"i8x_layout": 1, "i8x_chunk": 8}),
                        ("L2C4", {"i8x_layout": 2, "i8x_chunk": 4}), ("L2C8", {"i8x_layout": 2, "i8x_chunk": 8}), ("L2C16", {"i8x_layout": 2, "i8x_chunk": 16}),
                        ("L0C4", {"i8x_layout": 0, "i8x_chunk": 4})):
            oo = dict(o)
            if name != "k_fir8":
                oo["i8x_pair_max_log2"] = 28
            ms, kind = timeit(api, oo, ns, steps=steps, mix=True)
            row.append(f"{name} {ms * 1e3:8.1f} us ({kind[1]})")
        print(f"round {rnd} 2^{lg}: " + "  ".join(row), flush=True)
