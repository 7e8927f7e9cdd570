"""Where does the fused /64 pair change from the matrix-core kernel to the vector kernel?  Kernel time by events for the API's
250 kS/s pair at 2^24 .. 2^28 samples: k_fir_i8x (layout 2, chunks of 8) against k_fir8 under its walks.
usage (GPU box): python tools/pair_crossover.py"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from i8x_time import timeit, pkg
api = [(d, t) for d, t, _l in pkg.api_plan(250000)][:2]
variants = [("i8x L2C8", {"i8x_pair_max_log2": 28}, {}),
            ("k_fir8 static", {"i8x_pair": 0}, {"fir8_walk": 0}),
            ("k_fir8 rr K=8", {"i8x_pair": 0}, {"fir8_walk": 1, "fir8_chunk": 8}),
            ("k_fir8 rr K=16", {"i8x_pair": 0}, {"fir8_walk": 1, "fir8_chunk": 16}),
            ("k_fir8 rr K=32", {"i8x_pair": 0}, {"fir8_walk": 1, "fir8_chunk": 32}),
            ("k_fir8 default", {"i8x_pair": 0}, {})]
for rnd in range(3):
    for lg in (24, 25, 26, 27, 28):
        ns = 1 << lg
        row = []
        for name, opts, tun in variants:
            for k, v in tun.items():
                pkg.set_tunable(k, v)
            ms, kind = timeit(api, dict(opts), ns, steps=200 if lg <= 25 else 60, mix=True)
            for k in tun:
                pkg.set_tunable(k, -1 if k in ("fir8_walk", "fir8_dyn_pct") else 0)
            row.append(f"{name} {kind[2] * 1e3:7.1f}")
        print(f"round {rnd} 2^{lg} kernel us: " + " | ".join(row), flush=True)
