#!/usr/bin/env python3
"""Alternative stage orders for a rate plan, timed at the kernel ABI like tools/plan_rates.py (taps designed on the spot
with tools/design_plans.py's rule): python tools/plan_alt.py <rate> <d0,d1,...> [<d0,d1,...> ...] [--log2n 28]"""
import argparse, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import design_plans as dp
pkg = importlib.import_module("libperseus-sdr_amd")
ap = argparse.ArgumentParser()
ap.add_argument("rate", type=int)
ap.add_argument("orders", nargs="+")
ap.add_argument("--log2n", type=int, default=28)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--overlap", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
ns = 1 << a.log2n
d_in = pkg.synth_lcg(6 * ns, 12345, 0, dev)
st = torch.cuda.current_stream(dev).cuda_stream
torch.cuda.synchronize()
time.sleep(3.0)
first = True
for rnd in range(2):
    for order in a.orders:
        dec = [int(x) for x in order.split(",")]
        fs, fpass, stages = dp.FS, 0.4 * a.rate, []
        for i, D in enumerate(dec):
            fs_out = fs / D
            fst = 0.6 * a.rate if i == len(dec) - 1 else fs_out - fpass
            h, att, rip = dp.design(fs, fpass, fst, 1)
            stages.append((D, h))
            fs = fs_out
        pipe = pkg.Pipeline(stages, mix=True)
        pipe.set_freg(381178347)
        if a.overlap:
            pipe.set_overlap(True)
        out = torch.empty((pipe.max_output(ns) + 8, 2), dtype=torch.float32, device=dev)
        for _ in range(150 if first else 5):
            pipe.process_ptr(d_in.data_ptr(), ns, out.data_ptr(), out.shape[0], st)
        first = False
        pipe.fence(st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            pipe.process_ptr(d_in.data_ptr(), ns, out.data_ptr(), out.shape[0], st)
        pipe.fence(st)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.iters * 1e3
        print(json.dumps({"rate": a.rate, "order": dec, "ntaps": [int(h.size) for _, h in stages], "ms": round(ms, 4),
                          "GS_per_s": round(ns / ms / 1e6, 1), "on_i8": pipe.on_i8(ns), "round": rnd}), flush=True)
        pipe.close()
