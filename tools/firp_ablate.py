#!/usr/bin/env python3
"""k_firp's packed /10 (or /5) first stage by itself, 2^28 device-resident samples: with / without the NCO and with
20 .. 160 taps -- which of the block's phases (staging at the input rate, filter at ntaps/D per sample) the time follows.
Usage: python tools/firp_ablate.py [--log2n 28] [--D 10]"""
import argparse, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("libperseus-sdr_amd")
ap = argparse.ArgumentParser()
ap.add_argument("--log2n", type=int, default=28)
ap.add_argument("--D", type=int, default=10)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
ns = 1 << a.log2n
d_in = pkg.synth_lcg(6 * ns, 12345, 0, dev)
st = torch.cuda.current_stream(dev).cuda_stream
for mix in (True, False):
    for nt in (20, 40, 69, 100, 160):
        k = np.arange(nt) - (nt - 1) / 2
        h = (np.sinc(2 * 0.04 * k) * np.hamming(nt)).astype(np.float32)
        pipe = pkg.Pipeline([(a.D, h / h.sum())], mix=mix)
        if mix:
            pipe.set_freg(381178347)
        out = torch.empty((pipe.max_output(ns) + 8, 2), dtype=torch.float32, device=dev)
        for _ in range(3):
            pipe.process_ptr(d_in.data_ptr(), ns, out.data_ptr(), out.shape[0], st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            pipe.process_ptr(d_in.data_ptr(), ns, out.data_ptr(), out.shape[0], st)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.iters * 1e3
        print(json.dumps({"D": a.D, "mix": mix, "ntaps": nt, "ms": round(ms, 4), "GS_per_s": round(ns / ms / 1e6, 1),
                          "TB_per_s": round(ns * (6 + 8 / a.D) / ms / 1e9, 2)}), flush=True)
        pipe.close()
