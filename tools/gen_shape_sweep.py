#!/usr/bin/env python3
"""Block-shape sweep of k_fir_generic on the GPU box: for the second-stage shapes of the API's rate plans
(decimation, tap count) time a 2-stage pipeline [(8, 48 taps) -> (D, ntaps)] at 2^26 input samples for every
(threads, outputs per thread) shape (PDDC_GEN_SHAPE) and for the launcher's own choice."""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

pkg = importlib.import_module("libperseus-sdr_amd")
dev = torch.device("cuda:0")
ns = 1 << 26
d_in = pkg.synth_lcg(6 * ns, 1, 0, dev)
st = torch.cuda.current_stream(dev).cuda_stream
rng = np.random.default_rng(0)
h0 = (rng.standard_normal(48) / 48).astype(np.float32)


def run(stages, iters=30):
    pipe = pkg.Pipeline(stages)
    out = torch.empty((pipe.max_output(ns) + 8, 2), dtype=torch.float32, device=dev)
    for _ in range(3):
        pipe.process_ptr(d_in.data_ptr(), ns, out.data_ptr(), out.shape[0], st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        pipe.process_ptr(d_in.data_ptr(), ns, out.data_ptr(), out.shape[0], st)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    pipe.close()
    return ms


base = run([(8, h0)])
print(f"stage 0 alone: {base:.4f} ms")
for D, nt in ((10, 287), (10, 63), (5, 144), (4, 29), (5, 32), (5, 48), (10, 61)):
    h = (rng.standard_normal(nt) / nt).astype(np.float32)
    pkg.set_tunable("gen_shape_nt", 0)
    pkg.set_tunable("gen_shape_p", 0)
    auto = run([(8, h0), (D, h)]) - base
    row = {"D": D, "ntaps": nt, "auto_us": round(auto * 1e3, 1)}
    for NT in (256, 128, 64):
        for P in (1, 2, 3, 4):
            if (D % 2 == 1 and P in (2, 4)) or (D % 2 == 0 and P == 3):
                continue
            pkg.set_tunable("gen_shape_nt", NT)
            pkg.set_tunable("gen_shape_p", P)
            try:
                row[f"{NT},{P}"] = round((run([(8, h0), (D, h)]) - base) * 1e3, 1)
            except Exception as e:
                row[f"{NT},{P}"] = None
    print(json.dumps(row), flush=True)
