#!/usr/bin/env python3
"""Placement, part 7: the x320 cascade writes only 1/48 of what it reads (its fused pair leaves 32 MiB per 1.5 GiB
batch in a library-allocated buffer).  Does the placement of that small buffer matter?  New pipelines, each created
after another 8 GiB spacer, so that their internal buffers land further and further away from the input."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
pkg = importlib.import_module("libperseus-sdr_amd")
dev = torch.device("cuda:0")
ns = 1 << 28
wl = bench.workload_def("c320")
st = torch.cuda.current_stream(dev).cuda_stream
d_in = pkg.synth_lcg(6 * ns, 12345, 0, dev)
spacers = []
res = []
for k in range(20):
    pipe = pkg.Pipeline(wl["stages"], mix=True)
    pipe.set_freg(wl["freg"])
    out = torch.empty((pipe.max_output(ns) + 8, 2), dtype=torch.float32, device=dev)
    for _ in range(40 if k else 300):
        pipe.process_ptr(d_in.data_ptr(), ns, out.data_ptr(), out.shape[0], st)
    pipe.time_stage0_inline(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        pipe.process_ptr(d_in.data_ptr(), ns, out.data_ptr(), out.shape[0], st)
    e1.record()
    torch.cuda.synchronize()
    km, n = pipe.stage0_time()
    res.append((e0.elapsed_time(e1) / 30, km))
    print(f"pipeline {k:2d} (internal buffers behind {8 * k:3d} GiB of spacers): step {res[-1][0]:.4f} ms, fused pair {km:.4f} ms", flush=True)
    pipe.close()
    del out
    spacers.append(torch.empty(8 << 30, dtype=torch.uint8, device=dev))
os._exit(0)
