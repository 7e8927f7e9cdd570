#!/usr/bin/env python3
"""Step time of the x320 cascade by batch size, fused cascade (one kernel) vs fused pair + tail kernel.
usage (GPU box): python tools/cascade_sizes.py [log2n ...]"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

pkg = importlib.import_module("libperseus-sdr_amd")
dev = torch.device("cuda:0")
wl = bench.workload_def("c320")
sizes = [int(a) for a in sys.argv[1:]] or [20, 21, 22, 23, 24, 26]
for log2n in sizes:
    ns = 1 << log2n
    d_in = pkg.synth_lcg(6 * ns, 12345, 0, dev)
    row = {}
    for mode in ("cascade", "pair+tail"):
        if mode == "cascade":
            os.environ["PDDC_FUSE3"] = "1"
        else:
            os.environ.pop("PDDC_FUSE3", None)
        pipe = pkg.Pipeline(wl["stages"], mix=True)
        pipe.set_freg(wl["freg"])
        out = torch.empty((pipe.max_output(ns) + 8, 2), dtype=torch.float32, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        assert pipe.fused_cascade(ns) == (mode == "cascade")
        n = max(50, min(3000, int(0.15 / (ns * 1.2e-12 + 2e-5))))
        for _ in range(n):
            pipe.process_ptr(d_in.data_ptr(), ns, out.data_ptr(), out.shape[0], st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            pipe.process_ptr(d_in.data_ptr(), ns, out.data_ptr(), out.shape[0], st)
        e1.record()
        e1.synchronize()
        pipe.check(st)
        row[mode] = e0.elapsed_time(e1) / n * 1e3
        pipe.close()
    print(f"2^{log2n}: cascade {row['cascade']:8.1f} us   pair+tail {row['pair+tail']:8.1f} us   "
          f"({ns / row['cascade'] / 1e3:.0f} vs {ns / row['pair+tail'] / 1e3:.0f} GS/s)", flush=True)
