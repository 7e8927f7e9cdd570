#!/usr/bin/env python3
"""Throughput of the drop-in API's ten rate plans at the kernel ABI: device-resident
2^26-sample batches through pddc_pipeline_process, NCO on.  For each plan: GS/s, ms per batch, whether stage 0
reads the packed samples itself, and -- with --no-fast -- the unfused path (unpack kernel -> float2 -> generic).
Run under `rocprofv3 --kernel-trace --stats` to get the per-kernel split (k_resample's share of a plan)."""
import argparse
import ctypes as C
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
ap = argparse.ArgumentParser()
ap.add_argument("--log2n", type=int, default=26)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--rates", default="")
ap.add_argument("--no-fast", action="store_true")
ap.add_argument("--overlap", action="store_true", help="pddc_pipeline_set_overlap: the last stage rides along with the next batch's launch")
ap.add_argument("--place", action="store_true", help="pddc_pipeline_place_buffers: the pipeline's inter-stage buffers in another HBM extent class than the input")
ap.add_argument("--arena-gib", type=int, default=0, help="cut input, inter-stage workspace and output from ONE arena of that many GiB and place the workspace/output side by probing the plan itself at every 2 GiB (what bench.py does for its cascade; 0: first-come allocations)")
ap.add_argument("--alt", action="append", default=[], help="rate:d0,d1,... -- time that stage order for the rate instead of the API's plan (taps designed on the spot by tools/design_plans.py's rule; needs scipy)")
ap.add_argument("--warm-s", type=float, default=1.0, help="run every plan that long before it is timed (a pipeline's freshly allocated inter-stage buffers are slow for their first second)")
ap.add_argument("--opt", action="append", default=[], help="pipeline option name=value (pddc_pipeline_set_option), repeatable")
a = ap.parse_args()

L = pkg.sdr_lib()
os.environ["PERSEUS_AMD_DEVICES"] = "1"
L.perseus_set_debug(0)
assert L.perseus_init() == 1
d = L.perseus_open(0)
L.perseus_firmware_download(d, None)
rates = (C.c_int * 12)()
L.perseus_get_sampling_rates(None, rates, 12)
want = [int(r) for r in a.rates.split(",") if r] or [r for r in rates if r]
dev = torch.device("cuda:0")
ns = 1 << a.log2n
st = torch.cuda.current_stream(dev).cuda_stream
arena = None
if a.arena_gib:
    arena = torch.empty(a.arena_gib << 30, dtype=torch.uint8, device=dev)
    pkg.check(pkg.ddc_lib().pddc_synth_lcg(arena.data_ptr(), 6 * ns, 12345, 0, st))
    d_in = arena[:6 * ns]
else:
    d_in = pkg.synth_lcg(6 * ns, 12345, 0, dev)
# the first second behind a large allocation is slow on these boxes whatever runs in it (NOTEBOOK.md rounds 1-3, 6: the first
# plan of a process measured 0.405 ms for a first stage that takes 0.348 in every later one): rest, as bench.py does, and
# give the first plan a long warm-up
torch.cuda.synchronize()
time.sleep(3.0)
res = []
for rate in want:
    L.perseus_set_sampling_rate(d, rate)
    dec, nt, it = (C.c_int * 4)(), (C.c_int * 4)(), (C.c_int * 4)()
    n = L.perseus_amd_get_plan(d, dec, nt, None)
    taps = [np.zeros(nt[i], np.float32) for i in range(n)]
    arr = (C.POINTER(C.c_float) * 4)(*([t.ctypes.data_as(C.POINTER(C.c_float)) for t in taps] + [None] * (4 - n)))
    L.perseus_amd_get_plan(d, dec, nt, arr)
    L.perseus_amd_get_plan_interp(d, it)
    stages = [(dec[i], taps[i], it[i]) for i in range(n)]
    for alt in a.alt:
        r_, order = alt.split(":")
        if int(r_) == rate:
            sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
            import design_plans as dp
            ds = [int(x) for x in order.split(",")]
            fs, fpass, stages = dp.FS, 0.4 * rate, []
            for i, D in enumerate(ds):
                fs_out = fs / D
                h, _, _ = dp.design(fs, fpass, 0.6 * rate if i == len(ds) - 1 else fs_out - fpass, 1)
                stages.append((D, h, 1))
                fs = fs_out
            n = len(ds)
            dec, nt, it = ds, [int(h.size) for _, h, _ in stages], [1] * n
    pipe = pkg.Pipeline(stages, mix=True, no_fast=a.no_fast)
    pipe.set_freg(381178347)
    for o in a.opt:
        k, v = o.split("=")
        pipe.set_option(k, int(v))
    if a.overlap:
        pipe.set_overlap(True)
    rows = pipe.max_output(ns) + 8
    out = torch.empty((rows, 2), dtype=torch.float32, device=dev)
    placed_at, probes = None, None
    if arena is not None:
        # workspace + output as one block, tried at every 2 GiB behind the input; the plan itself is the probe
        ws = (pipe.workspace_size(ns) + 255) & ~255
        need, step = ws + rows * 8, 2 << 30
        first_off = ((6 * ns + step - 1) // step) * step
        best, probes = None, []
        for off in range(first_off, (a.arena_gib << 30) - need, step):
            base = arena.data_ptr() + off
            pipe.fence(st)
            torch.cuda.synchronize()
            if ws:
                pipe.set_workspace(base, ws, ns)
            o_ptr = base + ws
            for _ in range(4):
                pipe.process_ptr(d_in.data_ptr(), ns, o_ptr, rows, st)
            pipe.fence(st)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(8):
                pipe.process_ptr(d_in.data_ptr(), ns, o_ptr, rows, st)
            pipe.fence(st)
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / 8 * 1e3
            probes.append(round(t, 4))
            if best is None or t < best[0]:
                best = (t, off)
        placed_at = best[1]
        base = arena.data_ptr() + placed_at
        pipe.fence(st)
        torch.cuda.synchronize()
        if ws:
            pipe.set_workspace(base, ws, ns)
        out = arena[placed_at + ws:placed_at + ws + rows * 8].view(torch.float32).view(rows, 2)
    if a.place:
        pipe.place_buffers(d_in.data_ptr(), ns, st)
    for _ in range(3 if res else 150):
        pipe.process_ptr(d_in.data_ptr(), ns, out.data_ptr(), out.shape[0], st)
    pipe.fence(st)
    torch.cuda.synchronize()
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < a.warm_s:
        for _ in range(20):
            pipe.process_ptr(d_in.data_ptr(), ns, out.data_ptr(), out.shape[0], st)
        pipe.fence(st)
        torch.cuda.synchronize()
    pipe.time_stage0_inline(True)
    t0 = time.perf_counter()
    for _ in range(a.iters):
        pipe.process_ptr(d_in.data_ptr(), ns, out.data_ptr(), out.shape[0], st)
    pipe.fence(st)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.iters * 1e3
    s0_ms = pipe.stage0_time()[0]
    pipe.time_stage0_inline(False)
    r = {"rate": rate, "plan": "*".join(f"{dec[i]}" + (f"(x{it[i]})" if it[i] > 1 else "") for i in range(n)),
         "ntaps": [nt[i] for i in range(n)], "ms_per_2^%d" % a.log2n: round(ms, 4), "GS_per_s": round(ns / ms / 1e6, 1), "stage0_kernel_ms": round(s0_ms, 4),
         "stage0_reads_packed": pipe.stage0_reads_packed, "fused8": pipe.fused, "fused_pair": pipe.fused_pair(ns),
         "on_i8": pipe.on_i8(ns), "overlap": a.overlap, "placed": a.place, "arena_offset_GiB": None if placed_at is None else placed_at / 2**30, "probe_ms_every_2GiB": probes, "opts": a.opt}
    res.append(r)
    print(json.dumps(r), flush=True)
    pipe.close()
L.perseus_exit()
