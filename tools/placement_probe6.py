#!/usr/bin/env python3
"""Placement, part 6: a 200 GiB slab.  Input at +0 (and at +100 GiB); k_fir8 with the output at +k GiB, and a plain
256 MiB streaming copy from the input to +k GiB: how many regions, and does a copy see them too?"""
import importlib, os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("libperseus-sdr_amd")
dev = torch.device("cuda:0")
ns = 1 << 28
h = np.fromfile(os.path.join(ROOT, "tests", "golden", "taps_d8_127.f32"), dtype=np.float32)
pipe = pkg.Pipeline([(8, h)])
cap = pipe.max_output(ns) + 8
st = torch.cuda.current_stream(dev).cuda_stream
src = pkg.synth_lcg(6 * ns, 12345, 0, dev)
G = 1 << 30
NG = 200
slab = torch.empty(NG * G, dtype=torch.uint8, device=dev)

def tk(ip, op, n=16, warm=4):
    for _ in range(warm):
        pipe.process_ptr(ip, ns, op, cap, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        pipe.process_ptr(ip, ns, op, cap, st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

tk(src.data_ptr(), slab.data_ptr(), 300, 0)
for a in (0, 100):
    slab[a * G:a * G + 6 * ns].copy_(src)
    ip = slab.data_ptr() + a * G
    fir, cp = [], []
    for k in range(0, NG - 1):
        if a <= k < a + 2 or k == a - 1:
            fir.append(" . "); cp.append("  . ")
            continue
        fir.append("%3d" % int(round((tk(ip, slab.data_ptr() + k * G) - 0.33) * 1000)))
        ms = pkg.measure_copy(slab.data_ptr() + k * G, ip, 256 << 20, 8, st)
        cp.append("%4.1f" % (2 * (256 << 20) / ms / 1e9))
    print(f"input at +{a} GiB; k_fir8 (ms-0.330)*1000 per output GiB:\n  " + " ".join(fir))
    print(f"  streaming copy of 256 MiB input -> +k GiB, TB/s:\n  " + " ".join(cp), flush=True)
os._exit(0)
