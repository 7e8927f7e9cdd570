#!/usr/bin/env python3
"""Placement, part 5: input AND output inside ONE 48 GiB allocation.  For the input at slab + a GiB, scan the output over
the slab in 256 MiB steps: where is it slow?"""
import importlib, os, sys
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
M = 1 << 20
NG = 48
slab = torch.empty(NG * G, dtype=torch.uint8, device=dev)

def timeit(ip, op, n=24, warm=6):
    for _ in range(warm):
        pipe.process_ptr(ip, ns, op, cap, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        pipe.process_ptr(ip, ns, op, cap, st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

timeit(src.data_ptr(), slab.data_ptr(), 300, 0)
print("slab @ %#x" % slab.data_ptr())
for a in (0, 7, 20, 33):
    slab[a * G:a * G + 6 * ns].copy_(src)
    ip = slab.data_ptr() + a * G
    line = []
    for k in range(0, NG * 4 - 1):
        off = k * 256 * M
        if off + cap * 8 > NG * G or (off < a * G + 6 * ns and off + cap * 8 > a * G):
            line.append(" . ")
            continue
        t = timeit(ip, slab.data_ptr() + off)
        line.append("%3d" % int(round((t - 0.33) * 1000)))
    print(f"input at +{a:2d} GiB; output at +k*256MiB, (ms-0.330)*1000:\n  " + " ".join(line), flush=True)
os._exit(0)
