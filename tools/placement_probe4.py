#!/usr/bin/env python3
"""Placement, part 4: the output at offsets k * 1 GiB inside ONE 40 GiB allocation (input fixed), then the input
at offsets inside another slab (output fixed at its fastest offset): is there a period?"""
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
d_in = pkg.synth_lcg(6 * ns, 12345, 0, dev)
G = 1 << 30
slab = torch.empty(40 * G, dtype=torch.uint8, device=dev)

def timeit(ip, op, n=30, warm=10):
    for _ in range(warm):
        pipe.process_ptr(ip, ns, op, cap, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        pipe.process_ptr(ip, ns, op, cap, st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

timeit(d_in.data_ptr(), slab.data_ptr(), 200, 0)
print("slab @ %#x, input @ %#x" % (slab.data_ptr(), d_in.data_ptr()))
row = [timeit(d_in.data_ptr(), slab.data_ptr() + k * G) for k in range(39)]
print("output at slab + k GiB:", " ".join(f"{t:.3f}" for t in row), flush=True)
row2 = [timeit(d_in.data_ptr(), slab.data_ptr() + k * (G // 8)) for k in range(32)]
print("output at slab + k*128 MiB:", " ".join(f"{t:.3f}" for t in row2), flush=True)
kbest = int(np.argmin(row))
# input inside the slab (copy the bytes there), output at the best offset of a second small buffer
out2 = torch.empty((cap, 2), dtype=torch.float32, device=dev)
for k in (0, 2, 4, 8, 16, 24, 32):
    slab[k * G:k * G + 6 * ns].copy_(d_in)
    print(f"input at slab + {k} GiB -> separate output: {timeit(slab.data_ptr() + k * G, out2.data_ptr()):.4f}", flush=True)
os._exit(0)
