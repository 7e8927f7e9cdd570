#!/usr/bin/env python3
"""Placement, part 3: after picking the fastest of 6 output buffers, does the INPUT buffer's placement matter too?"""
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
ins = [pkg.synth_lcg(6 * ns, 12345, 0, dev)]
outs = [torch.empty((cap, 2), dtype=torch.float32, device=dev) for _ in range(6)]
ins += [pkg.synth_lcg(6 * ns, 12345, 0, dev) for _ in range(3)]

def timeit(i, o, n=40, warm=15):
    for _ in range(warm):
        pipe.process_ptr(i.data_ptr(), ns, o.data_ptr(), cap, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        pipe.process_ptr(i.data_ptr(), ns, o.data_ptr(), cap, st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

timeit(ins[0], outs[0], 200, 0)
for rep in range(2):
    print("matrix (rows = inputs, cols = outputs):")
    for i in ins:
        print("  " + " ".join(f"{timeit(i, o):.4f}" for o in outs), flush=True)
os._exit(0)
