#!/usr/bin/env python3
"""pddc_malloc_apart vs a plain allocation, judged by the real kernel (k_fir8 127 taps /8, 2^28 samples)."""
import importlib, os, sys, ctypes as C, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("libperseus-sdr_amd")
L = pkg.ddc_lib()
dev = torch.device("cuda:0")
ns = 1 << 28
h = np.fromfile(os.path.join(ROOT, "tests", "golden", "taps_d8_127.f32"), dtype=np.float32)
pipe = pkg.Pipeline([(8, h)])
cap = pipe.max_output(ns) + 8
st = torch.cuda.current_stream(dev).cuda_stream
d_in = pkg.synth_lcg(6 * ns, 12345, 0, dev)

def tk(op, n=40, warm=200):
    for _ in range(warm):
        pipe.process_ptr(d_in.data_ptr(), ns, op, cap, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        pipe.process_ptr(d_in.data_ptr(), ns, op, cap, st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

plain = C.c_void_p()
pkg.check(L.pddc_malloc(C.byref(plain), cap * 8))
apart = C.c_void_p()
fast, slow = C.c_float(0), C.c_float(0)
t0 = time.perf_counter()
pkg.check(L.pddc_malloc_apart(C.byref(apart), cap * 8, d_in.data_ptr(), 6 * ns, 24, C.byref(fast), C.byref(slow)))
dt = time.perf_counter() - t0
print(f"pddc_malloc_apart took {dt:.2f} s; probe {fast.value:.4f} ms (slowest candidate {slow.value:.4f} ms)", flush=True)
print(f"k_fir8 into the plain allocation: {tk(plain.value):.4f} ms;  into the one placed apart: {tk(apart.value, warm=40):.4f} ms", flush=True)
os._exit(0)
