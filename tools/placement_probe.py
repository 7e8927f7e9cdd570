#!/usr/bin/env python3
"""Does the kernel's speed depend on WHERE its output (or input) buffer lies?  Same kernel, same input,
several output buffers allocated in different ways; 60 back-to-back launches each, HIP events."""
import importlib, os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("libperseus-sdr_amd")
L = pkg.ddc_lib()
dev = torch.device("cuda:0")
ns = 1 << 28
h = np.fromfile(os.path.join(ROOT, "tests", "golden", "taps_d8_127.f32"), dtype=np.float32)
d_in = pkg.synth_lcg(6 * ns, 12345, 0, dev)
pipe = pkg.Pipeline([(8, h)])
cap = pipe.max_output(ns) + 8
st = torch.cuda.current_stream(dev).cuda_stream

def timeit(in_ptr, out_ptr, n=60):
    for _ in range(300):
        pipe.process_ptr(in_ptr, ns, out_ptr, cap, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        pipe.process_ptr(in_ptr, ns, out_ptr, cap, st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

outs = [torch.empty((cap, 2), dtype=torch.float32, device=dev) for _ in range(4)]
for i, o in enumerate(outs):
    print(f"torch out[{i}] @ {o.data_ptr():#x} (in @ {d_in.data_ptr():#x}): {timeit(d_in.data_ptr(), o.data_ptr()):.4f} ms", flush=True)
# raw hipMalloc buffers
for i in range(3):
    p = C.c_void_p()
    pkg.check(L.pddc_malloc(C.byref(p), cap * 8))
    print(f"hipMalloc out @ {p.value:#x}: {timeit(d_in.data_ptr(), p.value):.4f} ms", flush=True)
# offsets inside one big buffer
big = torch.empty(cap * 8 + (64 << 20), dtype=torch.uint8, device=dev)
for off in (0, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 8192, 16 << 20, 33 << 20):
    print(f"big+{off:#x} @ {big.data_ptr() + off:#x}: {timeit(d_in.data_ptr(), big.data_ptr() + off):.4f} ms", flush=True)
# a second input buffer
d_in2 = pkg.synth_lcg(6 * ns, 12345, 0, dev)
print(f"second input @ {d_in2.data_ptr():#x} -> out[0]: {timeit(d_in2.data_ptr(), outs[0].data_ptr()):.4f} ms")
print(f"first  input again            -> out[0]: {timeit(d_in.data_ptr(), outs[0].data_ptr()):.4f} ms")
