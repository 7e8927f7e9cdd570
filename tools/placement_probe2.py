#!/usr/bin/env python3
"""Placement sensitivity, part 2: many output buffers (hipMalloc, allocated in sequence) and many input buffers."""
import importlib, os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("libperseus-sdr_amd")
L = pkg.ddc_lib()
dev = torch.device("cuda:0")
ns = 1 << 28
h = np.fromfile(os.path.join(ROOT, "tests", "golden", "taps_d8_127.f32"), dtype=np.float32)
torch.zeros(1, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
order = sys.argv[1] if len(sys.argv) > 1 else "in_first"

def dmalloc(n):
    p = C.c_void_p()
    pkg.check(L.pddc_malloc(C.byref(p), n))
    return p.value

cap = ns // 8 + 8
if order == "out_first":
    outs = [dmalloc(cap * 8) for _ in range(8)]
    ins = [dmalloc(ns * 6) for _ in range(3)]
else:
    ins = [dmalloc(ns * 6) for _ in range(3)]
    outs = [dmalloc(cap * 8) for _ in range(8)]
for p in ins:
    pkg.check(L.pddc_synth_lcg(p, ns * 6, 12345, 0, st))
pipe = pkg.Pipeline([(8, h)])

def timeit(in_ptr, out_ptr, n=50):
    for _ in range(200):
        pipe.process_ptr(in_ptr, ns, out_ptr, cap, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        pipe.process_ptr(in_ptr, ns, out_ptr, cap, st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

print("order", order)
for i, ip in enumerate(ins):
    row = " ".join(f"{timeit(ip, op):.4f}" for op in outs)
    print(f"in[{i}] @ {ip:#x}: {row}", flush=True)
print("outs @", " ".join(f"{o:#x}" for o in outs))
os._exit(0)
