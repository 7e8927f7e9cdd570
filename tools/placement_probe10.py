"""One big arena instead of separate allocations: input candidates at a few 8 GiB slots of ONE allocation, the output
tried in every slot.  Does every process then find a fast pair?  python tools/placement_probe10.py [arena GiB] [workload]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
pkg = importlib.import_module("libperseus-sdr_amd")
gib = int(sys.argv[1]) if len(sys.argv) > 1 else 192
wl = bench.workload_def(sys.argv[2] if len(sys.argv) > 2 else "d8_127")
dev = torch.device("cuda", 0)
ns = 1 << 28
stream = torch.cuda.current_stream(dev).cuda_stream
pipe = pkg.Pipeline(wl["stages"], device=0, mix=False)
rows = pipe.max_output(ns) + 8
arena = torch.empty(gib << 30, dtype=torch.uint8, device=dev)
nslot = gib // 8
base = arena.data_ptr()
def in_ptr(k): return base + (k << 33)
def out_ptr(k): return base + (k << 33) + (2 << 30)
in_slots = [0, nslot // 3, 2 * nslot // 3]
for k in in_slots:
    pkg.check(pkg.ddc_lib().pddc_synth_lcg(in_ptr(k), 6 * ns, 12345, 0, stream))
def t(i, o, n=24):
    for _ in range(30):
        pipe.process_ptr(in_ptr(i), ns, out_ptr(o), rows, stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        pipe.process_ptr(in_ptr(i), ns, out_ptr(o), rows, stream)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
for _ in range(150):
    pipe.process_ptr(in_ptr(0), ns, out_ptr(0), rows, stream)
best = 9
for i in in_slots:
    row = [t(i, o) for o in range(nslot)]
    best = min(best, min(row))
    print(f"in@{8 * i:3d}GiB:", " ".join(f"{v:.3f}"[1:] for v in row))
print("best", f"{best:.4f}")
