"""Do the pipeline's own small buffers (taps, histories, scheduler words) matter once input and output are placed?
Arena, best pair by a coarse search, then six pipeline objects created one after the other (small allocations in
between) timed on that same pair.  python tools/placement_probe11.py"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
pkg = importlib.import_module("libperseus-sdr_amd")
wl = bench.workload_def("d8_127")
dev = torch.device("cuda", 0)
ns = 1 << 28
stream = torch.cuda.current_stream(dev).cuda_stream
pipes = [pkg.Pipeline(wl["stages"], device=0, mix=False)]
rows = pipes[0].max_output(ns) + 8
gib, slot = 160, 8 << 30
arena = torch.empty(gib << 30, dtype=torch.uint8, device=dev)
base = arena.data_ptr()
pkg.check(pkg.ddc_lib().pddc_synth_lcg(base, 6 * ns, 12345, 0, stream))
def t(p, o, n=48):
    for _ in range(30):
        p.process_ptr(base, ns, base + o * slot + (2 << 30), rows, stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        p.process_ptr(base, ns, base + o * slot + (2 << 30), rows, stream)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
for _ in range(150):
    pipes[0].process_ptr(base, ns, base + (2 << 30), rows, stream)
tab = [t(pipes[0], o, 24) for o in range(gib * (1 << 30) // slot)]
bo = min(range(len(tab)), key=tab.__getitem__)
print("output slots:", " ".join(f"{v:.3f}" for v in tab), "best", bo)
keep = []
for k in range(5):
    keep.append(torch.empty((3 + 7 * k) << 20, dtype=torch.uint8, device=dev))
    pipes.append(pkg.Pipeline(wl["stages"], device=0, mix=False))
for rep in range(2):
    print("pipelines on the best pair:", " ".join(f"{t(p, bo):.4f}" for p in pipes))
