"""Why is k_fir8 slow for EVERY input/output pair in some processes (best pair 0.364 ms instead of 0.340)?
Per process: a 3x3 walk of input/output candidates 8 GiB apart; then fresh pipeline objects (new taps / history /
scheduler-counter allocations, the old ones kept alive) re-timed on the same best pair; then small spacers before
another pipeline.  Run several times: python tools/placement_probe9.py"""
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

def mk():
    return pkg.Pipeline(wl["stages"], device=0, mix=False)

def t(pipe, i, o, n=24):
    for _ in range(30):
        pipe.process_ptr(i.data_ptr(), ns, o.data_ptr(), o.shape[0], stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        pipe.process_ptr(i.data_ptr(), ns, o.data_ptr(), o.shape[0], stream)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n

pipe = mk()
rows = pipe.max_output(ns) + 8
ins, outs, keep = [], [], []
for k in range(3):
    ins.append(pkg.synth_lcg(6 * ns, 12345, 0, dev))
    outs.append(torch.empty((rows, 2), dtype=torch.float32, device=dev))
    keep.append(torch.empty(8 << 30, dtype=torch.uint8, device=dev))
for _ in range(150):
    pipe.process_ptr(ins[0].data_ptr(), ns, outs[0].data_ptr(), rows, stream)
m = {(i, o): t(pipe, ins[i], outs[o]) for i in range(3) for o in range(3)}
(bi, bo), best = min(m.items(), key=lambda kv: kv[1])
print("matrix", " ".join(f"{m[(i, o)]:.4f}" for i in range(3) for o in range(3)), "best", f"{best:.4f}")
pipes = [pipe]
for k in range(4):
    if k >= 2:
        keep.append(torch.empty((64 << 20) * (k + 1), dtype=torch.uint8, device=dev))
    q = mk()
    pipes.append(q)
    print(f"  fresh pipeline {k}: best pair {t(q, ins[bi], outs[bo]):.4f}  (the first pipeline again: {t(pipe, ins[bi], outs[bo]):.4f})")
