#!/usr/bin/env python3
"""How long do the chip's slow phases last?  6000 back-to-back launches of the headline kernel (2 s), a HIP event after
every 10: the series of 600 readings, each 3.2 ms long, with the runs of readings more than 2.5 % above the median."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("libperseus-sdr_amd")
dev = torch.device("cuda:0")
ns = 1 << 28
h = np.fromfile(os.path.join(ROOT, "tests", "golden", "taps_d8_127.f32"), dtype=np.float32)
pipe = pkg.Pipeline([(8, h)])
rows = pipe.max_output(ns) + 8
st = torch.cuda.current_stream(dev).cuda_stream
# input at the start of one 80 GiB allocation, output behind it (first come) or 64 GiB up (usually another extent class)
arena = torch.empty(80 << 30, dtype=torch.uint8, device=dev)
pkg.check(pkg.ddc_lib().pddc_synth_lcg(arena.data_ptr(), 6 * ns, 12345, 0, st))
where = int(sys.argv[1]) if len(sys.argv) > 1 else 64
out_ptr = arena.data_ptr() + (where << 30) + (2 << 30)
step = lambda: pipe.process_ptr(arena.data_ptr(), ns, out_ptr, rows, st)
print(f"output {where} GiB behind the input", flush=True)
for rep in range(4):
    for _ in range(300):
        step()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(601)]
    evs[0].record()
    for k in range(600):
        for _ in range(10):
            step()
        evs[k + 1].record()
    torch.cuda.synchronize()
    t = np.array([evs[k].elapsed_time(evs[k + 1]) / 10 for k in range(600)])
    med = float(np.median(t))
    slow = t > 1.025 * med
    runs, k = [], 0
    while k < 600:
        if slow[k]:
            j = k
            while j < 600 and slow[j]:
                j += 1
            runs.append((k, j - k, float(t[k:j].mean())))
            k = j
        else:
            k += 1
    print(f"rep {rep}: median {med:.4f} ms, min {t.min():.4f}, max {t.max():.4f}, mean {t.mean():.4f}; {int(slow.sum())} of 600 readings slow; "
          f"slow runs (start reading, length in readings of 10 launches, mean ms): {[(a, b, round(c, 4)) for a, b, c in runs][:40]}", flush=True)

# the bench's situation: sustained load, a device synchronize, then 20 timed launches -- fifty times
for pause_us, nsettle in ((0, 150), (200, 150), (0, 800), (0, 2400)):
    import time
    res = []
    for rep in range(50 if nsettle <= 800 else 20):
        for _ in range(nsettle):
            step()
        torch.cuda.synchronize()
        if pause_us:
            time.sleep(pause_us * 1e-6)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            step()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20)
    r = np.array(res)
    print(f"{nsettle} launches, a synchronize (+{pause_us} us), then {len(res)} regions of 20 launches: median {np.median(r):.4f} ms, min {r.min():.4f}, max {r.max():.4f}; "
          f"regions more than 2.5 % above the median: {int((r > 1.025 * np.median(r)).sum())}; sorted tail {np.sort(r)[-6:].round(4).tolist()}", flush=True)
