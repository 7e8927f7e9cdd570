#!/usr/bin/env python3
"""How the chip's power management answers short pauses: after `settle` back-to-back launches of the 127-tap /8
kernel, pause P microseconds (host sleep after a stream sync), then time each of the next N launches with its own
HIP events.  Prints the per-launch times in groups, for several P."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
pkg = importlib.import_module("libperseus-sdr_amd")
dev = torch.device("cuda:0")
ns = 1 << 28
h = np.fromfile(os.path.join(ROOT, "tests", "golden", "taps_d8_127.f32"), dtype=np.float32)
d_in = pkg.synth_lcg(6 * ns, 12345, 0, dev)
pipe = pkg.Pipeline([(8, h)])
out = torch.empty((pipe.max_output(ns) + 8, 2), dtype=torch.float32, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
step = lambda: pipe.process_ptr(d_in.data_ptr(), ns, out.data_ptr(), out.shape[0], st)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 160
for pause_us in (0, 100, 300, 1000, 3000, 10000):
    for _ in range(800):                      # ~0.3 s sustained
        step()
    torch.cuda.synchronize()
    if pause_us:
        time.sleep(pause_us * 1e-6)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    evs[0].record()
    for k in range(N):
        step()
        evs[k + 1].record()
    torch.cuda.synchronize()
    t = np.array([evs[k].elapsed_time(evs[k + 1]) for k in range(N)])
    g = [t[i:i + 10].mean() for i in range(0, N, 10)]
    print(f"pause {pause_us:6d} us: first {t[0]:.3f}; means per 10 launches: " + " ".join(f"{v:.3f}" for v in g) +
          f"  | total {t.sum():.2f} ms", flush=True)
