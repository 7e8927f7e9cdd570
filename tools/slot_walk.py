#!/usr/bin/env python3
"""Does walking over the output slots of one arena (what the placement scan does) switch the chip into the state in which
EVERY pair is slow?  Input at the start of one 80 GiB allocation, outputs in slots 1..9 (8 GiB each): 30 rounds over all
slots, 10 + 24 launches of the headline kernel per slot and round; prints one row of readings per round."""
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
arena = torch.empty(80 << 30, dtype=torch.uint8, device=dev)
pkg.check(pkg.ddc_lib().pddc_synth_lcg(arena.data_ptr(), 6 * ns, 12345, 0, st))
in_span = (6 * ns + 255) // 256 * 256
def ms(o, n=24):
    ptr = arena.data_ptr() + (o << 33) + in_span
    for _ in range(10):
        pipe.process_ptr(arena.data_ptr(), ns, ptr, rows, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        pipe.process_ptr(arena.data_ptr(), ns, ptr, rows, st)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n
if os.environ.get("SLOT_WALK_SLEEP"):             # let whatever the driver does behind a fresh 80 GiB allocation finish first
    import time
    torch.cuda.synchronize()
    time.sleep(float(os.environ["SLOT_WALK_SLEEP"]))
for _ in range(150):
    ms(1, 1)
mode = sys.argv[1] if len(sys.argv) > 1 else "walk"
for r in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    if mode == "walk":
        row = [ms(o) for o in range(1, 10)]
    else:                                      # stay: the same number of launches on two slots only
        row = [ms(1 if k % 2 == 0 else 8) for k in range(9)]
    print(mode, r, " ".join(f"{v:.4f}" for v in row), flush=True)
