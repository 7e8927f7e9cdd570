import sys, os, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import load_taps
pkg = importlib.import_module("libperseus-sdr_amd")
dev = torch.device("cuda:0")
ns = 1 << 28
d_in = pkg.synth_lcg(6 * ns, 1, 0, dev)
h1, h2, h3 = load_taps("c320_s1_d8_32"), load_taps("c320_s2_d8_64"), load_taps("c320_s3_d5_161")
junk = []
for trial in range(14):
    pipe = pkg.Pipeline([(8, h1), (8, h2), (5, h3)], mix=True)
    pipe.set_freg(381178347)
    out = None
    for _ in range(30):
        out = pipe.process(d_in)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            out = pipe.process(d_in)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 100)
    k = pipe.time_stage0(d_in.data_ptr(), ns, out.data_ptr(), 50)
    print(f"trial {trial}: ms/step {[round(r, 4) for r in res]}  stage0-only {k:.4f}", flush=True)
    pipe.close()
    junk.append(torch.empty((trial + 1) * 12345677, dtype=torch.uint8, device=dev))   # shift later allocations
