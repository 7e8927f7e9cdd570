import sys, os, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import load_taps
pkg = importlib.import_module("libperseus-sdr_amd")
dev = torch.device("cuda:0")
ns = 1 << 27
d_in = pkg.synth_lcg(6 * ns, 1, 0, dev)
for name in ("d8_127", "d8_255"):
    for mix in (False, True):
        for R in ("4", "8"):
            os.environ["PDDC_FIR8_R"] = R
            pipe = pkg.Pipeline([(8, load_taps(name))], mix=mix)
            pipe.set_freg(381178347)
            for _ in range(30):
                pipe.process(d_in)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                pipe.process(d_in)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 200
            print(f"{name} mix={mix} R={R}: {ms:.4f} ms per 2^27  ({ns/ms/1e6:.1f} GS/s)")
            pipe.close()
