# time k_fir_generic alone at several sizes (run under rocprofv3 --kernel-trace --stats)
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, importlib
import numpy as np
pkg = importlib.import_module("libperseus-sdr_amd")
h = np.fromfile("tests/golden/taps_c320_s3_d5_161.f32", dtype=np.float32)
dev = torch.device("cuda:0")
for ns in [419430 * 8, 4194304, 4194304 * 8]:
    d_in = pkg.synth_lcg(6 * ns, 1, 0, dev)
    pipe = pkg.Pipeline([(5, h)])
    for _ in range(5):
        pipe.process(d_in)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        pipe.process(d_in)
    e1.record(); torch.cuda.synchronize()
    print(ns, "ms per process:", e0.elapsed_time(e1) / 20)
    pipe.close()
