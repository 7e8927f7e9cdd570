"""Per-launch time of the x320 cascade and its parts at small batches (2^18 .. 2^22 samples): pair + carried tail, pair\nonly, first stage only, each with and without the NCO -- where the fixed ~17 us of a small x320 batch go (first stage\nalone 7.6 us, the fused second stage with its priming tile +7.5 us, the NCO prologue +1.8 us, the carried tail +0).\nUsage on the GPU box: python tools/c320_small_ablate.py"""
import importlib, sys, os, time, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from conftest import load_taps
pkg = importlib.import_module("libperseus-sdr_amd")
dev = torch.device("cuda:0")
h1, h2, h3 = load_taps("c320_s1_d8_32"), load_taps("c320_s2_d8_64"), load_taps("c320_s3_d5_161")
st = torch.cuda.current_stream(dev).cuda_stream
for log2n in (18, 20, 22):
    ns = 1 << log2n
    d_in = pkg.synth_lcg(6 * ns, 1, 0, dev)
    for name, stages, mix, ov in (("pair+tail mix ov", [(8, h1), (8, h2), (5, h3)], True, True), ("pair+tail mix", [(8, h1), (8, h2), (5, h3)], True, False),
                                  ("pair+tail nomix ov", [(8, h1), (8, h2), (5, h3)], False, True), ("pair only mix", [(8, h1), (8, h2)], True, False),
                                  ("pair only nomix", [(8, h1), (8, h2)], False, False), ("stage1 only mix", [(8, h1)], True, False), ("stage1 only nomix", [(8, h1)], False, False)):
        pipe = pkg.Pipeline(stages, mix=mix)
        if mix:
            pipe.set_freg(381178347)
        if ov:
            pipe.set_overlap(True)
        out = torch.empty((pipe.max_output(ns) + 8, 2), dtype=torch.float32, device=dev)
        for _ in range(50):
            pipe.process_ptr(d_in.data_ptr(), ns, out.data_ptr(), out.shape[0], st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2000):
            pipe.process_ptr(d_in.data_ptr(), ns, out.data_ptr(), out.shape[0], st)
        pipe.fence(st)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 2000 * 1e6
        print(f"2^{log2n} {name:20s} {us:7.2f} us per batch", flush=True)
        pipe.close()
