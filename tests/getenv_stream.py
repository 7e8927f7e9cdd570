"""child of tests/test_gpu_api.py::test_a_running_stream_never_calls_getenv (run under LD_PRELOAD=getenv_count.so):
pipelines are created (the one place that may read PDDC_* variables), then a stream runs -- process(), push_synth_async,
a gang round, a retune, an option change -- while the interposed getenv counts."""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("libperseus-sdr_amd")
libc = C.CDLL(None)
libc.getenv.restype = C.c_char_p
dev = torch.device("cuda:0")


def lowpass(ntaps, cutoff):
    k = np.arange(ntaps) - (ntaps - 1) / 2.0
    h = np.sinc(2 * cutoff * k) * np.hamming(ntaps)
    return (h / h.sum()).astype(np.float32)


stages = [(8, lowpass(48, 0.05)), (8, lowpass(56, 0.05)), (5, lowpass(144, 0.08))]
ns = 1 << 18
pipes = [pkg.Pipeline(stages, mix=True) for _ in range(3)] + [pkg.Pipeline([(8, lowpass(127, 0.05))]), pkg.Pipeline([(10, lowpass(97, 0.04)), (5, lowpass(81, 0.08))], mix=True)]
for p in pipes:
    if p is not pipes[3]:
        p.set_freg(381178347)
gang = pkg.Gang(0)
d_in = pkg.synth_lcg(6 * ns, 1, 0, dev)
bufs = [[pkg.PinnedBuffer((p.max_output(ns) + 8) * 8) for _ in range(2)] for p in pipes]
pkg.get_tunable("fir8_chunk")                       # (the process-wide knobs read their environment at first use: before the watch)
torch.cuda.synchronize()
libc.getenv(b"__PDDC_WATCH_ON__")
for k in range(6):
    for p in pipes:
        p.process(d_in)
    if k == 2:
        pipes[0].set_freg(123456789)
        pipes[1].set_option("i8x", 0)
    items = [{"pipe": p, "h_out": bufs[i][k & 1].ptr, "out_cap": p.max_output(ns) + 8, "seed": 5 + i, "byte_offset": 6 * ns * k}
             for i, p in enumerate(pipes[:3])]
    res, ng = gang.push_async(items, ns)
    for p, (n_out, t) in zip(pipes[:3], res):
        p.wait_ticket(t)
    n_out, t = pipes[4].push_synth_async(9, 6 * ns * k, ns, bufs[4][k & 1].ptr, pipes[4].max_output(ns) + 8)
    pipes[4].wait_ticket(t)
torch.cuda.synchronize()
libc.getenv(b"__PDDC_WATCH_OFF__")
print("GETENV_CALLS", libc.getenv(b"__PDDC_WATCH_COUNT__").decode())
