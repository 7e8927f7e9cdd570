"""per-role clock probe of k_fir_i8x (library built with tools/ubench/i8x_probe_r05.patch as libperseus_ddc.so):
one launch of 2^27 samples per form, ticks per tile by wave.  waves 0..3 matrix (layout 2: 0, 1 matrix, 2, 3 finishing), 4..11 loaders"""
import ctypes as C, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from i8x_time import taps, lowpass, pkg
dev = torch.device("cuda:0")
L = pkg.ddc_lib()
L.pddc_i8x_probe_dump.argtypes = [C.c_int]
api = [(d, t) for d, t, _l in pkg.api_plan(250000)][:2]
ns = 1 << 27
cases = [("plain 127, matrix waves finish", [(8, taps("d8_127"))], False, {}),
         ("tuned 32, matrix waves finish", [api[0]], True, {}),
         ("tuned 127, matrix waves finish", [(8, taps("d8_127"))], True, {}),
         ("tuned 127, loaders finish", [(8, taps("d8_127"))], True, {"i8x_layout": 1}),
         ("tuned 255, loaders finish", [(8, taps("d8_255"))], True, {}),
         ("pair 32/41, two matrix + two finishing waves, chunks of 8", api, True, {"i8x_pair_max_log2": 28}),
         ("pair 32/41, loaders finish, chunks of 4", api, True, {"i8x_pair_max_log2": 28, "i8x_layout": 1, "i8x_chunk": 4})]
d_in = pkg.synth_lcg(6 * ns, 12345, 0, dev)
st = torch.cuda.current_stream(dev).cuda_stream
for name, stages, mix, opts in cases:
    pipe = pkg.Pipeline(stages, mix=mix)
    for k, v in opts.items():
        pipe.set_option(k, v)
    if mix:
        pipe.set_center_freq(7.1e6)
    out = torch.empty((pipe.max_output(ns) + 8, 2), dtype=torch.float32, device=dev)
    for _ in range(20):
        pipe.process_ptr(d_in.data_ptr(), ns, out.data_ptr(), out.shape[0], st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        pipe.process_ptr(d_in.data_ptr(), ns, out.data_ptr(), out.shape[0], st)
    e1.record()
    torch.cuda.synchronize()
    L.pddc_i8x_probe_dump(1)
    pipe.process_ptr(d_in.data_ptr(), ns, out.data_ptr(), out.shape[0], st)
    torch.cuda.synchronize()
    print(f"== {name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per 2^27 samples (probe build), kernels {pipe.on_i8(ns), pipe.fused_pair(ns)}", file=sys.stderr, flush=True)
    L.pddc_i8x_probe_dump(0)
    pipe.close()
