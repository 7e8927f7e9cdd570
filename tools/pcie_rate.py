# PCIe-inclusive rate of the host-buffer path (never the bench's `value`; NOTEBOOK.md rounds 1-3 5):
# synchronous push_host from pageable memory vs push_host_async, pinned, two batches in flight.
import sys, os, importlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from conftest import load_taps
pkg = importlib.import_module("libperseus-sdr_amd")
from oracle import oracle as O
h = load_taps("d8_127")
for log2n in (22, 24, 26):
    ns = 1 << log2n
    src = O.lcg_bytes(6 * ns, 1)
    pipe = pkg.Pipeline([(8, h)])
    for _ in range(3):
        pipe.push_host(src)
    reps = max(4, (1 << 28) // ns)
    t0 = time.perf_counter()
    for _ in range(reps):
        pipe.push_host(src)
    t_sync = (time.perf_counter() - t0) / reps
    cap = pipe.max_output(ns) + 1
    hin = [pkg.PinnedBuffer(6 * ns) for _ in range(2)]
    hout = [pkg.PinnedBuffer(8 * cap) for _ in range(2)]
    for b in hin:
        b.array[:] = src
    for k in range(4):
        pipe.push_host_async(hin[k & 1].ptr, ns, hout[k & 1].ptr, cap)
    pipe.wait()
    t0 = time.perf_counter()
    for k in range(reps):
        n, t = pipe.push_host_async(hin[k & 1].ptr, ns, hout[k & 1].ptr, cap)
    pipe.wait()
    t_async = (time.perf_counter() - t0) / reps
    print(f"2^{log2n}: sync pageable {ns / t_sync / 1e9:.2f} GS/s ({6 * ns / t_sync / 1e9:.1f} GB/s in)   "
          f"async pinned x2 {ns / t_async / 1e9:.2f} GS/s ({6 * ns / t_async / 1e9:.1f} GB/s in)")
    pipe.close()
    for b in hin + hout:
        b.free()
