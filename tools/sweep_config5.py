#!/usr/bin/env python3
"""BASELINE config 5 as a sweep on the kernels that ship (round 5 review, item 5): untuned decimate-by-8 first stages of
31 .. 255 taps x {fp32-stored, binary16-stored taps} x 2^22 .. 2^30 samples per launch, first-come buffers.  Per point: the
first-stage kernel's time (HIP events round the kernel, pddc_pipeline_time_stage0), GS/s, algorithmic GB/s (7 B per input
sample) and its fraction of 8 TB/s, which kernel ran, the history length the tap count selects and the matrix instructions
a matrix wave issues per tile -- the quantity that, beyond the loaders' own 0.256 ms per 2^28 samples, the time follows
(NOTEBOOK R5.4: up to 72 the loaders are the longer chain and the time sits on a plateau; at 108 -- 129 .. 256 taps -- the
matrix waves are).  Parity of every point:
the last launch's output against the double oracle, every output up to 2^26 samples, 2^22-sample prefix above.
HBM traffic per launch (PMC, both legs, 2^28 samples) comes from tools/pmc_traffic.sh: profiles/pmc_traffic.json.
usage (GPU box): python tools/sweep_config5.py [--out gpurun_out/sweep_config5.json] [--quick]"""
import argparse
import importlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("libperseus-sdr_amd")
from oracle import oracle as O  # noqa: E402  (checker only)

dev = torch.device("cuda:0")
GOLD = os.path.join(ROOT, "tests", "golden")


def lowpass(ntaps, cutoff=0.05):
    if ntaps in (127, 255):
        return np.fromfile(os.path.join(GOLD, f"taps_d8_{ntaps}.f32"), dtype=np.float32)
    k = np.arange(ntaps) - (ntaps - 1) / 2.0
    h = np.sinc(2 * cutoff * k) * np.hamming(ntaps)
    return (h / h.sum()).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sweep_config5.json"))
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    taps_list = [31, 63, 95, 127, 159, 191, 223, 255] if not a.quick else [127, 255]
    logs = [22, 24, 26, 28, 30] if not a.quick else [24, 28]
    st = torch.cuda.current_stream(dev).cuda_stream
    rows = []
    big = pkg.synth_lcg(6 * (1 << max(logs)), 12345, 0, dev)
    out = torch.empty(((1 << max(logs)) // 8 + 8, 2), dtype=torch.float32, device=dev)
    for lg in logs:
        ns = 1 << lg
        host_in = big[:6 * min(ns, 1 << 26)].cpu().numpy()
        for nt in taps_list:
            h = lowpass(nt)
            for storage in ("fp32", "binary16"):
                fp16 = storage == "binary16"
                pipe = pkg.Pipeline([(8, h)], taps_fp16=fp16)
                kind = pipe.on_i8(ns)
                iters = max(20, min(400, int(0.1 / (0.32e-3 * ns / (1 << 28) + 4e-6))))
                pipe.time_stage0(big.data_ptr(), ns, out.data_ptr(), max(10, iters // 4), st)
                ms = min(pipe.time_stage0(big.data_ptr(), ns, out.data_ptr(), iters, st) for _ in range(3))
                # parity of this very configuration (one batch from zero history)
                pipe.reset()
                n = pipe.process_ptr(big.data_ptr(), ns, out.data_ptr(), out.shape[0], st)
                torch.cuda.synchronize()
                href = h.astype(np.float16).astype(np.float32) if fp16 else h
                nchk = min(ns, 1 << 26)
                r = O.chain_check(host_in, 0, nchk, [(8, href)], out[:nchk // 8].cpu().numpy())
                pipe.close()
                hist = 32 if nt <= 32 else 64 if nt <= 64 else 128 if nt <= 128 else 256
                ksteps = (120 + hist + 63) // 64
                mfma = 2 * ksteps * 9                    # two column blocks a matrix wave and tile, nine plane products a k-step
                rows.append({"ntaps": nt, "taps": storage, "log2n": lg, "kernel_ms": round(ms, 5),
                             "GS_per_s": round(ns / ms / 1e6, 1), "GBps": round(7.0 * ns / ms / 1e6, 1),
                             "frac_of_8TBps": round(7.0 * ns / ms / 1e6 / 8000.0, 4),
                             "kernel": "k_fir_i8x (plain form)" if kind == 2 else "k_fir8", "hist": hist,
                             "mfma_per_wave_tile": mfma if kind == 2 else None,
                             "outputs": int(n), "parity": {"compared": r["n"], "max_rel_err": float(f"{r['max_rel_err']:.2e}"), "ok": r["ok"]}})
                print(json.dumps(rows[-1]), flush=True)
    json.dump(rows, open(a.out, "w"), indent=0)
    # the table a reader wants: time at 2^28 by tap count and storage, and where it leaves the plateau
    print("\n2^28 samples, kernel ms (fp32-stored / binary16-stored), of 8 TB/s:")
    for nt in taps_list:
        rr = {r["taps"]: r for r in rows if r["ntaps"] == nt and r["log2n"] == 28}
        if rr:
            print(f"  {nt:3d} taps (hist {rr['fp32']['hist']:3d}, {rr['fp32']['mfma_per_wave_tile']} matrix instr / wave-tile): "
                  f"{rr['fp32']['kernel_ms']:.4f} / {rr['binary16']['kernel_ms']:.4f} ms  "
                  f"{100 * rr['fp32']['frac_of_8TBps']:.1f} / {100 * rr['binary16']['frac_of_8TBps']:.1f} %")
    assert all(r["parity"]["ok"] for r in rows), "a sweep point failed parity"


if __name__ == "__main__":
    main()
