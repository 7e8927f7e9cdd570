#!/usr/bin/env python3
"""Config-5 style sweep on the GPU box: batch size x workload (127 / 255 taps,
fp32 vs fp16-rounded tap storage, unpack only, x320 cascade).  Writes
gpurun_out/sweep.json and prints a table (copy into profiles/ to keep)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
runs = []
for wl, extra in (("d8_127", []), ("d8_127", ["--taps-fp16"]), ("d8_255", []), ("d8_255", ["--taps-fp16"]),
                  ("unpack", []), ("c320", [])):
    for log2n in (22, 24, 26, 28, 30):
        steps = max(30, 1 << max(0, 33 - log2n))      # a timed region of at least ~2^33 samples: 2048 steps at 2^22
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu", "--workload", wl, "--log2n", str(log2n),
               "--steps", str(steps), "--warmup", "5", "--settle-ms", "30"] + extra
        p = subprocess.run(cmd, capture_output=True, text=True)
        try:
            d = json.loads(p.stdout.strip().splitlines()[-1])
        except Exception:
            print("FAILED", cmd, p.stderr[-300:])
            continue
        r = {"workload": wl, "taps": "fp16" if extra else "fp32", "log2n": log2n, "MS_per_s": d["value"],
             "ms_per_step": d["ms_per_step"], "kernel_ms": d["roofline"]["kernel_ms"],
             "GBps": d["roofline"]["achieved"], "frac_of_8TBps": d["roofline"]["frac"],
             "kernel": d["roofline"]["kernel"].split(" ")[0], "verified_ok": bool(d["verified"]["ok"]) if d.get("verified") else None}
        runs.append(r)
        print("%-7s %-4s 2^%-2d  %10.1f MS/s  %8.4f ms  %7.1f GB/s  %.3f  %s %s" %
              (wl, r["taps"], log2n, r["MS_per_s"], r["ms_per_step"], r["GBps"], r["frac_of_8TBps"], r["kernel"],
               "ok" if r["verified_ok"] else "NOT VERIFIED"), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(runs, open(os.path.join(ROOT, "gpurun_out", "sweep.json"), "w"), indent=1)
