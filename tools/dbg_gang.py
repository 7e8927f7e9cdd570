import sys, os, importlib
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
pkg = importlib.import_module("libperseus-sdr_amd")
import test_gpu_gang as T
plan = sys.argv[1] if len(sys.argv) > 1 else "8*10"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
stages = T.plans()[plan]
seeds = [777 + 13 * i for i in range(n)]
ys = T.run_gang(pkg, [stages] * n, seeds, T.SIZES, max(T.SIZES))
dec = int(np.prod([s[0] for s in stages]))
for i in range(n):
    solo = T.run_solo(pkg, stages, seeds[i], T.SIZES, max(T.SIZES))
    a, b = ys[i].reshape(-1, 2), solo.reshape(-1, 2)
    bad = np.nonzero((a.view(np.uint32) != b.view(np.uint32)).any(axis=1))[0]
    print(i, "outputs", a.shape[0], "mismatching", bad.size, "max abs diff", float(np.abs(a - b).max()),
          "first/last bad out idx", (bad[:5], bad[-5:]) if bad.size else None)
    if bad.size:
        cuts = np.cumsum([0] + T.SIZES) // dec
        print("   batch output boundaries", cuts, "bad in batches", sorted(set(np.searchsorted(cuts, bad, side='right') - 1)))
        # runs
        runs = np.split(bad, np.nonzero(np.diff(bad) > 1)[0] + 1)
        print("   runs:", [(int(r[0]), int(r[-1])) for r in runs[:12]])
