"""runner for tools/ubench/i8x_mixing_loaders.patch (apply, build, run on the GPU box): the mixing-loader form (option i8x_mixl,
default 1 with the patch) against the NCO-in-the-taps form for tuned first stages of 65..256 taps: errors against the oracle on a
ragged stream, then time at 2^28"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from i8x_time import timeit, taps, lowpass, pkg
from oracle import oracle as O
dev = torch.device("cuda:0")
FREG = 381178347
cases = [("tuned 100", [(8, lowpass(100, 0.05))]), ("tuned 127", [(8, taps("d8_127"))]), ("tuned 160", [(8, lowpass(160, 0.04))]), ("tuned 255", [(8, taps("d8_255"))])]
sizes = [8192 * 3, 8192 + 8, 264, 8192 * 40 + 4096 + 16, 8192 * 600, 8192 * 2 - 8, 8192 * 257]
cuts = np.concatenate([[0], np.cumsum(sizes)])
packed = O.lcg_bytes(6 * int(cuts[-1]), 2027)
for name, stages in cases:
    ref = O.ddc_chain(packed, stages, freg=FREG, mix=True)
    for mixl in (0, 1):
        pipe = pkg.Pipeline(stages, mix=True)
        pipe.set_option("i8x_mixl", mixl)
        pipe.set_freg(FREG)
        y = np.concatenate([pipe.process(torch.from_numpy(packed[6 * a:6 * b]).to(dev)).cpu().numpy().reshape(-1) for a, b in zip(cuts[:-1], cuts[1:])])
        pipe.close()
        e = np.abs(y.astype(np.float64) - ref.astype(np.float64))
        print(f"{name:10s} mixl {mixl}: rel err {O.rel_err(y, ref):.3e}  max abs {e.max():.3e}  rms {np.sqrt((e * e).mean()):.3e}", flush=True)
for rnd in range(3):
    for name, stages in cases:
        for mixl in (0, 1):
            ms, kind = timeit(stages, {"i8x_mixl": mixl}, 1 << 28, mix=True)
            print(f"round {rnd} {name:10s} mixl {mixl}: {ms:.4f} ms  stage0 {kind[2]} ms", flush=True)
