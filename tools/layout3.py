"""layout 3 (matrix waves finish the tile before inside this tile's matrix pass) against the form's default layout: bits
against the oracle on a ragged stream first, then time at 2^28.  usage: python tools/layout3.py"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from i8x_time import timeit, taps, lowpass, pkg
from oracle import oracle as O
dev = torch.device("cuda:0")
FREG = 381178347
cases = [("plain 127", [(8, taps("d8_127"))], False), ("plain 255", [(8, taps("d8_255"))], False), ("tuned 32", [(8, lowpass(32, 0.05))], True),
         ("tuned 127", [(8, taps("d8_127"))], True), ("tuned 255", [(8, taps("d8_255"))], True)]
lay = [int(x) for x in sys.argv[1:]] or [-1, 3]
sizes = [8192 * 3, 8192 + 8, 264, 8192 * 40 + 4096 + 16, 8192 * 600, 8192 * 2 - 8, 8192 * 257]
cuts = np.concatenate([[0], np.cumsum(sizes)])
packed = O.lcg_bytes(6 * int(cuts[-1]), 2027)
for name, stages, mix in cases:
    ref = O.ddc_chain(packed, stages, freg=FREG if mix else 0, mix=mix)
    for l in lay:
        pipe = pkg.Pipeline(stages, mix=mix)
        pipe.set_option("i8x_layout", l)
        if mix:
            pipe.set_freg(FREG)
        y = np.concatenate([pipe.process(torch.from_numpy(packed[6 * a:6 * b]).to(dev)).cpu().numpy().reshape(-1) for a, b in zip(cuts[:-1], cuts[1:])])
        pipe.close()
        print(f"{name:10s} layout {l:2d}: rel err {O.rel_err(y, ref):.3e} ({y.size} == {ref.size})", flush=True)
for rnd in range(2):
    for name, stages, mix in cases:
        for l in lay:
            ms, kind = timeit(stages, {"i8x_layout": l}, 1 << 28, mix=mix)
            print(f"round {rnd} {name:10s} layout {l:2d}: {ms:.4f} ms  stage0 {kind[2]} ms", flush=True)
