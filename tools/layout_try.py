"""a candidate layout against the default on the forms given: bits against the oracle on a ragged stream first, then time at
2^28.  usage: python tools/layout_try.py <layout> [case-substring ...]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from i8x_time import timeit, taps, lowpass, pkg
from oracle import oracle as O
dev = torch.device("cuda:0")
FREG = 381178347
cases = [("plain 127", [(8, taps("d8_127"))], False), ("plain 48", [(8, lowpass(48, 0.05))], False), ("plain 255", [(8, taps("d8_255"))], False),
         ("tuned 32", [(8, lowpass(32, 0.05))], True), ("tuned 48", [(8, lowpass(48, 0.05))], True),
         ("tuned 127", [(8, taps("d8_127"))], True), ("tuned 255", [(8, taps("d8_255"))], True)]
lay = [-1, int(sys.argv[1])]
sel = sys.argv[2:]
cases = [c for c in cases if not sel or any(x in c[0] for x in sel)]
sizes = [8192 * 3, 8192 + 8, 264, 8192 * 40 + 4096 + 16, 8192 * 600, 8192 * 2 - 8, 8192 * 257]
cuts = np.concatenate([[0], np.cumsum(sizes)])
packed = O.lcg_bytes(6 * int(cuts[-1]), 2027)
for name, stages, mix in cases:
    ref = O.ddc_chain(packed, stages, freg=FREG if mix else 0, mix=mix)
    outs = []
    for l in lay:
        pipe = pkg.Pipeline(stages, mix=mix)
        pipe.set_option("i8x_layout", l)
        if mix:
            pipe.set_freg(FREG)
        y = np.concatenate([pipe.process(torch.from_numpy(packed[6 * a:6 * b]).to(dev)).cpu().numpy().reshape(-1) for a, b in zip(cuts[:-1], cuts[1:])])
        pipe.close()
        outs.append(y)
        print(f"{name:10s} layout {l:2d}: rel err {O.rel_err(y, ref):.3e} ({y.size} == {ref.size}); same bits as the default: {np.array_equal(y.view(np.uint32), outs[0].view(np.uint32))}", flush=True)
for rnd in range(3):
    for name, stages, mix in cases:
        for l in lay:
            ms, kind = timeit(stages, {"i8x_layout": l}, 1 << 28, mix=mix)
            print(f"round {rnd} {name:10s} layout {l:2d}: {ms:.4f} ms  stage0 {kind[2]} ms", flush=True)
