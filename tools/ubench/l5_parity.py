import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
from i8x_time import lowpass, pkg
from oracle import oracle as O
dev = torch.device("cuda:0")
FREG = 381178347
for mix in (True, False):
    for t12 in ((32, 41), (48, 56), (64, 64)):
        stages = [(8, lowpass(t12[0], 0.05)), (8, lowpass(t12[1], 0.05))]
        sizes = [8192, 8192 * 3, 8192 * 300, 8192 * 7, 8192 * 513]
        cuts = np.concatenate([[0], np.cumsum(sizes)])
        packed = O.lcg_bytes(6 * int(cuts[-1]), 4242)
        ref = O.ddc_chain(packed, stages, freg=FREG if mix else 0, mix=mix)
        outs = []
        for lay in (1, 5):
            pipe = pkg.Pipeline(stages, mix=mix)
            pipe.set_option("i8x_layout", lay)
            if mix:
                pipe.set_freg(FREG)
            y = np.concatenate([pipe.process(torch.from_numpy(packed[6 * a:6 * b]).to(dev)).cpu().numpy().reshape(-1) for a, b in zip(cuts[:-1], cuts[1:])])
            assert pipe.fused_pair(8192 * 3) == 2
            pipe.close()
            outs.append(y)
            print(f"mix {mix} taps {t12} layout {lay}: rel err {O.rel_err(y, ref):.3e}; same bits as layout 1: {np.array_equal(y.view(np.uint32), outs[0].view(np.uint32))}", flush=True)
