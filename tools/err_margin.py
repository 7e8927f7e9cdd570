# print the actual max relative error (max|y-ref|/max|ref|) of the GPU path against the oracle
import sys, os, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import load_taps
pkg = importlib.import_module("libperseus-sdr_amd")
from oracle import oracle as O
dev = torch.device("cuda:0")
h1, h2, h3 = load_taps("c320_s1_d8_32"), load_taps("c320_s2_d8_64"), load_taps("c320_s3_d5_161")
cases = [("d8_127", [(8, load_taps("d8_127"))], False), ("d8_127+nco", [(8, load_taps("d8_127"))], True),
         ("d8_255+nco", [(8, load_taps("d8_255"))], True), ("c320 pair+nco", [(8, h1), (8, h2)], True),
         ("c320 full+nco", [(8, h1), (8, h2), (5, h3)], True)]
for seed, freg in ((7, 381178347), (8, 0xFFFFFFF0), (9, 0x7FFFFFFF), (10, 1)):
    for name, stages, mix in cases:
        ns = 4096 * 40
        packed = O.lcg_bytes(6 * ns, seed)
        ref = O.ddc_chain(packed, stages, freg=freg, mix=mix)
        pipe = pkg.Pipeline(stages, mix=mix)
        pipe.set_freg(freg)
        # advance the absolute sample counter first so the NCO phase is far from zero
        y = []
        for a in range(0, ns, 4096 * 10):
            y.append(pipe.process(torch.from_numpy(packed[6 * a:6 * (a + 4096 * 10)].copy()).to(dev)).cpu().numpy().reshape(-1))
        y = np.concatenate(y)
        print(f"freg {freg:#010x} {name:16s} rel_err {O.rel_err(y, ref[:y.size]):.3e}")
        pipe.close()
