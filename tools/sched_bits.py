#!/usr/bin/env python3
"""Are a plan's output BITS independent of the tile schedule (PDDC_FIR8_BLOCKS / DYN_PCT / CHUNK)?  The fused pair's should
be: a chunk's first tile only primes the second stage's history (its own early outputs, the ones that see the chunk-first
NCO path, are not used).  Usage on the GPU box: python tools/sched_bits.py"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import load_taps
pkg = importlib.import_module("libperseus-sdr_amd")
dev = torch.device("cuda:0")
h1, h2, h3 = load_taps("c320_s1_d8_32"), load_taps("c320_s2_d8_64"), load_taps("c320_s3_d5_161")
plans = {"8*8*5": [(8, h1), (8, h2), (5, h3)], "8*8": [(8, h1), (8, h2)], "8 (56 taps)": [(8, load_taps("d8_127")[:56] * 2)],
         "8*5": [(8, load_taps("d8_127")[:56] * 2), (5, h3)]}
for ns in (1 << 22, (1 << 22) + 4096 * 37, 1 << 20):
    d_in = pkg.synth_lcg(6 * ns * 2, 5, 0, dev)
    for name, stages in plans.items():
        outs = {}
        for blocks, dyn, chunk in ((0, None, None), (64, None, None), (64, 0, 1), (128, 50, 2), (16, None, None), (512, 100, 1), (3, None, None)):
            if blocks:
                os.environ["PDDC_FIR8_BLOCKS"] = str(blocks)
            else:
                os.environ.pop("PDDC_FIR8_BLOCKS", None)
            pkg.set_tunable("fir8_dyn_pct", -1 if dyn is None else dyn)       # (process-wide launcher knobs, not environment)
            pkg.set_tunable("fir8_chunk", 0 if chunk is None else chunk)
            pipe = pkg.Pipeline(stages, mix=True)
            pipe.set_freg(381178347)
            y = torch.cat([pipe.process(d_in[:6 * ns]).clone(), pipe.process(d_in[6 * ns:]).clone()])
            outs[(blocks, dyn, chunk)] = y
            pipe.close()
        ref = outs[(0, None, None)]
        same = {k: bool(torch.equal(v, ref)) for k, v in outs.items()}
        print(f"ns {ns} plan {name}: fused_pair-capable {len(stages) >= 2 and stages[1][0] == 8}: bit-identical to the default schedule: {same}", flush=True)
