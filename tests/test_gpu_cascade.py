"""GPU tests (-m gpu) of the fused cascade: stages 0, 1 and 2 (packed -> [NCO] -> /8 -> /8 -> /D3) as ONE kernel
(k_fir8<.., FUSE3>).  What is new in that kernel is exchanged between thread blocks INSIDE the launch:
a chunk of tiles starts without the third stage's history, holds its first outputs back and completes them from the
last outputs of the chunk in front of it, published by another block.  So the tests put many chunk seams into small
batches (few blocks, several static / dynamic splits), cut the stream at whole-tile and odd places (the path falls
back to the unfused kernels and returns, on one shared state), run it under a competing load, and compare with the
CPU oracle (the FIR chain is authored here, tolerance 1e-6 of full scale) and with the unfused path."""
import importlib
import os

import numpy as np
import pytest

from conftest import load_taps

pytestmark = pytest.mark.gpu
FIR_TOL = 1e-6
TILE = 4096                     # inputs per tile of the fused pair at R = 4


def to_dev(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def lowpass(ntaps, cutoff):
    k = np.arange(ntaps) - (ntaps - 1) / 2.0
    h = np.sinc(2 * cutoff * k) * np.hamming(ntaps)
    return (h / h.sum()).astype(np.float32)


def plans():
    h1, h2, h3 = load_taps("c320_s1_d8_32"), load_taps("c320_s2_d8_64"), load_taps("c320_s3_d5_161")
    return {
        "8*8*5": [(8, h1), (8, h2), (5, h3)],                               # BASELINE config 3 (250 kS/s)
        "8*8*10": [(8, lowpass(48, 0.05)), (8, lowpass(51, 0.05)), (10, lowpass(287, 0.04))],   # the 125 kS/s plan's shape
        "8*8*4": [(8, h1), (8, h2), (4, lowpass(33, 0.1))],                 # one tile per group
    }


@pytest.mark.parametrize("plan", ["8*8*5", "8*8*10", "8*8*4"])
@pytest.mark.parametrize("mix", [False, True])
@pytest.mark.parametrize("sched", ["3,-1,0", "7,0,0", "5,100,0", "4,50,5", "512,-1,0"])
def test_fused_cascade_vs_oracle(pkg, dev, O, monkeypatch, tune, plan, mix, sched):
    blocks, dyn, chunk = sched.split(",")
    monkeypatch.setenv("PDDC_FUSE3", "1")                # the one-kernel cascade is opt-in (slower than pair + tail); read at create
    monkeypatch.setenv("PDDC_FIR8_BLOCKS", blocks)       # (read when a pipeline is created)
    tune("fir8_dyn_pct", int(dyn))
    tune("fir8_chunk", int(chunk))
    stages = plans()[plan]
    # batches in tiles; 0.25 = a batch that is not whole tiles (unfused path on the same state)
    sizes = [3, 37, 5, 0.25, 1, 64, 12, 0.5, 23]
    cuts = [0]
    for t in sizes:
        cuts.append(cuts[-1] + int(t * TILE))
    ns = cuts[-1]
    packed = O.lcg_bytes(6 * ns, 4242)
    ref = O.ddc_chain(packed, stages, freg=381178347, mix=mix)
    pipe = pkg.Pipeline(stages, mix=mix)
    pipe.set_freg(381178347)
    parts, used = [], []
    for a, b in zip(cuts[:-1], cuts[1:]):
        used.append(pipe.fused_cascade(b - a))
        parts.append(pipe.process(to_dev(packed[6 * a:6 * b], dev)).cpu().numpy().reshape(-1))
    pipe.check()
    assert used == [float(t).is_integer() for t in sizes]
    y = np.concatenate(parts)
    assert y.size == ref.size
    assert O.rel_err(y, ref) <= FIR_TOL, (plan, mix, sched, O.rel_err(y, ref))
    pipe.close()


def test_fused_cascade_equals_unfused_and_is_deterministic_under_load(pkg, dev, O, monkeypatch):
    """2^24 samples on the full grid: the fused cascade agrees with the unfused path to 1e-6 of full scale, and its
    own output does not depend on how the blocks are timed -- which chunk a block draws changes who computes an
    output, never its arithmetic -- also while another stream keeps the chip busy (uneven load on the seams)."""
    import torch
    stages = plans()["8*8*5"]
    ns = 1 << 24
    d_in = pkg.synth_lcg(6 * ns, 5150, 0, dev)
    monkeypatch.setenv("PDDC_FUSE3", "1")
    fused = pkg.Pipeline(stages, mix=True)
    fused.set_center_freq(7.1e6)
    assert fused.fused_cascade(ns)
    a1 = fused.process(d_in).clone()
    a2 = fused.process(d_in).clone()
    fused.check()
    monkeypatch.delenv("PDDC_FUSE3")
    plain = pkg.Pipeline(stages, mix=True)
    plain.set_center_freq(7.1e6)
    assert plain.fused_pair(ns) and not plain.fused_cascade(ns)
    b1 = plain.process(d_in).clone()
    b2 = plain.process(d_in).clone()
    monkeypatch.setenv("PDDC_FUSE3", "1")
    scale = float(b2.abs().max())
    assert a1.shape == b1.shape
    assert float((a1 - b1).abs().max()) / scale <= FIR_TOL
    assert float((a2 - b2).abs().max()) / scale <= FIR_TOL
    # the same two batches again, with a competing stream of unpack kernels on the chip
    side = torch.cuda.Stream(device=dev)
    junk_in = pkg.synth_lcg(6 * (1 << 22), 1, 0, dev)
    for rep in range(6):
        fused.reset()
        with torch.cuda.stream(side):
            for _ in range(4 + 3 * rep):
                pkg.unpack24_f32(junk_in, stream=side.cuda_stream)
        c1 = fused.process(d_in).clone()
        c2 = fused.process(d_in).clone()
        fused.check()
        side.synchronize()
        assert torch.equal(c1, a1) and torch.equal(c2, a2), rep
    fused.close()
    plain.close()


def test_fused_cascade_checkpoint_and_retune(pkg, dev, O, monkeypatch):
    """The fused cascade shares its stream state with the unfused kernels: a checkpoint taken behind a fused batch
    restores into a fresh pipeline, and a retune between two fused batches is phase-continuous (oracle's retuned NCO)."""
    monkeypatch.setenv("PDDC_FUSE3", "1")
    stages = plans()["8*8*5"]
    nb = 20 * TILE
    packed = O.lcg_bytes(6 * 3 * nb, 99)
    f1, f2 = 381178347, 123456789
    ref = O.ddc_chain_retuned(packed, stages, [(0, f1), (2 * nb, f2)])
    pipe = pkg.Pipeline(stages, mix=True)
    pipe.set_freg(f1)
    y0 = pipe.process(to_dev(packed[:6 * nb], dev)).cpu().numpy().reshape(-1)
    blob = pipe.save_state()
    pipe.close()
    pipe = pkg.Pipeline(stages, mix=True)
    pipe.set_freg(f1)
    pipe.restore_state(blob)
    assert pipe.fused_cascade(nb)
    y1 = pipe.process(to_dev(packed[6 * nb:12 * nb], dev)).cpu().numpy().reshape(-1)
    pipe.set_freg(f2)
    y2 = pipe.process(to_dev(packed[12 * nb:], dev)).cpu().numpy().reshape(-1)
    pipe.check()
    y = np.concatenate([y0, y1, y2])
    assert y.size == ref.size
    assert O.rel_err(y, ref) <= FIR_TOL
    pipe.close()


# ------------------------------------------------------------------ overlap mode (the tail under the next batch's pair)
@pytest.mark.parametrize("i8x", [1, 0])
@pytest.mark.parametrize("plan", ["8*8*5", "8*8*10", "8*8*4*5", "8*10", "8*5", "8*7", "10*5", "5*4"])
def test_overlap_mode_is_bit_identical_and_fenced(pkg, dev, O, plan, i8x):
    """pddc_pipeline_set_overlap: the last stage -- behind the fused pair, or behind an unfused /8 first stage -- rides
    along with the NEXT batch's launch (extra thread blocks of the first stage's grid).  Same arithmetic, same order per stage -- so the outputs are bit-identical to the
    in-line pipeline, whatever mix of whole-tile batches (carried) and odd ones (in line, fenced by the library) the
    stream is cut into; a retune in between; outputs read only behind pddc_pipeline_fence.  (The four-stage plan is
    not carried: it must simply still be right with the mode switched on.)  i8x = 1: tuned first stages take k_fir_i8x where
    the batch allows (whole 8192-sample tiles: no tail is carried then, it runs in line); i8x = 0: the vector kernels
    throughout, every whole-tile batch carried.  The second size list grows the batch right behind a carried one: the
    buffers a held-back tail reads must not be freed under it."""
    import torch
    pl = plans()
    pl["8*8*4*5"] = pl["8*8*4"] + [(5, lowpass(41, 0.08))]
    pl["8*10"] = [(8, lowpass(56, 0.05)), (10, lowpass(287, 0.04))]          # the 1 MS/s plan's shape: k_firp's body rides along
    pl["8*5"] = [(8, lowpass(56, 0.05)), (5, lowpass(144, 0.08))]            # 2 MS/s
    pl["8*7"] = [(8, lowpass(56, 0.05)), (7, lowpass(99, 0.06))]             # a decimation k_firp is not built for: generic body
    pl["10*5"] = [(10, lowpass(69, 0.04)), (5, lowpass(144, 0.08))]          # 1.6 MS/s: k_firp reads the packed samples and carries
    pl["5*4"] = [(5, lowpass(41, 0.08)), (4, lowpass(33, 0.1))]
    stages = pl[plan]
    for sizes in ([8, 8, 3, 0.25, 16, 1, 1, 1, 0.5, 40, 8], [8, 8, 32, 8, 3, 3, 96, 3, 3, 3]):
        cuts = [0]
        for t in sizes:
            cuts.append(cuts[-1] + int(t * TILE))
        d_in = pkg.synth_lcg(6 * cuts[-1], 777, 0, dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        outs = {}
        for mode in ("inline", "overlap"):
            pipe = pkg.Pipeline(stages, mix=True)
            pipe.set_option("i8x", i8x)
            pipe.set_freg(381178347)
            if mode == "overlap":
                pipe.set_overlap(True)
            bufs = []
            for k, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
                if k == 5:
                    pipe.set_freg(123456789)
                o = torch.zeros((pipe.max_output(b - a) + 1, 2), dtype=torch.float32, device=dev)
                n = pipe.process_ptr(d_in[6 * a:6 * b].data_ptr(), b - a, o.data_ptr(), o.shape[0], st)
                bufs.append((o, n))
            pipe.fence(st)
            torch.cuda.synchronize()
            outs[mode] = torch.cat([o[:n] for o, n in bufs])
            pipe.close()
        assert outs["inline"].shape == outs["overlap"].shape and outs["inline"].shape[0] > 0
        assert torch.equal(outs["inline"], outs["overlap"])
        packed = d_in.cpu().numpy()
        ref = O.ddc_chain_retuned(packed, stages, [(0, 381178347), (cuts[5], 123456789)])
        assert O.rel_err(outs["overlap"].cpu().numpy().reshape(-1), ref) <= FIR_TOL


def test_overlap_mode_with_a_caller_workspace_and_checkpoint(pkg, dev, O):
    """the double-buffered workspace half comes from the caller's workspace too (set_overlap before set_workspace), and a
    checkpoint taken in overlap mode (save_state synchronises) resumes in an in-line pipeline bit-identically"""
    import torch
    stages = plans()["8*8*5"]
    nb = 32 * TILE
    d_in = pkg.synth_lcg(6 * 4 * nb, 31, 0, dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    ref_pipe = pkg.Pipeline(stages, mix=True)
    ref_pipe.set_option("i8x", 0)              # (the carried tail belongs to k_fir8's pair; k_fir_i8x runs its tail in line)
    ref_pipe.set_freg(381178347)
    want = torch.cat([ref_pipe.process(d_in[6 * k * nb:6 * (k + 1) * nb]).clone() for k in range(4)])
    ref_pipe.close()
    pipe = pkg.Pipeline(stages, mix=True)
    pipe.set_option("i8x", 0)
    pipe.set_freg(381178347)
    small = pipe.workspace_size(nb)
    pipe.set_overlap(True)
    need = pipe.workspace_size(nb)
    assert need > small
    ws = torch.empty(need + 256, dtype=torch.uint8, device=dev)
    base = (ws.data_ptr() + 255) // 256 * 256
    pipe.set_workspace(base, need, nb)
    outs = []
    for k in range(2):
        o = torch.zeros((pipe.max_output(nb) + 1, 2), dtype=torch.float32, device=dev)
        n = pipe.process_ptr(d_in[6 * k * nb:].data_ptr(), nb, o.data_ptr(), o.shape[0], st)
        outs.append((o, n))
    with pytest.raises(Exception):
        pipe.save_state()                      # a tail is still held back: the state would miss it
    pipe.fence(st)
    blob = pipe.save_state()
    pipe.close()
    torch.cuda.synchronize()
    got = [o[:n] for o, n in outs]
    pipe = pkg.Pipeline(stages, mix=True)
    pipe.set_option("i8x", 0)
    pipe.set_freg(381178347)
    pipe.restore_state(blob)
    got += [pipe.process(d_in[6 * k * nb:6 * (k + 1) * nb]).clone() for k in (2, 3)]
    pipe.close()
    assert torch.equal(torch.cat(got), want)
