"""GPU tests (-m gpu) of k_fir_i8x (ddc_fir_i8.hip): the tuned decimate-by-8 first stage on the int8 matrix cores with the
NCO folded into the taps -- y[m] = LO(n0 + 8m) sum_k (h[k] e^{+j theta k}) x_raw[8m - k] -- and the cascade's first two
stages as its fused pair.  This is the kernel every pipeline of the drop-in API runs behind perseus_set_ddc_center_freq /
perseus_start_async_input (perseus-sdr.c:556-692).  Bar: max|y - ref| / max|ref| <= 1e-6 against the CPU oracle's
mix-then-filter definition, through the C ABI, on the stream state k_fir8 keeps (packed history, 64 mixed
first-stage outputs), so the kernels can alternate batch by batch -- which they do around a retune."""
import numpy as np
import pytest

from conftest import load_taps

pytestmark = pytest.mark.gpu
FIR_TOL = 1e-6
FREG = 381178347                                   # 7.1 MHz (perseus-sdr.c:584), BASELINE config 3
TILE = 8192


def to_dev(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def lowpass(ntaps, cutoff):
    k = np.arange(ntaps) - (ntaps - 1) / 2.0
    h = np.sinc(2 * cutoff * k) * np.hamming(ntaps)
    return (h / h.sum()).astype(np.float32)


def run(pkg, dev, stages, packed, cuts, freg=FREG, mix=True, opts=None, retune_at=None, record=None):
    pipe = pkg.Pipeline(stages, mix=mix)
    for k, v in (opts or {}).items():
        pipe.set_option(k, v)
    if mix:
        pipe.set_freg(freg)
    parts = []
    for k, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        if retune_at and k in retune_at:
            pipe.set_freg(retune_at[k])
        if record is not None:
            record.append((pipe.on_i8(b - a), pipe.fused_pair(b - a)))
        parts.append(pipe.process(to_dev(packed[6 * a:6 * b], dev)).cpu().numpy().reshape(-1))
    pipe.close()
    return np.concatenate(parts)


@pytest.mark.parametrize("ntaps", [9, 32, 33, 48, 64, 65, 100, 127, 128, 160, 255, 256])
def test_tuned_first_stage_vs_oracle_ragged_batches(pkg, dev, O, ntaps):
    """Every geometry (32 / 64 / 128 / 256 samples of history; two tap sets per wave up to 64 taps, split over waves
    above), batches of whole tiles, ragged ones, more tiles than CUs (several tiles per block: the history travels inside
    LDS), tiny ones (below the history length: k_fir8's path on the same state)."""
    h = load_taps("d8_255") if ntaps == 255 else load_taps("d8_127") if ntaps == 127 else lowpass(ntaps, 0.05)
    hist = 32 if ntaps <= 32 else 64 if ntaps <= 64 else 128 if ntaps <= 128 else 256
    sizes = [TILE * 3, TILE + 8, 264, 8, 128, 256, TILE * 40 + 4096 + 16, TILE * 600, TILE * 2 - 8, TILE * 257]
    cuts = np.concatenate([[0], np.cumsum(sizes)])
    packed = O.lcg_bytes(6 * int(cuts[-1]), 2027)
    ref = O.ddc_chain(packed, [(8, h)], freg=FREG, mix=True)
    rec = []
    y = run(pkg, dev, [(8, h)], packed, cuts, record=rec)
    assert [r[0] for r in rec] == [2 if s >= hist else 0 for s in sizes]
    assert y.size == ref.size
    assert O.rel_err(y, ref) <= FIR_TOL, (ntaps, O.rel_err(y, ref))


@pytest.mark.parametrize("word", [1, 0x7FFFFFFF, 0xFFFFFFF0, 0x40000000, 123456789])
def test_tuning_words_at_the_edges(pkg, dev, O, word):
    h = load_taps("d8_127")
    ns = TILE * 5 + 808
    packed = O.lcg_bytes(6 * ns, 11)
    ref = O.ddc_chain(packed, [(8, h)], freg=word, mix=True)
    y = run(pkg, dev, [(8, h)], packed, [0, TILE * 2, ns], freg=word)
    assert O.rel_err(y, ref) <= FIR_TOL, (word, O.rel_err(y, ref))


@pytest.mark.parametrize("taps12", [(48, 56), (32, 64), (17, 33), (64, 64), (127, 56), (100, 40)])
@pytest.mark.parametrize("mix", [True, False])
def test_fused_pair_vs_oracle(pkg, dev, O, taps12, mix):
    """Stages 0 and 1 in one kernel: batches of one tile, a few, more tiles than CUs (warm-up tiles in front of every
    block's range), and a ragged batch in between (unfused path, same state)."""
    h1, h2 = lowpass(taps12[0], 0.05), lowpass(taps12[1], 0.05)
    sizes = [TILE, TILE * 3, TILE * 2 + 64, TILE * 300, TILE * 7, 512, TILE * 513]
    cuts = np.concatenate([[0], np.cumsum(sizes)])
    packed = O.lcg_bytes(6 * int(cuts[-1]), 4242)
    ref = O.ddc_chain(packed, [(8, h1), (8, h2)], freg=FREG if mix else 0, mix=mix)
    rec = []
    y = run(pkg, dev, [(8, h1), (8, h2)], packed, cuts, mix=mix, opts=None if mix else {"i8x_plain": 1}, record=rec)
    assert [r[1] for r in rec] == [2 if s % TILE == 0 else 0 for s in sizes], rec
    assert y.size == ref.size
    assert O.rel_err(y, ref) <= FIR_TOL, O.rel_err(y, ref)


def test_cascade_x320_on_the_matrix_cores_vs_oracle_and_vs_the_vector_kernels(pkg, dev, O):
    """BASELINE config 3's shape with the drop-in API's tap counts (48 / 56 / 144): pair on k_fir_i8x, the /5 tail in line;
    against the oracle and against the same plan on the vector kernels (option i8x = 0)."""
    stages = [(8, lowpass(48, 0.05)), (8, lowpass(56, 0.05)), (5, lowpass(144, 0.08))]
    sizes = [TILE * 40, TILE * 5, TILE * 320, TILE * 3]
    cuts = np.concatenate([[0], np.cumsum(sizes)])
    packed = O.lcg_bytes(6 * int(cuts[-1]), 320)
    ref = O.ddc_chain(packed, stages, freg=FREG, mix=True)
    y = run(pkg, dev, stages, packed, cuts)
    yv = run(pkg, dev, stages, packed, cuts, opts={"i8x": 0})
    assert y.size == ref.size == yv.size
    assert O.rel_err(y, ref) <= FIR_TOL, O.rel_err(y, ref)
    assert O.rel_err(yv, ref) <= FIR_TOL
    assert O.rel_err(y, yv) <= FIR_TOL


def test_retune_between_batches_goes_through_the_vector_kernel_once(pkg, dev, O):
    """A new tuning word takes effect at a batch boundary, phase-continuously; the batch behind it sees two words in its
    history window and runs on k_fir8 (which re-mixes its packed history with the old word), the next one is back on the
    matrix cores with tables for the new word -- single stage and fused pair."""
    f1, f2, f3 = FREG, 123456789, 0xC0000001
    for stages in ([(8, load_taps("d8_127"))], [(8, lowpass(48, 0.05)), (8, lowpass(56, 0.05))]):
        nb = TILE * 6
        packed = O.lcg_bytes(6 * 6 * nb, 77)
        ref = O.ddc_chain_retuned(packed, stages, [(0, f1), (2 * nb, f2), (4 * nb, f3)])
        rec = []
        y = run(pkg, dev, stages, packed, [k * nb for k in range(7)], freg=f1, retune_at={2: f2, 4: f3}, record=rec)
        assert [r[0] for r in rec] == [2, 2, 0, 2, 0, 2], rec
        assert O.rel_err(y, ref) <= FIR_TOL, O.rel_err(y, ref)


def test_matrix_and_vector_kernels_alternate_on_one_stream(pkg, dev, O):
    """option i8x flipped between calls: k_fir_i8x and k_fir8 (single stage and fused pair) continue each other's stream"""
    import torch
    for stages in ([(8, load_taps("d8_127"))], [(8, lowpass(48, 0.05)), (8, lowpass(56, 0.05))]):
        sizes = [TILE * 4, TILE * 9, TILE * 2, TILE * 300, TILE, TILE * 5]
        cuts = np.concatenate([[0], np.cumsum(sizes)])
        packed = O.lcg_bytes(6 * int(cuts[-1]), 5)
        ref = O.ddc_chain(packed, stages, freg=FREG, mix=True)
        pipe = pkg.Pipeline(stages, mix=True)
        pipe.set_freg(FREG)
        parts = []
        for k, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
            pipe.set_option("i8x", k % 2)
            assert pipe.on_i8(b - a) == (2 if k % 2 else 0)
            parts.append(pipe.process(to_dev(packed[6 * a:6 * b], dev)).cpu().numpy().reshape(-1))
        pipe.close()
        assert O.rel_err(np.concatenate(parts), ref) <= FIR_TOL


def test_checkpoint_resume_and_set_taps_on_the_matrix_core_path(pkg, dev, O):
    import ctypes as C
    stages = [(8, lowpass(48, 0.05)), (8, lowpass(56, 0.05))]
    ns = TILE * 12
    packed = O.lcg_bytes(6 * ns, 8)
    ref = O.ddc_chain(packed, stages, freg=FREG, mix=True)
    a = pkg.Pipeline(stages, mix=True)
    a.set_freg(FREG)
    half = TILE * 5
    y1 = a.process(to_dev(packed[:6 * half], dev)).cpu().numpy().reshape(-1)
    blob = a.save_state()
    b = pkg.Pipeline(stages, mix=True)
    b.restore_state(blob)
    assert b.fused_pair(ns - half) == 2
    y2 = b.process(to_dev(packed[6 * half:], dev)).cpu().numpy().reshape(-1)
    assert O.rel_err(np.concatenate([y1, y2]), ref) <= FIR_TOL
    # new taps: the tables follow
    g = lowpass(40, 0.03)
    pkg.check(pkg.ddc_lib().pddc_pipeline_set_taps(a._h, 0, g.ctypes.data_as(C.POINTER(C.c_float)), g.size))
    a.reset()
    y = a.process(to_dev(packed, dev)).cpu().numpy().reshape(-1)
    assert O.rel_err(y, O.ddc_chain(packed, [(8, g), stages[1]], freg=FREG, mix=True)) <= FIR_TOL
    a.close()
    b.close()


def test_plain_form_agrees_with_the_vector_kernel(pkg, dev, O):
    """Two independent kernels, one filter: the untuned form of k_fir_i8x (the default: exact integer accumulation on the
    matrix cores) and k_fir8 (option i8x_plain = 0: packed fp32 FMAs) on the same stream.  Both within their bars against
    the oracle -- 2e-7 and 1e-6 -- and within 1e-6 of each other.  (Until round 5 this compared with round 3's k_fir_i8,
    the same arithmetic in another kernel; that kernel is retired and its binary16-stored taps live on as the plain form's
    second operand source, compared bit for bit in test_gpu_i8.py.)"""
    for name in ("d8_127", "d8_255"):
        h = load_taps(name)
        n = TILE * 40 + 264
        packed = O.lcg_bytes(6 * n, 11)
        rec = []
        y_new = run(pkg, dev, [(8, h)], packed, [0, n], mix=False, opts={"i8x_plain": 1}, record=rec)
        y_old = run(pkg, dev, [(8, h)], packed, [0, n], mix=False, opts={"i8x_plain": 0}, record=rec)
        assert [int(k) for k, _ in rec] == [2, 0]
        ref = O.ddc_chain(packed, [(8, h)])
        assert O.rel_err(y_new, y_old) <= FIR_TOL
        assert O.rel_err(y_new, ref) <= 2e-7 and O.rel_err(y_old, ref) <= FIR_TOL


def test_untuned_first_stage_on_the_same_kernel(pkg, dev, O):
    """the no-NCO form of k_fir_i8x against the oracle, ragged batches, 48 / 127 / 255 taps"""
    for ntaps in (127, 255, 48):
        h = load_taps("d8_255") if ntaps == 255 else load_taps("d8_127") if ntaps == 127 else lowpass(ntaps, 0.05)
        sizes = [TILE * 3 + 8, TILE * 290, 264, TILE * 2]
        cuts = np.concatenate([[0], np.cumsum(sizes)])
        packed = O.lcg_bytes(6 * int(cuts[-1]), 6)
        ref = O.ddc_chain(packed, [(8, h)])
        y = run(pkg, dev, [(8, h)], packed, cuts, mix=False, opts={"i8x_plain": 1})
        assert O.rel_err(y, ref) <= FIR_TOL, (ntaps, O.rel_err(y, ref))


def test_full_scale_extremes_through_the_tuned_stage(pkg, dev, O):
    """every sample at +- full scale, taps of one sign at the 64-tap limit (two band products in one accumulator set) and at
    256 taps (four float partial products)"""
    rng = np.random.default_rng(5)
    ns = TILE * 6 + 40
    v = np.where(rng.random((ns, 2)) < 0.5, -(1 << 23), (1 << 23) - 1).astype(np.int64)
    v[: ns // 3] = (1 << 23) - 1
    v[ns // 3: 2 * ns // 3] = -(1 << 23)
    b = np.zeros((ns, 2, 3), np.uint8)
    for i in range(3):
        b[:, :, i] = (v >> (8 * i)) & 0xFF
    packed = b.reshape(-1)
    for n in (64, 256):
        h = (np.ones(n, np.float32) / n * (1 + 1e-3 * np.arange(n))).astype(np.float32)
        for word in (1 << 29, 3):
            ref = O.ddc_chain(packed, [(8, h)], freg=word, mix=True)
            y = run(pkg, dev, [(8, h)], packed, [0, TILE * 2, ns], freg=word)
            assert O.rel_err(y, ref) <= FIR_TOL, (n, word, O.rel_err(y, ref))


def test_every_walk_gives_the_same_bits(pkg, dev, O):
    """The chunk size of the round-robin walk (option i8x_chunk: 1 = tile-interleaved, large = contiguous ranges) and the
    grid size change which tiles take their history from memory and which from LDS, and which tiles compute their own
    porch for the fused second stage -- never a bit of the output.  The layout (option i8x_layout: which waves load,
    multiply and finish a tile) leaves the single stage's bits alone too; the fused second stage sums in another order in
    layout 1 (scalar FMAs on the loader waves), so layouts are compared with the oracle's tolerance there."""
    for stages in ([(8, load_taps("d8_127"))], [(8, lowpass(48, 0.05)), (8, lowpass(56, 0.05))]):
        sizes = [TILE * 300, TILE * 7, TILE * 1030]
        cuts = np.concatenate([[0], np.cumsum(sizes)])
        packed = O.lcg_bytes(6 * int(cuts[-1]), 12)
        ref = O.ddc_chain(packed, stages, freg=FREG, mix=True)
        first = None
        for lay in (0, 1, 2):
            base = None
            for opts in ({}, {"i8x_chunk": 1}, {"i8x_chunk": 2}, {"i8x_chunk": 3}, {"i8x_chunk": 100000}, {"i8x_blocks": 7, "i8x_chunk": 5},
                         {"i8x_blocks": 1}):
                y = run(pkg, dev, stages, packed, cuts, opts=dict(opts, i8x_layout=lay))
                if base is None:
                    base = y
                    assert O.rel_err(y, ref) <= FIR_TOL, (lay, O.rel_err(y, ref))
                assert np.array_equal(y.view(np.uint32), base.view(np.uint32)), (lay, opts)
            if first is None:
                first = base
            elif len(stages) == 1:
                assert np.array_equal(base.view(np.uint32), first.view(np.uint32)), lay
            else:
                assert O.rel_err(base, first) <= FIR_TOL


@pytest.mark.parametrize("layout", [0, 1, 2])
def test_layouts_stay_clean_under_repetition(pkg, dev, O, layout):
    """Sporadic faults need repetition to show (the packed-fp32 / store hazards beside matrix waves; with the
    SLP vectoriser on, this test's layout 1 fails within the first few runs -- tools/hazard_ab.sh): the same 600-tile
    batch SIXTY times per layout -- single stage in all three NCO forms (48, 127 and 255 taps), the fused pair, the
    decimate-by-10 form --, every run against the first run's bits and the first run against the oracle."""
    cases = ([(8, load_taps("d8_127"))], [(8, lowpass(48, 0.05))], [(8, load_taps("d8_255"))],
             [(8, lowpass(48, 0.05)), (8, lowpass(56, 0.05))], [(10, lowpass(51, 0.04))])
    for stages in cases:
        ns = (TILE if stages[0][0] == 8 else 10240) * 600
        packed = O.lcg_bytes(6 * ns, 2027)
        ref = O.ddc_chain(packed, stages, freg=FREG, mix=True)
        d_in = to_dev(packed, dev)
        pipe = pkg.Pipeline(stages, mix=True)
        pipe.set_option("i8x_layout", layout)
        pipe.set_freg(FREG)
        assert pipe.on_i8(ns) == 2
        first = None
        for rep in range(60):
            pipe.reset()
            y = pipe.process(d_in).cpu().numpy().reshape(-1)
            if first is None:
                first = y
                assert O.rel_err(y, ref) <= FIR_TOL
            assert np.array_equal(y.view(np.uint32), first.view(np.uint32)), (layout, len(stages), stages[0][0], len(stages[0][1]), rep)
        pipe.close()


@pytest.mark.parametrize("ntaps", [51, 57, 20])
def test_decimate_by_ten_first_stage_on_the_matrix_cores(pkg, dev, O, ntaps):
    """The 1.6 MS/s plan's first stage (decimate by 10, tuned; perseus-sdr.c:776-892 picks the rate) on k_fir_i8x's paired
    rows: batches of any multiple of 8 samples, so the decimation phase of a batch's first output walks through 0 .. 9
    (it goes into the taps as a delay of 0 .. 7 samples); a retune sends the one batch behind it through the vector kernel;
    with the plan's second stage behind it.  Against the oracle, and the kernel that ran is asserted."""
    h1, h2 = lowpass(ntaps, 0.04), lowpass(117, 0.08)
    sizes = [10240 * 3 + 8, 8, 10240 * 260 + 16, 64, 10240 * 2 + 24, 10240 * 5, 1016, 10240 * 300 + 8, 10240 * 7 + 32]
    cuts = np.concatenate([[0], np.cumsum(sizes)])
    packed = O.lcg_bytes(6 * int(cuts[-1]), 77)
    for stages in ([(10, h1)], [(10, h1), (5, h2)]):
        rec = []
        y = run(pkg, dev, stages, packed, cuts, record=rec)
        assert all(int(k) == 2 for k, _ in rec), rec
        ref = O.ddc_chain(packed, stages, freg=FREG, mix=True)
        assert y.size == ref.size and O.rel_err(y, ref) <= FIR_TOL, (len(stages), O.rel_err(y, ref))
        # the vector kernel on the same stream: the two agree to the tolerance's order, and i8x = 0 really is the other kernel
        rec0 = []
        y0 = run(pkg, dev, stages, packed, cuts, opts={"i8x": 0}, record=rec0)
        assert all(int(k) == 0 for k, _ in rec0)
        assert O.rel_err(y0, ref) <= FIR_TOL and not np.array_equal(y, y0)
    # a retune between batches: sample-accurate and phase-continuous, as on every path
    word2 = O.nco_freg(14.2e6)
    rec = []
    y = run(pkg, dev, [(10, h1), (5, h2)], packed, cuts, retune_at={3: word2, 6: FREG}, record=rec)
    ref = O.ddc_chain_retuned(packed, [(10, h1), (5, h2)], [(0, FREG), (int(cuts[3]), word2), (int(cuts[6]), FREG)])
    assert O.rel_err(y, ref) <= FIR_TOL, O.rel_err(y, ref)
    assert [int(k) for k, _ in rec].count(0) >= 1 and [int(k) for k, _ in rec].count(2) >= 5, rec


def test_decimate_by_ten_retune_storm(pkg, dev, O):
    """A retune every third batch while the batches' decimation phases cycle through all eight tap delays: every delay's
    table set is rebuilt for every new word, in the double buffers behind their `left` events (ddc_pipeline.cpp
    i8x_d10_prepare; until round 5 each rebuild was a hipDeviceSynchronize) -- against the retuned oracle, and the same stream
    again without waiting between the batches' submissions (the rebuilds then overlap launches still in flight)."""
    import torch
    h1, h2 = lowpass(51, 0.04), lowpass(117, 0.08)
    stages = [(10, h1), (5, h2)]
    nb = 36
    sizes = [10240 * 4 + 8 * (1 + k % 9) for k in range(nb)]                     # first outputs on every sample 0 .. 9 of a batch
    cuts = np.concatenate([[0], np.cumsum(sizes)])
    packed = O.lcg_bytes(6 * int(cuts[-1]), 31)
    words = [O.nco_freg(f) for f in (7.1e6, 14.2e6, 3.55e6, 21.0e6, 1.8e6, 10.1e6)]
    retune_at = {k: words[(k // 3) % len(words)] for k in range(3, nb, 3)}
    segs = [(0, FREG)] + [(int(cuts[k]), w) for k, w in sorted(retune_at.items())]
    ref = O.ddc_chain_retuned(packed, stages, segs)
    rec = []
    y = run(pkg, dev, stages, packed, cuts, retune_at=retune_at, record=rec)
    assert y.size == ref.size and O.rel_err(y, ref) <= FIR_TOL, O.rel_err(y, ref)
    assert [int(k) for k, _ in rec].count(2) >= nb // 2, rec                      # (the batch behind each retune takes the vector kernel)
    # the same without a host wait between batches: outputs stay on the device until the end
    pipe = pkg.Pipeline(stages, mix=True)
    pipe.set_freg(FREG)
    d_in = to_dev(packed, dev)
    outs = []
    for k, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        if k in retune_at:
            pipe.set_freg(retune_at[k])
        outs.append(pipe.process(d_in[6 * a:6 * b]))
    torch.cuda.synchronize()
    y2 = np.concatenate([o.cpu().numpy().reshape(-1) for o in outs])
    pipe.close()
    assert np.array_equal(y2.view(np.uint32), y.view(np.uint32))


def test_decimate_by_ten_walks_and_checkpoint(pkg, dev, O):
    """every walk / grid / layout of the decimate-by-10 form gives the same bits, and a checkpoint taken between batches
    continues them"""
    h1 = lowpass(51, 0.04)
    n = 10240 * 70 + 24
    packed = O.lcg_bytes(6 * 2 * n, 5)
    outs = []
    for opts in ({}, {"i8x_layout": 1}, {"i8x_layout": 2}, {"i8x_chunk": 3}, {"i8x_blocks": 7, "i8x_chunk": 2}):
        outs.append(run(pkg, dev, [(10, h1)], packed, [0, n, 2 * n], opts=opts))
    assert all(np.array_equal(outs[0], o) for o in outs[1:])
    a = pkg.Pipeline([(10, h1)], mix=True)
    a.set_freg(FREG)
    y1 = a.process(to_dev(packed[:6 * n], dev)).cpu().numpy().reshape(-1)
    blob = a.save_state()
    b = pkg.Pipeline([(10, h1)], mix=True)
    b.restore_state(blob)
    y2 = b.process(to_dev(packed[6 * n:], dev)).cpu().numpy().reshape(-1)
    assert np.array_equal(np.concatenate([y1, y2]), outs[0])
    a.close()
    b.close()
