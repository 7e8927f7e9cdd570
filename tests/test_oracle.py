"""CPU tests: the oracle against the reference outputs recorded in
tests/golden and against its own second (numpy) restatement."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLD, load_taps


@pytest.fixture(scope="module")
def gold():
    return json.load(open(os.path.join(GOLD, "unpack_golden.json")))


def test_unpack_kat_reference_table(O, gold):
    for k in gold["kat"]:
        p = O.pack24(np.array([k["code24"]]), np.array([k["code24"]]))
        for fn in (O.unpack24_f32, O.unpack24_f32_numpy):
            f = fn(p)
            assert f.view(np.uint32)[0] == k["float_bits"] and f.view(np.uint32)[1] == k["float_bits"]
        for fn in (O.unpack24_i32, O.unpack24_i32_numpy):
            assert fn(p)[0] == k["int32"] and fn(p)[1] == k["int32"]


def test_unpack_range_and_extremes(O):
    p = O.pack24(np.array([0x7FFFFF, 0x800000]), np.array([0x800000, 0x7FFFFF]))
    f = O.unpack24_f32(p)
    assert f[0] == 1.0 and f[3] == 1.0
    assert f[1].view(np.uint32) == 0xBF800001      # -1.00000012, reference range note
    assert O.unpack24_f32(np.zeros(5, np.uint8)).size == 0   # buf_size/6 samples, tail ignored
    assert O.unpack24_f32(p[:11]).size == 2


def test_unpack_exhaustive_sha256_matches_reference(O, gold):
    v = np.arange(1 << 24, dtype=np.int64)
    packed = O.pack24(v, (~v) & 0xFFFFFF)
    f = O.unpack24_f32(packed)
    assert hashlib.sha256(f.tobytes()).hexdigest() == gold["sha256"]["exhaustive_f32"]
    i = O.unpack24_i32(packed)
    assert hashlib.sha256(i.tobytes()).hexdigest() == gold["sha256"]["exhaustive_i32"]
    # second restatement (numpy divide) agrees bit for bit on a slice of it
    sl = packed[: 6 * 300000]
    assert np.array_equal(O.unpack24_f32_numpy(sl).view(np.uint32), f[:600000].view(np.uint32))


def test_lcg_buffer_fixture(O, gold):
    b = O.lcg_bytes(6144, 12345)
    assert b[:12].tobytes().hex() == gold["sha256"]["lcg_6144_in_first12"]
    assert hashlib.sha256(b.tobytes()).hexdigest().startswith(gold["sha256"]["lcg_6144_in_prefix"])
    assert np.array_equal(b, np.fromfile(os.path.join(GOLD, "lcg_6144.in"), dtype=np.uint8))
    out = O.unpack24_f32(b)
    assert hashlib.sha256(out.tobytes()).hexdigest() == gold["sha256"]["lcg_6144_out_f32"]
    assert np.array_equal(out, np.fromfile(os.path.join(GOLD, "lcg_6144.f32.out"), dtype=np.float32))
    assert np.array_equal(O.lcg_bytes_numpy(50000, 777), O.lcg_bytes(50000, 777))


def test_nco_word_kats(O, gold):
    for hz, w in gold["nco_freg_kat"].items():
        assert O.nco_freg(float(hz)) == w
    assert O.nco_freg(7.1e6) == 0x16B851EB


def test_rate_selection_semantics(O):
    R = O.REFERENCE_RATES
    for i, r in enumerate(R):
        assert O.rate_index(r) == i
    assert O.rate_index(1) == 0 and O.rate_index(10 ** 9) == len(R) - 1
    # midpoint goes to the LOWER rate (perseus-sdr.c:799-807)
    assert O.rate_index((48000 + 95000) // 2) == 0
    assert O.rate_index((48000 + 95000) // 2 + 1) == 1
    assert O.rate_index(1800000) == 8 and O.rate_index(1800001) == 9


def test_preselector_choice(O):
    assert O.presel_id(7.1e6) == 5          # FLT_6
    assert O.presel_id(1.0e6) == 0 and O.presel_id(1.7e6) == 1
    assert O.presel_id(32e6) == 10 and O.presel_id(5e6, False) == 10


def test_fir_two_restatements_agree(O):
    rng = np.random.default_rng(3)
    x = rng.standard_normal(2 * 5000)
    for D, nt in ((8, 127), (5, 161), (1, 4), (3, 50)):
        h = rng.standard_normal(nt).astype(np.float32)
        a = O.fir_decim(x, h, D)
        b = O.fir_decim_numpy(x, h, D)
        assert a.size == b.size
        assert np.max(np.abs(a - b)) <= 1e-9 * np.max(np.abs(b))


def test_nco_mix_properties(O):
    x = np.zeros(2 * 64, np.float32)
    x[0::2] = 1.0
    assert np.allclose(O.nco_mix(x, 0), x.astype(np.float64))
    y = O.nco_mix(x, 1 << 30).reshape(-1, 2)            # fs/4: 1, -j, -1, j
    assert np.allclose(y[:4], [[1, 0], [0, -1], [-1, 0], [0, 1]], atol=1e-12)
    # phase is a pure function of the absolute index
    a = O.nco_mix(x, 381178347, n0=0).reshape(-1, 2)
    b = O.nco_mix(x[: 2 * 32], 381178347, n0=32).reshape(-1, 2)
    assert np.allclose(a[32:], b, atol=1e-12)


@pytest.mark.parametrize("name", ["d8_127", "d8_255"])
def test_ddc_golden_fixture_regression(O, name):
    meta = json.load(open(os.path.join(GOLD, "ddc_golden.json")))
    packed = O.lcg_bytes(6 * meta["ddc_d8_lcg_samples"], meta["lcg_seed"])
    y = O.ddc_chain(packed, [(8, load_taps(name))])
    exp = np.fromfile(os.path.join(GOLD, f"ddc_{name}_lcg.f32"), dtype=np.float32)
    assert np.array_equal(y, exp)
    # the float baseline path stays within the FIR tolerance of the double oracle
    yf = O.stage1_f32(packed, load_taps(name), 8, 2)
    assert O.rel_err(yf, exp) <= 1e-6


def test_cascade_fixture_regression(O):
    meta = json.load(open(os.path.join(GOLD, "ddc_golden.json")))
    tone = np.fromfile(os.path.join(GOLD, "tone_7101k.in"), dtype=np.uint8)
    st = [(d, load_taps(n)) for d, n in meta["c320_stages"]]
    y = O.ddc_chain(tone, st, freg=meta["freg"], mix=True)
    exp = np.fromfile(os.path.join(GOLD, "ddc_c320_tone.f32"), dtype=np.float32)
    assert np.array_equal(y, exp)


def test_taps_manifest(O):
    man = json.load(open(os.path.join(GOLD, "taps_manifest.json")))
    for name, m in man.items():
        h = load_taps(name)
        assert h.size == m["ntaps"] and abs(float(h.astype(np.float64).sum()) - 1.0) < 1e-6
        assert m["stop_atten_db"] >= 85


def test_fir_and_resampler_against_scipy_upfirdn(O):
    """Third, independent statement of the decimator and of the rational resampler: scipy's
    polyphase upfirdn (upsample by L, FIR, keep every D-th) on the same data.  The oracle is the
    definition of the FIR/NCO arithmetic (no reference arithmetic exists for it); this pins its
    indexing conventions (zero history, y[m] = sum h[k] x[mD-k], phase of the L/M stage)."""
    signal = pytest.importorskip("scipy.signal")
    rng = np.random.default_rng(7)
    x = rng.standard_normal(2 * 4000)                       # interleaved I/Q, float64
    xc = x[0::2] + 1j * x[1::2]
    for D, nt in ((8, 127), (5, 161), (1, 9), (10, 77)):
        h = (rng.standard_normal(nt) / nt).astype(np.float32)
        y = O.fir_decim(x, h, D)
        ref = signal.upfirdn(h.astype(np.float64), xc, up=1, down=D)[: y.size // 2]
        got = y[0::2] + 1j * y[1::2]
        assert np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max()), (D, nt)
    for L, M, per in ((12, 25, 20), (3, 2, 16), (19, 40, 30)):
        g = (rng.standard_normal(L * per) / per).astype(np.float32)
        y = O.resample(x, g, L, M)
        ref = signal.upfirdn(g.astype(np.float64), xc, up=L, down=M)[: y.size // 2]
        got = y[0::2] + 1j * y[1::2]
        assert got.size == (xc.size * L + M - 1) // M
        assert np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max()), (L, M)


def test_retuned_nco_is_a_phase_accumulator(O):
    """N4: the NCO retuned while it runs.  C statement vs the independent numpy one; one segment
    equals the plain NCO; the phase is continuous at a switch (no jump), and a window of the stream
    can be computed on its own (n0)."""
    x = O.unpack24_f32(O.lcg_bytes(6 * 6000, 3))
    assert np.array_equal(O.nco_mix_retuned(x, [(0, 381178347)]), O.nco_mix(x, 381178347, 0))
    segs = [(0, 381178347), (1024, 123456789), (3000, 4000000000), (3008, 1)]
    c = O.nco_mix_retuned(x, segs)
    assert np.max(np.abs(c - O.nco_mix_retuned_numpy(x, segs))) < 1e-12
    assert np.array_equal(O.nco_mix_retuned(x[2 * 2000:], segs, 2000), c[4000:])
    # continuity: with a constant input the output's phase advances by exactly the word in force
    ones = np.tile(np.array([1.0, 0.0], np.float32), 4000)
    z = O.nco_mix_retuned(ones, segs)
    ph = np.unwrap(np.angle(z[0::2] + 1j * z[1::2]))
    step = -np.diff(ph) / (2 * np.pi) * 2 ** 32
    for (a, w), b in zip(segs, [s[0] for s in segs[1:]] + [4000]):
        ww = w if w < 2 ** 31 else w - 2 ** 32
        assert np.allclose(step[a:b - 1], ww, atol=1e-3 * 2 ** 32 / (2 * np.pi) * 1e-6 + 2.0)


@pytest.mark.parametrize("case", ["d8", "c320", "d10x4"])
def test_chain_check_agrees_with_the_chain(O, case):
    """orc_chain_check -- every output of one batch of the periodic stream, in chunks with their halos (what the
    full-size GPU tests and bench.py's `verified` use) -- is the same definition as orc_ddc_chain: the chain's float32
    result over two periods of the stream passes for both batches, at a batch length that is not a multiple of the
    decimation (the second batch starts in another phase), and ONE wrong output anywhere is found and located."""
    rng = np.random.default_rng(7)

    def lp(n, c):
        k = np.arange(n) - (n - 1) / 2.0
        h = np.sinc(2 * c * k) * np.hamming(n)
        return (h / h.sum()).astype(np.float32)

    stages = {"d8": [(8, lp(127, 0.05))], "c320": [(8, lp(32, 0.05)), (8, lp(41, 0.05)), (5, lp(117, 0.08))],
              "d10x4": [(10, lp(54, 0.04)), (4, lp(93, 0.1))]}[case]
    dtot = int(np.prod([d for d, _ in stages]))
    ns = 8 * 70000 + (8 if case != "d8" else 0)                 # c320 / d10x4: not a multiple of the decimation
    packed = O.lcg_bytes(6 * ns, 4711)
    mix = case != "d8"
    freg = 381178347
    y = O.ddc_chain(np.concatenate([packed, packed]), stages, freg=freg, mix=mix).reshape(-1, 2)
    m1 = -(-ns // dtot)
    for first, seg in ((0, y[:m1]), (ns, y[m1:])):
        r = O.chain_check(packed, first, ns, stages, seg, freg=freg, mix=mix)
        assert r["n"] == seg.shape[0] and r["ok"] and r["max_rel_err"] < 2e-7, r     # float32 rounding of the chain's result
    bad = y[m1:].copy()
    k = int(rng.integers(0, bad.shape[0]))
    bad[k, 1] += 4e-6 * np.abs(y).max()
    r = O.chain_check(packed, ns, ns, stages, bad, freg=freg, mix=mix)
    assert not r["ok"] and r["n_bad"] == 1 and r["first_bad"] == k, (r, k)
    bad[k, 1] = np.nan
    assert not O.chain_check(packed, ns, ns, stages, bad, freg=freg, mix=mix)["ok"]
    with pytest.raises(RuntimeError):
        O.chain_check(packed, ns, ns, stages, bad[:-1], freg=freg, mix=mix)       # fewer outputs than the batch makes


@pytest.mark.parametrize("case", ["d8_127", "c320"])
def test_callback_style_stream_is_the_same_chain(O, case):
    """bench.py's single-thread CPU leg (orc_stream_f32_callback_style: 6144-byte callbacks as perseus-in.c:206-207
    delivers them, each unpacked as examples/perseustest.c:466-502 does, streaming float FIR stages) computes the chain the
    double oracle defines -- to float accumulation -- for a stream that does not end on a buffer or decimation boundary,
    and the vectorised unpack of the all-core leg (orc_stage1_f32) is bit-identical to a scalar float FIR on the scalar unpack."""
    stages = {"d8_127": [(8, load_taps("d8_127"))],
              "c320": [(8, load_taps("c320_s1_d8_32")), (8, load_taps("c320_s2_d8_64")), (5, load_taps("c320_s3_d5_161"))]}[case]
    mix = case == "c320"
    ns = 1024 * 37 + 8 * 11
    packed = O.lcg_bytes(6 * ns, 99)
    ref = O.ddc_chain(packed, stages, freg=381178347, mix=mix)
    got = O.stream_callback_style(packed, stages, freg=381178347, mix=mix)
    assert got.size == ref.size
    assert O.rel_err(got, ref) < 3e-6          # float accumulation and a float phasor recurrence (a CPU baseline, not the parity oracle)
    if case == "d8_127":
        y = O.stage1_f32(packed, stages[0][1], 8, 2)
        assert O.rel_err(y, ref) < 3e-6
        x = O.unpack24_f32(packed).reshape(-1, 2)
        h = stages[0][1]
        m = 3000                                # one output by hand from the scalar unpack: same products, same taps
        want = float(np.dot(h.astype(np.float64), x[8 * m - np.arange(h.size), 0].astype(np.float64)))
        assert abs(float(y[2 * m]) - want) < 1e-5
