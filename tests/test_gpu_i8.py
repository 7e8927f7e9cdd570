"""GPU tests (-m gpu) of k_fir_i8x's PLAIN form: the untuned decimate-by-8 first stage on the int8 matrix cores (DESIGN.md 4;
until round 5 this file was about round 3's k_fir_i8, whose work -- binary16-stored taps included -- the plain form took
over).  The wire bytes are the operand -- three int8 planes per component --, the taps are four planes of
balanced base-256 digits, int32 accumulation is exact; what is left is the tap quantisation (2^-31 of the largest tap)
and three dropped low-order plane products.  Same bar as every FIR path here: max|y - ref| / max|ref| <= 1e-6 against
the CPU oracle, on the same stream state as k_fir8 (history, hist_out), so the two kernels can alternate
batch by batch."""
import numpy as np
import pytest

from conftest import load_taps

pytestmark = pytest.mark.gpu
FIR_TOL = 1e-6


def to_dev(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def lowpass(ntaps, cutoff):
    k = np.arange(ntaps) - (ntaps - 1) / 2.0
    h = np.sinc(2 * cutoff * k) * np.hamming(ntaps)
    return (h / h.sum()).astype(np.float32)


def run(pkg, dev, stages, packed, cuts, monkeypatch=None, no_i8_at=(), fp16=False):
    pipe = pkg.Pipeline(stages, taps_fp16=fp16)
    parts, on = [], []
    for k, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        if monkeypatch is not None:
            pipe.set_option("no_i8", 1 if k in no_i8_at else 0)      # kernel selection is API state, per pipeline
        on.append(bool(pipe.on_i8(b - a)))
        parts.append(pipe.process(to_dev(packed[6 * a:6 * b], dev)).cpu().numpy().reshape(-1))
    pipe.close()
    return np.concatenate(parts), on


@pytest.mark.parametrize("fp16", [False, True])
@pytest.mark.parametrize("ntaps", [20, 33, 65, 100, 127, 128, 129, 160, 200, 255, 256])
def test_i8_first_stage_vs_oracle_ragged_batches(pkg, dev, O, ntaps, fp16):
    """Tap counts over the kernel's whole range (all four history lengths); batches of whole tiles (8192), ragged ones, one
    of a single group of 8 behind the history length, and tiny ones (below the history: k_fir8's generic history path on the
    same state).  Both ways the operand reaches the matrix waves: the host's table, and binary16-STORED taps quantised by
    the waves themselves (PDDC_F_TAPS_FP16; the oracle then filters with the rounded taps)."""
    h = load_taps("d8_255") if ntaps == 255 else load_taps("d8_127") if ntaps == 127 else lowpass(ntaps, 0.05)
    if fp16:
        h = h.astype(np.float16).astype(np.float32)
    hist = 32 if ntaps <= 32 else 64 if ntaps <= 64 else 128 if ntaps <= 128 else 256
    sizes = [8192 * 3, 8192 + 8, 264, 8, 128, 256, 8192 * 40 + 4096 + 16, 1 << 20, 8192 * 2 - 8]
    cuts = np.concatenate([[0], np.cumsum(sizes)])
    packed = O.lcg_bytes(6 * int(cuts[-1]), 2026)
    ref = O.ddc_chain(packed, [(8, h)])
    y, on = run(pkg, dev, [(8, h)], packed, cuts, fp16=fp16)
    assert on == [s >= hist for s in sizes]
    assert y.size == ref.size
    assert O.rel_err(y, ref) <= FIR_TOL, (ntaps, O.rel_err(y, ref))


def test_i8_and_fp32_kernels_alternate_on_one_stream(pkg, dev, O, monkeypatch):
    """k_fir_i8x and k_fir8 keep the same stream state (256 packed history samples): switching between them batch by
    batch -- option no_i8 flipped between calls -- still gives the oracle's stream, and the two agree with each other to 1e-6."""
    h = load_taps("d8_255")
    sizes = [1 << 16, 8192 * 5 + 24, 1 << 15, 1 << 17, 8192, 1 << 16]
    cuts = np.concatenate([[0], np.cumsum(sizes)])
    packed = O.lcg_bytes(6 * int(cuts[-1]), 77)
    ref = O.ddc_chain(packed, [(8, h)])
    y_mixed, on = run(pkg, dev, [(8, h)], packed, cuts, monkeypatch, no_i8_at=(1, 3, 4))
    assert on == [True, False, True, False, False, True]
    y_i8, _ = run(pkg, dev, [(8, h)], packed, cuts, monkeypatch)
    y_f32, on = run(pkg, dev, [(8, h)], packed, cuts, monkeypatch, no_i8_at=range(len(sizes)))
    assert not any(on)
    for y in (y_mixed, y_i8, y_f32):
        assert O.rel_err(y, ref) <= FIR_TOL
    assert O.rel_err(y_i8, y_f32) <= FIR_TOL


def test_i8_extreme_samples_and_one_signed_taps(pkg, dev, O):
    """The accumulators' bound and the dropped plane products at their worst: every sample at +-full scale (bytes 0xff /
    0x00 in every plane: the planes' digits at -128 / +127), taps all of one sign and nearly equal -- every term of every
    plane product has the same sign."""
    h = (np.ones(256, np.float32) / 256 * (1 + 1e-3 * np.arange(256))).astype(np.float32)
    rng = np.random.default_rng(5)
    ns = 8192 * 6 + 40
    v = np.where(rng.random((ns, 2)) < 0.5, -(1 << 23), (1 << 23) - 1).astype(np.int64)
    v[: ns // 3] = (1 << 23) - 1                                 # a long run of the positive extreme ...
    v[ns // 3: 2 * ns // 3] = -(1 << 23)                         # ... and of the negative one
    b = np.zeros((ns, 2, 3), np.uint8)
    for i in range(3):
        b[:, :, i] = (v >> (8 * i)) & 0xFF
    packed = b.reshape(-1)
    ref = O.ddc_chain(packed, [(8, h)])
    y, on = run(pkg, dev, [(8, h)], packed, [0, 8192 * 2, ns])
    assert all(on)
    assert O.rel_err(y, ref) <= FIR_TOL, O.rel_err(y, ref)


def test_i8_binary16_taps_set_taps_and_checkpoint(pkg, dev, O):
    """The kernel's tap table follows the pipeline's taps: binary16-rounded values (config 5's fp16 leg), new taps from
    pddc_pipeline_set_taps, and a checkpoint taken in one pipeline and restored into another continue the stream."""
    import ctypes as C
    h = load_taps("d8_255")
    h16 = h.astype(np.float16).astype(np.float32)
    ns = 8192 * 9
    packed = O.lcg_bytes(6 * ns, 31)
    pipe = pkg.Pipeline([(8, h)], taps_fp16=True)
    assert pipe.on_i8(ns)
    y = pipe.process(to_dev(packed, dev)).cpu().numpy().reshape(-1)
    assert O.rel_err(y, O.ddc_chain(packed, [(8, h16)])) <= FIR_TOL
    pipe.close()
    # binary16 STORAGE: with the flag the device holds 2 bytes a tap and the kernel's matrix waves quantise them into their
    # operand registers; without it the host builds the int8 operand table from the same values -- the same integers,
    # so the same bits, for every history length (1..32, ..64, ..128, ..256 taps) and for taps that need the rounding
    # (subnormal binary16 values far below the largest tap)
    for g in (h16, lowpass(100, 0.05).astype(np.float16).astype(np.float32), lowpass(48, 0.05).astype(np.float16).astype(np.float32),
              lowpass(24, 0.05).astype(np.float16).astype(np.float32),
              (h16 * (1.0 + 0.0 * h16) * np.where(np.arange(h16.size) % 7 == 0, 2.0 ** -14, 1.0)).astype(np.float16).astype(np.float32)):
        outs = []
        for flag in (False, True):
            pipe = pkg.Pipeline([(8, g)], taps_fp16=flag)
            assert pipe.on_i8(ns) == 2
            outs.append(pipe.process(to_dev(packed, dev)).cpu().numpy().reshape(-1))
            pipe.close()
        assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
    # new taps under the flag: the binary16 array on the device is replaced, not a stale table read
    g = lowpass(233, 0.03)
    g16 = g.astype(np.float16).astype(np.float32)
    pipe = pkg.Pipeline([(8, h)], taps_fp16=True)
    pkg.check(pkg.ddc_lib().pddc_pipeline_set_taps(pipe._h, 0, g.ctypes.data_as(C.POINTER(C.c_float)), g.size))
    pipe.reset()
    assert pipe.on_i8(ns)
    y_new = pipe.process(to_dev(packed, dev)).cpu().numpy().reshape(-1)
    pipe.close()
    pipe = pkg.Pipeline([(8, g16)])
    y_ref = pipe.process(to_dev(packed, dev)).cpu().numpy().reshape(-1)
    pipe.close()
    assert np.array_equal(y_new.view(np.uint32), y_ref.view(np.uint32))
    pipe = pkg.Pipeline([(8, h)])
    pkg.check(pkg.ddc_lib().pddc_pipeline_set_taps(pipe._h, 0, g.ctypes.data_as(C.POINTER(C.c_float)), g.size))
    pipe.reset()
    half = 8192 * 4 + 808
    y1 = pipe.process(to_dev(packed[:6 * half], dev)).cpu().numpy().reshape(-1)
    blob = pipe.save_state()
    other = pkg.Pipeline([(8, g)])
    other.restore_state(blob)
    y2 = other.process(to_dev(packed[6 * half:], dev)).cpu().numpy().reshape(-1)
    assert O.rel_err(np.concatenate([y1, y2]), O.ddc_chain(packed, [(8, g)])) <= FIR_TOL
    pipe.close()
    other.close()


def test_which_first_stages_run_where(pkg, dev):
    h = load_taps("d8_255")
    p = pkg.Pipeline([(8, h)], mix=True)
    assert p.on_i8(1 << 20) == 2                                 # tuned: k_fir_i8x, the NCO folded into the taps
    p.set_option("i8x", 0)
    assert p.on_i8(1 << 20) == 0                                 # ... unless switched off: k_fir8 mixes in floats
    p.close()
    p = pkg.Pipeline([(8, h)])
    assert p.on_i8(1 << 20) == 2                                 # untuned: k_fir_i8x's plain form (round 4)
    p.set_option("i8x_plain", 0)
    assert p.on_i8(1 << 20) == 0                                 # ... or the vector kernel (round 3's k_fir_i8 is gone)
    p.close()
    p = pkg.Pipeline([(8, h)], taps_fp16=True)
    assert p.on_i8(1 << 20) == 2                                 # binary16 tap STORAGE: the plain form's matrix waves quantise them
    p.close()
    p = pkg.Pipeline([(8, h)], taps_fp16=True, mix=True)
    assert p.on_i8(1 << 20) == 2                                 # ... tuned: tables from the host, built from the rounded values
    p.close()
    p = pkg.Pipeline([(8, load_taps("c320_s1_d8_32"))])
    assert p.on_i8(1 << 20) == 2                                 # short untuned stages too
    p.set_option("i8x_plain", 0)
    assert p.on_i8(1 << 20) == 0                                 # ... or the vector kernel
    p.close()
    p = pkg.Pipeline([(8, h)], no_fast=True)
    assert not p.on_i8(1 << 20)
    p.close()
    with pytest.raises(pkg.PddcError):
        pkg.Pipeline([(8, h)]).set_option("no_such_option", 1)

def _tone_packed(ns, f_cyc_per_sample, amp=(1 << 23) - 1):
    """a full-scale complex tone as exact 24-bit integers (what the ADC side would deliver), packed"""
    n = np.arange(ns, dtype=np.float64)
    v = np.stack([np.rint(amp * np.cos(2 * np.pi * f_cyc_per_sample * n)), np.rint(amp * np.sin(2 * np.pi * f_cyc_per_sample * n))],
                 axis=1).astype(np.int64)
    b = np.zeros((ns, 2, 3), np.uint8)
    for i in range(3):
        b[:, :, i] = (v >> (8 * i)) & 0xFF
    return b.reshape(-1)


def _floor(y, ref, skip):
    """(max |y - ref| absolute, highest residual spur in dBFS: Hann-windowed spectrum, a full-scale tone = 0 dBFS)"""
    r = (y.astype(np.float64) - ref.astype(np.float64)).reshape(-1, 2)[skip:]
    c = r[:, 0] + 1j * r[:, 1]
    n = 1 << int(np.log2(c.size))
    w = np.hanning(n)
    spec = np.abs(np.fft.fft(c[:n] * w)) / w.sum()
    return float(np.abs(r).max()), 20 * np.log10(max(spec.max(), 1e-300))


@pytest.mark.parametrize("fp16", [False, True])
@pytest.mark.parametrize("design", ["d8_255", "kaiser120"])
def test_int8_path_numeric_floor_in_dbfs(pkg, dev, O, design, fp16):
    """The floor of the int8 matrix-core path measured ABSOLUTELY, not relative to max|ref|: a full-scale tone far out of
    band (the output is the filter's stop band, -80 .. -120 dB: whatever the kernel adds shows) and a one-signed full-scale
    DC (every term of every plane product with the same sign) through the 255-tap fixture and a 120-dB Kaiser design.
    max |y - oracle| <= 2e-7 of full scale for both, and no residual spur of the tone case above -150 dBFS -- with the
    host-built table and with binary16-stored taps (oracle on the same binary16 values)."""
    from scipy.signal import firwin
    h = load_taps("d8_255") if design == "d8_255" else firwin(255, 0.8 / 16, window=("kaiser", 0.1102 * (120 - 8.7))).astype(np.float32)
    if fp16:
        h = h.astype(np.float16).astype(np.float32)
    ns = 8192 * 72
    for name, packed in (("tone", _tone_packed(ns, 0.3137)),
                         ("dc", _tone_packed(ns, 0.0)[: 6 * ns].reshape(-1, 6).copy().reshape(-1))):
        if name == "dc":                                          # I = +FS, Q = -FS - 1: both rails one-signed
            v = np.zeros((ns, 2), np.int64)
            v[:, 0], v[:, 1] = (1 << 23) - 1, -(1 << 23)
            b = np.zeros((ns, 2, 3), np.uint8)
            for i in range(3):
                b[:, :, i] = (v >> (8 * i)) & 0xFF
            packed = b.reshape(-1)
        ref = O.ddc_chain(packed, [(8, h)])
        pipe = pkg.Pipeline([(8, h)], taps_fp16=fp16)
        assert pipe.on_i8(ns) == 2                              # k_fir_i8x's plain form either way (binary16 STORAGE: quantised in the kernel)
        y = pipe.process(to_dev(packed, dev)).cpu().numpy().reshape(-1)
        pipe.close()
        err, spur = _floor(y, ref, skip=64)
        assert err <= 2e-7, (design, fp16, name, err)
        if name == "tone":
            assert float(np.abs(ref.reshape(-1, 2)[64:]).max()) < 3e-4       # (the tone really is in the stop band)
            assert spur <= -150.0, (design, fp16, spur)


@pytest.mark.parametrize("ntaps", [48, 127, 255])
def test_tuned_matrix_core_path_numeric_floor_in_dbfs(pkg, dev, O, ntaps):
    """The same floor for k_fir_i8x (the NCO folded into the taps, one rotation per output): a full-scale tone that the
    mix leaves far out of band; max |y - oracle| <= 2e-7 absolute, no residual spur above -150 dBFS."""
    from scipy.signal import firwin
    h = firwin(ntaps, 0.8 / 16, window=("kaiser", 0.1102 * (100 - 8.7))).astype(np.float32)
    ns = 8192 * 72
    freg = 381178347                                              # 7.1 MHz: 0.08875 cycles per sample
    packed = _tone_packed(ns, 0.08875 + 0.2931)
    ref = O.ddc_chain(packed, [(8, h)], freg=freg, mix=True)
    pipe = pkg.Pipeline([(8, h)], mix=True)
    pipe.set_freg(freg)
    assert pipe.on_i8(ns) == 2
    y = pipe.process(to_dev(packed, dev)).cpu().numpy().reshape(-1)
    pipe.close()
    err, spur = _floor(y, ref, skip=64)
    assert float(np.abs(ref.reshape(-1, 2)[64:]).max()) < 3e-3
    assert err <= 2e-7 and spur <= -150.0, (ntaps, err, spur)
