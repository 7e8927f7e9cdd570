"""GPU parity tests (-m gpu): the HIP path, called through the C ABI
(include/perseus_ddc.h), against the CPU oracle on the same inputs.

Bars: unpack bit-exact; FIR / NCO stages max|y-ref|/max|ref| <= 1e-6
(BASELINE.json north_star; metric defined in oracle.rel_err / DESIGN.md).
"""
import hashlib
import importlib
import json
import os

import numpy as np
import pytest

from conftest import GOLD, load_taps

pytestmark = pytest.mark.gpu
FIR_TOL = 1e-6


def _torch():
    import torch
    return torch


def to_dev(a, dev):
    return _torch().from_numpy(np.ascontiguousarray(a)).to(dev)


# ------------------------------------------------------------------ plumbing
def test_native_library_loaded_and_device_visible(pkg, dev):
    L = pkg.ddc_lib()
    assert L.pddc_device_count() >= 1
    with open("/proc/self/maps") as f:
        assert "libperseus_ddc.so" in f.read()


# -------------------------------------------------------------------- unpack
def test_unpack_golden_kat(pkg, dev, O):
    g = json.load(open(os.path.join(GOLD, "unpack_golden.json")))
    codes = np.array([k["code24"] for k in g["kat"]], dtype=np.int64)
    # pad to a multiple of 8 samples; Q carries the reversed list
    packed = O.pack24(codes, codes[::-1])
    f = pkg.unpack24_f32(to_dev(packed, dev)).cpu().numpy()
    i = pkg.unpack24_i32(to_dev(packed, dev)).cpu().numpy()
    for n, k in enumerate(g["kat"]):
        assert f[n, 0].view(np.uint32) == k["float_bits"]
        assert i[n, 0] == k["int32"]
        assert f[len(codes) - 1 - n, 1].view(np.uint32) == k["float_bits"]


def test_unpack_lcg_fixture(pkg, dev):
    b = np.fromfile(os.path.join(GOLD, "lcg_6144.in"), dtype=np.uint8)
    exp = np.fromfile(os.path.join(GOLD, "lcg_6144.f32.out"), dtype=np.float32)
    got = pkg.unpack24_f32(to_dev(b, dev)).cpu().numpy().reshape(-1)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
    sha = json.load(open(os.path.join(GOLD, "unpack_golden.json")))["sha256"]["lcg_6144_out_f32"]
    assert hashlib.sha256(got.tobytes()).hexdigest() == sha


def test_unpack_exhaustive_2p24_bit_exact(pkg, dev, O):
    """Every 24-bit code in I, its complement in Q: the reference's own
    exhaustive vector, hashes recorded from the reference."""
    sha = json.load(open(os.path.join(GOLD, "unpack_golden.json")))["sha256"]
    v = np.arange(1 << 24, dtype=np.int64)
    packed = O.pack24(v, (~v) & 0xFFFFFF)
    d = to_dev(packed, dev)
    f = pkg.unpack24_f32(d).cpu().numpy()
    assert hashlib.sha256(f.tobytes()).hexdigest() == sha["exhaustive_f32"]
    i = pkg.unpack24_i32(d).cpu().numpy()
    assert hashlib.sha256(i.tobytes()).hexdigest() == sha["exhaustive_i32"]
    ref = O.unpack24_f32(packed)
    assert np.array_equal(f.reshape(-1).view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("ns", [0, 1, 7, 8, 9, 1023, 1024, 4099])
def test_unpack_ragged_sizes(pkg, dev, O, ns):
    packed = O.lcg_bytes(6 * ns + 16, 99)[: 6 * ns]
    buf = _torch().zeros(6 * ns + 64, dtype=_torch().uint8, device=dev)
    buf[: 6 * ns] = to_dev(packed, dev) if ns else buf[:0]
    got = pkg.unpack24_f32(buf[: 6 * ns]).cpu().numpy().reshape(-1)
    ref = O.unpack24_f32(packed) if ns else np.zeros(0, np.float32)
    assert got.size == 2 * ns
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_synth_lcg_matches_oracle(pkg, dev, O):
    n = 6 * 8 * 2000 + 5
    ref = O.lcg_bytes(n, 12345)
    got = pkg.synth_lcg(n, 12345, 0, dev).cpu().numpy()
    assert np.array_equal(got, ref)
    off = 6 * 1234 + 3
    got2 = pkg.synth_lcg(1000, 12345, off, dev).cpu().numpy()
    assert np.array_equal(got2, ref[off:off + 1000])
    # more chunks than the grid has threads (8192 blocks x 256 threads x 16 bytes = 32 MiB): every thread then walks on
    # by the grid stride with one affine jump; ragged end, odd offset beyond 2^32 steps
    big = (80 << 20) + 7
    refb = O.lcg_bytes(big, 777)
    gotb = pkg.synth_lcg(big, 777, 0, dev).cpu().numpy()
    assert np.array_equal(gotb, refb)
    far = (1 << 32) + 12345
    # the LCG has period 2^32 in its state: byte k + 2^32 equals byte k
    got3 = pkg.synth_lcg(4096, 777, far, dev).cpu().numpy()
    assert np.array_equal(got3, refb[12345:12345 + 4096])


# ------------------------------------------------------------- fused /8 FIR
@pytest.mark.parametrize("name", ["d8_127", "d8_255", "c320_s1_d8_32", "c320_s2_d8_64"])
def test_fused_decimate8_vs_oracle(pkg, dev, O, name):
    h = load_taps(name)
    ns = 8 * 9000 + 8 * 3                      # several tiles + ragged tail
    packed = O.lcg_bytes(6 * ns, 12345)
    pipe = pkg.Pipeline([(8, h)])
    assert pipe.fused
    y = pipe.process(to_dev(packed, dev)).cpu().numpy().reshape(-1)
    ref = O.ddc_chain(packed, [(8, h)])
    assert y.size == ref.size == 2 * (ns // 8)
    assert O.rel_err(y, ref) <= FIR_TOL
    pipe.close()


@pytest.mark.parametrize("name", ["d8_127", "d8_255"])
def test_fused_golden_fixture(pkg, dev, name):
    meta = json.load(open(os.path.join(GOLD, "ddc_golden.json")))
    from oracle import oracle as O
    ns = meta["ddc_d8_lcg_samples"]
    packed = O.lcg_bytes(6 * ns, meta["lcg_seed"])
    exp = np.fromfile(os.path.join(GOLD, f"ddc_{name}_lcg.f32"), dtype=np.float32)
    pipe = pkg.Pipeline([(8, load_taps(name))])
    y = pipe.process(to_dev(packed, dev)).cpu().numpy().reshape(-1)
    assert y.size == exp.size
    assert O.rel_err(y, exp) <= FIR_TOL
    pipe.close()


def test_stream_seams_match_single_shot(pkg, dev, O):
    """FIR history + NCO counter carried across process() calls: pushing the
    stream in uneven batches must equal one big batch (and the oracle)."""
    h = load_taps("d8_127")
    ns = 8 * 6000
    packed = O.lcg_bytes(6 * ns, 777)
    ref = O.ddc_chain(packed, [(8, h)], freg=381178347, mix=True)
    pipe = pkg.Pipeline([(8, h)], mix=True)
    pipe.set_center_freq(7.1e6)
    assert pipe.freg == 381178347
    cuts = [0, 8 * 16, 8 * 16 + 8 * 1024, 8 * 3000, 8 * 3001, ns]
    parts = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        parts.append(pipe.process(to_dev(packed[6 * a:6 * b], dev)).cpu().numpy().reshape(-1))
    y = np.concatenate(parts)
    assert y.size == ref.size
    assert O.rel_err(y, ref) <= FIR_TOL
    pipe.reset()
    y1 = pipe.process(to_dev(packed, dev)).cpu().numpy().reshape(-1)
    assert O.rel_err(y1, ref) <= FIR_TOL
    pipe.close()


def test_generic_path_equals_oracle_and_fused(pkg, dev, O):
    h = load_taps("d8_127")
    ns = 8 * 2500
    packed = O.lcg_bytes(6 * ns, 4242)
    ref = O.ddc_chain(packed, [(8, h)])
    slow = pkg.Pipeline([(8, h)], no_fast=True)
    assert not slow.fused
    y = slow.process(to_dev(packed, dev)).cpu().numpy().reshape(-1)
    assert O.rel_err(y, ref) <= FIR_TOL
    slow.close()


@pytest.mark.parametrize("D,nt", [(5, 161), (10, 77), (1, 9), (3, 1), (40, 401)])
def test_generic_decimators(pkg, dev, O, D, nt):
    rng = np.random.default_rng(D * 1000 + nt)
    h = (rng.standard_normal(nt) / nt).astype(np.float32)
    ns = 8 * 1300
    packed = O.lcg_bytes(6 * ns, 31337)
    ref = O.ddc_chain(packed, [(D, h)])
    pipe = pkg.Pipeline([(D, h)])
    # two uneven pushes exercise the decimation phase carry
    a = 8 * 401
    y = np.concatenate([pipe.process(to_dev(packed[:6 * a], dev)).cpu().numpy().reshape(-1),
                        pipe.process(to_dev(packed[6 * a:], dev)).cpu().numpy().reshape(-1)])
    assert y.size == ref.size
    assert O.rel_err(y, ref) <= FIR_TOL
    pipe.close()


# ------------------------------------------------------- config 3: x320 + NCO
def test_cascade_320_with_nco_fixture_and_oracle(pkg, dev, O):
    meta = json.load(open(os.path.join(GOLD, "ddc_golden.json")))
    tone = np.fromfile(os.path.join(GOLD, "tone_7101k.in"), dtype=np.uint8)
    exp = np.fromfile(os.path.join(GOLD, "ddc_c320_tone.f32"), dtype=np.float32)
    stages = [(d, load_taps(n)) for d, n in meta["c320_stages"]]
    pipe = pkg.Pipeline(stages, mix=True)
    pipe.set_freg(meta["freg"])
    assert pipe.decim == 320
    y = pipe.process(to_dev(tone, dev)).cpu().numpy().reshape(-1)
    assert y.size == exp.size
    assert O.rel_err(y, exp) <= FIR_TOL
    # spectral sanity: a 7.101 MHz tone lands at ~1 kHz at 250 kS/s
    z = y.reshape(-1, 2)[100:, 0] + 1j * y.reshape(-1, 2)[100:, 1]
    f_est = np.mean(np.diff(np.unwrap(np.angle(z)))) * 250e3 / (2 * np.pi)
    assert abs(f_est - 1000.0) < 1.0
    # same stream in batches that are NOT multiples of 320
    pipe.reset()
    cuts = [0, 8 * 1000, 8 * 1000 + 8 * 77, tone.size // 6]
    parts = [pipe.process(to_dev(tone[6 * a:6 * b], dev)).cpu().numpy().reshape(-1)
             for a, b in zip(cuts[:-1], cuts[1:])]
    y2 = np.concatenate(parts)
    assert y2.size == exp.size
    assert O.rel_err(y2, exp) <= FIR_TOL
    pipe.close()


def test_nco_mix_lcg_noise(pkg, dev, O):
    """NCO on full-scale noise at an awkward tuning word, generic + fused."""
    h = load_taps("c320_s1_d8_32")
    ns = 8 * 4096
    packed = O.lcg_bytes(6 * ns, 5)
    for freg in (1, 0x80000000, 0xFFFFFFFF, 381178347, 123456789):
        ref = O.ddc_chain(packed, [(8, h)], freg=freg, mix=True)
        pipe = pkg.Pipeline([(8, h)], mix=True)
        pipe.set_freg(freg)
        y = pipe.process(to_dev(packed, dev)).cpu().numpy().reshape(-1)
        assert O.rel_err(y, ref) <= FIR_TOL, freg
        pipe.close()


def test_push_host_roundtrip(pkg, dev, O):
    h = load_taps("d8_127")
    ns = 8 * 1024 * 3
    packed = O.lcg_bytes(6 * ns, 11)
    ref = O.ddc_chain(packed, [(8, h)])
    pipe = pkg.Pipeline([(8, h)])
    y = np.concatenate([pipe.push_host(packed[: 6 * 8192]).reshape(-1),
                        pipe.push_host(packed[6 * 8192:]).reshape(-1)])
    assert O.rel_err(y, ref) <= FIR_TOL
    pipe.close()


# ------------------------------------------------------------ error behaviour
def test_push_host_async_double_buffered(pkg, dev, O):
    """Pinned buffers, two batches in flight (H2D of k+1, kernels of k, D2H of k-1 on three
    streams), slots reused five times, a short last batch: the stream is the oracle's."""
    stages = [(8, load_taps("c320_s1_d8_32")), (8, load_taps("c320_s2_d8_64")), (5, load_taps("c320_s3_d5_161"))]
    batch = 4096 * 20
    cuts = [batch] * 5 + [320 * 7]
    total = sum(cuts)
    stream = O.lcg_bytes(6 * total, 4242)
    ref = O.ddc_chain(stream, stages, freg=381178347, mix=True)
    pipe = pkg.Pipeline(stages, mix=True)
    pipe.set_freg(381178347)
    cap = pipe.max_output(batch) + 1
    h_in = [pkg.PinnedBuffer(6 * batch) for _ in range(2)]
    h_out = [pkg.PinnedBuffer(8 * cap) for _ in range(2)]
    parts, pending, pos = [], None, 0
    for k, ns in enumerate(cuts):
        b = k & 1
        h_in[b].array[:6 * ns] = stream[6 * pos:6 * (pos + ns)]
        n, t = pipe.push_host_async(h_in[b].ptr, ns, h_out[b].ptr, cap)
        assert t == b
        if pending is not None:                       # collect batch k-1 while batch k is in flight
            pb, pn, pt = pending
            pipe.wait_ticket(pt)
            parts.append(h_out[pb].array[:8 * pn].view(np.float32).copy())
        pending = (b, n, t)
        pos += ns
    pipe.wait()
    pb, pn, pt = pending
    parts.append(h_out[pb].array[:8 * pn].view(np.float32).copy())
    y = np.concatenate(parts)
    assert y.size == ref.size
    assert O.rel_err(y, ref) <= FIR_TOL
    with pytest.raises(Exception):
        pipe.wait_ticket(5)
    pipe.close()
    for hb in h_in + h_out:
        hb.free()


def test_argument_errors(pkg, dev, O):
    h = load_taps("d8_127")
    pipe = pkg.Pipeline([(8, h)])
    t = _torch()
    buf = t.zeros(6 * 64 + 16, dtype=t.uint8, device=dev)
    out = t.zeros(64, dtype=t.float32, device=dev)
    with pytest.raises(pkg.PddcError) as e:          # not a multiple of 8 samples
        pipe.process_ptr(buf.data_ptr(), 12, out.data_ptr(), 32)
    assert e.value.code == pkg.PDDC_EINVAL
    with pytest.raises(pkg.PddcError) as e:          # misaligned input
        pipe.process_ptr(buf.data_ptr() + 2, 8, out.data_ptr(), 32)
    assert e.value.code == pkg.PDDC_EINVAL
    with pytest.raises(pkg.PddcError) as e:          # output too small
        pipe.process_ptr(buf.data_ptr(), 64, out.data_ptr(), 4)
    assert e.value.code == pkg.PDDC_ECAPACITY
    with pytest.raises(pkg.PddcError):
        pipe.set_center_freq(41e6)                   # perseus-sdr.c:575 range
    assert pipe.process_ptr(buf.data_ptr(), 0, out.data_ptr(), 32) == 0
    pipe.close()
    with pytest.raises(pkg.PddcError):
        pkg.Pipeline([(8, h)], device=99)


# ------------------------------------------- full size, size-independent checks
def test_full_size_properties(pkg, dev, O):
    """BASELINE config 2 size (2^26 here to bound test time; bench runs 2^28):
    (1) prefix equals the oracle, (2) linearity: DDC(a)+DDC(b) == DDC(a+b) for
    24-bit inputs whose sum does not overflow, (3) DC gain = sum(h)."""
    t = _torch()
    h = load_taps("d8_127")
    ns = 1 << 26
    d_in = pkg.synth_lcg(6 * ns, 12345, 0, dev)
    pipe = pkg.Pipeline([(8, h)])
    y = pipe.process(d_in)
    assert y.shape[0] == ns // 8
    pre = 8 * 20000
    ref = O.ddc_chain(O.lcg_bytes(6 * pre, 12345), [(8, h)])
    assert O.rel_err(y[: pre // 8].cpu().numpy().reshape(-1), ref) <= FIR_TOL
    # interior seam far from the start: compare a window against the oracle
    s0 = (ns // 2 // 8192) * 8192 - 8 * 300
    win = d_in[6 * (s0 - 8 * 64): 6 * (s0 + 8 * 700)].cpu().numpy()
    refw = O.ddc_chain(win, [(8, h)])[2 * 64:]
    got = y[s0 // 8: s0 // 8 + 700].cpu().numpy().reshape(-1)
    assert O.rel_err(got, refw[: got.size]) <= FIR_TOL
    del y, d_in
    # DC gain
    n2 = 8 * 8192 * 4
    half = np.full(n2, 0x200000, dtype=np.int64)
    packed = O.pack24(half, -half)
    pipe.reset()
    ydc = pipe.process(to_dev(packed, dev)).cpu().numpy()
    g = float(np.sum(h.astype(np.float64)))
    assert abs(ydc[-1, 0] - 0.25 * g * 8388608 / 8388607) < 1e-6
    assert abs(ydc[-1, 1] + 0.25 * g * 8388608 / 8388607) < 1e-6
    # linearity
    rng = np.random.default_rng(1)
    a = rng.integers(-(1 << 22), 1 << 22, size=(2, n2))
    b = rng.integers(-(1 << 22), 1 << 22, size=(2, n2))
    outs = []
    for u in (a, b, a + b):
        pipe.reset()
        outs.append(pipe.process(to_dev(O.pack24(u[0], u[1]), dev)).cpu().numpy().astype(np.float64))
    scale = np.max(np.abs(outs[2]))
    assert np.max(np.abs(outs[0] + outs[1] - outs[2])) / scale <= 3e-6
    pipe.close()


# ----------------------------------- drop-in API, DDC mode (GPU behind callbacks)
def test_perseus_api_ddc_mode_vs_oracle(pkg, dev, O, monkeypatch):
    """perseus_* callback API with the GPU doing the FPGA's job: 80 MS/s LCG
    source -> NCO at 7.1 MHz -> plan for 250 kS/s (8*8*5) -> float32 callbacks."""
    import ctypes as C
    import time
    monkeypatch.setenv("PERSEUS_AMD_PACE", "0")
    monkeypatch.delenv("PERSEUS_AMD_DEVICES", raising=False)
    L = pkg.sdr_lib()
    L.perseus_set_debug(0)
    assert L.perseus_init() == 1
    d = L.perseus_open(0)
    assert L.perseus_firmware_download(d, None) == 0
    assert L.perseus_set_sampling_rate(d, 250000) == 0
    assert L.perseus_set_ddc_center_freq(d, C.c_double(7.1e6), 1) == 0
    cfg = pkg.AmdConfig()
    L.perseus_amd_get_config(d, C.byref(cfg))
    cfg.mode, cfg.pace, cfg.batch_samples, cfg.max_buffers = 1, 0, 8 * 40000, 3
    assert L.perseus_amd_set_config(d, C.byref(cfg)) == 0
    dec, nt = (C.c_int * 4)(), (C.c_int * 4)()
    n = L.perseus_amd_get_plan(d, dec, nt, None)
    taps = [np.zeros(nt[i], np.float32) for i in range(n)]
    arr = (C.POINTER(C.c_float) * 4)(*[t.ctypes.data_as(C.POINTER(C.c_float)) for t in taps], None)
    L.perseus_amd_get_plan(d, dec, nt, arr)
    got = []
    cb = pkg.PERSEUS_CALLBACK(lambda b, nbytes, x: got.append(C.string_at(b, nbytes)) or 0)
    assert L.perseus_start_async_input(d, 6144, cb, None) == 0, L.perseus_errorstr()
    t0 = time.time()
    while L.perseus_amd_source_running(d) and time.time() - t0 < 60:
        time.sleep(0.005)
    assert L.perseus_stop_async_input(d) == 0
    L.perseus_exit()
    assert len(got) == 3
    y = np.frombuffer(b"".join(got), dtype=np.float32)
    nout = y.size // 2                                       # 3 * 768 complex samples
    need = nout * 320
    packed = O.lcg_bytes(6 * need, 12345)
    ref = O.ddc_chain(packed, [(dec[i], taps[i]) for i in range(n)], freg=381178347, mix=True)
    assert O.rel_err(y, ref[: y.size]) <= FIR_TOL


# ------------------------------------------------ N1: float -> 24-bit repack
def test_pack24_roundtrip_exhaustive_and_edges(pkg, dev, O):
    t = _torch()
    v = np.arange(1 << 24, dtype=np.int64)
    packed = O.pack24(v, (~v) & 0xFFFFFF)
    d = to_dev(packed, dev)
    f = pkg.unpack24_f32(d)
    back = pkg.pack24_f32(f).cpu().numpy()
    assert np.array_equal(back, packed)                       # pack(unpack(c)) == c for all 2^24 codes
    rng = np.random.default_rng(9)
    x = (rng.standard_normal(2 * 100003) * 0.7).astype(np.float32)   # odd count: ragged tail, saturation
    x[:8] = [2.0, -2.0, np.nan, 0.5, 1e-9, -1e-9, 1.0, -1.00000012]
    got = pkg.pack24_f32(to_dev(x, dev)).cpu().numpy()
    assert np.array_equal(got, O.pack24_f32(x))


def test_pipeline_packed_output(pkg, dev, O):
    h = load_taps("d8_127")
    ns = 8 * 5000
    packed = O.lcg_bytes(6 * ns, 21)
    pf = pkg.Pipeline([(8, h)])
    pp = pkg.Pipeline([(8, h)], out_packed=True)
    yf = pf.process(to_dev(packed, dev)).cpu().numpy().reshape(-1)
    yp = pp.process(to_dev(packed, dev)).cpu().numpy()
    assert yp.size == 6 * (ns // 8)
    assert np.array_equal(yp, O.pack24_f32(yf))               # same floats, oracle quantiser
    ref = O.ddc_chain(packed, [(8, h)])
    assert O.rel_err(O.unpack24_f32(yp), ref) <= FIR_TOL + 1.0 / 8388607 / np.max(np.abs(ref))
    pf.close()
    pp.close()


def test_perseus_api_fpga_emulation_mode(pkg, dev, O, monkeypatch):
    """DDC_WIRE: an unmodified reference-style client (24-bit packed callbacks,
    client-side unpack) running on the GPU path at 2 MS/s (80 MS/s / 40)."""
    import ctypes as C
    import time
    monkeypatch.setenv("PERSEUS_AMD_PACE", "0")
    monkeypatch.delenv("PERSEUS_AMD_DEVICES", raising=False)
    L = pkg.sdr_lib()
    L.perseus_set_debug(0)
    assert L.perseus_init() == 1
    d = L.perseus_open(0)
    L.perseus_firmware_download(d, None)
    assert L.perseus_set_sampling_rate(d, 2000000) == 0
    assert L.perseus_set_ddc_center_freq(d, C.c_double(7.05e6), 1) == 0
    cfg = pkg.AmdConfig()
    L.perseus_amd_get_config(d, C.byref(cfg))
    cfg.mode, cfg.pace, cfg.batch_samples, cfg.max_buffers = 2, 0, 8 * 20000, 4
    assert L.perseus_amd_set_config(d, C.byref(cfg)) == 0
    dec, nt = (C.c_int * 4)(), (C.c_int * 4)()
    n = L.perseus_amd_get_plan(d, dec, nt, None)
    assert list(dec)[:n] == [10, 4]                          # (2 MS/s = 80 MS/s / 40: decimate by 10 on the matrix cores, then by 4)
    taps = [np.zeros(nt[i], np.float32) for i in range(n)]
    arr = (C.POINTER(C.c_float) * 4)(*[t.ctypes.data_as(C.POINTER(C.c_float)) for t in taps], None, None)
    L.perseus_amd_get_plan(d, dec, nt, arr)
    got = []
    cb = pkg.PERSEUS_CALLBACK(lambda b, nbytes, x: got.append(C.string_at(b, nbytes)) or 0)
    assert L.perseus_start_async_input(d, 6144, cb, None) == 0, L.perseus_errorstr()
    t0 = time.time()
    while L.perseus_amd_source_running(d) and time.time() - t0 < 60:
        time.sleep(0.005)
    assert L.perseus_stop_async_input(d) == 0
    L.perseus_exit()
    assert len(got) == 4 and all(len(g) == 6144 for g in got)
    wire = np.frombuffer(b"".join(got), dtype=np.uint8)
    y = O.unpack24_f32(wire)                                   # what the client's callback would compute
    nout = y.size // 2
    ref = O.ddc_chain(O.lcg_bytes(6 * nout * 40, 12345), [(dec[i], taps[i]) for i in range(n)],
                      freg=O.nco_freg(7.05e6), mix=True)
    assert O.rel_err(y, ref[: y.size]) <= FIR_TOL + 1.0 / 8388607 / np.max(np.abs(ref))


# ------------------------------------- N3: rational resampler / non-integer rates
@pytest.mark.parametrize("L,M,K", [(12, 25, 40), (19, 40, 30), (6, 25, 33), (3, 2, 16)])
def test_rational_resampler_stage_vs_oracle(pkg, dev, O, L, M, K):
    rng = np.random.default_rng(L * 100 + M)
    h1 = load_taps("c320_s1_d8_32")
    g = (rng.standard_normal(K * L) / K).astype(np.float32)
    ns = 8 * 2600
    packed = O.lcg_bytes(6 * ns, 606)
    stages = [(8, h1), (M, g, L)]
    ref = O.ddc_chain(packed, stages)
    pipe = pkg.Pipeline(stages)
    cuts = [0, 8 * 700, 8 * 701, 8 * 1900, ns]                  # uneven pushes: phase carry
    y = np.concatenate([pipe.process(to_dev(packed[6 * a:6 * b], dev)).cpu().numpy().reshape(-1)
                        for a, b in zip(cuts[:-1], cuts[1:])])
    assert y.size == ref.size
    assert O.rel_err(y, ref) <= FIR_TOL
    pipe.close()


@pytest.mark.parametrize("rate", [96000, 95000])
def test_perseus_api_non_integer_rate(pkg, dev, O, monkeypatch, rate):
    import ctypes as C
    import time
    monkeypatch.setenv("PERSEUS_AMD_PACE", "0")
    monkeypatch.delenv("PERSEUS_AMD_DEVICES", raising=False)
    L = pkg.sdr_lib()
    L.perseus_set_debug(0)
    assert L.perseus_init() == 1
    d = L.perseus_open(0)
    L.perseus_firmware_download(d, None)
    assert L.perseus_set_sampling_rate(d, rate) == 0
    assert L.perseus_set_ddc_center_freq(d, C.c_double(7.0e6), 1) == 0
    cfg = pkg.AmdConfig()
    L.perseus_amd_get_config(d, C.byref(cfg))
    cfg.mode, cfg.pace, cfg.batch_samples, cfg.max_buffers = 1, 0, 8 * 100000, 1
    assert L.perseus_amd_set_config(d, C.byref(cfg)) == 0
    dec, nt, it = (C.c_int * 4)(), (C.c_int * 4)(), (C.c_int * 4)()
    n = L.perseus_amd_get_plan(d, dec, nt, None)
    L.perseus_amd_get_plan_interp(d, it)
    taps = [np.zeros(nt[i], np.float32) for i in range(n)]
    arr = (C.POINTER(C.c_float) * 4)(*([t.ctypes.data_as(C.POINTER(C.c_float)) for t in taps] + [None] * (4 - n)))
    L.perseus_amd_get_plan(d, dec, nt, arr)
    got = []
    cb = pkg.PERSEUS_CALLBACK(lambda b, nbytes, x: got.append(C.string_at(b, nbytes)) or 0)
    assert L.perseus_start_async_input(d, 6144, cb, None) == 0, L.perseus_errorstr()
    t0 = time.time()
    while L.perseus_amd_source_running(d) and time.time() - t0 < 60:
        time.sleep(0.005)
    assert L.perseus_stop_async_input(d) == 0
    L.perseus_exit()
    assert len(got) == 1
    y = np.frombuffer(got[0], dtype=np.float32)                 # 768 complex outputs
    need = int(np.ceil(768 * 80e6 / rate)) + 8
    need += (-need) % 8
    stages = [(dec[i], taps[i], it[i]) for i in range(n)]
    ref = O.ddc_chain(O.lcg_bytes(6 * need, 12345), stages, freg=O.nco_freg(7.0e6), mix=True)
    assert O.rel_err(y, ref[: y.size]) <= FIR_TOL


# ----------------------------------------------- persistence / state edge cases
def test_long_tile_runs_per_block_all_variants(pkg, dev, O, monkeypatch):
    """Few persistent blocks => many consecutive tiles per block: exercises the
    LDS-carried history of every kernel variant (packed, packed+NCO, float2
    second stage) and the ragged last tile."""
    monkeypatch.setenv("PDDC_FIR8_BLOCKS", "3")
    ns = 8 * (4096 * 5 + 333)                                  # 40+ tiles over 3 blocks, ragged tail
    packed = O.lcg_bytes(6 * ns, 2024)
    h1, h2, h255 = load_taps("c320_s1_d8_32"), load_taps("c320_s2_d8_64"), load_taps("d8_255")
    for stages, mix in (([(8, load_taps("d8_127"))], False), ([(8, h255)], True), ([(8, h1), (8, h2)], True)):
        for R in ("4", "8"):
            monkeypatch.setenv("PDDC_FIR8_R", R)
            ref = O.ddc_chain(packed, stages, freg=123456789, mix=mix)
            pipe = pkg.Pipeline(stages, mix=mix)
            pipe.set_freg(123456789)
            y = pipe.process(to_dev(packed, dev)).cpu().numpy().reshape(-1)
            assert y.size == ref.size
            assert O.rel_err(y, ref) <= FIR_TOL, (len(stages), mix, R)
            pipe.close()


@pytest.mark.parametrize("dyn,chunk,walk", [("100", "1", 0), ("100", "3", 0), ("60", "2", 0), ("0", "5", 0), ("100", "64", 0),
                                            ("0", "1", 1), ("0", "3", 1), ("40", "2", 1), ("-1", "0", 1), ("0", "64", 1), ("100", "4", 1)])
def test_tile_scheduler_variants(pkg, dev, O, monkeypatch, tune, dyn, chunk, walk):
    """k_fir8 hands part of the tiles out dynamically (atomic chunk counter, the
    history of a chunk's first tile re-read from global memory; the fused pair's
    chunks start with a porch: the second stage's history computed from the 544
    samples in front).  Every schedule must give the same stream: all-dynamic with
    single-tile chunks, odd chunk sizes, mostly static, one chunk larger than the
    batch -- and the round-robin walk (chunk j -> block j mod nblocks, fir8_walk 1)
    with single-tile chunks, odd chunks, a dynamic tail, its default chunk, chunks
    larger than the batch, everything from the counter; the pair also with a 48-tap
    first stage (eight tap blocks: a porch of 72 groups, bursts of 12 tiles);
    several launches in a row check that the counters are left at zero."""
    tune("fir8_dyn_pct", int(dyn))
    tune("fir8_chunk", int(chunk))
    tune("fir8_walk", walk)
    monkeypatch.setenv("PDDC_FIR8_BLOCKS", "5")
    h1, h2 = load_taps("c320_s1_d8_32"), load_taps("c320_s2_d8_64")
    k48 = np.arange(48) - 23.5
    h48 = np.sinc(0.1 * k48) * np.hamming(48)
    h48 = (h48 / h48.sum()).astype(np.float32)
    for stages, mix, R in (([(8, load_taps("d8_127"))], False, "4"), ([(8, load_taps("d8_255"))], True, "8"),
                           ([(8, h1), (8, h2)], True, "4"), ([(8, h1), (8, h2)], False, "8"), ([(8, h48), (8, h2)], True, "4")):
        monkeypatch.setenv("PDDC_FIR8_R", R)
        tile = 1024 * int(R)
        cuts = [0, 23 * tile, 23 * tile + 2 * tile, 60 * tile + (0 if len(stages) == 2 else 8 * 41)]
        packed = O.lcg_bytes(6 * cuts[-1], 99)
        ref = O.ddc_chain(packed, stages, freg=987654321, mix=mix)
        pipe = pkg.Pipeline(stages, mix=mix)
        pipe.set_option("no_i8", 1)                      # (this is about k_fir8's scheduler: the vector kernels throughout)
        pipe.set_freg(987654321)
        y = np.concatenate([pipe.process(to_dev(packed[6 * a:6 * b], dev)).cpu().numpy().reshape(-1)
                            for a, b in zip(cuts[:-1], cuts[1:])])
        assert y.size == ref.size
        assert O.rel_err(y, ref) <= FIR_TOL, (len(stages), mix, R, dyn, chunk)
        pipe.close()


def test_set_taps_reset_and_tiny_batches(pkg, dev, O):
    h = load_taps("d8_127")
    g = (np.arange(100, dtype=np.float32) - 50) / 5000
    ns = 8 * 700
    packed = O.lcg_bytes(6 * ns, 8)
    pipe = pkg.Pipeline([(8, h)])
    # batches of 8 samples (far below the history length) take the generic history path
    y = np.concatenate([pipe.process(to_dev(packed[6 * a: 6 * (a + 8)], dev)).cpu().numpy().reshape(-1)
                        for a in range(0, 8 * 40, 8)] +
                       [pipe.process(to_dev(packed[6 * 8 * 40:], dev)).cpu().numpy().reshape(-1)])
    assert O.rel_err(y, O.ddc_chain(packed, [(8, h)])) <= FIR_TOL
    import ctypes as C
    pkg.check(pkg.ddc_lib().pddc_pipeline_set_taps(pipe._h, 0, g.ctypes.data_as(C.POINTER(C.c_float)), g.size))
    pipe.reset()
    y2 = pipe.process(to_dev(packed, dev)).cpu().numpy().reshape(-1)
    assert O.rel_err(y2, O.ddc_chain(packed, [(8, g)])) <= FIR_TOL
    with pytest.raises(pkg.PddcError):                          # does not fit the geometry fixed at create time
        big = np.ones(400, np.float32)
        pkg.check(pkg.ddc_lib().pddc_pipeline_set_taps(pipe._h, 0, big.ctypes.data_as(C.POINTER(C.c_float)), 400))
    pipe.close()


def test_fp16_tap_storage_rounding(pkg, dev, O):
    h = load_taps("d8_255")
    h16 = h.astype(np.float16).astype(np.float32)
    ns = 8 * 3000
    packed = O.lcg_bytes(6 * ns, 55)
    pipe = pkg.Pipeline([(8, h)], taps_fp16=True)
    y = pipe.process(to_dev(packed, dev)).cpu().numpy().reshape(-1)
    assert O.rel_err(y, O.ddc_chain(packed, [(8, h16)])) <= FIR_TOL      # oracle with the same rounded taps
    assert O.rel_err(y, O.ddc_chain(packed, [(8, h)])) > FIR_TOL          # and visibly not the fp32 taps
    pipe.close()


@pytest.mark.parametrize("R", ["4", "8"])
def test_whole_buffer_fused_vs_generic_path(pkg, dev, O, monkeypatch, R):
    """Every output of a 2^22-sample batch: the fused kernel (asm stores, LDS
    carry, persistent tiles) against the independent generic kernels."""
    t = _torch()
    monkeypatch.setenv("PDDC_FIR8_R", R)
    h = load_taps("d8_127")
    ns = (1 << 22) + 8 * 77
    d_in = pkg.synth_lcg(6 * ns, 4711, 0, dev)
    fast = pkg.Pipeline([(8, h)])
    slow = pkg.Pipeline([(8, h)], no_fast=True)
    a, b = fast.process(d_in), slow.process(d_in)
    assert a.shape == b.shape
    scale = float(b.abs().max())
    assert float((a - b).abs().max()) / scale <= 1e-6
    # and twice in a row on the same state-advancing pipelines (history hand-over between launches)
    a2, b2 = fast.process(d_in), slow.process(d_in)
    assert float((a2 - b2).abs().max()) / scale <= 1e-6
    assert not t.equal(a[:64], a2[:64])              # the second call really saw the first call's history
    fast.close()
    slow.close()


# ------------------------------------------- stages 0+1 fused into one kernel
@pytest.mark.parametrize("R", ["4", "8"])
@pytest.mark.parametrize("mix", [False, True])
def test_fused_stage_pair_vs_oracle(pkg, dev, O, monkeypatch, R, mix):
    """/8 -> /8 (-> /5) with the first intermediate kept in LDS: warm-up tiles at
    block-range starts (few blocks), history hand-over between calls, and the
    switch to and from the unfused path when a batch is not whole tiles."""
    monkeypatch.setenv("PDDC_FIR8_R", R)
    monkeypatch.setenv("PDDC_FIR8_BLOCKS", "3")
    tile = 1024 * int(R)
    h1, h2, h3 = load_taps("c320_s1_d8_32"), load_taps("c320_s2_d8_64"), load_taps("c320_s3_d5_161")
    for stages in ([(8, h1), (8, h2)], [(8, h1), (8, h2), (5, h3)]):
        cuts = [0, 7 * tile, 7 * tile + 8 * 64, 7 * tile + 8 * 64 + 5 * tile, 30 * tile + 8 * 64]
        ns = cuts[-1]
        packed = O.lcg_bytes(6 * ns, 17)
        ref = O.ddc_chain(packed, stages, freg=381178347, mix=mix)
        pipe = pkg.Pipeline(stages, mix=mix)
        pipe.set_freg(381178347)
        assert pipe.fused_pair(7 * tile) and not pipe.fused_pair(8 * 64)
        parts, used = [], []
        for a, b in zip(cuts[:-1], cuts[1:]):
            used.append(bool(pipe.fused_pair(b - a)))
            parts.append(pipe.process(to_dev(packed[6 * a:6 * b], dev)).cpu().numpy().reshape(-1))
        assert used == [True, False, True, True]
        y = np.concatenate(parts)
        assert y.size == ref.size
        assert O.rel_err(y, ref) <= FIR_TOL, (len(stages), R, mix)
        pipe.close()


def test_fused_stage_pair_equals_unfused_whole_buffer(pkg, dev, O, monkeypatch):
    t = _torch()
    h1, h2 = load_taps("c320_s1_d8_32"), load_taps("c320_s2_d8_64")
    ns = 1 << 22
    d_in = pkg.synth_lcg(6 * ns, 5150, 0, dev)
    fused = pkg.Pipeline([(8, h1), (8, h2)], mix=True)
    fused.set_center_freq(7.1e6)
    assert fused.fused_pair(ns)
    a = fused.process(d_in).clone()
    a2 = fused.process(d_in).clone()
    monkeypatch.setenv("PDDC_NO_FUSE2", "1")
    plain = pkg.Pipeline([(8, h1), (8, h2)], mix=True)
    plain.set_center_freq(7.1e6)
    assert not plain.fused_pair(ns)
    b = plain.process(d_in).clone()
    b2 = plain.process(d_in).clone()
    scale = float(b.abs().max())
    assert float((a - b).abs().max()) / scale <= 1e-6
    assert float((a2 - b2).abs().max()) / scale <= 1e-6
    fused.close()
    plain.close()


# ------------------------------------- ONE stream cut into time chunks for several GPUs
@pytest.mark.parametrize("name", ["d8_127", "c320"])
def test_time_chunk_sharding_on_one_gpu(pkg, dev, O, name):
    """four "ranks" (run one after the other here) each take a contiguous
    time chunk of the same stream: seek(chunk start - halo), process halo + chunk, drop the
    halo's outputs.  Stitched, that is the single-stream result -- with the NCO, whose phase
    comes from the absolute sample index alone."""
    shard = importlib.import_module("libperseus-sdr_amd.shard")
    if name == "d8_127":
        stages, mix, unit = [(8, load_taps("d8_127"))], True, 8192
    else:
        stages, mix, unit = [(8, load_taps("c320_s1_d8_32")), (8, load_taps("c320_s2_d8_64")),
                             (5, load_taps("c320_s3_d5_161"))], True, 4096 * 5
    dtot = int(np.prod([d for d, _ in stages]))
    total = unit * 14
    stream = O.lcg_bytes(6 * total, 12345)
    ref = O.ddc_chain(stream, stages, freg=381178347, mix=mix)
    halo = shard.cascade_halo(stages, align=unit)                 # whole tiles: stays on the fused kernels
    parts = []
    for start, length in shard.time_chunks(total, 4, unit):
        h = min(halo, start)
        pipe = pkg.Pipeline(stages, mix=mix)
        pipe.set_freg(381178347)
        y = shard.process_time_chunk(pipe, to_dev(stream[6 * (start - h):6 * (start + length)], dev), start, halo, dtot)
        parts.append(y.cpu().numpy().reshape(-1))
        pipe.close()
    y = np.concatenate(parts)
    assert y.size == ref.size
    assert O.rel_err(y, ref) <= FIR_TOL
    # a position off the output grid is refused
    pipe = pkg.Pipeline(stages, mix=mix)
    with pytest.raises(Exception):
        pipe.seek(dtot + 4)
    pipe.close()


# --------------------------------------------- randomized chunking (state machine)
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_chunking_is_equivalent_to_one_batch(pkg, dev, O, seed):
    """Property: any split of the stream into batches (multiples of 8 samples)
    gives the single-batch result, for integer, fused-pair and rational plans."""
    rng = np.random.default_rng(seed)
    h1, h2, h3 = load_taps("c320_s1_d8_32"), load_taps("c320_s2_d8_64"), load_taps("c320_s3_d5_161")
    g = (rng.standard_normal(12 * 20) / 20).astype(np.float32)
    plans = [[(8, load_taps("d8_127"))], [(8, h1), (8, h2), (5, h3)], [(8, h1), (5, h3[:41]), (25, g, 12)],
             [(10, h3[:77])]]
    ns = 8 * 4096 * 3 + 8 * int(rng.integers(1, 500))
    packed = O.lcg_bytes(6 * ns, 1000 + seed)
    for stages in plans:
        ref = O.ddc_chain(packed, stages, freg=987654321, mix=True)
        cuts = sorted(set([0, ns] + [8 * int(c) for c in rng.integers(1, ns // 8, size=6)] +
                          [4096 * int(c) for c in rng.integers(1, ns // 4096, size=3)]))
        pipe = pkg.Pipeline(stages, mix=True)
        pipe.set_freg(987654321)
        y = np.concatenate([pipe.process(to_dev(packed[6 * a:6 * b], dev)).cpu().numpy().reshape(-1)
                            for a, b in zip(cuts[:-1], cuts[1:])])
        assert y.size == ref.size, (seed, len(stages))
        assert O.rel_err(y, ref) <= FIR_TOL, (seed, len(stages), cuts)
        pipe.close()


def test_perseus_api_ddc_mode_uses_fused_pair_with_default_style_batches(pkg, dev, O, monkeypatch):
    """125 kS/s plan (8*8*10) with power-of-two batches: stages 0+1 run fused."""
    import ctypes as C
    import time
    monkeypatch.setenv("PERSEUS_AMD_PACE", "0")
    monkeypatch.delenv("PERSEUS_AMD_DEVICES", raising=False)
    L = pkg.sdr_lib()
    L.perseus_set_debug(0)
    assert L.perseus_init() == 1
    d = L.perseus_open(0)
    L.perseus_firmware_download(d, None)
    assert L.perseus_set_sampling_rate(d, 125000) == 0
    assert L.perseus_set_ddc_center_freq(d, C.c_double(14.2e6), 1) == 0
    cfg = pkg.AmdConfig()
    L.perseus_amd_get_config(d, C.byref(cfg))
    cfg.mode, cfg.pace, cfg.batch_samples, cfg.max_buffers = 1, 0, 1 << 18, 2
    assert L.perseus_amd_set_config(d, C.byref(cfg)) == 0
    dec, nt = (C.c_int * 4)(), (C.c_int * 4)()
    n = L.perseus_amd_get_plan(d, dec, nt, None)
    assert list(dec)[:n] == [8, 8, 10] and nt[1] <= 64
    taps = [np.zeros(nt[i], np.float32) for i in range(n)]
    arr = (C.POINTER(C.c_float) * 4)(*([t.ctypes.data_as(C.POINTER(C.c_float)) for t in taps] + [None]))
    L.perseus_amd_get_plan(d, dec, nt, arr)
    got = []
    cb = pkg.PERSEUS_CALLBACK(lambda b, nbytes, x: got.append(C.string_at(b, nbytes)) or 0)
    assert L.perseus_start_async_input(d, 6144, cb, None) == 0, L.perseus_errorstr()
    t0 = time.time()
    while L.perseus_amd_source_running(d) and time.time() - t0 < 60:
        time.sleep(0.005)
    assert L.perseus_stop_async_input(d) == 0
    L.perseus_exit()
    y = np.frombuffer(b"".join(got), dtype=np.float32)
    assert y.size == 2 * 2 * 768
    need = (y.size // 2) * 640
    ref = O.ddc_chain(O.lcg_bytes(6 * need, 12345), [(dec[i], taps[i]) for i in range(n)],
                      freg=O.nco_freg(14.2e6), mix=True)
    assert O.rel_err(y, ref[: y.size]) <= FIR_TOL


# ------------------------------------------------------------ spectral sanity
def test_spectral_sanity_two_tone(pkg, dev, O):
    """a pass-band tone comes through at unity gain,
    a stop-band tone that would alias into the pass-band is rejected by the
    127-tap filter's ~88 dB (24-bit quantisation noise floor permitting)."""
    h = load_taps("d8_127")
    n = 8 * 16384
    t = np.arange(n, dtype=np.float64)
    fs = 80e6
    # both tones sit on FFT bin centres of the 8192-point output spectrum (no scalloping loss)
    f_pass = 819 * 10e6 / 8192
    f_alias = -2212 * 10e6 / 8192            # where the stop-band tone lands after folding
    f_stop = 10e6 + f_alias                  # ~7.3 MHz
    z = 0.4 * np.exp(2j * np.pi * f_pass / fs * t) + 0.4 * np.exp(2j * np.pi * f_stop / fs * t)
    packed = O.pack24(np.round(z.real * 8388607).astype(np.int64), np.round(z.imag * 8388607).astype(np.int64))
    pipe = pkg.Pipeline([(8, h)])
    y = pipe.process(to_dev(packed, dev)).cpu().numpy().astype(np.float64)
    pipe.close()
    w = y[256:, 0] + 1j * y[256:, 1]
    w = w[: 8192] * np.hanning(8192)
    S = np.abs(np.fft.fft(w)) / (0.5 * 8192)
    fo = np.fft.fftfreq(8192, 1 / 10e6)
    peak = lambda f: S[np.argmin(np.abs(fo - f))]
    assert abs(20 * np.log10(peak(f_pass) / 0.4)) < 0.05                 # unity pass-band gain
    assert 20 * np.log10(peak(f_alias) / 0.4) < -80.0                    # alias of the stop-band tone


# --------------------- the unmodified reference client on the GPU path (full circle)
def test_reference_client_on_gpu_path_fpga_emulation(pkg, dev, O, tmp_path, monkeypatch):
    """oracle/_ref/perseustest_ref (the reference's examples/perseustest.c, compiled
    unmodified) with PERSEUS_AMD_MODE=ddc-wire: the GPU does NCO + /320, re-quantises
    to the 24-bit wire format, and the REFERENCE's own callback unpacks it."""
    import ctypes as C
    import subprocess
    ref_bin = os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref", "perseustest_ref")
    if not os.path.exists(ref_bin):
        pytest.skip("oracle/_ref not built")
    out = tmp_path / "gpu_ref.bin"
    env = dict(os.environ, PERSEUS_AMD_PACE="0", PERSEUS_AMD_MODE="ddc-wire", PERSEUS_AMD_SOURCE="lcg:12345",
               PERSEUS_AMD_MAX_BUFFERS="3", PERSEUS_AMD_BATCH=str(1 << 18))
    env.pop("PERSEUS_AMD_DEVICES", None)
    p = subprocess.run([ref_bin, "-a", "-t", "2", "-p", "-s", "250000", "-f", "7100000", "-d", "3", "-o", str(out)],
                       env=env, capture_output=True, text=True, timeout=120)
    assert "Elapsed time:" in p.stderr, p.stderr[-500:]
    y = np.fromfile(out, dtype=np.float32)
    assert y.size == 3 * 2048                                  # 3 transfers of 1024 samples
    # the plan the library used for 250 kS/s (deterministic), through the in-process API
    monkeypatch.setenv("PERSEUS_AMD_PACE", "0")
    L = pkg.sdr_lib()
    L.perseus_set_debug(0)
    assert L.perseus_init() == 1
    d = L.perseus_open(0)
    L.perseus_firmware_download(d, None)
    L.perseus_set_sampling_rate(d, 250000)
    dec, nt = (C.c_int * 4)(), (C.c_int * 4)()
    n = L.perseus_amd_get_plan(d, dec, nt, None)
    taps = [np.zeros(nt[i], np.float32) for i in range(n)]
    arr = (C.POINTER(C.c_float) * 4)(*([t.ctypes.data_as(C.POINTER(C.c_float)) for t in taps] + [None]))
    L.perseus_amd_get_plan(d, dec, nt, arr)
    L.perseus_exit()
    ref = O.ddc_chain(O.lcg_bytes(6 * (y.size // 2) * 320, 12345), [(dec[i], taps[i]) for i in range(n)],
                      freg=O.nco_freg(7.1e6), mix=True)
    # float path tolerance + one 24-bit quantisation step of the wire format
    assert O.rel_err(y, ref[: y.size]) <= FIR_TOL + 1.0 / 8388607 / np.max(np.abs(ref))


# ------------------------------ any first decimation reads the packed samples itself (no float2 intermediate)
@pytest.mark.parametrize("D,nt,mix", [(10, 69, True), (10, 69, False), (5, 161, True), (3, 17, True), (1, 9, True),
                                      (40, 401, False), (7, 250, True)])
def test_packed_generic_first_stage(pkg, dev, O, D, nt, mix):
    """Stage 0 of a plan whose first decimation is not 8 (the reference's 1.6 MS/s rate is 80 MS/s / (10*5)):
    the generic decimator unpacks and mixes while it stages its input span; tiny and uneven batches carry the
    packed history and the decimation phase."""
    rng = np.random.default_rng(D * 1000 + nt)
    h = (rng.standard_normal(nt) / np.sqrt(nt)).astype(np.float32)
    ns = 8 * 2600
    packed = O.lcg_bytes(6 * ns, 4242)
    freg = 0x9E3779B1
    ref = O.ddc_chain(packed, [(D, h)], freg=freg, mix=mix)
    pipe = pkg.Pipeline([(D, h)], mix=mix)
    assert pipe.stage0_reads_packed and not pipe.fused
    if mix:
        pipe.set_freg(freg)
    cuts = [0, 8, 16, 8 * 30, 8 * 31, 8 * 700, 8 * 1999, ns]          # 8-sample batches: no output, history only
    y = np.concatenate([pipe.process(to_dev(packed[6 * a:6 * b], dev)).cpu().numpy().reshape(-1)
                        for a, b in zip(cuts[:-1], cuts[1:])])
    assert y.size == ref.size
    assert O.rel_err(y, ref) <= FIR_TOL
    slow = pkg.Pipeline([(D, h)], mix=mix, no_fast=True)              # unpack kernel -> float2 -> generic
    assert not slow.stage0_reads_packed
    if mix:
        slow.set_freg(freg)
    ys = slow.process(to_dev(packed, dev)).cpu().numpy().reshape(-1)
    assert O.rel_err(ys, ref) <= FIR_TOL and O.rel_err(y, ys) <= FIR_TOL
    pipe.close()
    slow.close()


def test_perseus_api_1600k_plan_is_fused_end_to_end(pkg, dev, O, monkeypatch):
    """The 1.6 MS/s rate (10*5): first stage /10 through the packed-input generic kernel, vs the oracle."""
    import ctypes as C
    import time
    monkeypatch.setenv("PERSEUS_AMD_PACE", "0")
    monkeypatch.delenv("PERSEUS_AMD_DEVICES", raising=False)
    L = pkg.sdr_lib()
    L.perseus_set_debug(0)
    assert L.perseus_init() == 1
    d = L.perseus_open(0)
    L.perseus_firmware_download(d, None)
    assert L.perseus_set_sampling_rate(d, 1600000) == 0
    assert L.perseus_set_ddc_center_freq(d, C.c_double(21.3e6), 1) == 0
    cfg = pkg.AmdConfig()
    L.perseus_amd_get_config(d, C.byref(cfg))
    cfg.mode, cfg.pace, cfg.batch_samples, cfg.max_buffers = 1, 0, 8 * 25000, 12
    assert L.perseus_amd_set_config(d, C.byref(cfg)) == 0
    dec, nt = (C.c_int * 4)(), (C.c_int * 4)()
    n = L.perseus_amd_get_plan(d, dec, nt, None)
    assert list(dec)[:n] == [10, 5]
    taps = [np.zeros(nt[i], np.float32) for i in range(n)]
    arr = (C.POINTER(C.c_float) * 4)(*[t.ctypes.data_as(C.POINTER(C.c_float)) for t in taps], None, None)
    L.perseus_amd_get_plan(d, dec, nt, arr)
    got = []
    cb = pkg.PERSEUS_CALLBACK(lambda b, nbytes, x: got.append(C.string_at(b, nbytes)) or 0)
    assert L.perseus_start_async_input(d, 6144, cb, None) == 0, L.perseus_errorstr()
    t0 = time.time()
    while L.perseus_amd_source_running(d) and time.time() - t0 < 60:
        time.sleep(0.005)
    assert L.perseus_stop_async_input(d) == 0
    L.perseus_exit()
    y = np.frombuffer(b"".join(got), dtype=np.float32)
    assert y.size == 12 * 1536
    need = (y.size // 2) * 50
    ref = O.ddc_chain(O.lcg_bytes(6 * need, 12345), [(dec[i], taps[i]) for i in range(n)], freg=O.nco_freg(21.3e6),
                      mix=True)
    assert O.rel_err(y, ref[:y.size]) <= FIR_TOL


# ------------------------------------------------------------ failure atomicity
def test_a_failed_batch_leaves_the_stream_where_it_was(pkg, dev, O):
    """process() commits its state only after every launch was accepted: a failure half way (injected at
    each stage in turn) changes nothing, the same batch is simply pushed again, and the stream goes on
    bit-identically to one that never failed."""
    import ctypes as C
    stages = [(8, load_taps("c320_s1_d8_32")), (5, load_taps("c320_s3_d5_161")[:41]), (4, load_taps("c320_s1_d8_32")),
              (25, (np.random.default_rng(3).standard_normal(120) / 10).astype(np.float32), 12)]
    ns = 8 * 4000
    batches = [to_dev(O.lcg_bytes(6 * ns, 100 + k), dev) for k in range(6)]
    good = pkg.Pipeline(stages, mix=True)
    good.set_freg(123456789)
    want = [good.process(b).cpu().numpy() for b in batches]
    flaky = pkg.Pipeline(stages, mix=True)
    flaky.set_freg(123456789)
    L = pkg.ddc_lib()
    for k, b in enumerate(batches):
        if k >= 1:
            stage = (k - 1) % len(stages)
            assert L.pddc_pipeline_inject_failure(flaky._h, stage) == 0
            with pytest.raises(pkg.PddcError) as e:
                flaky.process(b)
            assert e.value.code == pkg.PDDC_EHIP and b"injected failure" in L.pddc_last_error()
        got = flaky.process(b).cpu().numpy()                      # the retry
        assert np.array_equal(got, want[k]), k
    good.close()
    flaky.close()


def test_too_small_output_buffer_is_refused_before_anything_runs(pkg, dev, O):
    import ctypes as C
    h = load_taps("d8_127")
    ns = 8 * 3000
    a, b = O.lcg_bytes(6 * ns, 1), O.lcg_bytes(6 * ns, 2)
    pipe = pkg.Pipeline([(8, h)])
    ya = pipe.push_host(a)
    out = np.empty((ns // 8, 2), np.float32)
    n = C.c_size_t(0)
    L = pkg.ddc_lib()
    rc = L.pddc_pipeline_push_host(pipe._h, b.ctypes.data, ns, out.ctypes.data, ns // 8 - 1, C.byref(n))
    assert rc == pkg.PDDC_ECAPACITY and n.value == 0
    yb = pipe.push_host(b)                                          # same batch again, with room
    ref = O.ddc_chain(np.concatenate([a, b]), [(8, h)])
    assert O.rel_err(np.concatenate([ya, yb]).reshape(-1), ref) <= FIR_TOL
    assert pipe.next_output(ns) == ns // 8
    pipe.close()


# ------------------------------------------------------------ checkpoint / resume
@pytest.mark.parametrize("plan", ["d8_255", "c320", "rational", "generic10"])
def test_checkpoint_and_resume_continue_bit_identically(pkg, dev, O, plan):
    """The stream state -- FIR histories, decimation phases, sample counter, NCO word and phase offset -- saved
    after some batches and restored into a fresh pipeline of the same plan: the continuation is bit-identical
    to the stream that was never interrupted (also right after a retune, and with uneven batches)."""
    g = (np.random.default_rng(4).standard_normal(120) / 10).astype(np.float32)
    plans = {"d8_255": [(8, load_taps("d8_255"))],
             "c320": [(8, load_taps("c320_s1_d8_32")), (8, load_taps("c320_s2_d8_64")), (5, load_taps("c320_s3_d5_161"))],
             "rational": [(8, load_taps("c320_s1_d8_32")), (5, load_taps("c320_s3_d5_161")[:41]), (25, g, 12)],
             "generic10": [(10, load_taps("c320_s3_d5_161")[:77]), (5, load_taps("c320_s3_d5_161"))]}
    stages = plans[plan]
    sizes = [8192 * 3, 8 * 777, 8192, 8 * 12, 8192 * 2 + 8 * 5, 4096]
    packed = [to_dev(O.lcg_bytes(6 * n, 700 + k), dev) for k, n in enumerate(sizes)]
    words = [381178347, 381178347, 99999999, 99999999, 4000000000, 4000000000]
    a = pkg.Pipeline(stages, mix=True)
    blobs, outs = [], []
    for b, w in zip(packed, words):
        a.set_freg(w)
        blobs.append(a.save_state())
        outs.append(a.process(b).cpu().numpy())
    for cut in (1, 2, 3, 5):                                  # resume from the state saved BEFORE batch `cut`
        b2 = pkg.Pipeline(stages, mix=True)
        b2.restore_state(blobs[cut])
        for k in range(cut, len(sizes)):
            b2.set_freg(words[k])
            assert np.array_equal(b2.process(packed[k]).cpu().numpy(), outs[k]), (plan, cut, k)
        b2.close()
    other = pkg.Pipeline([(8, load_taps("d8_127"))], mix=True)
    with pytest.raises(pkg.PddcError) as e:
        other.restore_state(blobs[2])                         # another plan: refused
    assert e.value.code == pkg.PDDC_ESTATE
    with pytest.raises(pkg.PddcError):
        a.restore_state(blobs[2][:40])
    other.close()
    a.close()


def test_malloc_apart_returns_a_usable_buffer(pkg, dev, O):
    """pddc_malloc_apart (placement of a stream's output away from its input, include/perseus_ddc.h): the pointer
    works like any other allocation; small requests are plain allocations; the probe times are reported."""
    import ctypes as C
    L = pkg.ddc_lib()
    ns = 1 << 26                                                # 64 MiB of output: large enough to be probed
    packed = O.lcg_bytes(6 * ns, 5)
    d_in = to_dev(packed, dev)
    h = load_taps("d8_127")
    pipe = pkg.Pipeline([(8, h)])
    cap = pipe.max_output(ns) + 8
    p, fast, slow = C.c_void_p(), C.c_float(-1), C.c_float(-1)
    assert L.pddc_malloc_apart(C.byref(p), cap * 8, d_in.data_ptr(), 6 * ns, 3, C.byref(fast), C.byref(slow)) == 0
    assert p.value and fast.value > 0 and slow.value >= fast.value
    st = _torch().cuda.current_stream(dev).cuda_stream
    n = pipe.process_ptr(d_in.data_ptr(), ns, p.value, cap, st)
    y = np.empty((n, 2), np.float32)
    pkg.check(L.pddc_memcpy_d2h(y.ctypes.data, p.value, n * 8, st))
    pkg.check(L.pddc_stream_sync(st))
    ref = O.ddc_chain(packed[:6 * 8192 * 4], [(8, h)])
    assert O.rel_err(y.reshape(-1)[:ref.size], ref) <= FIR_TOL
    assert L.pddc_free(p) == 0
    q = C.c_void_p()
    assert L.pddc_malloc_apart(C.byref(q), 4096, d_in.data_ptr(), 6 * ns, 8, C.byref(fast), C.byref(slow)) == 0
    assert q.value and fast.value == 0.0                       # too small to matter: no probing
    assert L.pddc_free(q) == 0
    assert L.pddc_malloc_apart(None, 4096, None, 0, 1, None, None) == pkg.PDDC_EINVAL
    pipe.close()


@pytest.mark.gpu
def test_workspace_from_the_caller(pkg, dev, O):
    """pddc_pipeline_set_workspace: the inter-stage buffers of a cascade live in memory the HOST provides (so the host
    decides where in HBM they lie).  Same outputs bit for bit as with the pipeline's own buffers, in one shot and in
    ragged batches, unfused and fused routes; a batch beyond what the workspace was sized for is refused; NULL goes
    back to own allocations without touching the stream state."""
    torch = _torch()
    meta = json.load(open(os.path.join(GOLD, "ddc_golden.json")))
    stages = [(d, load_taps(n)) for d, n in meta["c320_stages"]]
    ns = 8 * 8192 * 5 + 8 * 123                                     # fused-pair tiles plus a ragged rest
    packed = O.lcg_bytes(6 * ns, 77)
    d_in = to_dev(packed, dev)
    own = pkg.Pipeline(stages, mix=True)
    own.set_freg(meta["freg"])
    ref = own.process(d_in).cpu().numpy()
    p = pkg.Pipeline(stages, mix=True)
    p.set_freg(meta["freg"])
    need = p.workspace_size(ns)
    assert need >= (ns // 8 + ns // 64) * 8 and need % 256 == 0
    ws = torch.empty(need + 256, dtype=torch.uint8, device=dev)
    base = (ws.data_ptr() + 255) & ~255
    with pytest.raises(pkg.PddcError) as e:
        p.set_workspace(base, need - 256, ns)                       # too small
    assert e.value.code == pkg.PDDC_EINVAL
    with pytest.raises(pkg.PddcError):
        p.set_workspace(base + 8, need, ns)                         # misaligned
    p.set_workspace(base, need, ns)
    y = p.process(d_in).cpu().numpy()
    assert np.array_equal(y, ref)
    assert O.rel_err(y.reshape(-1)[:2 * 64], O.ddc_chain(packed[:6 * 320 * 200], stages, freg=meta["freg"], mix=True)[:2 * 64]) <= FIR_TOL
    # ragged batches through the same workspace, stream state carried as usual
    p.reset()
    cuts = [0, 8 * 8192 * 2, 8 * 8192 * 2 + 8 * 31, ns]
    own.reset()
    parts = [p.process(d_in[6 * a:6 * b]).cpu().numpy() for a, b in zip(cuts[:-1], cuts[1:])]
    own_parts = [own.process(d_in[6 * a:6 * b]).cpu().numpy() for a, b in zip(cuts[:-1], cuts[1:])]
    assert np.array_equal(np.concatenate(parts), np.concatenate(own_parts))     # the same routes: the same bits
    assert O.rel_err(np.concatenate(parts).reshape(-1), ref.reshape(-1)) <= FIR_TOL
    # a larger batch than the workspace was sized for: refused, state untouched
    big = to_dev(O.lcg_bytes(6 * (ns + 8 * 8192 * 8), 78), dev)
    with pytest.raises(pkg.PddcError) as e:
        p.process(big)
    assert e.value.code == pkg.PDDC_ECAPACITY
    # back to own allocations in the middle of a stream: the state is kept
    p.reset()
    own.reset()
    a1 = p.process(d_in[:6 * cuts[1]]).cpu().numpy()
    p.set_workspace(None)
    a2 = p.process(d_in[6 * cuts[1]:]).cpu().numpy()
    b1 = own.process(d_in[:6 * cuts[1]]).cpu().numpy()
    b2 = own.process(d_in[6 * cuts[1]:]).cpu().numpy()
    assert np.array_equal(np.concatenate([a1, a2]), np.concatenate([b1, b2]))
    p.process(big)                                                  # and large batches are fine again
    del ws
    p.close()
    own.close()
    # a single-stage pipeline has no inter-stage buffer: size 0, and setting an (empty) workspace is a no-op
    one = pkg.Pipeline([(8, load_taps("d8_127"))])
    assert one.workspace_size(1 << 20) == 0
    one.set_workspace(None)
    y1 = one.process(d_in[:6 * 8192]).cpu().numpy()
    assert O.rel_err(y1.reshape(-1), O.ddc_chain(packed[:6 * 8192], [(8, load_taps("d8_127"))])) <= FIR_TOL
    one.close()
