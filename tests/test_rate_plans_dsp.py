"""CPU test of the DSP quality of the drop-in API's ten rate plans.  The reference
has no tap values (they live in FPGA bitstreams), so the plans are authored here (perseus_api.c plan_build:
Kaiser designs).  Parity tests only show the GPU computes what the plan says; this one shows the plans are
receivers: through the CPU oracle, a tone inside the wanted band comes out at unit gain and alone (no images
of the rational resampler, no spurs), and tones that sit where each stage would fold them onto the wanted band
are rejected."""
import ctypes as C

import numpy as np
import pytest


def _plans(pkg):
    L = pkg.sdr_lib()
    L.perseus_set_debug(0)
    assert L.perseus_init() >= 1
    d = L.perseus_open(0)
    L.perseus_firmware_download(d, None)
    out = {}
    rates = (C.c_int * 12)()
    L.perseus_get_sampling_rates(None, rates, 12)
    for rate in [r for r in rates if r]:
        assert L.perseus_set_sampling_rate(d, rate) == 0
        dec, nt, it = (C.c_int * 4)(), (C.c_int * 4)(), (C.c_int * 4)()
        n = L.perseus_amd_get_plan(d, dec, nt, None)
        taps = [np.zeros(nt[i], np.float32) for i in range(n)]
        arr = (C.POINTER(C.c_float) * 4)(*([t.ctypes.data_as(C.POINTER(C.c_float)) for t in taps] + [None] * (4 - n)))
        L.perseus_amd_get_plan(d, dec, nt, arr)
        L.perseus_amd_get_plan_interp(d, it)
        out[rate] = [(dec[i], taps[i], it[i]) for i in range(n)]
    L.perseus_exit()
    return out


def _run(O, stages, tones_hz, n_in, amp=0.2):
    n = np.arange(n_in)
    x = sum(amp * np.exp(2j * np.pi * (f / 80e6) * n) for f in tones_hz)
    packed = O.pack24(np.rint(x.real * 8388607).astype(np.int64), np.rint(x.imag * 8388607).astype(np.int64))
    y = O.ddc_chain(packed, stages)                                   # no NCO: frequencies relative to the LO
    return y[0::2] + 1j * y[1::2]


NFFT = 1024


def _spectrum(z, skip):
    z = z[skip:skip + NFFT]
    assert z.size == NFFT
    w = np.blackman(z.size)
    return np.abs(np.fft.fft(z * w)) / np.sum(w)


@pytest.mark.parametrize("rate", [48000, 95000, 96000, 125000, 192000, 250000, 500000, 1000000, 1600000, 2000000])
def test_rate_plan_is_a_clean_receiver(pkg, O, rate):
    stages = _plans(pkg)[rate]
    n_out = 2048
    tot = 80e6 / rate
    n_in = int((n_out + 64) * tot) // 8 * 8
    skip = 512                                                          # filter transients
    amp = 0.2
    # 1. a tone inside the wanted band: unit gain, and nothing else in the output above -75 dBc
    f0 = round(0.23 * NFFT) / NFFT * rate                               # on an FFT bin: no scalloping loss
    z = _run(O, stages, [f0], n_in, amp)
    sp = _spectrum(z, skip)
    nfft = sp.size
    k0 = int(round(f0 / rate * nfft)) % nfft
    peak = sp[max(k0 - 3, 0):k0 + 4].max()
    assert abs(20 * np.log10(peak / amp)) < 0.15, (rate, 20 * np.log10(peak / amp))
    mask = np.ones(nfft, bool)
    for k in range(-8, 9):
        mask[(k0 + k) % nfft] = False                                   # the tone and its window skirt
    spur = sp[mask].max()
    assert 20 * np.log10(spur / peak) < -75.0, (rate, 20 * np.log10(spur / peak))
    # 2. tones that each stage would fold onto the wanted band: at every intermediate rate fs_i, the
    #    frequencies fs_i -/+ 0.2*rate alias to -/+0.2*rate there unless the filters in front removed them
    fs, bad = 80e6, []
    for D, _, Lr in stages[:-1] if len(stages) > 1 else stages:
        fs = fs * max(Lr, 1) / D
        bad += [fs - 0.2 * rate, -(fs + 0.3 * rate)]
    bad = [f for f in bad if abs(f) < 40e6]
    a_each = 0.9 / len(bad)                                             # the sum stays inside the 24-bit range
    z = _run(O, stages, bad, n_in, a_each)
    leak = _spectrum(z, skip).max()
    assert 20 * np.log10(leak / a_each) < -75.0, (rate, bad, 20 * np.log10(leak / a_each))
