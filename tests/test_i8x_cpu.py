"""CPU tests of k_fir_i8x's arithmetic (ddc_fir_i8.hip; operands from pddc_fir_i8x_tables / pddc_fir_i8x_taps2, host
arithmetic, no GPU).  The kernel folds the NCO into the taps,
    y[m] = LO(n0 + 8 m) * sum_k (h[k] e^{+j theta k}) x_raw[8 m - k],      theta = 2 pi freg / 2^32,
so that the int8 matrix cores work on the wire bytes themselves and the phase is applied once per output.  Everything it
does is integer and exact up to the recombination, so it can be restated in numpy from the tables alone: byte planes,
digit planes of the cosine and sine tap sets, the band products, their combination in the kernel's two forms (mode 1,
129..256 taps: four partial products meet as floats; mode 2, up to 128 taps: ONE operand holds both tap sets -- rows 0..7
the band of eight outputs for one set, rows 8..15 for the other -- and [c ; s] on the I planes plus [-s ; c] on the Q planes
go into one set of integer accumulators: rows 0..7 come out as uI, rows 8..15 as uQ), the rotation with
the exact 32-bit phase, and the fused second stage with its complex taps.  Checked against the oracle's mix-then-filter
definition (perseus-sdr.c:584 for the tuning word)."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_taps

FREG = 381178347              # 7.1 MHz (perseus-sdr.c:584), BASELINE config 3


def lowpass(ntaps, cutoff):
    k = np.arange(ntaps) - (ntaps - 1) / 2.0
    h = np.sinc(2 * cutoff * k) * np.hamming(ntaps)
    return (h / h.sum()).astype(np.float32)


def form(hist, mix):
    return 0 if not mix else 3 if hist <= 128 else 1


def tables(pkg, h, hist, mix, freg):
    L = pkg.ddc_lib()
    ks = ((56 if form(hist, mix) == 3 else 120) + hist + 63) // 64
    ntab = 2 if mix else 1
    tab = np.zeros(ntab * 4 * ks * 64 * 16, np.int8)
    sc, ct = C.c_float(), (C.c_float * 2)()
    h = np.ascontiguousarray(h, np.float32)
    n = pkg.check(L.pddc_fir_i8x_tables(h.ctypes.data_as(C.POINTER(C.c_float)), h.size, hist, int(mix), freg,
                                        tab.ctypes.data, tab.nbytes, C.byref(sc), ct))
    assert n == ntab
    return tab.reshape(ntab, 4, ks, 64, 16), sc.value, (ct[0], ct[1])


def band(tab):
    """one table -> the integer Toeplitz band T[16][64 ksteps] (digits recombined)"""
    ks = tab.shape[1]
    T = np.zeros((4, 16, 64 * ks), np.int64)
    for k in range(ks):
        for lane in range(64):
            T[:, lane & 15, 64 * k + 16 * (lane >> 4):64 * k + 16 * (lane >> 4) + 16] = tab[:, k, lane, :]
    return T


def byte_planes(packed, hist, K):
    b = packed.reshape(-1, 2, 3).astype(np.int64)
    pl = np.stack([b[:, :, 0] - 128, b[:, :, 1] - 128, np.where(b[:, :, 2] >= 128, b[:, :, 2] - 256, b[:, :, 2])])
    zero = np.array([-128, -128, 0], np.int64)[:, None, None]            # the planes of a zero sample
    return np.concatenate([np.broadcast_to(zero, (3, hist, 2)), pl, np.broadcast_to(zero, (3, K + 128, 2))], axis=1)


def plane_products(T, X):
    """the nine plane products of one band with one component's planes X[i][c] -> the float the kernel recombines"""
    acc = np.zeros((4, 16), np.int64)
    for i in range(3):
        for j in range(4):
            if i + j >= 2:
                acc[i + j - 2] += T[j] @ X[i]
    return acc


def recombine(acc):
    a = acc.astype(np.float32)
    return ((a[0] * np.float32(65536.0) + a[1] * np.float32(16777216.0)) +
            (a[2] * np.float32(4294967296.0) + a[3] * np.float32(1099511627776.0)))


def lo(freg, n, off=0):
    ph = (np.asarray(n, np.uint64) * np.uint64(freg) + np.uint64(off)) & np.uint64(0xFFFFFFFF)
    th = 2.0 * np.pi * ph.astype(np.float64) / 4294967296.0
    return np.cos(th).astype(np.float32), (-np.sin(th)).astype(np.float32)      # nco_lo's c, s: LO = c + j s


def restated_first_stage(tabs, scale, ct, hist, packed, mix):
    """u[m] (before the rotation) of a batch that starts a stream, as the kernel forms it"""
    Ts = [band(t) for t in tabs]
    K = Ts[0].shape[2]
    xp = byte_planes(packed, hist, K)
    n_out = packed.size // 48
    u = np.zeros((n_out, 2), np.float32)
    scale = np.float32(scale)
    if mix and hist <= 128:                                      # mode 2: columns of eight outputs, both rails from one pass
        for col in range((n_out + 7) // 8):
            X = xp[:, 64 * col:64 * col + K, :]
            acc = plane_products(Ts[0], X[:, :, 0]) + plane_products(Ts[1], X[:, :, 1])
            assert np.abs(acc).max() < 1 << 24
            y = recombine(acc) * scale
            m = min(8, n_out - 8 * col)
            u[8 * col:8 * col + m, 0] = (y[:8] + np.float32(ct[0]))[:m]
            u[8 * col:8 * col + m, 1] = (y[8:] + np.float32(ct[1]))[:m]
        return u
    for col in range((n_out + 15) // 16):
        X = xp[:, 128 * col:128 * col + K, :]
        XI, XQ = X[:, :, 0], X[:, :, 1]
        if not mix:
            y = [recombine(plane_products(Ts[0], Xc)) * scale + np.float32(ct[c]) for c, Xc in enumerate((XI, XQ))]
        else:                                                    # mode 1: four float partial products
            P = [[recombine(plane_products(Ts[t], Xc)) * scale for Xc in (XI, XQ)] for t in range(2)]
            y = [(P[0][0] - P[1][1]) + np.float32(ct[0]), (P[1][0] + P[0][1]) + np.float32(ct[1])]
        m = min(16, n_out - 16 * col)
        u[16 * col:16 * col + m, 0] = y[0][:m]
        u[16 * col:16 * col + m, 1] = y[1][:m]
    return u


def rotate(u, freg, step, off=0, n0=0):
    c, s = lo(freg, n0 + step * np.arange(u.shape[0], dtype=np.uint64), off)
    return np.stack([u[:, 0] * c - u[:, 1] * s, u[:, 0] * s + u[:, 1] * c], axis=1).astype(np.float32)


@pytest.mark.parametrize("ntaps,hist", [(127, 128), (255, 256), (100, 128), (48, 64), (64, 64), (32, 32), (17, 32)])
def test_restated_nco_first_stage_matches_the_oracle(pkg, O, ntaps, hist):
    h = load_taps("d8_127") if ntaps == 127 else load_taps("d8_255") if ntaps == 255 else lowpass(ntaps, 0.05)
    packed = O.lcg_bytes(6 * 8 * 600, 9)
    ref = O.ddc_chain(packed, [(8, h)], freg=FREG, mix=True)
    tabs, scale, ct = tables(pkg, h, hist, True, FREG)
    assert tabs.shape[0] == 2
    y = rotate(restated_first_stage(tabs, scale, ct, hist, packed, True), FREG, 8).reshape(-1)
    assert y.size == ref.size and O.rel_err(y, ref) <= 3e-7, O.rel_err(y, ref)


def test_tables_hold_the_rotated_taps_and_their_negative(pkg):
    h = lowpass(56, 0.05)
    tabs, scale, ct = tables(pkg, h, 64, True, FREG)
    E = 30 - int(np.ceil(np.log2(np.abs(h).max())))
    th = 2 * np.pi * ((np.arange(56, dtype=np.uint64) * np.uint64(FREG)) & np.uint64(0xFFFFFFFF)).astype(np.float64) / 2.0 ** 32
    want_c = np.rint(np.ldexp(h.astype(np.float64), E) * np.cos(th)).astype(np.int64)
    want_s = np.rint(np.ldexp(h.astype(np.float64), E) * np.sin(th)).astype(np.int64)
    # table 0 = [c ; s], table 1 = [-s ; c]: rows 0 and 8 are the bands of the block's first output (tt = c = hist - k)
    for t, row, want in ((0, 0, want_c), (0, 8, want_s), (1, 0, -want_s), (1, 8, want_c)):
        T = band(tabs[t])
        H = np.array([sum(int(T[j, row, 64 - k]) << (8 * j) for j in range(4)) for k in range(56)])
        assert np.abs(H - want).max() <= 1, (t, row)                                             # (rint vs llround on exact halves)
    T = band(tabs[0])                                            # rows r and r + 8: the same band, eight samples further on per row
    assert np.array_equal(T[:, 3, 24:24 + 64], T[:, 0, 0:64]) and np.array_equal(T[:, 11, 24:24 + 64], T[:, 8, 0:64])
    unit = np.ldexp(1.0, -E) / 8388607.0
    assert abs(scale - unit) <= 1e-7 * unit
    assert abs(ct[0] - float(want_c.sum() - want_s.sum()) * 32896.0 * unit) <= 1e-6 * abs(ct[0]) + 1e-9
    assert abs(ct[1] - float(want_c.sum() + want_s.sum()) * 32896.0 * unit) <= 1e-6 * abs(ct[1]) + 1e-9


def test_without_the_nco_the_table_is_k_fir_i8s(pkg):
    h = load_taps("d8_127")
    tabs, scale, ct = tables(pkg, h, 128, False, 0)
    L = pkg.ddc_lib()
    old = np.zeros(4 * 4 * 64 * 16, np.int8)
    sc, c0 = C.c_float(), C.c_float()
    pkg.check(L.pddc_fir_i8_table(h.ctypes.data_as(C.POINTER(C.c_float)), h.size, 128, old.ctypes.data, old.nbytes,
                                  C.byref(sc), C.byref(c0)))
    assert np.array_equal(tabs[0].reshape(-1), old)
    assert abs(scale - sc.value) <= 1e-7 * scale and abs(ct[0] - c0.value) <= 1e-6 * abs(c0.value) + 1e-12 and ct[0] == ct[1]


@pytest.mark.parametrize("mix", [True, False])
def test_restated_fused_pair_matches_the_oracle(pkg, O, mix):
    """second stage on the first stage's u values: complex taps g2[k] = h2[k] e^{+j 8 theta k} in descending order, one
    rotation with the phase of input sample 64 P -- as thread p of the kernel walks its window"""
    L = pkg.ddc_lib()
    h1, h2 = lowpass(48, 0.05), lowpass(56, 0.05)
    freg = FREG if mix else 0
    packed = O.lcg_bytes(6 * 64 * 150, 3)
    ref = O.ddc_chain(packed, [(8, h1), (8, h2)], freg=freg, mix=mix)
    tabs, scale, ct = tables(pkg, h1, 64, mix, freg)
    u = restated_first_stage(tabs, scale, ct, 64, packed, mix)
    g = np.zeros(136, np.float32)
    pkg.check(L.pddc_fir_i8x_taps2(h2.ctypes.data_as(C.POINTER(C.c_float)), h2.size, int(mix), freg,
                                   g.ctypes.data_as(C.POINTER(C.c_float)), g.size))
    gre, gim = g[:65].astype(np.float64), g[68:133].astype(np.float64)
    if not mix:
        assert not gim.any() and np.array_equal(g[64 - np.arange(56)], h2)
    up = np.concatenate([np.zeros((64, 2), np.float32), u]).astype(np.float64)           # the porch: zeros at a stream's start
    nz = u.shape[0] // 8
    z = np.zeros((nz, 2), np.float32)
    for P in range(nz):
        w = up[8 * P:8 * P + 65]                                                          # u[8P - 64 .. 8P]
        zr = (gre * w[:, 0]).sum() - (gim * w[:, 1]).sum()
        zi = (gre * w[:, 1]).sum() + (gim * w[:, 0]).sum()
        z[P] = (zr, zi)
    if mix:
        z = rotate(z, freg, 64)
    assert z.size == ref.size and O.rel_err(z.reshape(-1), ref) <= 3e-7, O.rel_err(z.reshape(-1), ref)


def test_extremes_stay_inside_the_accumulators(pkg, O):
    """mode 2 adds two band products into one accumulator set: taps of one sign at the 128-tap limit, samples at the
    extremes, a tuning word that keeps cos and sin near 0.7 -- the accumulators stay below 2^24 (exact float conversion;
    asserted inside restated_first_stage)"""
    h = (np.ones(128, np.float32) / 128 * (1 + 1e-3 * np.arange(128))).astype(np.float32)
    freg = 1 << 29                                                # 45 degrees per sample
    ns = 8 * 200
    v = np.full((ns, 2), (1 << 23) - 1, np.int64)
    v[ns // 2:] = -(1 << 23)
    b = np.zeros((ns, 2, 3), np.uint8)
    for i in range(3):
        b[:, :, i] = (v >> (8 * i)) & 0xFF
    packed = b.reshape(-1)
    tabs, scale, ct = tables(pkg, h, 128, True, freg)
    y = rotate(restated_first_stage(tabs, scale, ct, 128, packed, True), freg, 8).reshape(-1)
    ref = O.ddc_chain(packed, [(8, h)], freg=freg, mix=True)
    assert O.rel_err(y, ref) <= 3e-7, O.rel_err(y, ref)


def test_tables_refuse_what_they_cannot_hold(pkg):
    L = pkg.ddc_lib()
    buf = np.zeros(3 * 4 * 6 * 1024, np.int8)
    sc, ct = C.c_float(), (C.c_float * 2)()
    h = np.zeros(60, np.float32)

    def call(n, hist, mix, nbytes):
        return L.pddc_fir_i8x_tables(h.ctypes.data_as(C.POINTER(C.c_float)), n, hist, mix, FREG, buf.ctypes.data, nbytes,
                                     C.byref(sc), ct)

    assert call(60, 64, 1, buf.nbytes) < 0                          # all-zero taps
    h[5] = 0.25
    assert call(60, 64, 1, buf.nbytes) == 2
    assert call(60, 128, 1, buf.nbytes) == 2
    assert call(60, 256, 1, buf.nbytes) == 2
    assert call(60, 64, 0, buf.nbytes) == 1
    assert call(60, 32, 1, buf.nbytes) < 0                          # more taps than the history reaches
    assert call(60, 96, 1, buf.nbytes) < 0                          # no such geometry
    assert call(60, 64, 1, 2 * 4 * 2 * 1024 - 1) < 0                # buffer too small


@pytest.mark.parametrize("first", range(10))
def test_restated_decimate_by_ten_matches_the_oracle_at_every_phase(pkg, O, first):
    """The tuned decimate-by-10 first stage (the 1.6 MS/s plan's, perseus-sdr.c:776-892 picks the rate) on the paired-rows form:
    columns of 8 outputs 80 samples apart, tt = c - 10 (r & 7), 3 k-steps, hist = 64.  A batch's first output may sit on any
    sample 0 .. 9 (the stream's decimation phase), the loaders' groups sit on multiples of 8: the taps are delayed by
    delay = (-first) mod 8, the windows end on in_off = first + delay, and the rotation takes the phase of THAT sample.
    Restated from pddc_fir_i8x_d10_tables and compared with the oracle's mix-then-filter on the outputs
    y[m] = sum_k h[k] (x LO)[first + 10 m - k]."""
    L = pkg.ddc_lib()
    h = lowpass(51, 0.04)
    delay = (8 - first % 8) % 8
    in_off = first + delay
    tab = np.zeros(2 * 4 * 3 * 64 * 16, np.int8)
    sc, ct = C.c_float(), (C.c_float * 2)()
    assert pkg.check(L.pddc_fir_i8x_d10_tables(h.ctypes.data_as(C.POINTER(C.c_float)), h.size, delay, FREG, tab.ctypes.data,
                                               tab.nbytes, C.byref(sc), ct)) == 2
    Ts = [band(t) for t in tab.reshape(2, 4, 3, 64, 16)]
    K = Ts[0].shape[2]
    assert K == 192
    ns = 8 * 500
    packed = O.lcg_bytes(6 * ns, 21)
    n_out = (ns - 1 - first) // 10 + 1
    # planes of [64 zero samples of history | the batch from in_off on]
    xp = byte_planes(packed[6 * in_off:], 64, K)
    u = np.zeros((n_out, 2), np.float32)
    for col in range((n_out + 7) // 8):
        X = xp[:, 80 * col:80 * col + K, :]
        acc = plane_products(Ts[0], X[:, :, 0]) + plane_products(Ts[1], X[:, :, 1])
        assert np.abs(acc).max() < 1 << 24
        y = recombine(acc) * np.float32(sc.value)
        m = min(8, n_out - 8 * col)
        u[8 * col:8 * col + m, 0] = (y[:8] + np.float32(ct[0]))[:m]
        u[8 * col:8 * col + m, 1] = (y[8:] + np.float32(ct[1]))[:m]
    # a stream that starts with this batch has zeros in front of sample 0: the window of output 0 reaches back to
    # in_off - 63 < 0 only through zero taps or zero history, except that byte_planes() started the planes AT in_off:
    # the first in_off samples of the batch are the history's last -- put them there
    if in_off:
        hist = byte_planes(packed[:6 * in_off], 64 - in_off, 0)[:, :64, :]
        xp[:, :64, :] = hist
        for col in range(2):                                     # only the first columns reach into the history
            X = xp[:, 80 * col:80 * col + K, :]
            acc = plane_products(Ts[0], X[:, :, 0]) + plane_products(Ts[1], X[:, :, 1])
            y = recombine(acc) * np.float32(sc.value)
            m = min(8, n_out - 8 * col)
            u[8 * col:8 * col + m, 0] = (y[:8] + np.float32(ct[0]))[:m]
            u[8 * col:8 * col + m, 1] = (y[8:] + np.float32(ct[1]))[:m]
    y = rotate(u, FREG, 10, n0=in_off).reshape(-1)
    # oracle: mix, then y[m] = sum_k h[k] x[first + 10 m - k]: its decimator gives outputs at 10 m' + 9 - ... -- take the
    # full-rate filter output and pick the samples
    full = O.ddc_chain(packed, [(1, h)], freg=FREG, mix=True).reshape(-1, 2)
    ref = full[first::10][:n_out].reshape(-1)
    assert y.size == ref.size and O.rel_err(y, ref) <= 3e-7, O.rel_err(y, ref)
