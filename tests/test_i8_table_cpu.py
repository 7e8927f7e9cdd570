"""CPU tests of the tap operand of k_fir_i8x's plain form (pddc_fir_i8_table, include/perseus_ddc.h: host arithmetic, no GPU;
until round 5 the operand of round 3's k_fir_i8, same layout).  The kernel's arithmetic is integer and exact, so all of it can be restated in numpy from the table alone: byte planes of
the packed samples, digit planes of the taps, the banded Toeplitz product in the matrix instruction's lane order, the
nine plane products, the float recombination.  Checked here: the digits reconstruct the quantised taps, the band sits
where out[m] = sum_k h[k] x[8m - k] needs it, and the restated kernel agrees with the oracle to 1e-6 of full scale -- also
with the three dropped products at their worst."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_taps


def table(pkg, h, hist):
    L = pkg.ddc_lib()
    ks = (120 + hist + 63) // 64
    tab = np.zeros(4 * ks * 64 * 16, np.int8)
    sc, ct = C.c_float(), C.c_float()
    h = np.ascontiguousarray(h, np.float32)
    pkg.check(L.pddc_fir_i8_table(h.ctypes.data_as(C.POINTER(C.c_float)), h.size, hist, tab.ctypes.data, tab.nbytes,
                                  C.byref(sc), C.byref(ct)))
    return tab.reshape(4, ks, 64, 16), sc.value, ct.value


def band(tab, hist):
    """table -> T[j][r][c], the 16 x (64 ksteps) matrix of tap plane j"""
    ks = tab.shape[1]
    T = np.zeros((4, 16, 64 * ks), np.int64)
    for k in range(ks):
        for lane in range(64):
            T[:, lane & 15, 64 * k + 16 * (lane >> 4):64 * k + 16 * (lane >> 4) + 16] = tab[:, k, lane, :]
    return T


@pytest.mark.parametrize("name,hist", [("d8_255", 256), ("d8_127", 128)])
def test_digits_reconstruct_the_quantised_taps_and_the_band_is_in_place(pkg, name, hist):
    h = load_taps(name)
    tab, scale, cterm = table(pkg, h, hist)
    T = band(tab, hist)
    E = 30 - int(np.ceil(np.log2(np.abs(h).max())))
    x = np.ldexp(h.astype(np.float64), E)
    H = (np.sign(x) * np.floor(np.abs(x) + 0.5)).astype(np.int64)        # llround: halves away from zero
    assert np.abs(tab).max() <= 128 and np.abs(H).max() <= 1 << 30
    for r in range(16):
        for c in range(T.shape[2]):
            tt = c - 8 * r
            v = sum(int(T[j, r, c]) << (8 * j) for j in range(4))
            want = int(H[hist - tt]) if 1 <= tt <= hist and hist - tt < h.size else 0
            assert v == want, (r, c)
    unit = np.ldexp(1.0, -E) * 256.0 / 2147483391.0
    assert abs(scale - unit) <= 1e-7 * unit
    assert abs(cterm - float(H.sum()) * 32896.0 * unit) <= 1e-6 * abs(cterm) + 1e-12


def restated_kernel(tab, scale, cterm, hist, packed):
    """what the plain form computes for a batch that starts a stream (zero history), from the table alone"""
    T = band(tab, hist)
    K = T.shape[2]
    b = packed.reshape(-1, 2, 3).astype(np.int64)
    ns = b.shape[0]
    planes = np.stack([b[:, :, 0] - 128, b[:, :, 1] - 128, np.where(b[:, :, 2] >= 128, b[:, :, 2] - 256, b[:, :, 2])])
    zero = np.array([-128, -128, 0], np.int64)[:, None, None]            # the planes of a zero sample
    pad = K + 128
    xp = np.concatenate([np.broadcast_to(zero, (3, hist, 2)), planes, np.broadcast_to(zero, (3, pad, 2))], axis=1)
    n_out = ns // 8
    out = np.zeros((n_out, 2), np.float32)
    for col in range((n_out + 15) // 16):
        X = xp[:, 128 * col:128 * col + K, :]                          # [plane i][c][component]
        acc = np.zeros((4, 16, 2), np.int64)
        for i in range(3):
            for j in range(4):
                if i + j >= 2:
                    acc[i + j - 2] += T[j] @ X[i]
        assert np.abs(acc).max() < 1 << 24                               # the float conversions are exact
        a = acc.astype(np.float32)
        y = ((a[0] * np.float32(65536.0) + a[1] * np.float32(16777216.0)) +
             (a[2] * np.float32(4294967296.0) + a[3] * np.float32(1099511627776.0))) * np.float32(scale) + np.float32(cterm)
        m = min(16, n_out - 16 * col)
        out[16 * col:16 * col + m] = y[:m]
    return out.reshape(-1)


@pytest.mark.parametrize("name,hist", [("d8_255", 256), ("d8_127", 128)])
def test_restated_int8_kernel_matches_the_oracle(pkg, O, name, hist):
    h = load_taps(name)
    tab, scale, cterm = table(pkg, h, hist)
    packed = O.lcg_bytes(6 * 8 * 700, 4)
    y = restated_kernel(tab, scale, cterm, hist, packed)
    ref = O.ddc_chain(packed, [(8, h)])
    assert y.size == ref.size and O.rel_err(y, ref) <= 2e-7, O.rel_err(y, ref)


def test_dropped_plane_products_at_their_worst(pkg, O):
    """taps of one sign, samples at the extremes: every term of every plane product has the same sign; the three
    products that are not computed (i + j < 2) then cost the most they can -- still far inside 1e-6 of full scale"""
    h = (np.ones(256, np.float32) / 256 * (1 + 1e-3 * np.arange(256))).astype(np.float32)
    tab, scale, cterm = table(pkg, h, 256)
    ns = 8 * 300
    v = np.full((ns, 2), (1 << 23) - 1, np.int64)
    v[ns // 2:] = -(1 << 23)
    b = np.zeros((ns, 2, 3), np.uint8)
    for i in range(3):
        b[:, :, i] = (v >> (8 * i)) & 0xFF
    y = restated_kernel(tab, scale, cterm, 256, b.reshape(-1))
    ref = O.ddc_chain(b.reshape(-1), [(8, h)])
    assert O.rel_err(y, ref) <= 3e-7, O.rel_err(y, ref)


def test_table_refuses_what_it_cannot_hold(pkg):
    L = pkg.ddc_lib()
    buf = np.zeros(4 * 6 * 1024, np.int8)
    sc, ct = C.c_float(), C.c_float()
    h = np.zeros(200, np.float32)

    def call(taps, n, hist, nbytes):
        return L.pddc_fir_i8_table(taps.ctypes.data_as(C.POINTER(C.c_float)), n, hist, buf.ctypes.data, nbytes,
                                   C.byref(sc), C.byref(ct))

    assert call(h, 200, 256, buf.nbytes) != 0                      # all-zero taps
    h[3] = 0.5
    assert call(h, 200, 256, buf.nbytes) == 0
    assert call(h, 200, 128, buf.nbytes) != 0                      # more taps than the history reaches
    assert call(h, 200, 200, buf.nbytes) != 0                      # a history the kernel has no form for
    assert call(h, 200, 256, 1000) != 0                            # buffer too small


@pytest.mark.parametrize("name,hist", [("d8_255", 256), ("d8_127", 128)])
def test_binary16_stored_taps_quantise_to_the_host_table(pkg, name, hist):
    """PDDC_F_TAPS_FP16 on k_fir_i8: the device holds the taps as binary16 (pddc_fir_i8_taps16) and the kernel's matrix
    waves build their operand from them.  Restated here lane by lane as the kernel does it -- two 16-byte loads of eight
    values per k-step at 128 + 64 ks + 16 kq - 8 r, float32 product with 2^E, halves away from zero, low byte of r, then
    (r + 128) >> 8 -- the result must be the table the host builds from the same binary16 values, byte for byte; also
    with taps far below the largest one (subnormal binary16 values, halves that need the rounding)."""
    L = pkg.ddc_lib()
    h = load_taps(name).astype(np.float64)
    h[::7] *= 2.0 ** -13                                           # some taps thirteen octaves down: binary16 subnormals
    h[3::11] = np.ldexp(np.round(np.ldexp(h[3::11], 20)) + 0.5, -20)   # ... and some that end in a half after scaling
    h = h.astype(np.float32)
    h16 = h.astype(np.float16).astype(np.float32)
    tab, _, _ = table(pkg, h16, hist)
    G = np.zeros(512, np.uint16)
    two_e = C.c_double()
    pkg.check(L.pddc_fir_i8_taps16(h.ctypes.data_as(C.POINTER(C.c_float)), h.size, hist, G.ctypes.data, G.size, C.byref(two_e)))
    E = 30 - int(np.ceil(np.log2(np.abs(h16).max())))
    assert two_e.value == 2.0 ** E
    assert not G[:129].any() and not G[129 + hist:].any()
    assert np.array_equal(G[129:129 + hist].view(np.float16).astype(np.float32)[::-1][:h.size], h16)
    vals = G.view(np.float16).astype(np.float32)
    ks = tab.shape[1]
    got = np.zeros_like(tab)
    for k in range(ks):
        for lane in range(64):
            r0, kq = lane & 15, lane >> 4
            i0 = 128 + 64 * k + 16 * kq - 8 * r0
            assert i0 % 8 == 0 and 0 <= i0 and i0 + 16 <= G.size
            x = vals[i0:i0 + 16] * np.float32(two_e.value)              # exact in binary32
            r = (x + np.copysign(np.float32(0.5), x)).astype(np.int32)  # conversion truncates: halves away from zero
            for j in range(4):
                got[j, k, lane, :] = (r & 255).astype(np.uint8).view(np.int8)
                r = (r + 128) >> 8
            assert not r.any()
    assert np.array_equal(got, tab)
