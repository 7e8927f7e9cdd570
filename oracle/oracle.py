"""ctypes front-end and numpy restatement of the CPU oracle.

TEST INFRASTRUCTURE ONLY (see perseus_oracle.h): imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product.

Two independent restatements of the reference unpack are kept on purpose:
the C one (perseus_oracle.c, follows examples/perseustest.c:411-502) and the
vectorised numpy one below; tests check them against each other and against
the reference outputs recorded in tests/golden/unpack_golden.json.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libperseus_oracle.so")
_lib = None

# sampling-rate table of the reference (generate_fpga_code.sh:119-202 sorts the
# perseus*.rbs images by rate)
REFERENCE_RATES = (48000, 95000, 96000, 125000, 192000, 250000, 500000,
                   1000000, 1600000, 2000000)


def build(force: bool = False) -> str:
    """Compile the C oracle (gcc); building the checker is not using it."""
    src = os.path.join(_HERE, "perseus_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        u8p, f32p, f64p, i32p = (C.POINTER(C.c_uint8), C.POINTER(C.c_float),
                                 C.POINTER(C.c_double), C.POINTER(C.c_int32))
        L.orc_lcg_fill.argtypes = [u8p, C.c_size_t, C.c_uint32]
        L.orc_lcg_fill.restype = C.c_uint32
        L.orc_unpack24_f32.argtypes = [u8p, C.c_size_t, f32p]
        L.orc_unpack24_f32.restype = None
        L.orc_unpack24_i32.argtypes = [u8p, C.c_size_t, i32p]
        L.orc_unpack24_i32.restype = None
        L.orc_unpack24_f32_callback_style.argtypes = [u8p, C.c_size_t, f32p, C.c_size_t]
        L.orc_unpack24_f32_callback_style.restype = None
        L.orc_pack24_f32.argtypes = [f32p, C.c_size_t, u8p]
        L.orc_pack24_f32.restype = None
        L.orc_nco_freg.argtypes = [C.c_double, C.c_double]
        L.orc_nco_freg.restype = C.c_uint32
        L.orc_presel_id.argtypes = [C.c_double, C.c_int]
        L.orc_presel_id.restype = C.c_int
        L.orc_rate_index.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int]
        L.orc_rate_index.restype = C.c_int
        L.orc_nco_mix_f64.argtypes = [f32p, C.c_size_t, C.c_uint64, C.c_uint32, f64p]
        L.orc_nco_mix_f64.restype = None
        L.orc_nco_mix_retuned_f64.argtypes = [f32p, C.c_size_t, C.c_uint64, C.POINTER(C.c_uint64),
                                              C.POINTER(C.c_uint32), C.c_int, f64p]
        L.orc_nco_mix_retuned_f64.restype = None
        L.orc_fir_decim_f64.argtypes = [f64p, C.c_size_t, f32p, C.c_int, C.c_int, f64p]
        L.orc_fir_decim_f64.restype = C.c_size_t
        L.orc_ddc_chain.argtypes = [u8p, C.c_size_t, C.c_uint32, C.c_int, C.c_int,
                                    C.POINTER(C.c_int), C.POINTER(C.c_int),
                                    C.POINTER(f32p), C.POINTER(C.c_int), f32p, C.c_size_t]
        L.orc_resample_f64.argtypes = [f64p, C.c_size_t, f32p, C.c_int, C.c_int, C.c_int, f64p]
        L.orc_resample_f64.restype = C.c_size_t
        L.orc_ddc_chain.restype = C.c_size_t
        L.orc_stage1_f32.argtypes = [u8p, C.c_size_t, f32p, C.c_int, C.c_int, f32p, C.c_int]
        L.orc_stage1_f32.restype = C.c_size_t
        L.orc_max_threads.argtypes = []
        L.orc_max_threads.restype = C.c_int
        L.orc_stream_f32_callback_style.argtypes = [u8p, C.c_size_t, C.c_size_t, C.c_uint32, C.c_int, C.c_int,
                                                    C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(f32p), f32p, C.c_size_t]
        L.orc_stream_f32_callback_style.restype = C.c_size_t
        L.orc_chain_check.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64, C.c_size_t, C.c_uint32, C.c_int, C.c_int,
                                      C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(f32p), C.c_void_p, C.c_size_t,
                                      C.c_double, C.POINTER(CheckStats)]
        L.orc_chain_check.restype = C.c_longlong
        _lib = L
    return _lib


class CheckStats(C.Structure):
    _fields_ = [("max_err", C.c_double), ("max_ref", C.c_double), ("worst_chunk_ratio", C.c_double),
                ("n_compared", C.c_longlong), ("n_bad", C.c_longlong), ("first_bad", C.c_longlong),
                ("chunk_outputs", C.c_longlong)]


def _p(a: np.ndarray, ty):
    return a.ctypes.data_as(C.POINTER(ty))


# ---------------------------------------------------------------- synthetic
def lcg_bytes(nbytes: int, seed: int = 12345) -> np.ndarray:
    out = np.empty(nbytes, dtype=np.uint8)
    lib().orc_lcg_fill(_p(out, C.c_uint8), nbytes, seed & 0xFFFFFFFF)
    return out


def lcg_bytes_numpy(nbytes: int, seed: int = 12345) -> np.ndarray:
    """Closed-form LCG (affine powers), independent of the C loop."""
    a, c = 1664525, 1013904223
    # state after k steps: A_k*seed + C_k (mod 2^32), built by doubling
    n = nbytes
    A = np.empty(n, dtype=np.uint64)
    Cc = np.empty(n, dtype=np.uint64)
    M = np.uint64(0xFFFFFFFF)
    A[0], Cc[0] = a, c
    filled = 1
    while filled < n:
        m = min(filled, n - filled)
        # steps (filled+i+1) = steps(filled) o steps(i+1)
        Af, Cf = A[filled - 1], Cc[filled - 1]
        A[filled:filled + m] = (A[:m] * Af) & M
        Cc[filled:filled + m] = (A[:m] * Cf + Cc[:m]) & M
        filled += m
    s = (A * np.uint64(seed) + Cc) & M
    return (s >> np.uint64(24)).astype(np.uint8)


def pack24(i24: np.ndarray, q24: np.ndarray) -> np.ndarray:
    """Wire format (perseustest.c:449-455): I0 I1 I2 Q0 Q1 Q2, 24-bit LE."""
    i = np.asarray(i24).astype(np.int64) & 0xFFFFFF
    q = np.asarray(q24).astype(np.int64) & 0xFFFFFF
    out = np.empty((i.size, 6), dtype=np.uint8)
    for b in range(3):
        out[:, b] = (i >> (8 * b)) & 0xFF
        out[:, 3 + b] = (q >> (8 * b)) & 0xFF
    return out.reshape(-1)


# ------------------------------------------------------------------- unpack
def unpack24_i32_numpy(packed: np.ndarray) -> np.ndarray:
    """A1/A3: bytes 1..3 of a LE int32, byte 0 zero (perseustest.c:411-460)."""
    b = np.asarray(packed, dtype=np.uint8)
    ns = b.size // 6
    b = b[:ns * 6].reshape(ns, 2, 3).astype(np.uint32)
    u = (b[:, :, 0] << np.uint32(8)) | (b[:, :, 1] << np.uint32(16)) | (b[:, :, 2] << np.uint32(24))
    return u.astype(np.uint32).view(np.int32).reshape(-1)


def unpack24_f32_numpy(packed: np.ndarray) -> np.ndarray:
    """A2: (float)int32 / (float)(INT_MAX-256) (perseustest.c:496-497)."""
    v = unpack24_i32_numpy(packed).astype(np.float32)          # exact: 24 significant bits
    return v / np.float32(2147483391)                            # rounds to 2147483392.0f


def unpack24_f32(packed: np.ndarray) -> np.ndarray:
    b = np.ascontiguousarray(packed, dtype=np.uint8)
    out = np.empty(2 * (b.size // 6), dtype=np.float32)
    lib().orc_unpack24_f32(_p(b, C.c_uint8), b.size, _p(out, C.c_float))
    return out


def unpack24_i32(packed: np.ndarray) -> np.ndarray:
    b = np.ascontiguousarray(packed, dtype=np.uint8)
    out = np.empty(2 * (b.size // 6), dtype=np.int32)
    lib().orc_unpack24_i32(_p(b, C.c_uint8), b.size, _p(out, C.c_int32))
    return out


def pack24_f32(x_iq: np.ndarray) -> np.ndarray:
    """N1 (authored): float32 I/Q -> wire bytes, inverse of unpack24_f32."""
    x = np.ascontiguousarray(x_iq, dtype=np.float32).reshape(-1)
    out = np.empty(3 * x.size, dtype=np.uint8)
    lib().orc_pack24_f32(_p(x, C.c_float), x.size // 2, _p(out, C.c_uint8))
    return out


# ------------------------------------------------------------ control-plane
def nco_freg(hz: float, fclk: float = 80e6) -> int:
    return int(lib().orc_nco_freg(float(hz), float(fclk)))


def presel_id(hz: float, enable: bool = True) -> int:
    return int(lib().orc_presel_id(float(hz), int(bool(enable))))


def rate_index(sps: int, table=REFERENCE_RATES) -> int:
    t = (C.c_int * len(table))(*table)
    return int(lib().orc_rate_index(int(sps), t, len(table)))


# ----------------------------------------------------------------- DSP path
def nco_mix(x_iq: np.ndarray, freg: int, n0: int = 0) -> np.ndarray:
    x = np.ascontiguousarray(x_iq, dtype=np.float32)
    out = np.empty(x.size, dtype=np.float64)
    lib().orc_nco_mix_f64(_p(x, C.c_float), x.size // 2, n0, freg, _p(out, C.c_double))
    return out


def nco_mix_retuned(x_iq: np.ndarray, segments, n0: int = 0) -> np.ndarray:
    """NCO as a phase accumulator retuned while running: segments = [(first_sample, freg), ...],
    first_sample ascending and 0 for the first; phase-continuous at every switch."""
    x = np.ascontiguousarray(x_iq, dtype=np.float32)
    starts = (C.c_uint64 * len(segments))(*[int(a) for a, _ in segments])
    words = (C.c_uint32 * len(segments))(*[int(f) & 0xFFFFFFFF for _, f in segments])
    out = np.empty(x.size, dtype=np.float64)
    lib().orc_nco_mix_retuned_f64(_p(x, C.c_float), x.size // 2, n0, starts, words, len(segments),
                                  _p(out, C.c_double))
    return out


def nco_mix_retuned_numpy(x_iq: np.ndarray, segments, n0: int = 0) -> np.ndarray:
    """Second, independent statement of the retuned accumulator (cumulative sum of the word)."""
    x = np.asarray(x_iq, dtype=np.float64).reshape(-1, 2)
    n = np.arange(n0, n0 + x.shape[0], dtype=np.uint64)
    starts = np.array([a for a, _ in segments], dtype=np.uint64)
    words = np.array([f for _, f in segments], dtype=np.uint64)
    acc0 = np.zeros(len(segments), dtype=np.uint64)
    for i in range(1, len(segments)):
        acc0[i] = (acc0[i - 1] + (starts[i] - starts[i - 1]) * words[i - 1]) & np.uint64(0xFFFFFFFF)
    sg = np.searchsorted(starts, n, side="right") - 1
    ph = (acc0[sg] + (n - starts[sg]) * words[sg]) & np.uint64(0xFFFFFFFF)
    a = ph.astype(np.float64) * (2.0 * np.pi / 4294967296.0)
    z = (x[:, 0] + 1j * x[:, 1]) * np.exp(-1j * a)
    return np.stack([z.real, z.imag], axis=1).reshape(-1)


def ddc_chain_retuned(packed: np.ndarray, stages, segments) -> np.ndarray:
    """unpack -> retuned NCO -> FIR chain (plain decimators or rational stages), double, result float32."""
    x = nco_mix_retuned(unpack24_f32(packed), segments, 0)
    for st in stages:
        L = int(st[2]) if len(st) > 2 and st[2] and int(st[2]) > 1 else 1
        x = resample(x, st[1], L, int(st[0])) if L > 1 else fir_decim(x, st[1], int(st[0]))
    return np.asarray(x, dtype=np.float32)


def fir_decim(x_iq: np.ndarray, taps: np.ndarray, D: int) -> np.ndarray:
    x = np.ascontiguousarray(x_iq, dtype=np.float64)
    h = np.ascontiguousarray(taps, dtype=np.float32)
    ns = x.size // 2
    out = np.empty(2 * ((ns + D - 1) // D), dtype=np.float64)
    n = lib().orc_fir_decim_f64(_p(x, C.c_double), ns, _p(h, C.c_float), h.size, D, _p(out, C.c_double))
    return out[:2 * n]


def ddc_chain(packed: np.ndarray, stages, freg: int = 0, mix: bool = False) -> np.ndarray:
    """stages: sequence of (D, taps) or (D, taps, L) -- L > 1 makes the stage a
    rational L/D resampler.  Returns float32 I/Q."""
    b = np.ascontiguousarray(packed, dtype=np.uint8)
    ns = b.size // 6
    ds = [int(st[0]) for st in stages]
    ls = [int(st[2]) if len(st) > 2 and st[2] and st[2] > 1 else 1 for st in stages]
    hs = [np.ascontiguousarray(st[1], dtype=np.float32) for st in stages]
    n = ns
    for d, l in zip(ds, ls):
        n = (n * l + d - 1) // d
    out = np.empty(2 * max(n, 1), dtype=np.float32)
    k = max(len(ds), 1)
    Darr = (C.c_int * k)(*ds)
    Larr = (C.c_int * k)(*ls)
    Narr = (C.c_int * k)(*[h.size for h in hs])
    Tarr = (C.POINTER(C.c_float) * k)(*[_p(h, C.c_float) for h in hs])
    r = lib().orc_ddc_chain(_p(b, C.c_uint8), ns, freg & 0xFFFFFFFF, int(bool(mix)),
                            len(ds), Darr, Narr, Tarr, Larr, _p(out, C.c_float), max(n, 1))
    if r == C.c_size_t(-1).value:
        raise RuntimeError("orc_ddc_chain failed")
    return out[:2 * r]


def stream_callback_style(packed: np.ndarray, stages, freg: int = 0, mix: bool = False, buf_bytes: int = 6144) -> np.ndarray:
    """The chain as the reference would run it on a CPU: ONE thread, callbacks of buf_bytes, each unpacked the way
    examples/perseustest.c:466-502 does and pushed through streaming float FIR stages.  float32 I/Q."""
    b = np.ascontiguousarray(packed, dtype=np.uint8)
    ds = [int(st[0]) for st in stages]
    hs = [np.ascontiguousarray(st[1], dtype=np.float32) for st in stages]
    k = len(ds)
    n = b.size // 6
    for d in ds:
        n = (n + d - 1) // d
    out = np.empty(2 * (n + 8), dtype=np.float32)
    Darr = (C.c_int * k)(*ds)
    Narr = (C.c_int * k)(*[h.size for h in hs])
    Tarr = (C.POINTER(C.c_float) * k)(*[_p(h, C.c_float) for h in hs])
    r = lib().orc_stream_f32_callback_style(_p(b, C.c_uint8), b.size, buf_bytes, freg & 0xFFFFFFFF, int(bool(mix)), k, Darr,
                                            Narr, Tarr, _p(out, C.c_float), n + 8)
    if r == C.c_size_t(-1).value:
        raise RuntimeError("orc_stream_f32_callback_style failed")
    return out[:2 * r]


def chain_check(packed: np.ndarray, first_in: int, n_in: int, stages, got_iq: np.ndarray, freg: int = 0,
                mix: bool = False, tol: float = 1e-6) -> dict:
    """EVERY output of the batch that starts at absolute sample first_in (n_in samples) of the periodic stream
    `packed`, `packed`, ... against the double oracle (orc_chain_check; plain decimators).  got_iq: float32 pairs, the
    batch's outputs in order.  -> {"n": outputs compared, "max_rel_err": max|got-ref| / max|ref| over the whole batch,
    "worst_chunk_rel_err": the same per chunk of `chunk_outputs` outputs, worst chunk, "n_bad", "first_bad", "ok"}."""
    b = np.ascontiguousarray(packed, dtype=np.uint8)
    g = np.ascontiguousarray(got_iq, dtype=np.float32).reshape(-1)
    ds = [int(st[0]) for st in stages]
    if any(len(st) > 2 and st[2] and int(st[2]) > 1 for st in stages):
        raise ValueError("chain_check: plain decimators only")
    hs = [np.ascontiguousarray(st[1], dtype=np.float32) for st in stages]
    k = len(ds)
    Darr = (C.c_int * k)(*ds)
    Narr = (C.c_int * k)(*[h.size for h in hs])
    Tarr = (C.POINTER(C.c_float) * k)(*[_p(h, C.c_float) for h in hs])
    st = CheckStats()
    n = lib().orc_chain_check(b.ctypes.data, b.size // 6, int(first_in), int(n_in), freg & 0xFFFFFFFF, int(bool(mix)), k,
                              Darr, Narr, Tarr, g.ctypes.data, g.size // 2, float(tol), C.byref(st))
    if n < 0:
        raise RuntimeError("orc_chain_check: bad arguments (or fewer outputs than the batch produces)")
    rel = st.max_err / st.max_ref if st.max_ref > 0 else st.max_err
    return {"n": int(n), "max_rel_err": float(rel), "worst_chunk_rel_err": float(st.worst_chunk_ratio),
            "n_bad": int(st.n_bad), "first_bad": int(st.first_bad), "chunk_outputs": int(st.chunk_outputs),
            "tol": tol, "ok": bool(st.n_bad == 0 and n > 0)}


def resample(x_iq: np.ndarray, taps: np.ndarray, L: int, M: int) -> np.ndarray:
    x = np.ascontiguousarray(x_iq, dtype=np.float64)
    h = np.ascontiguousarray(taps, dtype=np.float32)
    ns = x.size // 2
    out = np.empty(2 * ((ns * L + M - 1) // M), dtype=np.float64)
    n = lib().orc_resample_f64(_p(x, C.c_double), ns, _p(h, C.c_float), h.size, L, M, _p(out, C.c_double))
    return out[:2 * n]


def fir_decim_numpy(x_iq: np.ndarray, taps: np.ndarray, D: int) -> np.ndarray:
    """Second, independent statement of y[m] = sum_k h[k] x[mD-k] (float64)."""
    x = np.asarray(x_iq, dtype=np.float64).reshape(-1, 2)
    z = x[:, 0] + 1j * x[:, 1]
    full = np.convolve(z, np.asarray(taps, dtype=np.float64))[:z.size]
    y = full[::D]
    return np.stack([y.real, y.imag], axis=1).reshape(-1)


def stage1_f32(packed: np.ndarray, taps: np.ndarray, D: int, threads: int = 0) -> np.ndarray:
    b = np.ascontiguousarray(packed, dtype=np.uint8)
    h = np.ascontiguousarray(taps, dtype=np.float32)
    ns = b.size // 6
    out = np.empty(2 * ((ns + D - 1) // D), dtype=np.float32)
    n = lib().orc_stage1_f32(_p(b, C.c_uint8), ns, _p(h, C.c_float), h.size, D,
                             _p(out, C.c_float), threads)
    return out[:2 * n]


def max_threads() -> int:
    return int(lib().orc_max_threads())


def rel_err(y: np.ndarray, ref: np.ndarray) -> float:
    """The FIR parity metric: max|y-ref| / max|ref| (DESIGN.md 'Tolerance')."""
    ref = np.asarray(ref, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    m = np.max(np.abs(ref)) if ref.size else 0.0
    if m == 0.0:
        return float(np.max(np.abs(y))) if y.size else 0.0
    return float(np.max(np.abs(y - ref)) / m)
