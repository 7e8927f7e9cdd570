#!/usr/bin/env python3
"""Regenerate tests/golden/* data fixtures (inputs + expected outputs).

Provenance of each fixture:
  unpack_golden.json  -- reference outputs recorded in SURVEY.md 8c (produced
                         there by the reference's own callbacks,
                         examples/perseustest.c:432-502); this script only
                         re-checks that the oracle reproduces them.
  lcg_6144.in/.out    -- the 6144-byte LCG buffer of SURVEY.md 8c and its float
                         unpack (SHA-256 checked against the survey's value).
  ddc_*.f32           -- outputs of the in-repo oracle (authored definition,
                         parity unpinned by the reference) for small inputs.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

SURVEY_KAT = [  # 24-bit code, int32 (MSB aligned), float bits
    (0x000000, 0, 0x00000000), (0x000001, 256, 0x34000001),
    (0x000002, 512, 0x34800001), (0x123456, 305419776, 0x3E11A2B1),
    (0x400000, 1073741824, 0x3F000001), (0x7FFFFF, 2147483392, 0x3F800000),
    (0x800000, -2147483648, 0xBF800001), (0x800001, -2147483392, 0xBF800000),
    (0xC00000, -1073741824, 0xBF000001), (0xFFFFFF, -256, 0xB4000001),
]
SURVEY_SHA = {
    "exhaustive_f32": "7e5c094b42dc0377503af871672c1c62981e357cdf19d0952db4a94c0db42884",
    "exhaustive_i32": "fb09d271bb13d3bc14172f15eb09a1ab3ac215b431913515755b4faf9cfa38cb",
    "lcg_6144_out_f32": "d44dcef7ae568bff3085585ae46432d24d93ca068de41107a5246a9f64eea978",
    "lcg_6144_in_prefix": "945cfe8e",
    "lcg_6144_in_first12": "05048ba2e81c7e8c98c80abe",
}
NCO_KAT = {"7100000": 381178347, "7050000": 378493992, "7000000": 375809638,
           "40000000": 2147483648, "0": 0}


def taps(name):
    return np.fromfile(os.path.join(GOLD, f"taps_{name}.f32"), dtype=np.float32)


def tone_packed(n, f_hz, fs=80e6, amp=0.5, seed=7):
    """complex tone + small noise, quantised to 24 bit, wire-packed."""
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64)
    z = amp * np.exp(2j * np.pi * f_hz / fs * t)
    z += 1e-3 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    i24 = np.clip(np.round(z.real * 8388607), -8388608, 8388607).astype(np.int64)
    q24 = np.clip(np.round(z.imag * 8388607), -8388608, 8388607).astype(np.int64)
    return O.pack24(i24, q24)


def main():
    os.makedirs(GOLD, exist_ok=True)
    with open(os.path.join(GOLD, "unpack_golden.json"), "w") as f:
        json.dump({"provenance": "SURVEY.md 8c: outputs of the reference callbacks "
                                 "examples/perseustest.c:432-502 compiled in the survey container",
                   "kat": [{"code24": c, "int32": i, "float_bits": b} for c, i, b in SURVEY_KAT],
                   "sha256": SURVEY_SHA, "nco_freg_kat": NCO_KAT,
                   "exhaustive_input": "sample v: I=v, Q=(~v)&0xFFFFFF, v=0..2^24-1"}, f, indent=1)

    b = O.lcg_bytes(6144, 12345)
    assert b[:12].tobytes().hex() == SURVEY_SHA["lcg_6144_in_first12"]
    out = O.unpack24_f32(b)
    assert hashlib.sha256(out.tobytes()).hexdigest() == SURVEY_SHA["lcg_6144_out_f32"]
    b.tofile(os.path.join(GOLD, "lcg_6144.in"))
    out.tofile(os.path.join(GOLD, "lcg_6144.f32.out"))

    # oracle-generated DSP fixtures (small)
    n = 8 * 4096 + 8 * 40                       # not a multiple of any tile
    lcg = O.lcg_bytes(6 * n, 12345)
    y = O.ddc_chain(lcg, [(8, taps("d8_127"))])
    y.tofile(os.path.join(GOLD, "ddc_d8_127_lcg.f32"))
    y = O.ddc_chain(lcg, [(8, taps("d8_255"))])
    y.tofile(os.path.join(GOLD, "ddc_d8_255_lcg.f32"))

    n3 = 320 * 200
    tone = tone_packed(n3, 7.1e6 + 1000.0)
    tone.tofile(os.path.join(GOLD, "tone_7101k.in"))
    stages = [(8, taps("c320_s1_d8_32")), (8, taps("c320_s2_d8_64")), (5, taps("c320_s3_d5_161"))]
    y = O.ddc_chain(tone, stages, freg=381178347, mix=True)
    y.tofile(os.path.join(GOLD, "ddc_c320_tone.f32"))
    meta = {"ddc_d8_lcg_samples": n, "lcg_seed": 12345, "tone_samples": n3,
            "tone_hz": 7101000.0, "freg": 381178347,
            "c320_stages": [[8, "c320_s1_d8_32"], [8, "c320_s2_d8_64"], [5, "c320_s3_d5_161"]]}
    with open(os.path.join(GOLD, "ddc_golden.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("golden fixtures written to", GOLD)


if __name__ == "__main__":
    main()
