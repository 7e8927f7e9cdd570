"""The reference's OWN example client, compiled from /root/reference/examples
(oracle/Makefile target `ref`, binaries in oracle/_ref/), running unmodified
against this repository's drop-in library.

Two things are established here, on the CPU:
  1. drop-in: the reference client builds against include/perseus-sdr.h and
     runs its whole call sequence against libperseus-sdr.so;
  2. oracle pinning: the bytes written by the REFERENCE's unpack callbacks
     (examples/perseustest.c:432-502, executed on this machine) are bit-equal to
     the oracle's, for the LCG stream and for the exhaustive 2^24 vector whose
     SHA-256 the survey recorded.
The binaries are prebuilt where the reference tree exists and travel to the
GPU box; nothing here reads /root/reference at run time.
"""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLD, ROOT

REF_BIN = os.path.join(ROOT, "oracle", "_ref", "perseustest_ref")
pytestmark = pytest.mark.skipif(not os.path.exists(REF_BIN),
                                reason="oracle/_ref not built (needs the reference tree at build time)")


def run_ref(tmp_path, args, env_extra, seconds=1):
    out = tmp_path / "out.bin"
    env = dict(os.environ, PERSEUS_AMD_PACE="0", PERSEUS_AMD_MODE="wire", **env_extra)
    env.pop("PERSEUS_AMD_DEVICES", None)
    p = subprocess.run([REF_BIN, "-a", "-t", str(seconds), "-s", "95000", "-d", "3", "-o", str(out)] + args,
                       env=env, capture_output=True, text=True, timeout=120)
    return p, np.fromfile(out, dtype=np.uint8)


def test_reference_client_float_callback_pins_the_oracle(tmp_path, O):
    p, raw = run_ref(tmp_path, ["-p"], {"PERSEUS_AMD_MAX_BUFFERS": "300", "PERSEUS_AMD_SOURCE": "lcg:12345"})
    assert "1 Perseus receivers found" in p.stderr or "Perseus receivers found" in p.stderr
    assert "Elapsed time:" in p.stderr and "Rate:" in p.stderr          # the library's stop line
    f = raw.view(np.float32)
    assert f.size == 300 * 2048
    ref = O.unpack24_f32(O.lcg_bytes(300 * 6144, 12345))
    assert np.array_equal(f.view(np.uint32), ref.view(np.uint32))
    sha = json.load(open(os.path.join(GOLD, "unpack_golden.json")))["sha256"]["lcg_6144_out_f32"]
    assert hashlib.sha256(f[:2048].tobytes()).hexdigest() == sha


def test_reference_client_int32_callback_pins_the_oracle(tmp_path, O):
    p, raw = run_ref(tmp_path, [], {"PERSEUS_AMD_MAX_BUFFERS": "100", "PERSEUS_AMD_SOURCE": "lcg:777"})
    i = raw.view(np.int32)
    assert i.size == 100 * 2048
    assert np.array_equal(i, O.unpack24_i32(O.lcg_bytes(100 * 6144, 777)))


def test_reference_client_exhaustive_2p24(tmp_path, O):
    """Every 24-bit code through the reference's float callback, fed from a raw
    capture file: the SHA-256 the survey recorded for the reference."""
    sha = json.load(open(os.path.join(GOLD, "unpack_golden.json")))["sha256"]
    v = np.arange(1 << 24, dtype=np.int64)
    packed = O.pack24(v, (~v) & 0xFFFFFF)                     # 96 MiB = 16384 transfers of 6144 bytes
    cap = tmp_path / "exhaustive.raw"
    packed.tofile(cap)
    p, raw = run_ref(tmp_path, ["-p"], {"PERSEUS_AMD_SOURCE": f"file:{cap}"}, seconds=6)
    assert raw.size == (1 << 24) * 8, p.stderr[-400:]
    assert hashlib.sha256(raw.tobytes()).hexdigest() == sha["exhaustive_f32"]
    assert np.array_equal(raw.view(np.uint32), O.unpack24_f32(packed).view(np.uint32))


def test_second_reference_client_simple_c(tmp_path, O):
    """examples/simple.c, the reference's minimal client (its own copy of the int32 callback,
    simple.c:33-61; 96 kS/s, 7.05 MHz, a fixed 10 s run, output ./perseusdata.bin): unmodified,
    against the drop-in library, bit-equal to the oracle."""
    simple = os.path.join(ROOT, "oracle", "_ref", "simple_ref")
    if not os.path.exists(simple):
        pytest.skip("simple_ref not built")
    env = dict(os.environ, PERSEUS_AMD_PACE="0", PERSEUS_AMD_MODE="wire", PERSEUS_AMD_SOURCE="lcg:4321",
               PERSEUS_AMD_MAX_BUFFERS="64")
    env.pop("PERSEUS_AMD_DEVICES", None)
    p = subprocess.run([simple], env=env, cwd=tmp_path, capture_output=True, text=True, timeout=60)
    assert "1 Perseus receiver(s) found" in p.stderr and "Bye" in p.stderr
    assert p.returncode == 1                                   # simple.c:145 returns 1 on the normal path
    i = np.fromfile(tmp_path / "perseusdata.bin", dtype=np.int32)
    assert i.size == 64 * 2048
    assert np.array_equal(i, O.unpack24_i32(O.lcg_bytes(64 * 6144, 4321)))
