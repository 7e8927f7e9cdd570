"""CPU tests of the bench harness pieces that do not need a GPU: argument
defaults, the cpu_baseline leg (oracle float path) and the loud failure of the
product loader when the HIP library is missing."""
import importlib
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _bench():
    sys.path.insert(0, ROOT)
    return importlib.import_module("bench")


def test_defaults_follow_the_contract(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse()
    assert a.gpus == 1 and a.steps >= 10 and a.warmup >= 1 and a.workload == "d8_127" and a.log2n == 28
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "7", "--warmup", "2"])
    a = b.parse()
    assert (a.gpus, a.steps, a.warmup) == (8, 7, 2)
    assert b.HBM_PEAK_GBS == 8000.0


def test_cpu_baseline_leg_fields():
    b = _bench()
    r = b.cpu_baseline("d8_127", 0.2)
    assert set(r) == {"value", "unit", "cores", "kind", "sample"}
    assert r["unit"] == "MS/s" and r["kind"] == "port" and r["cores"] >= 1 and r["value"] > 0
    r1 = b.cpu_baseline("unpack", 0.1)
    assert r1["cores"] == 1 and r1["value"] > 0


def test_bench_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "no CPU path" in (p.stderr + p.stdout)


def test_loader_fails_loudly_without_the_hip_library(pkg, monkeypatch):
    monkeypatch.setattr(pkg, "_ddc", None)
    monkeypatch.setattr(pkg, "DDC_LIB", os.path.join(ROOT, "libperseus-sdr_amd", "does_not_exist.so"))
    with pytest.raises(FileNotFoundError) as e:
        pkg.ddc_lib()
    assert "no CPU fallback" in str(e.value)


def test_committed_bench_line_schema():
    d = json.load(open(os.path.join(ROOT, "profiles", "r01", "v8_final_bench.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["traffic"] and abs(rf["traffic"] / (7 * 2 ** 28) - 1) < 0.01      # no wasted re-reads
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}


@pytest.mark.gpu
def test_bench_line_end_to_end_on_the_gpu():
    """The real thing, small: one JSON line, last on stdout, metric string verbatim from
    BASELINE.json, roofline and cpu_baseline objects present and self-consistent."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "2", "--log2n", "24",
                        "--settle-ms", "20", "--cpu-seconds", "0.5"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    d = json.loads(lines[-1])
    assert d["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    assert d["unit"] == "MS/s" and d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert abs(d["value"] - (1 << 24) * 8 / (d["ms_per_step"] * 8 * 1e-3) / 1e6) / d["value"] < 0.01
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["kernel"] == "k_fir8"
    assert abs(rf["achieved"] - 7.0 * (1 << 24) / (rf["kernel_ms"] * 1e-3) / 1e9) / rf["achieved"] < 0.01
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "MS/s"
