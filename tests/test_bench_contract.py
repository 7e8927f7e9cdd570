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
    assert set(r) == {"value", "unit", "cores", "kind", "sample", "cpu_model", "single_thread"} and r["cpu_model"]
    assert r["unit"] == "MS/s" and r["kind"] == "port" and r["cores"] >= 1 and r["value"] > 0
    # beside the all-core figure the single-thread, 6144-byte-callback leg (the reference's own way of running)
    s1 = r["single_thread"]
    assert s1["cores"] == 1 and s1["unit"] == "MS/s" and 0 < s1["value"] and "6144-byte callbacks" in s1["sample"]
    rc = b.cpu_baseline("c320", 0.2)
    assert rc["single_thread"]["value"] > 0 and rc["value"] > 0 and "one single-threaded streaming float chain per core" in rc["sample"]
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


def _latest_committed_bench_line():
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "*final_bench*.json")))
    return json.load(open(files[-1]))


def test_committed_bench_line_schema():
    d = _latest_committed_bench_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    # dtype: the arithmetic the dominant kernel computes in -- int8 digit/byte planes into int32, float recombination (k_fir_i8x)
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] in ("f32", "i8xi8->i32, f32 out")
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}


def test_traffic_is_reported_only_with_matching_provenance(tmp_path, monkeypatch):
    """roofline.traffic comes from offline PMC passes; bench.py must refuse to print it for a
    kernel source or launch shape other than the one it was measured on."""
    b = _bench()
    sig = b.kernel_source_sig()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    json.dump({"d8_127": 1.88e9, "provenance": {"kernel_source_sha16": sig, "log2n": 28, "commit": "abc"}},
              open(prof / "pmc_traffic.json", "w"))
    v, src = b.traffic_from_profile("d8_127", sig, 28, False)
    assert v == 1.88e9 and "abc" in src
    assert b.traffic_from_profile("d8_127", "0" * 16, 28, False)[0] is None          # other kernel source
    assert "stale" in b.traffic_from_profile("d8_127", "0" * 16, 28, False)[1]
    assert b.traffic_from_profile("d8_127", sig, 24, False)[0] is None               # other launch shape
    assert b.traffic_from_profile("d8_127", sig, 28, True)[0] is None
    assert b.traffic_from_profile("c320", sig, 28, False)[0] is None                 # not measured
    json.dump({"d8_127": 1.88e9}, open(prof / "pmc_traffic.json", "w"))               # no provenance at all
    assert b.traffic_from_profile("d8_127", sig, 28, False)[0] is None


DRY = os.path.join(ROOT, "tests", "bench_dry.py")     # bench.py with its rank body replaced (the CPU stand-in lives in tests/)


def _dry_env():
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    return env


def test_plain_command_launcher_two_ranks_gloo():
    """`python bench.py --gpus 2` as a plain command: the parent starts the ranks itself (it never
    imports torch), relays ONE JSON line.  On CPU the children run the gloo plumbing mode."""
    env = _dry_env()
    p = subprocess.run([sys.executable, DRY, "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--workload", "c320"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["devices"] == ["cpu:0", "cpu:1"]
    assert d["gather"]["this_workload"]["root_blocks_match_each_ranks_stream"] is True
    assert "dry_run" in d and d["value"] == 0.0                   # nothing is claimed as measured


def test_driver_style_launch_under_torch_distributed_run():
    """What the driver does at N>1: python -m torch.distributed.run ... bench.py --gpus N.  The ranks find
    RANK/WORLD_SIZE in the environment and skip the launcher (CPU: gloo plumbing mode)."""
    env = _dry_env()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_bench()._free_port()),
                        DRY, "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2
    assert d["gather"]["this_workload"]["root_blocks_match_each_ranks_stream"] is True


def test_launcher_parent_makes_no_gpu_call():
    """The launching parent must not import torch (never exec/spawn from a process that touched the GPU)."""
    code = ("import sys, bench\n"
            "sys.argv = ['bench.py', '--gpus', '2']\n"
            "bench.launch_ranks = lambda n, argv: (print('torch' in sys.modules), 0)[1]\n"
            "bench.main()\n")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert p.stdout.strip() == "False", (p.stdout, p.stderr[-500:])


def test_launcher_reports_failure_when_ranks_fail():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode != 0 and "no CPU path" in p.stderr


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
    # (2^24-sample launches of the 127-tap stage run on the int8 matrix cores; from 2^26 on the vector kernel k_fir8)
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["kernel"].startswith("k_fir_i8")
    assert abs(rf["achieved"] - 7.0 * (1 << 24) / (rf["kernel_ms"] * 1e-3) / 1e9) / rf["achieved"] < 0.01
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "MS/s"
    # self-contained roofline: measured copy ceiling, labelled traffic, parity in the line
    assert rf["copy_ceiling_GBps"] and 2000 < rf["copy_ceiling_GBps"] < 8000
    assert abs(rf["frac_of_copy_ceiling"] - rf["achieved"] / rf["copy_ceiling_GBps"]) < 1e-3
    assert rf["traffic"] is None and rf["traffic_source"]        # the offline PMC figure is for 2^28 launches only
    v = d["verified"]
    assert v["ok"] is True and v["windows"] >= 20 and v["max_rel_err"] <= 1e-6 and "max|y-ref|" in v["metric"]
    # every output of the last timed step, not only windows (hosts with 16 cores or more), and what the parity is pinned to
    if (os.cpu_count() or 1) >= 16:
        ev = v["every_output"]
        assert ev["ok"] is True and ev["compared"] == v["n_outputs"] == (1 << 24) // 8 and ev["bad"] == 0 and ev["max_rel_err"] <= 1e-6
    assert "UNPINNED" in v["pinned"] and "bit-exact" in v["pinned"]
    assert d["cpu_baseline"]["single_thread"]["cores"] == 1 and d["phases_s"]["total"] > 0 and d["wall_budget_s"]["total"] > 0
    assert d["config"]["taps"].startswith("fp32")          # (fp32 values; on the int8 kernel: as four digit planes)
    # placement by the rule: a handful of probed pairs, the first-come time next to the chosen one (2^24-sample launches
    # live in the last-level cache, so there may be nothing to choose -- then no search is made at all)
    pl = d["placement"]
    # (four probes when the rule holds; every slot of the arena -- and of a larger one, "grown_after" -- when it does not)
    assert pl is None or ((pl["probe_pairs"] <= 12 or pl["grown_after"]) and pl["first_come_ms"] >= pl["chosen"]["ms"])
    assert pl is None or pl["arena_GiB"] <= 72                           # a quarter of the HBM at most, never grown by default
    assert "value_first_come" in d and (pl is None or 0 < d["value_first_come"] <= d["value"] * 1.1)


def test_eight_rank_dry_run_of_the_launcher():
    """The driver's N = 8 shape on CPU: eight gloo ranks through the plain-command launcher, one JSON line, every rank's
    block gathered and checked on rank 0 (8 ranks, not 2)."""
    env = _dry_env()
    p = subprocess.run([sys.executable, DRY, "--gpus", "8", "--steps", "1", "--warmup", "0",
                        "--workload", "c320", "--log2n", "12"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and d["devices"] == [f"cpu:{r}" for r in range(8)]
    assert d["gather"]["this_workload"]["root_blocks_match_each_ranks_stream"] is True
    # the wall-time budget of the real 8-GPU run (driver's arguments): every phase named, the whole well inside the few
    # minutes the driver allows a bench -- the gather legs are the link-bound part (225 steps of 268 MB per link)
    wb = d["wall_budget_s"]
    assert set(wb) >= {"start_import_rendezvous", "arena_alloc_and_rest", "placement_probes", "settle_warmup_timed", "verify",
                       "gather_leg_this_workload", "gather_leg_c320", "total"}
    assert 1.0 < wb["gather_leg_this_workload"] < 10 and wb["total"] < 120 and abs(wb["total"] - sum(v for k, v in wb.items() if k != "total")) < 0.5


def test_launcher_tears_the_other_ranks_down_when_one_dies(tmp_path):
    """a rank that dies before rendezvous must not leave the others waiting for their own timeouts."""
    wrap = tmp_path / "bench_dying.py"             # bench.py with a rank body of which rank 1 dies before the rendezvous
    wrap.write_text("import os, sys, time\n"
                    f"sys.path.insert(0, {ROOT!r})\n"
                    "import bench\n"
                    "def run(a):\n"
                    "    if os.environ['RANK'] == '1':\n"
                    "        sys.exit(7)\n"
                    "    time.sleep(600)\n"
                    "bench.run_rank = run\n"
                    "bench.main()\n")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    t0 = __import__("time").time()
    p = subprocess.run([sys.executable, str(wrap), "--gpus", "3", "--steps", "1"],
                       capture_output=True, text=True, timeout=120, env=env)
    assert p.returncode == 7 and __import__("time").time() - t0 < 60


def test_bench_holds_no_cpu_stand_in():
    """bench.py must contain no path in which the oracle produces the output: the oracle appears
    only in the cpu_baseline leg and in the parity check of the GPU's output."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "run_rank_dry" not in src and "PDDC_BENCH_BACKEND" not in src
    assert "PDDC_BENCH_RANK_HOOK" not in src and src.count("os.environ.get(") <= 6   # no hook that swaps the rank body from the environment
    assert src.count("from oracle import oracle") == 2            # cpu_baseline() and the verification of the last step


def test_arena_plan_for_eight_ranks_with_a_mocked_device():
    """The per-rank arena sizing (what bench.py does with torch.cuda.mem_get_info) as a pure function: every rank of an
    8-GPU node has its own 288 GB, so each plans the same arena; a GPU with little free memory gets none."""
    b = _bench()
    ns = 1 << 28
    full = b.arena_plan(free_bytes=280 << 30, in_bytes=6 * ns, out_bytes=(ns // 8 + 8) * 8, ws_bytes=0, arena_gib=192)
    assert full["gib"] == 192 and full["slot"] == 8 << 30 and full["nslot"] == 24
    casc = b.arena_plan(free_bytes=280 << 30, in_bytes=6 * ns, out_bytes=(ns // 320 + 8) * 8, ws_bytes=70 << 20, arena_gib=192)
    assert casc["gib"] == 192 and casc["in_span"] % (1 << 30) == 0 and casc["ws_span"] >= 70 << 20
    small = b.arena_plan(free_bytes=40 << 30, in_bytes=6 * ns, out_bytes=(ns // 8 + 8) * 8, ws_bytes=0, arena_gib=192)
    assert small["gib"] < 3 * 8                                   # fewer than three slots: no search, first-come buffers
    tiny = b.arena_plan(free_bytes=280 << 30, in_bytes=6 << 20, out_bytes=1 << 20, ws_bytes=0, arena_gib=192)
    assert tiny["search"] is False                                # small batches live in the last-level cache anyway


def test_bench_c320_is_the_api_plan():
    """ONE x320 filter set, benchmarked and shipped (round-3 review item 3): bench.py's c320 workload is the plan
    perseus_set_sampling_rate(250000) builds (perseus-sdr.c:776-892 picks the rate; plan_build in perseus_api.c the
    stages), tap for tap -- not a fixture of its own."""
    import importlib
    import numpy as np
    sys.path.insert(0, ROOT)
    b = importlib.import_module("bench")
    pkg = importlib.import_module("libperseus-sdr_amd")
    w = b.workload_def("c320", pkg)
    plan = pkg.api_plan(250000)
    assert [d for d, _ in w["stages"]] == [d for d, _, _ in plan] == [8, 8, 5]
    assert all(l == 1 for _, _, l in plan)
    for (_, t), (_, u, _) in zip(w["stages"], plan):
        assert np.array_equal(np.asarray(t, np.float32), u)
    assert [len(t) for _, t in w["stages"]] == [32, 41, 117]
    assert w["decim"] == 320 and w["mix"]
