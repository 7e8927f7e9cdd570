#!/usr/bin/env python3
"""bench.py -- throughput of the I/Q ingest + decimation hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N>1 either arrives already launched by torch.distributed.run (RANK / LOCAL_RANK /
  WORLD_SIZE in the environment), or -- as a plain command -- this process starts the N
  ranks itself as child processes BEFORE it has imported torch or touched a GPU, relays
  rank 0's JSON line and exits with the worst child status.
A "step" is one pass of the hot path over one device-resident batch:
  BASELINE.json configs[1]: 2^28 complex samples of synthetic 24-bit I/Q (LCG,
  seed 12345+rank) -> fused unpack + 127-tap polyphase decimate-by-8 -> float32.
The stream shards as independent streams (one per GPU), so there is no
data-path collective in the timed region: scaling is "weak".  At N>1 the bench also
runs BASELINE config 4 -- every rank's decimated output gathered on rank 0's GPU over
xGMI -- and reports it in the "gather" object (never in `value`): once for this
workload (/8: xGMI-link-bound) and once for the /320 cascade (where the gather vanishes).
All GPU collectives are the C library's own RCCL calls (pddc_comm_*, ddc_multi.cpp);
torch.distributed is only the rendezvous (a gloo group carrying the 128-byte RCCL id).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (k_fir8):
algorithmic bytes = 7 B per input sample (6 packed in + 8/8 out) over the
kernel's average duration measured with HIP events on the launch stream;
`roofline.copy_ceiling_GBps` is a device-to-device copy of the same number of bytes
measured in this run.  `placement`: input, a cascade's inter-stage workspace and the output
are cut from one arena at the pair of 8 GiB slots where they run fastest against each other
(different HBM extent classes; every pair's probe time is in the line; NOTEBOOK.md rounds 1-3 5 (o)-(r)).
`verified` is a parity check of the LAST timed step's output
against the CPU oracle on windows placed at the tile scheduler's seams (outside the
timed region).  `cpu_baseline` times the oracle's float path (oracle/perseus_oracle.c
orc_stage1_f32, kind "port") on this box's cores over a bounded sample of the same
workload (N=1, rank 0 only).
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PARITY_METRIC = "max|y-ref| / max|ref| per window (full-scale-relative), ref = CPU oracle with double accumulation"
PARITY_TOL = 1e-6               # BASELINE.json north_star: FIR within 1e-6 of the CPU reference


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle-ms", type=float, default=250.0,
                    help="untimed back-to-back launches before the W warmup steps: after idle the\n"
                         "chip's power management first boosts, then overshoots downwards for some tens\n"
                         "of ms (kernel trace in profiles/) before it settles; the timed region should see\n"
                         "the settled, sustained-load clocks")
    ap.add_argument("--log2n", type=int, default=28, help="log2 complex samples per GPU per step")
    ap.add_argument("--workload", default="d8_127",
                    choices=["d8_127", "d8_255", "c320", "c320_fixture", "unpack", "api250k"],
                    help="d8_127 = BASELINE configs[1] (default); others are sweep points.  c320 = BASELINE config 3 with the\n"
                         "plan perseus_set_sampling_rate(250000) builds (what the drop-in API ships); c320_fixture = the same\n"
                         "shape with the committed fixture taps (tests/golden); api250k = the reference's own call sequence\n"
                         "(perseustest.c:188-404) through libperseus-sdr.so with the on-device source, in a C client")
    ap.add_argument("--api-batch-log2", type=int, default=0,
                    help="api250k: GPU batch size of the API stream (0 = the library's own choice: 2^24 for an unpaced on-device source)")
    ap.add_argument("--taps-fp16", action="store_true",
                    help="binary16 taps (BASELINE config 5's fp16 leg).  The 127-/255-tap workloads run on k_fir_i8x's plain\n"
                         "form, which then holds the taps on the device as binary16 (2 bytes a tap) and whose matrix waves\n"
                         "quantise them into their operand themselves; the vector kernels keep the binary16 VALUES in fp32 registers (their\n"
                         "taps sit in SGPRs in the hot loop: no storage to halve)")
    ap.add_argument("--gather", action="store_true", help="run the gather leg at N=1 too (1-rank RCCL group)")
    ap.add_argument("--no-gather", action="store_true", help="skip the gather leg at N>1")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--recheck-placement", action="store_true",
                    help="diagnostics: right after the timed region, time every probed output slot again (24 launches\n"
                         "each, the chosen one first and last): the line gains `placement_recheck_ms`")
    ap.add_argument("--per-step-events", action="store_true",
                    help="diagnostics: a HIP event after every timed step; the line gains `per_step_ms` (the extra\n"
                         "records cost about a microsecond per step, so this is not the default)")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle window check of the last output")
    ap.add_argument("--no-verify-all", action="store_true",
                    help="windows only: skip the every-output comparison of the last step (hosts with fewer than 16 cores skip it anyway)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--out-candidates", type=int, default=24,
                    help="1 = no placement search: input and output as hipMalloc hands them out (anything else: search\n"
                         "for a pair in different HBM extent classes, 0.340 instead of 0.367 ms,\n"
                         "profiles/r02/i_placement_map.txt)")
    ap.add_argument("--arena-gib", type=int, default=72,
                    help="size of the allocation the placement cuts its 8 GiB slots from (less if less is free): a quarter of\n"
                         "the 288 GB at most by default -- a receiver cannot spend more of its HBM on buying 1-8 %%; 72 GiB\n"
                         "hold the four offsets the rule probes (+8, +32, +48, +64 GiB)")
    ap.add_argument("--arena-rest-s", type=float, default=3.0,
                    help="pause after the arena has been allocated, before the first launch into it.  During the first\n"
                         "second or so behind an 80 GiB allocation the chip sometimes (one process in five) runs every\n"
                         "stream 4-5 %% slower for some tenths of a second, whatever its placement -- what the driver\n"
                         "does to freshly handed-out memory, most likely -- and a 20-step timed region can fall into\n"
                         "that; with the pause none of thirty processes did (profiles/r03/n_slow_state_investigation.txt).\n"
                         "A receiver allocates once and streams for hours; 0 switches the pause off")
    ap.add_argument("--arena-grow-gib", type=int, default=0,
                    help="rule placement: if every slot of the first arena runs at the first-come speed (one extent class\n"
                         "over all of it), allocate this much instead (less if less is free) and look again; 0 (default\n"
                         "since round 5): never -- such a layout is reported as it is, `value` = the first-come speed")
    ap.add_argument("--placement", default="rule", choices=["rule", "full"],
                    help="rule: input at the start of the arena, output probed at +8 (first come), +32, +48, +64 GiB;\n"
                         "full: three input slots x every output slot (the map; use --arena-gib 192)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="cascades: run the stage behind the fused pair in line, as a kernel of its own, instead of holding it\n"
                         "back as extra thread blocks of the NEXT batch's first-stage launch (pddc_pipeline_set_overlap)")
    ap.add_argument("--gather-timeout", type=float, default=240.0,
                    help="watchdog for the gather leg: past this the line is printed without it")
    return ap.parse_args(argv)


def load_taps(name):
    import numpy as np
    return np.fromfile(os.path.join(ROOT, "tests", "golden", f"taps_{name}.f32"), dtype=np.float32)


# ----------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` as a plain command
# ----------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv, timeout_s=3000.0):
    """Start the N ranks as children of a parent that has made no GPU call (never re-exec a
    process that has touched the GPU), relay rank 0's JSON line, return the worst status."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # (the children run the script this process was started as: bench.py -- or a test's wrapper around it)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(sys.argv[0] if sys.argv and sys.argv[0].endswith(".py")
                                                                        else __file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + timeout_s
    failed = None
    while time.time() < deadline:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:                          # one rank died: the others would sit in rendezvous or a collective until
            failed = bad[0]              # their own timeouts, holding their GPUs and arenas
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.05)
    worst = 0
    for p in procs:
        if p.poll() is None:
            if failed is None and time.time() < deadline:
                continue
            p.terminate()                # the exact children we started
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    for p in procs:
        p.wait()
        if p.returncode != 0:
            worst = p.returncode if worst == 0 or p.returncode > 0 else worst
    if failed is not None and worst == 0:
        worst = failed
    reader.join(timeout=5)
    out0 = out0[0] if out0 else ""
    line = None
    for ln in (out0 or "").splitlines():
        t = ln.strip()
        if t.startswith("{") and t.endswith("}"):
            line = t
        elif t:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    elif worst == 0:
        worst = 1
    return worst


# ----------------------------------------------------------------------------------------
# CPU leg
# ----------------------------------------------------------------------------------------
def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(workload, seconds):
    """Oracle float path on the host cores, bounded sample (kind: port)."""
    import numpy as np
    from oracle import oracle as O
    O.build()
    threads = O.max_threads()
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "perseustest_ref")
    if workload == "unpack" and os.path.exists(ref_bin):
        # the reference's own client (compiled from its sources, oracle/Makefile `ref`):
        # user_data_callback_c_f with its per-sample fwrite, fed unpaced by the drop-in library
        import re
        env = dict(os.environ, PERSEUS_AMD_PACE="0", PERSEUS_AMD_SOURCE="zero", PERSEUS_AMD_MODE="wire")
        t = max(1, int(round(min(seconds, 5.0))))
        p = subprocess.run([ref_bin, "-a", "-t", str(t), "-p", "-s", "2000000", "-d", "3", "-o", "/dev/null"],
                           env=env, capture_output=True, text=True, timeout=120)
        m = re.search(r"Rate: ([0-9.]+) kS/s", p.stderr)
        if m:
            return {"value": round(float(m.group(1)) / 1e3, 2), "unit": "MS/s", "cores": 1, "kind": "reference",
                    "cpu_model": cpu_model(),
                    "sample": f"examples/perseustest.c user_data_callback_c_f (6144-byte buffers, per-sample fwrite to "
                              f"/dev/null) driven unpaced by libperseus-sdr.so for {t} s"}
    single = None
    if workload == "unpack":
        run = lambda b: O.unpack24_f32(b)
        label = "24-bit unpack only, 1 thread (reference callback style)"
        threads = 1
    else:
        w = workload_def(workload if workload in ("c320", "c320_fixture", "d8_255") else "d8_127")
        if workload in ("c320", "c320_fixture"):
            # all cores: one single-threaded streaming chain per core over its own 2^25 / cores samples -- independent
            # receivers, which is how the reference scales (perseus-sdr.c:43-47: eight descriptors, one callback each)
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(max_workers=threads)

            def run(b):
                per = (b.size // 6 // threads) // 1024 * 1024 * 6
                parts = [b[i * per:(i + 1) * per] for i in range(threads)]
                list(pool.map(lambda q: O.stream_callback_style(q, w["stages"], freg=w["freg"], mix=True, buf_bytes=6144), parts))
            label = ("unpack + NCO 7.1 MHz + cascade /320 (8*8*5): one single-threaded streaming float chain per core, each over "
                     "its own slice in 6144-byte callbacks")
        else:
            h = w["stages"][0][1]
            run = lambda b: O.stage1_f32(b, h, 8, threads)
            label = f"unpack (byte shuffles, 4 samples a step) + {h.size}-tap decimate-by-8, float accumulate, OpenMP"
        # the way the reference itself would run the path -- one thread, the stream arriving in 6144-byte
        # callbacks (perseus-in.c:206-207), each unpacked as examples/perseustest.c:466-502 does, streaming float FIR stages
        n1 = 1 << 22
        b1 = O.lcg_bytes(6 * n1, 12345)
        one = lambda: O.stream_callback_style(b1, w["stages"], freg=w["freg"], mix=w["mix"], buf_bytes=6144)
        one()
        p1, t1 = 0, time.perf_counter()
        while True:
            one()
            p1 += 1
            d1 = time.perf_counter() - t1
            if d1 >= max(2.0, seconds / 4) or p1 >= 4096:
                break
        single = {"value": round(n1 * p1 / d1 / 1e6, 2), "unit": "MS/s", "cores": 1,
                  "sample": f"{p1} passes over 2^22 samples, ONE thread, 6144-byte callbacks (perseus-in.c:206-207) unpacked as "
                            f"examples/perseustest.c:466-502 does and pushed through streaming float FIR stages "
                            f"(oracle/perseus_oracle.c orc_stream_f32_callback_style), {d1:.1f} s"}
    n = 1 << 25                                # 2^25 samples (192 MiB packed) per pass
    buf = O.lcg_bytes(6 * n, 12345)
    run(buf)                                   # warm (page-in, omp pool)
    passes, t0 = 0, time.perf_counter()
    while True:                                # repeat passes until ~`seconds` of CPU work
        run(buf)
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or passes >= 4096:
            break
    n_big = n * passes
    out = {"value": round(n_big / dt / 1e6, 2), "unit": "MS/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
           "sample": f"{passes} passes over 2^25 samples of the same LCG stream ({label}), {dt:.1f} s"}
    if single is not None:
        out["single_thread"] = single          # the reference's own shape of the work; `value` above is all cores
    return out


def _baseline_metric():
    """The metric string is BASELINE.json's, verbatim."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "input MS/s through unpack+decimate, 1/2/4/8 GPU; % HBM-roofline"


BASELINE_METRIC = _baseline_metric()


def workload_def(name, pkg=None):
    """-> dict(stages, mix, freg, bytes_per_sample, decim, label)."""
    if name in ("c320", "c320_fixture"):
        if name == "c320":
            # ONE x320 filter set, benchmarked and shipped: the plan the drop-in API builds for the reference's 250 kS/s
            # setting (perseus_set_sampling_rate(250000), perseus-sdr.c:776-892; taps: csrc/plan_taps.inc)
            if pkg is None:
                pkg = importlib.import_module("libperseus-sdr_amd")
            stages = [(d, t) for d, t, _l in pkg.api_plan(250000)]
            origin = "the drop-in API's 250 kS/s plan"
        else:
            stages = [(8, load_taps("c320_s1_d8_32")), (8, load_taps("c320_s2_d8_64")), (5, load_taps("c320_s3_d5_161"))]
            origin = "fixture taps"
        nt = "/".join(str(len(t)) for _d, t in stages)
        return {"stages": stages, "ntaps": [int(len(t)) for _d, t in stages],
                "mix": True, "freg": 381178347,        # 7.1 MHz at 80 MHz (perseus-sdr.c:584)
                "bytes_per_sample": 6.0 + 8.0 / 320.0, "decim": 320, "kernel_bytes_per_sample": 6.125,
                "label": f"80 MS/s synthetic 24-bit I/Q, NCO mix 7.1 MHz + cascade /320 (8*8*5, {nt} taps: {origin})"}
    if name == "unpack":
        return {"stages": None, "mix": False, "freg": 0, "bytes_per_sample": 14.0, "decim": 1,
                "kernel_bytes_per_sample": 14.0, "label": "24-bit packed I/Q -> float32 unpack only"}
    h = load_taps(name)
    return {"stages": [(8, h)], "ntaps": [int(h.size)], "mix": False, "freg": 0, "bytes_per_sample": 7.0, "decim": 8,
            "kernel_bytes_per_sample": 7.0,
            "label": f"80 MS/s synthetic 24-bit I/Q, unpack + {h.size}-tap polyphase decimate-by-8"}


# ----------------------------------------------------------------------------------------
# parity check of an output batch against the oracle, on windows (outside the timed region)
# ----------------------------------------------------------------------------------------
def window_positions(n_first, n_count, sched, dtot, width, k_random, seed):
    """Batch-relative output indices where windows start: the scheduler's seams + random."""
    import numpy as np
    pos = {0, max(0, n_count - width)}
    if sched:
        per_tile = sched["tile"] / dtot                      # outputs per stage-0 tile
        S, nb, nt = sched["S"], sched["nblocks"], sched["ntiles"]
        K = sched["K"]
        if S < 0:                                            # round-robin walk: chunk j -> block j mod nb, dynamic from chunk -S on
            seams = [K, 2 * K, nb * K, (nb + 1) * K, -S * K, (-S + 1) * K, nt - 1]
        else:
            seams = [nb * S, nb * S + K, nt - 1]             # first dynamic chunk, the next one, last tile
        if S > 0:
            seams += [S, (nb // 2) * S, (nb - 1) * S]        # block-range seams of the static part
        for t in seams:
            o = int(t * per_tile) - width // 2
            pos.add(min(max(0, o), max(0, n_count - width)))
    rng = np.random.default_rng(seed)
    for _ in range(k_random):
        pos.add(int(rng.integers(0, max(1, n_count - width))))
    return sorted(pos)


def verify_last_output(O, shard, fetch_bytes, out_np_fn, ns, wl, n0_last, sched, k_random=20, seed=2026):
    """Compare windows of the last step's output with the oracle.

    The pipeline saw the same `ns`-sample batch again and again, i.e. a periodic stream;
    the last batch starts at absolute sample n0_last.  Absolute output M of the cascade is
    produced in the batch that contains ADC sample dtot*M.  For a window of absolute outputs
    [M0, M1) the oracle runs from zero history over ADC samples [dtot*M0 - halo, dtot*M1)
    (NCO phase from the absolute index) and drops the halo's outputs.
    fetch_bytes(a, b) -> packed bytes of absolute samples [a, b); out_np_fn(j0, j1) -> the
    batch's outputs [j0, j1) as float32 [n, 2]."""
    import numpy as np
    stages, dtot = wl["stages"], wl["decim"]
    halo = shard.cascade_halo(stages)
    m_first = -(-n0_last // dtot)
    m_end = -(-(n0_last + ns) // dtot)
    n_count = m_end - m_first
    width = 256 if dtot <= 8 else 64
    worst, nwin = 0.0, 0
    for j0 in window_positions(m_first, n_count, sched, dtot, width, k_random, seed):
        j1 = min(j0 + width, n_count)
        M0, M1 = m_first + j0, m_first + j1
        a = max(dtot * M0 - halo, 0)                 # < 0 only in the very first batch: zero history
        seg = fetch_bytes(a, dtot * M1)
        x = O.unpack24_f32(seg)
        x = O.nco_mix(x, wl["freg"], a) if wl["mix"] else x.astype(np.float64)
        for st in stages:
            x = O.fir_decim(x, st[1], int(st[0]))
        ref = np.asarray(x, dtype=np.float64).reshape(-1, 2)[(dtot * M0 - a) // dtot:]
        got = out_np_fn(j0, j1).astype(np.float64)
        ref = ref[:got.shape[0]]
        den = np.abs(ref).max()
        err = float(np.abs(got - ref).max() / den) if den > 0 else float(np.abs(got).max())
        worst = max(worst, err)
        nwin += 1
    return {"windows": nwin, "window_outputs": width, "max_rel_err": float(f"{worst:.3e}"), "tol": PARITY_TOL,
            "ok": bool(worst <= PARITY_TOL), "n_outputs": int(n_count), "metric": PARITY_METRIC,
            "of": "last timed step's output, windows on the tile scheduler's seams + %d random" % k_random}


def traffic_from_profile(workload, kernel_sig, log2n, taps_fp16, kernel_in_use=None):
    """HBM bytes per launch from the committed offline PMC passes -- reported only when the
    file says it was measured for THIS kernel source and launch shape; otherwise null."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(path))
    except Exception:
        return None, "no profiles/pmc_traffic.json"
    key = workload + ("_fp16" if taps_fp16 else "")              # the binary16-stored leg has a PMC pass of its own
    ent = d.get(key)
    prov = d.get("provenance", {})
    if not isinstance(ent, (int, float)):
        return None, "workload not in profiles/pmc_traffic.json"
    if not prov:
        return None, "profiles/pmc_traffic.json carries no provenance"
    if prov.get("log2n") != log2n:
        return None, "offline PMC was taken at another launch shape"
    if prov.get("kernel_source_sha16") != kernel_sig:
        return None, ("stale: offline PMC was taken for kernel source %s, this build is %s"
                      % (prov.get("kernel_source_sha16"), kernel_sig))
    measured = (prov.get("kernels") or {}).get(key)
    if kernel_in_use is not None and measured is not None and not kernel_in_use.startswith(measured):
        return None, "stale: offline PMC was taken on kernel %s, this run used %s" % (measured, kernel_in_use.split(" ")[0])
    return float(ent), ("offline rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic.sh), commit %s, "
                        "kernel source %s" % (prov.get("commit", "?"), kernel_sig))


def kernel_source_sig():
    import hashlib
    h = hashlib.sha256()
    for f in ("ddc_kernels.hip", "ddc_kernels.h", "fir8_block.inc", "ddc_fir_i8.hip", "ddc_dev.h"):
        h.update(open(os.path.join(ROOT, "libperseus-sdr_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def arena_plan(free_bytes, in_bytes, out_bytes, ws_bytes, arena_gib):
    """Sizes of the placement arena (pure arithmetic, tested on CPU for the 8-rank shape): slots of at least 8 GiB, each
    holding the input's span (whole GiB), a cascade's workspace and the output; the arena takes what `arena_gib` asks
    for, leaving 32 GiB of the device alone.  search: False for launches small enough to live in the last-level cache."""
    in_span = -(-in_bytes // (1 << 30)) << 30
    ws_span = -(-ws_bytes // (2 << 20)) * (2 << 20)
    slot = max(8 << 30, -(-(in_span + ws_span + out_bytes) // (1 << 30)) << 30)
    gib = max(0, min(arena_gib, (free_bytes - (32 << 30)) >> 30))
    return {"in_span": in_span, "ws_span": ws_span, "slot": slot, "gib": gib, "nslot": (gib << 30) // slot,
            "search": out_bytes + ws_bytes >= (32 << 20)}


# ----------------------------------------------------------------------------------------
def wall_budget(world, steps, warmup, log2n, arena_rest_s=3.0, settle_ms=250.0, verify_all=True, gather=True):
    """What a run of this bench SHOULD take on the wall, by phase, in seconds -- so that the first real 8-GPU run has a
    figure to be boring against (round 5 review, item 7).  Constants: a 2^28-sample step 0.31 ms, scaled with the batch;
    process start with torch + RCCL 25 s (a fresh box pages the image in: up to 2 min more); an arena of 72 GiB allocates in
    about 1 s; 582 probe launches; the every-output check of one 2^28-sample batch about 3 s of 128 threads, shared by
    the ranks of a node; a gather step is bound by ONE xGMI link per peer (~153 GB/s nominal, 100 assumed here) for the
    decimate-by-8 output, by the kernel for the /320 cascade; each gather leg runs 200 + W untimed and K timed steps."""
    step = 0.31e-3 * 2.0 ** (log2n - 28)
    out8 = 8.0 * 2.0 ** log2n / 8.0
    b = {"start_import_rendezvous": 25.0, "arena_alloc_and_rest": 1.0 + arena_rest_s, "placement_probes": 582 * step + 0.2,
         "settle_warmup_timed": settle_ms * 1e-3 + (8 + warmup + steps) * step,
         "verify": (1.5 + (3.0 * world if verify_all else 0.3)) * 2.0 ** (log2n - 28)}
    if gather and world > 1:
        link = max(step, out8 / 100e9)
        b["gather_leg_this_workload"] = (200 + warmup + steps) * link + 1.0
        b["gather_leg_c320"] = (200 + warmup + steps) * max(step, out8 / 40 / 100e9) + 2.0
    if world == 1:
        b["cpu_baseline"] = 16.0
    b["total"] = round(sum(b.values()), 1)
    return {k: round(v, 2) for k, v in b.items()}


def run_rank(a):
    t_phase = [time.perf_counter()]
    phases = {}

    def phase(name):                      # wall seconds of the phase that just ended (this rank's host clock)
        now = time.perf_counter()
        phases[name] = round(phases.get(name, 0.0) + now - t_phase[0], 2)
        t_phase[0] = now
    import numpy as np
    import torch
    pkg = importlib.import_module("libperseus-sdr_amd")
    shard = importlib.import_module("libperseus-sdr_amd.shard")

    rank, world, local = shard.env_rank_world()
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_dist = os.environ.get("PDDC_BENCH_FORCE_DIST") == "1" or a.gather      # 1-rank RCCL group
    grp = shard.RcclGroup(pkg, rank, world, local) if (world > 1 or force_dist) else shard.Group()

    ns = 1 << a.log2n
    wl = workload_def(a.workload)
    stages = wl["stages"]
    inbox = [pkg.synth_lcg(6 * ns, shard.stream_seed(rank), 0, dev)]   # device resident before timing
    stream = torch.cuda.current_stream(dev).cuda_stream
    calls = [0]
    if stages is not None:
        # configuration (taps, NCO word, plan) comes from rank 0 over RCCL: a few KB, once
        pipe = grp.make_pipeline(pkg, stages, wl["freg"], wl["mix"], a.taps_fp16)
        # a cascade behind the fused pair: its tail (1/64 of the samples) is held back and rides along with the NEXT step's
        # first-stage launch as extra thread blocks; the K timed steps end with a fence, so all K tails are inside the timed region
        overlap = len(stages) > 2 and pipe.fused_pair(ns) and not a.no_overlap
        if overlap:
            pipe.set_overlap(True)
        out_rows = pipe.max_output(ns) + 8
        outbox = [torch.empty((out_rows, 2), dtype=torch.float32, device=dev)]

        def step():
            calls[0] += 1
            return pipe.process_ptr(inbox[0].data_ptr(), ns, outbox[0].data_ptr(), out_rows, stream)
    else:
        pipe = None
        overlap = False
        out_rows = ns
        outbox = [torch.empty((ns, 2), dtype=torch.float32, device=dev)]

        def step():
            calls[0] += 1
            pkg.check(pkg.ddc_lib().pddc_unpack24_f32(inbox[0].data_ptr(), ns, outbox[0].data_ptr(), stream))
            return ns

    # Where the buffers lie in HBM matters to this read/write stream.  HBM is laid out in a few classes of large
    # extents (tens of GiB each; three classes seen): with the input and the output in extents of the SAME class the
    # kernel takes 0.367 ms, in DIFFERENT classes 0.340 ms, on every box, for every pair tried (maps in
    # profiles/r02/i_placement_map.txt, tools/placement_probe.py --mode map) -- streams that share a class get in each
    # other's way (two write streams even more: tools/ubench/stream_classes.hip).  Buffers allocated one after the
    # other usually land in the same extent, and separate allocations 8 GiB apart do not reliably leave it (one
    # process saw a single class over 200 GiB of them).  Inside ONE large allocation the classes alternate every
    # 32-64 GiB in every process tried (tools/placement_probe.py --mode matrix), so: one arena (--arena-gib, default 72 = a quarter of the HBM), cut
    # into 8 GiB slots, the input at its start.  The rule (profiles/r03/f_placement_rule.txt; the library's own form is
    # pddc_pipeline_arena_place): the slot right behind the input is always in the input's class ("first come"), one of the slots
    # at +32 / +48 / +64 GiB nearly always in another (every slot is looked at when none of them gains 3 %) -- four probes of 24 back-to-back steps, the fastest kept, the first-come
    # time reported next to it.  --placement full scans every slot for three input places (1.5 s; what round 2 did).
    # A receiver allocates once and runs for hours; 288 GB of HBM make this affordable.
    placement = None
    phase("setup")
    arena = None
    in_bytes, out_bytes = 6 * ns, out_rows * 8
    # a cascade's inter-stage buffers come from the arena too (pddc_pipeline_set_workspace): the fused pair of the
    # x320 cascade writes 1/48 of what it reads and is as sensitive to where that goes as the single stage is
    ws_bytes = pipe.workspace_size(ns) if (stages is not None and len(stages) > 1) else 0
    plan = arena_plan(0, in_bytes, out_bytes, ws_bytes, a.arena_gib)
    in_span, ws_span, slot = plan["in_span"], plan["ws_span"], plan["slot"]
    if a.out_candidates > 1 and plan["search"]:
        inbox[0] = outbox[0] = None                               # the first-come pair goes back before the arena is made
        torch.cuda.empty_cache()
        free_b, _ = torch.cuda.mem_get_info(dev)
        plan = arena_plan(free_b, in_bytes, out_bytes, ws_bytes, a.arena_gib)
        gib = plan["gib"]
        while gib << 30 >= 3 * slot:
            try:
                arena = torch.empty(gib << 30, dtype=torch.uint8, device=dev)
                break
            except RuntimeError:
                gib = gib * 3 // 4
        if arena is not None and a.arena_rest_s > 0:
            torch.cuda.synchronize(dev)
            time.sleep(a.arena_rest_s)
        if arena is None:                                         # no room: first come, first served
            inbox[0] = pkg.synth_lcg(in_bytes, shard.stream_seed(rank), 0, dev)
            outbox[0] = torch.empty((out_rows, 2), dtype=torch.float32, device=dev)
    if arena is not None:
        nslot = (gib << 30) // slot

        def in_view(k):
            return arena[k * slot:k * slot + in_bytes]

        def out_view(k):
            if ws_bytes:                                          # the workspace goes where the output goes
                pipe.fence(stream)                                # (a tail still held back reads the present one)
                pipe.set_workspace(arena[k * slot + in_span:].data_ptr(), ws_bytes, ns)
            at = k * slot + in_span + ws_span
            return arena[at:at + out_bytes].view(torch.float32).view(out_rows, 2)

        full = a.placement == "full"
        in_slots = sorted({0, nslot // 3, 2 * nslot // 3}) if full else [0]
        for k in in_slots:                                        # the same LCG bytes in every input candidate
            pkg.check(pkg.ddc_lib().pddc_synth_lcg(in_view(k).data_ptr(), in_bytes, shard.stream_seed(rank), 0, stream))
        inbox[0], outbox[0] = in_view(0), out_view(0)
        for _ in range(150):                                      # the first pair is not to be measured on cold clocks
            step()
        table, launches = {}, [150]
        grown = None

        def probe(i, o):
            inbox[0], outbox[0] = in_view(i), out_view(o)
            for _ in range(30):
                step()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(24):
                step()
            e1.record()
            e1.synchronize()
            if overlap:
                pipe.fence(stream)
            launches[0] += 54
            table[(i, o)] = e0.elapsed_time(e1) / 24
            return table[(i, o)]

        if full:                                                  # the map: every pair (profiles/r02/k_arena_map.txt)
            for i in in_slots:
                for o in range(nslot):
                    probe(i, o)
        else:
            # The rule read off those maps (some twenty leases, profiles/r0[23]/*placement*, NOTEBOOK.md rounds 1-3 5 (u)): with the
            # input at the START of one allocation, the extent class it lies in reaches 32, 48 or 64 GiB up; the slot
            # right behind the input is nearly always in it ("first come"), and +32, +48 or +64 GiB nearly always in another one.
            # So: four probes.  Only if none of them gains (a workload that does not care, or a layout not seen yet)
            # the other slots are looked at too.
            def rule_scan():
                # four probes settle it when one of them is clearly in another class (the vector kernels lose 8-12 % in
                # the input's class); a smaller gain may be a THIRD class or the matrix-core kernel, whose classes are
                # only 3 % apart: then every slot is looked at (nine probes, 0.15 s)
                first = probe(0, 1 if nslot > 1 else 0)
                for o in (4, 6, 8):
                    if o < nslot:
                        probe(0, o)
                if min(table.values()) > 0.93 * first:
                    for o in range(2, nslot):
                        if (0, o) not in table:
                            probe(0, o)
                return first

            first_come = rule_scan()
            # One class over the whole arena (about one process in five draws such a layout for 80 GiB): a LARGER
            # allocation is laid out anew and has shown at least two classes every time (192-200 GiB arenas, round 2).
            # The memory is there (288 GB); the receiver allocates once.
            if min(table.values()) > 0.985 * first_come and a.arena_grow_gib > gib:
                small = {"arena_GiB": gib, "step_ms": {str(o): round(v, 4) for (_, o), v in sorted(table.items())}}
                inbox[0] = outbox[0] = None
                arena = None
                torch.cuda.empty_cache()
                free_b, _ = torch.cuda.mem_get_info(dev)
                big = min((free_b - (16 << 30)) >> 30, a.arena_grow_gib) // 8 * 8
                want = big if big >= gib + 32 else gib
                try:
                    arena = torch.empty(want << 30, dtype=torch.uint8, device=dev)
                except RuntimeError:
                    want = gib
                    arena = torch.empty(want << 30, dtype=torch.uint8, device=dev)
                if a.arena_rest_s > 0:
                    torch.cuda.synchronize(dev)
                    time.sleep(a.arena_rest_s)
                gib, nslot = want, (want << 30) // slot
                pkg.check(pkg.ddc_lib().pddc_synth_lcg(in_view(0).data_ptr(), in_bytes, shard.stream_seed(rank), 0, stream))
                table.clear()
                first_come = rule_scan()
                grown = small
        (bi, bo), best_ms = min(table.items(), key=lambda kv: kv[1])
        inbox[0], outbox[0] = in_view(bi), out_view(bo)
        placement = {"arena_GiB": gib, "slot_GiB": slot >> 30, "output_slots": nslot, "mode": a.placement,
                     "step_ms_by_input_slot": {f"in@{(i * slot) >> 30}GiB": {str(o): round(table[(i, o)], 4)
                                                                             for o in range(nslot) if (i, o) in table}
                                               for i in in_slots},
                     "first_come_ms": round(table.get((0, 1), table.get((0, 0))), 4),
                     "chosen": {"input_at_GiB": (bi * slot) >> 30, "output_slot": bo, "ms": round(best_ms, 4)},
                     "probe_pairs": len(table), "probe_launches": launches[0], "arena_rest_s": a.arena_rest_s,
                     "grown_after": grown,     # not None: the first, smaller arena showed ONE class only; its table
                     "note": "input and output (with a cascade's inter-stage workspace) cut from ONE allocation and placed in "
                             "different HBM extent classes: input at the start, the output probed right behind it (first "
                             "come: same class) and at +32 / +48 / +64 GiB (one of them is nearly always another class; if none gains 3 % every slot is probed); every "
                             "probed pair's step time is listed (key = output slot)"}
    out = outbox[0]
    d_in = inbox[0]

    phase("placement")
    # untimed settle: sustained-load clocks, not boost.  Eight calibration steps, then settle_ms worth of
    # launches queued back to back with NO host synchronisation in between (every idle gap, however short,
    # lets the power management raise the clock again for the next few milliseconds), running straight
    # into the W warm-up steps
    t_cal = time.perf_counter()
    for _ in range(8):
        step()
    torch.cuda.synchronize(dev)
    ms_est = max((time.perf_counter() - t_cal) * 1e3 / 8, 1e-3)
    for _ in range(min(int(a.settle_ms / ms_est) + 1, 100000)):
        step()
    for _ in range(a.warmup):
        step()
    # everything the timed region needs is made ready BEFORE the contract's synchronize + barrier, so that the GPU idles
    # no longer than those two take (the first timed step pays for the pause: 0.325-0.40 instead of 0.322 ms)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    step_evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps)] if a.per_step_events else None
    # the dominant kernel's duration is taken over THIS region: for a cascade the library brackets its stage-0
    # (fused-pair) kernel with HIP events on this stream; a single-kernel step needs nothing but ev0/ev1
    cascade = stages is not None and len(stages) > 2 and pipe.fused_cascade(ns)     # the whole cascade is ONE kernel
    # (overlap mode: the step is ONE launch -- the pair with the previous step's tail as extra blocks -- so ev0/ev1 do)
    multi_kernel = stages is not None and pipe.fused and len(stages) > 1 and not cascade and not overlap
    for e in [ev0, ev1] + (step_evs or []):             # torch makes the HIP event at the first record(): 10-100 us that would
        e.record()                                      # otherwise fall between t0 and the first timed launch
    ev_w = torch.cuda.Event()
    ev_w.record()
    while not ev_w.query():                             # polled, so that the host is back within microseconds of the last
        pass                                            # warm-up step's end (a blocking wait adds its wake-up latency to the idle gap)
    torch.cuda.synchronize(dev)
    t_idle0 = time.perf_counter()                       # the GPU is idle from here to the first timed launch
    grp.barrier()
    if multi_kernel:
        pipe.time_stage0_inline(True)
    t0 = time.perf_counter()
    idle_ms = (t0 - t_idle0) * 1e3
    ev0.record()
    marks = [time.perf_counter()]                       # host clock: after the first event, ...
    n_last = 0
    for k in range(a.steps):
        n_last = step()
        if step_evs:
            step_evs[k].record()
        if k == 0:
            marks.append(time.perf_counter())           # ... after the first launch call, ...
    if overlap:
        pipe.fence(stream)                              # the last steps' tails belong to the timed region
    ev1.record()
    marks.append(time.perf_counter())                   # ... after the last one, ...
    while not ev1.query():                              # the host waits by polling: a blocking synchronize adds its wake-up
        pass                                            # latency (30-200 us on these hosts) to a 6.5 ms region
    marks.append(time.perf_counter())                   # ... when the poll sees the end, ...
    torch.cuda.synchronize(dev)
    grp.barrier()
    dt = time.perf_counter() - t0
    marks.append(t0 + dt)                               # ... and behind synchronize + barrier
    ev_ms = ev0.elapsed_time(ev1)                       # HIP events on the launch stream
    # dominant-kernel duration, HIP events on the launch stream over the timed region itself (a separate
    # region after a host synchronisation would see other clocks: a pause of a few hundred microseconds buys
    # ~8 % faster kernels for the next milliseconds on this chip)
    if multi_kernel:
        kern_ms, n_k = pipe.stage0_time()           # the library keeps at most 8192 event pairs
        pipe.time_stage0_inline(False)
        assert n_k == min(a.steps, 8192), (n_k, a.steps)
    else:
        kern_ms = ev_ms / a.steps
    recheck = None
    if a.recheck_placement and placement is not None and not overlap:
        def again(o):
            outbox[0] = out_view(o)
            for _ in range(10):
                step()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(24):
                step()
            e1.record()
            e1.synchronize()
            return round(e0.elapsed_time(e1) / 24, 4)
        order = [bo] + [o for (i, o) in sorted(table) if i == bi and o != bo] + [bo]
        recheck = [(o, again(o)) for o in order]
        outbox[0] = out_view(bo)
    # measured copy ceiling: a device-to-device copy that moves as many bytes through HBM as
    # one launch of the dominant kernel does (nbytes read + nbytes written)
    copy_gbps = None
    try:
        import ctypes
        nb = int(wl["kernel_bytes_per_sample"] * ns / 2) // 256 * 256
        if arena is not None and nb <= slot:
            # under the same favourable placement as the kernel: source = the chosen input slot, destination = the
            # fastest of the other slots (never the one that holds the output still to be verified)
            src_ptr = arena[bi * slot:bi * slot + nb].data_ptr()
            quick = {o: pkg.measure_copy(arena[o * slot:o * slot + nb].data_ptr(), src_ptr, nb, 6, stream)
                     for o in range(nslot) if o not in (bi, bo)}
            od = min(quick, key=quick.get)
            ms = pkg.measure_copy(arena[od * slot:od * slot + nb].data_ptr(), src_ptr, nb, 40, stream)
        else:
            src = d_in[:nb] if nb <= d_in.numel() else torch.empty(nb, dtype=torch.uint8, device=dev)
            # destination in another HBM extent class than the source (pddc_malloc_apart)
            dst = ctypes.c_void_p()
            pkg.check(pkg.ddc_lib().pddc_malloc_apart(ctypes.byref(dst), nb, src.data_ptr(), nb, 24, None, None))
            ms = pkg.measure_copy(dst.value, src.data_ptr(), nb, 40, stream)
            pkg.check(pkg.ddc_lib().pddc_free(dst))
            del src
        copy_gbps = 2.0 * nb / (ms * 1e-3) / 1e9
    except Exception as e:                              # never lose the line over the extra figure
        print(f"[bench] copy ceiling failed: {e}", file=sys.stderr)

    dt_max = grp.max_seconds(dt)
    if pipe is not None:
        pipe.check(stream)                              # a kernel-side failure flag (fused cascade) fails the run

    phase("settle_warmup_timed")
    # ---- parity of what was just timed (outside the timed region, every rank its own stream)
    verified = None
    if not a.no_verify:
        if world > 1:                                   # the checker's OpenMP team: this rank's share of the node's cores
            os.environ.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // world)))
        from oracle import oracle as O
        if rank == 0:
            O.build()                                   # one rank (re)builds the checker, the others wait for it
        grp.barrier()

        def fetch(a0, b0):                              # absolute samples of the periodic stream
            idx = (torch.arange(6 * a0, 6 * b0, device=dev, dtype=torch.int64) % (6 * ns))
            return d_in[idx].cpu().numpy()

        if stages is not None:
            n0_last = (calls[0] - 1) * ns
            sched = pipe.schedule(ns) if pipe.fused else None
            wl_v = wl
            if a.taps_fp16:                             # the oracle gets the taps the pipeline really uses: binary16 values
                wl_v = dict(wl)
                wl_v["stages"] = [(st[0], np.asarray(st[1], np.float32).astype(np.float16).astype(np.float32)) + tuple(st[2:])
                                  for st in wl["stages"]]
            verified = verify_last_output(O, shard, fetch, lambda j0, j1: out[j0:j1].cpu().numpy(), ns, wl_v, n0_last,
                                          sched)
            verified["n_outputs_ok"] = bool(n_last == verified["n_outputs"])
            verified["ok"] = bool(verified["ok"] and verified["n_outputs_ok"])
            # what this parity is parity WITH: the unpack is pinned to the reference bit for bit; the NCO mix and
            # the FIR have no reference arithmetic (they live in FPGA bitstreams) -- the oracle is this repository's definition
            verified["pinned"] = ("unpack: bit-exact against the reference's compiled callbacks (tests/test_reference_client.py); "
                                  "NCO + FIR: UNPINNED -- no reference arithmetic exists, the double oracle (cross-checked against "
                                  "scipy.signal.upfirdn to 1e-12) is the definition")
            # ... and EVERY output of that step (oracle/perseus_oracle.c orc_chain_check: chunks with their halos on the host's
            # cores; 2^28 samples take a couple of seconds on 128 threads): sparse lane-level damage is what windows miss
            if not a.no_verify_all and (os.cpu_count() or 1) >= 16 and all(len(st) < 3 or not st[2] or int(st[2]) <= 1
                                                                           for st in wl_v["stages"]):
                t_all = time.perf_counter()
                ev = O.chain_check(d_in.cpu().numpy(), n0_last, ns, [(st[0], st[1]) for st in wl_v["stages"]],
                                   out[:n_last].cpu().numpy(), freg=wl_v["freg"], mix=wl_v["mix"], tol=PARITY_TOL)
                verified["every_output"] = {"compared": ev["n"], "bad": ev["n_bad"], "first_bad": ev["first_bad"],
                                            "max_rel_err": float(f"{ev['max_rel_err']:.3e}"),
                                            "worst_chunk_rel_err": float(f"{ev['worst_chunk_rel_err']:.3e}"),
                                            "chunk_outputs": ev["chunk_outputs"], "ok": bool(ev["ok"] and ev["n"] == n_last),
                                            "seconds": round(time.perf_counter() - t_all, 2),
                                            "rule": "an output is bad when |y - ref| > tol * (max |ref| of its chunk), or NaN"}
                verified["ok"] = bool(verified["ok"] and verified["every_output"]["ok"])
        else:                                           # unpack only: bit-exact windows
            rng = np.random.default_rng(2026)
            starts = [0, ns - 4096] + [int(v) for v in rng.integers(0, ns - 4096, 20)]
            bad = 0
            for s0 in starts:
                ref = O.unpack24_f32(fetch(s0, s0 + 4096)).view(np.uint32)
                got = out[s0:s0 + 4096].cpu().numpy().reshape(-1).view(np.uint32)
                bad += int((ref != got).sum())
            verified = {"windows": len(starts), "window_outputs": 4096, "mismatching_words": bad, "ok": bad == 0,
                        "metric": "bit-exact vs the CPU oracle (examples/perseustest.c:466-502)"}

    phase("verify")
    names = grp.all_gather_object(torch.cuda.get_device_name(dev))
    per_rank_ms = [float(v) * 1e3 / a.steps for v in grp.all_gather_object(dt)]
    oks = grp.all_gather_object(None if verified is None else bool(verified["ok"]))
    res = None
    if rank == 0:
        total_samples = world * ns * a.steps
        value = total_samples / dt_max / 1e6
        fused = stages is not None and pipe.fused
        # overlap mode: a launch is the pair of this batch plus the tail of the one before, i.e. one batch's worth of the
        # whole cascade: priced with the cascade's figure, not with the larger traffic it really moves
        bps = wl["bytes_per_sample"] if (cascade or overlap) else (wl["kernel_bytes_per_sample"] if fused
                                                                   else wl["bytes_per_sample"])
        achieved = bps * ns / (kern_ms * 1e-3) / 1e9
        step_achieved = wl["bytes_per_sample"] * ns / (dt_max / a.steps) / 1e9      # the whole step, gaps and tails included
        klabel = (("k_fir8 (fused cascade: all stages in one launch)" if cascade else
                   "k_fir_i8x (int8 matrix cores on the wire bytes%s, fused pair; the tail in line)" % (", NCO folded into the taps" if wl["mix"] else "") if pipe.fused_pair(ns) == 2
                   else "k_fir8 (fused pair + the previous batch's tail as extra blocks of the launch)" if overlap
                   else "k_fir_i8x (int8 matrix cores on the wire bytes%s)" % (", NCO folded into the taps" if wl["mix"] else "") if pipe.on_i8(ns) == 2
                   else "k_fir8") if fused else "k_unpack24" if stages is None else "pipeline")
        traffic, traffic_src = traffic_from_profile(a.workload, kernel_source_sig(), a.log2n, a.taps_fp16, klabel)
        if verified is not None:
            verified["all_ranks_ok"] = bool(all(o for o in oks))
        res = {
            "metric": BASELINE_METRIC,
            "value": round(value, 1), "unit": "MS/s",
            # what the same step does in the buffers as the allocator hands them out (no arena, no probing): from the
            # first-come probe of the placement (24 back-to-back steps at output slot 1, right behind the input)
            "value_first_come": (round(world * ns / (placement["first_come_ms"] * 1e-3) / 1e6, 1)
                                 if placement and placement.get("first_come_ms") else None),
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt_max / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("i8xi8->i32, f32 out" if (pipe is not None and pipe.on_i8(ns)) else "f32"), "data": "synthetic",
            "config": {"workload": wl["label"], "samples_per_gpu_per_step": ns, "ntaps": wl.get("ntaps"),
                       "input": "LCG bytes seed 12345+rank, device resident",
                       "sharding": "independent stream per GPU, no data-path collective",
                       "taps": (("binary16 STORAGE: 2 bytes a tap on the device, quantised into int8 digit planes by k_fir_i8x's matrix waves themselves (PDDC_F_TAPS_FP16)" if pipe is not None and pipe.on_i8(ns)
                                 else "binary16 VALUES held in fp32 registers (PDDC_F_TAPS_FP16)") if a.taps_fp16 else
                                ("fp32 values as four int8 digit planes, 2^-31 of the largest tap (k_fir_i8x)"
                                 if pipe is not None and pipe.on_i8(ns) else "fp32")),
                       "overlap": ("the stage behind the fused pair is carried by the next step's first-stage launch as extra "
                                   "thread blocks (pddc_pipeline_set_overlap); the timed region ends with a fence") if overlap else None},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "copy_ceiling_GBps": round(copy_gbps, 1) if copy_gbps else None,
                         "frac_of_copy_ceiling": round(achieved / copy_gbps, 4) if copy_gbps else None,
                         "kernel": klabel,
                         "kernel_ms": round(kern_ms, 4),
                         "algorithmic_bytes_per_sample": bps,
                         "step_achieved": round(step_achieved, 1), "step_frac": round(step_achieved / HBM_PEAK_GBS, 4),
                         "step_algorithmic_bytes_per_sample": wl["bytes_per_sample"]},
            "events_ms_per_step": round(ev_ms / a.steps, 4),
            "idle_before_timed_ms": round(idle_ms, 3),      # barrier + synchronize, as the contract asks: the clocks
                                                            # the first timed steps see depend on how long this was
            "host_marks_ms": [round((m - t0) * 1e3, 3) for m in marks],   # first event / first launch / last launch /
                                                                         # end seen by the poll / behind synchronize + barrier
            "placement_recheck_ms": recheck,
            "per_step_ms": ([round(x.elapsed_time(y), 4) for x, y in zip([ev0] + step_evs[:-1], step_evs)]
                            if step_evs else None),
            "placement": placement,
            "verified": verified,
            "ranks_seen": grp.comm_size(), "devices": names,
            "per_rank_ms_per_step": {"min": round(min(per_rank_ms), 4),
                                     "median": round(sorted(per_rank_ms)[len(per_rank_ms) // 2], 4),
                                     "max": round(max(per_rank_ms), 4)},
            "kernel_only_aggregate_MSps": round(world * ns / (kern_ms * 1e-3) / 1e6, 1),
            "collectives": "RCCL called from the C library (pddc_comm_*); torch.distributed = rendezvous only"
                           if grp.comm is not None else
                           ("gloo control plane only (plan broadcast, barrier, MAX of the step time; the data path has "
                            "no collective): the RCCL communicator could not be made -- " + grp.comm_error)
                           if grp.comm_error else None,
            "cpu_baseline": None,
        }
        PARTIAL["res"] = res

    # ---- BASELINE config 4: gather of every rank's output on rank 0's GPU (RCCL, C library)
    if grp.comm is not None and stages is not None and not a.no_gather:
        out2 = spare = None
        if arena is not None and in_span + ws_span + 2 * out_bytes + 256 <= slot:
            at = bo * slot + in_span + ws_span + ((out_bytes + 255) & ~255)      # right behind `out`, same slot
            out2 = arena[at:at + out_bytes].view(torch.float32).view(out_rows, 2)
            at2 = (at + out_bytes + 255) & ~255
            spare = arena[at2:(bo + 1) * slot]                                   # the rest of that slot
        g = guarded_gather_legs(a, pkg, shard, grp, dev, stream, ns, d_in, wl, pipe, out, out2, spare)
        phase("gather_legs")
        if res is not None:
            res["gather"] = g
            # the three figures the 1 / 2 / 4 / 8-GPU question is about, side by side (round 5 review, item 7): the kernels
            # alone (what `value` is: no data-path collective), and the same step with every rank's output landing on rank 0
            tw, tc = g.get("this_workload") or {}, g.get("c320") or {}
            res["scaling_legs"] = {
                "kernel_only": {"value": res["value"], "unit": "MS/s", "ms_per_step": res["ms_per_step"]},
                "with_gather_" + ("c320" if a.workload.startswith("c320") else "d8"): {
                    "value": tw.get("value"), "unit": "MS/s", "ms_per_step": tw.get("ms_per_step"),
                    "per_link_GBps": tw.get("per_link_GBps"),
                    "expected": ("link-bound: every peer's %.0f MB per step cross ONE xGMI link to rank 0 (~153 GB/s nominal "
                                 "each, SURVEY.md 8e)" % (tw.get("out_bytes_per_rank_per_step", 0) / 1e6)) if tw else None},
                "with_gather_c320": ({"value": tc.get("value"), "unit": "MS/s", "ms_per_step": tc.get("ms_per_step"),
                                      "per_link_GBps": tc.get("per_link_GBps"),
                                      "expected": "kernel-bound: 6.7 MB per rank and step, the gather disappears under the next step's kernels"}
                                     if tc else None)}
    elif res is not None and grp.comm_error and stages is not None and not a.no_gather:
        res["gather"] = {"skipped": "the gather is RCCL by definition and no communicator exists: " + grp.comm_error}
    if res is not None and world == 1 and not a.no_cpu:
        res["cpu_baseline"] = cpu_baseline(a.workload, a.cpu_seconds)
        phase("cpu_baseline")
    if res is not None:
        # rank 0's wall clock by phase beside what the phases should take (wall_budget): a first 8-GPU run that strays from
        # its budget says where
        res["phases_s"] = dict(phases, total=round(sum(phases.values()), 2))
        res["wall_budget_s"] = wall_budget(world, a.steps, a.warmup, a.log2n, a.arena_rest_s, a.settle_ms,
                                           not a.no_verify_all, grp.comm is not None and not a.no_gather)
    finish(grp, res)


def gather_leg(a, pkg, grp, dev, stream, ns, d_in, pipe, out_shape_rows, decim, label, placed=None):
    """hot path + gather of its output to rank 0, batch k's transfer under batch k+1's kernels.
    `placed`: two output buffers that already lie well against d_in (the arena's chosen slot), else fresh ones."""
    import torch
    world, rank = grp.world, grp.rank
    n_out = pipe.max_output(ns)
    nbytes = n_out * 8
    outs = placed if placed is not None else [torch.empty((out_shape_rows, 2), dtype=torch.float32, device=dev)
                                              for _ in range(2)]
    recv = torch.empty((world, n_out, 2), dtype=torch.float32, device=dev) if rank == 0 else None
    rptr = recv.data_ptr() if recv is not None else 0

    def gstep(k):
        o = outs[k & 1]
        grp.comm.gather_fence(stream)          # the transfer that last read this buffer pair is done
        pipe.process_ptr(d_in.data_ptr(), ns, o.data_ptr(), o.shape[0], stream)
        pipe.fence(stream)                     # (overlap mode) the gather reads this batch's output
        grp.comm.gather_async(o.data_ptr(), nbytes, rptr, 0, stream)

    # untimed: both buffers once, then enough back-to-back steps for sustained clocks (the legs before this one end in
    # verification and host work; the first dozen launches after an idle spell run up to 40 % slow, NOTEBOOK.md rounds 1-3 5 DVFS)
    for k in range(200 + a.warmup):         # the same count on every rank: each step is a collective
        gstep(k)
    grp.comm.gather_wait()
    torch.cuda.synchronize(dev)
    grp.barrier()
    t0 = time.perf_counter()
    for k in range(a.steps):
        gstep(k)
    grp.comm.gather_wait()
    torch.cuda.synchronize(dev)
    grp.barrier()
    tg = grp.max_seconds(time.perf_counter() - t0)
    # every block that reached the root against what its rank sent: checksums of the last batch travel over
    # the control plane (gloo), the blocks themselves came over xGMI
    last = outs[(a.steps - 1) & 1][:n_out]
    sums = grp.all_gather_object(int(last.view(torch.int32).to(torch.int64).sum().item()))
    ok = blocks_ok = None
    if rank == 0:                               # rank 0's own block is its own last output
        ok = bool(torch.equal(recv[0], last))
        blocks_ok = bool(all(int(recv[r].view(torch.int32).to(torch.int64).sum().item()) == sums[r]
                             for r in range(world)))
    return {"workload": label, "value": round(world * ns * a.steps / tg / 1e6, 1), "unit": "MS/s",
            "ms_per_step": round(tg / a.steps * 1e3, 4),
            "out_bytes_per_rank_per_step": int(nbytes),
            "root_ingest_GBps": round((world - 1) * nbytes * a.steps / tg / 1e9, 2),
            # xGMI is point to point: each peer reaches rank 0 over its own link
            # (None at one rank: no link carried anything -- a measured zero would be a number)
            "per_link_GBps": round(nbytes * a.steps / tg / 1e9, 2) if world > 1 else None,
            "root_block_matches_own_output": ok, "all_blocks_match_their_ranks_checksums": blocks_ok}


def guarded_gather_legs(a, pkg, shard, grp, dev, stream, ns, d_in, wl, pipe, out, out2=None, spare=None):
    """Both gather legs under a watchdog: a collective that never completes must not cost the
    run its line -- the watchdog prints it without the gather and ends the process."""
    import torch
    res = {"note": "hot path + RCCL gather (grouped ncclSend/ncclRecv from the C library, peer -> rank 0, one xGMI "
                   "link per peer) of the float32 output; batch k's transfer runs under batch k+1's kernels. "
                   "Never part of `value`."}
    state = {"fired": False}

    def on_timeout():
        state["fired"] = True
        print(f"[bench] rank {grp.rank}: gather leg exceeded {a.gather_timeout:.0f} s", file=sys.stderr, flush=True)
        if grp.rank == 0 and PARTIAL.get("res") is not None:
            r = PARTIAL["res"]
            r["gather"] = {"error": f"gather leg did not finish within {a.gather_timeout:.0f} s"}
            print(json.dumps(r), flush=True)
        os._exit(3)          # a collective that never completed is a failure: the launcher passes the status on

    timer = threading.Timer(a.gather_timeout, on_timeout)
    timer.daemon = True
    timer.start()
    try:
        res["this_workload"] = gather_leg(a, pkg, grp, dev, stream, ns, d_in, pipe, out.shape[0], wl["decim"],
                                          wl["label"], placed=[out, out2] if out2 is not None else None)
        if a.workload != "c320" and (grp.world > 1 or os.environ.get("PDDC_BENCH_GATHER_C320") == "1"):
            w2 = workload_def("c320")
            p2 = grp.make_pipeline(pkg, w2["stages"], w2["freg"], w2["mix"])
            rows2, placed2 = p2.max_output(ns) + 8, None
            ws2 = (p2.workspace_size(ns) + 255) & ~255
            ob2 = (rows2 * 8 + 255) & ~255
            if spare is not None and ws2 + 2 * ob2 <= spare.numel():
                # the cascade's workspace and outputs in the rest of the arena slot that already suits this input
                p2.set_workspace(spare.data_ptr(), ws2, ns)
                placed2 = [spare[ws2 + k * ob2:ws2 + k * ob2 + rows2 * 8].view(torch.float32).view(rows2, 2) for k in range(2)]
            res["c320"] = gather_leg(a, pkg, grp, dev, stream, ns, d_in, p2, rows2, 320, w2["label"], placed=placed2)
            p2.close()
    except Exception as e:
        res["error"] = f"{type(e).__name__}: {e}"
    finally:
        timer.cancel()
    return res


PARTIAL = {}


def run_api250k(a):
    """The reference's own call sequence (examples/perseustest.c:188-404: init, open, firmware, rate, tuning, start with a
    callback, stop, close, exit) through the drop-in library, in the C client (libperseus-sdr_amd/perseus_plumbing -B):
    250 kS/s setting, 7.1 MHz, float32 callback buffers from the GPU (mode ddc), the synthetic LCG stream generated on the
    device, unpaced.  A "step" is one GPU batch of the stream; the client stamps the callback that completes batch W and
    the one that completes batch W + K.  This process makes no GPU call (the child owns the device)."""
    here = os.path.dirname(os.path.abspath(__file__))
    exe = os.path.join(here, "libperseus-sdr_amd", "perseus_plumbing")
    if not os.path.exists(exe):
        raise SystemExit(f"{exe} is missing: run __graft_entry__.build()")
    env = dict(os.environ, PERSEUS_AMD_MODE="ddc", PERSEUS_AMD_PACE="0")
    if a.api_batch_log2 > 0:
        env["PERSEUS_AMD_BATCH"] = str(1 << a.api_batch_log2)
    cmd = [exe, "-s", "250000", "-f", "7100000", "-n", "2", "-b", "6144", "-o", "none", "-a", "-d", "0", "-t", "600",
           "-B", f"{a.warmup},{a.steps}"]
    t0 = time.perf_counter()
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    wall = time.perf_counter() - t0
    line = [l for l in out.stdout.splitlines() if l.startswith("api_bench")]
    if out.returncode != 0 or not line or "failed" in line[0]:
        raise SystemExit(f"api250k: client failed (rc {out.returncode}): {out.stdout[-400:]} {out.stderr[-800:]}")
    kv = dict(t.split("=") for t in line[0].split()[1:])
    ns = int(kv["batch_samples"])
    ms = float(kv["ms_per_step"])
    pkg = importlib.import_module("libperseus-sdr_amd")
    plan = pkg.api_plan(250000)
    bps = 6.0 + 8.0 / 320.0
    achieved = bps * ns / (ms * 1e-3) / 1e9
    res = {
        "metric": BASELINE_METRIC, "value": round(ns / ms / 1e3, 1), "unit": "MS/s", "n_gpus": 1, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        # (the pair of the 250 kS/s plan runs on the matrix-core kernel for whole-tile batches up to 2^25 samples -- the
        # pipeline's i8x_pair_max_log2 -- and on the vector pair above)
        "dtype": "i8xi8->i32, f32 out" if (ns <= (1 << 25) and ns % 8192 == 0) else "f32", "data": "synthetic",
        "config": {"workload": "drop-in API: perseus_init/open/firmware_download/set_sampling_rate(250000)/set_ddc_center_freq(7.1 MHz)/"
                               "start_async_input(12288 B, C callback) in libperseus-sdr_amd/perseus_plumbing; on-device LCG source, unpaced, "
                               "float32 buffers (mode ddc)",
                   "samples_per_gpu_per_step": ns, "ntaps": [int(t.size) for _d, t, _l in plan],
                   "api_batch": "the library's choice for an unpaced on-device source" if a.api_batch_log2 <= 0 else "PERSEUS_AMD_BATCH",
                   "gpu_batches": int(kv["gpu_batches"]), "gpu_source": int(kv["gpu_source"])},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                     "kernel": "whole API step (callbacks included): 6 + 8/320 B per ADC sample over the time between the callbacks "
                               "that complete batch W and batch W + K",
                     "bytes_per_sample": bps},
        "cpu_baseline": None, "client_wall_s": round(wall, 2),
    }
    print(json.dumps(res), flush=True)


def finish(grp, res):
    # RCCL writes its version banner to C stdout, which is block buffered on a pipe and would
    # otherwise come out at process exit -- after the JSON line, on any rank.  Every rank flushes
    # it now, then all ranks meet, then rank 0 prints the one JSON line last.
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    grp.barrier()
    grp.close()
    if res is not None:
        print(json.dumps(res), flush=True)
    if grp.must_hard_exit:                 # a helper thread is still inside ncclCommInitRank: no normal exit
        sys.stderr.flush()
        os._exit(0)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain command: become the launcher.  Nothing above has imported torch or touched a GPU.
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))
    if a.workload == "api250k":
        return run_api250k(a)
    run_rank(a)


if __name__ == "__main__":
    main()
