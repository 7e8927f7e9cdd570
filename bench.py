#!/usr/bin/env python3
"""bench.py -- throughput of the I/Q ingest + decimation hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N>1 is launched by torch.distributed.run (one rank per GPU, RCCL).
A "step" is one pass of the hot path over one device-resident batch:
  BASELINE.json configs[1]: 2^28 complex samples of synthetic 24-bit I/Q (LCG,
  seed 12345+rank) -> fused unpack + 127-tap polyphase decimate-by-8 -> float32.
The stream shards as independent streams (one per GPU), so there
is no data-path collective in the timed region: scaling is "weak".  `--gather`
additionally measures config 4's RCCL gather of the /8 output to rank 0 and
reports it in the "gather" object (never in `value`).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel
(k_fir8): algorithmic bytes = 7 B per input sample (6 packed in + 8/8 out,
SURVEY.md 8d) over the kernel's average duration measured with HIP events on
the launch stream.  `cpu_baseline` times the oracle's float path
(oracle/perseus_oracle.c orc_stage1_f32, kind "port") on this box's cores
over a bounded sample of the same workload (N=1, rank 0 only).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
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
                    choices=["d8_127", "d8_255", "c320", "unpack"],
                    help="d8_127 = BASELINE configs[1] (default); others are sweep points")
    ap.add_argument("--taps-fp16", action="store_true")
    ap.add_argument("--gather", action="store_true", help="also measure RCCL gather of the output (N>1)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def load_taps(name):
    import numpy as np
    return np.fromfile(os.path.join(ROOT, "tests", "golden", f"taps_{name}.f32"), dtype=np.float32)


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
        import subprocess
        env = dict(os.environ, PERSEUS_AMD_PACE="0", PERSEUS_AMD_SOURCE="zero", PERSEUS_AMD_MODE="wire")
        t = max(1, int(round(min(seconds, 5.0))))
        p = subprocess.run([ref_bin, "-a", "-t", str(t), "-p", "-s", "2000000", "-d", "3", "-o", "/dev/null"],
                           env=env, capture_output=True, text=True, timeout=120)
        m = re.search(r"Rate: ([0-9.]+) kS/s", p.stderr)
        if m:
            return {"value": round(float(m.group(1)) / 1e3, 2), "unit": "MS/s", "cores": 1, "kind": "reference",
                    "sample": f"examples/perseustest.c user_data_callback_c_f (6144-byte buffers, per-sample fwrite to "
                              f"/dev/null) driven unpaced by libperseus-sdr.so for {t} s"}
    if workload == "unpack":
        run = lambda b: O.unpack24_f32(b)
        label = "24-bit unpack only, 1 thread (reference callback style)"
        threads = 1
    else:
        h = load_taps("d8_255" if workload == "d8_255" else "d8_127")
        run = lambda b: O.stage1_f32(b, h, 8, threads)
        label = f"unpack + {h.size}-tap decimate-by-8, float accumulate, OpenMP"
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
    return {"value": round(n_big / dt / 1e6, 2), "unit": "MS/s", "cores": threads, "kind": "port",
            "sample": f"{passes} passes over 2^25 samples of the same LCG stream ({label}), {dt:.1f} s"}


def _baseline_metric():
    """The metric string is BASELINE.json's, verbatim."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "input MS/s through unpack+decimate, 1/2/4/8 GPU; % HBM-roofline"


BASELINE_METRIC = _baseline_metric()


def main():
    a = parse()
    import numpy as np
    import torch
    import torch.distributed as dist
    pkg = importlib.import_module("libperseus-sdr_amd")
    shard = importlib.import_module("libperseus-sdr_amd.shard")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} needs WORLD_SIZE={a.gpus} (launch with torch.distributed.run)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_dist = os.environ.get("PDDC_BENCH_FORCE_DIST") == "1"     # 1-rank RCCL group (testing)
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    ns = 1 << a.log2n
    # ---- workload ---------------------------------------------------------
    if a.workload == "c320":
        stages = [(8, load_taps("c320_s1_d8_32")), (8, load_taps("c320_s2_d8_64")),
                  (5, load_taps("c320_s3_d5_161"))]
        mix, bytes_per_sample, decim = True, 6.0 + 8.0 / 320.0, 320
        wl = "80 MS/s synthetic 24-bit I/Q, NCO mix 7.1 MHz + cascade /320 (8*8*5)"
    elif a.workload == "unpack":
        stages, mix, bytes_per_sample, decim = None, False, 14.0, 1
        wl = "24-bit packed I/Q -> float32 unpack only"
    else:
        h = load_taps(a.workload)
        stages, mix, bytes_per_sample, decim = [(8, h)], False, 7.0, 8
        wl = f"80 MS/s synthetic 24-bit I/Q, unpack + {h.size}-tap polyphase decimate-by-8"

    # configuration (taps, NCO word, plan) comes from rank 0 over RCCL: a few KB, once
    if stages is not None:
        cfg = shard.broadcast_config({"freg": pkg.ddc_lib().pddc_nco_freg(7.1e6, 80e6) if mix else 0,
                                      "stages": stages} if rank == 0 else None, dev)
        stages = cfg["stages"]
    d_in = pkg.synth_lcg(6 * ns, shard.stream_seed(rank), 0, dev)   # device resident before timing
    stream = torch.cuda.current_stream(dev).cuda_stream
    if stages is not None:
        pipe = pkg.Pipeline(stages, device=local, mix=mix, taps_fp16=a.taps_fp16)
        if mix:
            pipe.set_freg(cfg["freg"])
        out = torch.empty((pipe.max_output(ns) + 8, 2), dtype=torch.float32, device=dev)

        def step():
            return pipe.process_ptr(d_in.data_ptr(), ns, out.data_ptr(), out.shape[0], stream)
    else:
        out = torch.empty((ns, 2), dtype=torch.float32, device=dev)

        def step():
            pkg.check(pkg.ddc_lib().pddc_unpack24_f32(d_in.data_ptr(), ns, out.data_ptr(), stream))
            return ns

    barrier = shard.barrier

    t_settle = time.perf_counter()
    while (time.perf_counter() - t_settle) * 1e3 < a.settle_ms:    # untimed, back-to-back (no idle gaps):
        for _ in range(8):                                         # sustained-load clocks, not boost
            step()
        torch.cuda.synchronize(dev)
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize(dev)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(a.steps):
        step()
    ev1.record()
    torch.cuda.synchronize(dev)
    barrier()
    dt = time.perf_counter() - t0
    ev_ms = ev0.elapsed_time(ev1)                       # HIP events on the launch stream

    dt_max = shard.max_over_ranks(dt, dev)

    # dominant-kernel duration: stage-0 kernel alone, HIP events on the same stream
    kern_ms = None
    if stages is not None and pipe.fused:
        kern_ms = pipe.time_stage0(d_in.data_ptr(), ns, out.data_ptr(), max(a.steps, 5), stream)
    else:
        kern_ms = ev_ms / a.steps

    gather = None
    if a.gather and shard.is_dist() and stages is not None:
        # BASELINE config 4: gather every rank's /8 output on rank 0.  Double-buffered and
        # asynchronous: the xGMI transfer of batch k overlaps the kernels of batch k+1.
        n_out = pipe.max_output(ns)
        outs = [out, torch.empty_like(out)]
        bufs = [torch.empty((n_out, 2), dtype=torch.float32, device=dev) for _ in range(world)] if rank == 0 else None

        def gstep(k, pending):
            o = outs[k & 1]
            pipe.process_ptr(d_in.data_ptr(), ns, o.data_ptr(), o.shape[0], stream)
            if pending is not None:
                pending.wait()                       # batch k-1 has left before its buffer is reused at k+1
            return shard.gather_to_root_async(o[:n_out], bufs)

        pending = None
        for k in range(2):
            pending = gstep(k, pending)
        if pending is not None:
            pending.wait()
        torch.cuda.synchronize(dev)
        barrier()
        t0 = time.perf_counter()
        pending = None
        for k in range(a.steps):
            pending = gstep(k, pending)
        if pending is not None:
            pending.wait()
        torch.cuda.synchronize(dev)
        barrier()
        tgv = shard.max_over_ranks(time.perf_counter() - t0, dev)
        gather = {"value": round(world * ns * a.steps / tgv / 1e6, 1), "unit": "MS/s",
                  "note": "hot path + RCCL gather of the /%d float32 output to rank 0, gather of batch k "
                          "overlapped with the kernels of batch k+1" % decim,
                  "out_bytes_per_rank_per_step": int(n_out * 8),
                  "root_ingest_GBps": round((world - 1) * (n_out * 8) * a.steps / tgv / 1e9, 2),
                  # xGMI is point to point: each peer reaches rank 0 over its own link
                  "per_link_GBps": round((n_out * 8) * a.steps / tgv / 1e9, 2) if world > 1 else 0.0}

    if rank == 0:
        total_samples = world * ns * a.steps
        value = total_samples / dt_max / 1e6
        achieved = bytes_per_sample * ns / (kern_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc) and a.log2n == 28 and not a.taps_fp16:   # measured for this exact launch shape
            try:
                traffic = json.load(open(pmc)).get(a.workload)
            except Exception:
                traffic = None
        res = {
            "metric": BASELINE_METRIC,
            "value": round(value, 1), "unit": "MS/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt_max / a.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl, "samples_per_gpu_per_step": ns,
                       "input": "LCG bytes seed 12345+rank, device resident",
                       "sharding": "independent stream per GPU, no data-path collective",
                       "taps_storage": "fp16" if a.taps_fp16 else "fp32"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic,
                         "kernel": "k_fir8" if (stages is not None and pipe.fused) else "pipeline",
                         "kernel_ms": round(kern_ms, 4),
                         "algorithmic_bytes_per_sample": bytes_per_sample},
            "events_ms_per_step": round(ev_ms / a.steps, 4),
        }
        if gather:
            res["gather"] = gather
        if world == 1 and not a.no_cpu:
            res["cpu_baseline"] = cpu_baseline(a.workload, a.cpu_seconds)
        else:
            res["cpu_baseline"] = None
    else:
        res = None
    # RCCL writes its version banner to C stdout, which is block buffered on a pipe and would
    # otherwise come out at process exit -- after the JSON line, on any rank.  Every rank flushes
    # it now, then all ranks meet, then rank 0 prints the one JSON line last.
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if shard.is_dist():
        barrier()
        dist.destroy_process_group()
    if res is not None:
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
