"""Test scaffolding for bench.py's launcher path (tests/test_bench_contract.py): NOT part of the measurement tool.
tests/bench_dry.py imports bench.py, replaces its rank body by run() below and calls its main() -- so `python bench.py
--gpus N` (parent starts N children, gloo rendezvous, rank 0 relays ONE JSON line) can be exercised where no GPU exists."""
import time


def run(a, bench):
    """The rank body tests/bench_dry.py puts in place of bench.run_rank: launcher / rendezvous / relay on
    CPU.  No GPU exists there, so nothing is measured: the CPU oracle stands in for the pipeline only to give the
    ranks distinct data to gather, the line says so and claims a value of 0."""
    import importlib
    import numpy as np
    import torch
    from oracle import oracle as O
    shard = importlib.import_module("libperseus-sdr_amd.shard")
    rank, world, local = shard.env_rank_world()
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    workload_def, BASELINE_METRIC, finish = bench.workload_def, bench.BASELINE_METRIC, bench.finish
    O.build()
    grp = shard.TorchGroup(rank, world, local)
    wl = workload_def(a.workload)
    ns = 1 << min(a.log2n, 13)
    cfg = shard.broadcast_config({"freg": wl["freg"], "stages": wl["stages"]} if rank == 0 else None, grp.device)
    packed = O.lcg_bytes(6 * ns, shard.stream_seed(rank))
    t0 = time.perf_counter()
    y = None
    for _ in range(a.steps):
        y = O.ddc_chain(packed, cfg["stages"], cfg["freg"], wl["mix"])
    dt = grp.max_seconds(time.perf_counter() - t0)
    bufs = shard.gather_to_root(torch.from_numpy(y.copy()))
    ok = None
    if rank == 0:
        ok = all(np.array_equal(bufs[r].numpy(), O.ddc_chain(O.lcg_bytes(6 * ns, shard.stream_seed(r)), cfg["stages"],
                                                              cfg["freg"], wl["mix"])) for r in range(world))
    hosts = grp.all_gather_object(f"cpu:{rank}")
    res = None
    if rank == 0:
        res = {"metric": BASELINE_METRIC, "value": 0.0, "unit": "MS/s", "n_gpus": world, "steps": a.steps,
               "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": wl["label"], "samples_per_gpu_per_step": ns},
               "dry_run": "launcher/rendezvous plumbing test on CPU (gloo), the oracle stands in "
                          "for the HIP pipeline, nothing is measured",
               # what the REAL run of this shape (the driver's arguments, 2^28 samples a rank) should take on the wall
               "wall_budget_s": bench.wall_budget(world, 20, 5, 28),
               "ranks_seen": world, "devices": hosts,
               "gather": {"this_workload": {"root_blocks_match_each_ranks_stream": ok}},
               "roofline": None, "cpu_baseline": None}
    finish(grp, res)


