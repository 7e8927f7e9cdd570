"""world_size-2 gloo test (CPU) of the multi-GPU sharding logic: independent
stream per rank, configuration broadcast, MAX time reduction, gather to root."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_taps


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = importlib.import_module("libperseus-sdr_amd.shard")
    from oracle import oracle as O
    dev = torch.device("cpu")
    cfg = None
    if rank == 0:
        cfg = {"freg": 381178347, "stages": [(8, load_taps("c320_s1_d8_32")), (5, load_taps("c320_s3_d5_161"))]}
    cfg = shard.broadcast_config(cfg, dev)
    ok_cfg = cfg["freg"] == 381178347 and [d for d, _ in cfg["stages"]] == [8, 5] and \
        np.array_equal(cfg["stages"][1][1], load_taps("c320_s3_d5_161"))
    # each rank owns an independent stream; the checker (oracle) stands in for the GPU path here
    ns = 8 * 512
    packed = O.lcg_bytes(6 * ns, shard.stream_seed(rank))
    y = torch.from_numpy(O.ddc_chain(packed, [cfg["stages"][0]]).copy())
    tmax = shard.max_over_ranks(1.0 + rank, dev)
    shard.barrier()
    bufs = shard.gather_to_root(y)
    res = {"rank": rank, "ok_cfg": ok_cfg, "tmax": tmax, "gathered": None}
    if rank == 0:
        res["gathered"] = [b.numpy().copy() for b in bufs]
    q.put(res)
    dist.destroy_process_group()


def test_two_rank_sharding_gloo(O):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda r: r["rank"])
    assert all(r["ok_cfg"] for r in res)
    assert all(abs(r["tmax"] - 2.0) < 1e-12 for r in res)           # MAX over ranks
    g = res[0]["gathered"]
    assert len(g) == 2
    shard = importlib.import_module("libperseus-sdr_amd.shard")
    h = load_taps("c320_s1_d8_32")
    for r in range(2):                                              # rank order, distinct streams
        exp = O.ddc_chain(O.lcg_bytes(6 * 8 * 512, shard.stream_seed(r)), [(8, h)])
        assert np.array_equal(g[r], exp)
    assert not np.array_equal(g[0], g[1])


def test_single_process_helpers_are_noops():
    shard = importlib.import_module("libperseus-sdr_amd.shard")
    assert shard.max_over_ranks(0.25, torch.device("cpu")) == 0.25
    cfg = {"freg": 1, "stages": []}
    assert shard.broadcast_config(cfg, torch.device("cpu")) is cfg
    t = torch.arange(4)
    assert shard.gather_to_root(t)[0] is t
    assert shard.stream_seed(3) == 12348
