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
        # 2-tuples and a rational (D, taps, L) stage, as the ten-rate plans have them
        cfg = {"freg": 381178347, "stages": [(8, load_taps("c320_s1_d8_32")), (5, load_taps("c320_s3_d5_161")),
                                              (25, np.linspace(-1, 1, 48, dtype=np.float32), 12)]}
    cfg = shard.broadcast_config(cfg, dev)
    ok_cfg = cfg["freg"] == 381178347 and [(st[0], st[2]) for st in cfg["stages"]] == [(8, 1), (5, 1), (25, 12)] and \
        np.array_equal(cfg["stages"][1][1], load_taps("c320_s3_d5_161")) and \
        np.array_equal(cfg["stages"][2][1], np.linspace(-1, 1, 48, dtype=np.float32))
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


# ------------------------------------------------ ONE stream cut into time chunks
def _oracle_chunk(O, packed_seg, stages, freg, n0):
    """What a rank's pipeline computes after seek(n0): zero history, absolute NCO phase."""
    x = O.nco_mix(O.unpack24_f32(packed_seg).astype(np.float64), freg, n0)
    for d, h in stages:
        x = O.fir_decim(x, h, d)
    return np.asarray(x, dtype=np.float64)


def _time_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shard = importlib.import_module("libperseus-sdr_amd.shard")
    from oracle import oracle as O
    stages = [(8, load_taps("c320_s1_d8_32")), (8, load_taps("c320_s2_d8_64")), (5, load_taps("c320_s3_d5_161"))]
    freg, dtot = 381178347, 320
    total = 320 * 64 * 6
    halo = shard.cascade_halo(stages)
    start, length = shard.time_chunks(total, world, dtot)[rank]
    h = min(halo, start)
    stream = O.lcg_bytes(6 * total, 12345)                      # every rank can generate the ONE stream
    seg = stream[6 * (start - h):6 * (start + length)]
    y = _oracle_chunk(O, seg, stages, freg, start - h)[2 * (h // dtot):]
    t = torch.from_numpy(np.ascontiguousarray(y))
    bufs = [torch.empty(2 * (ln // dtot), dtype=torch.float64) for _, ln in shard.time_chunks(total, world, dtot)] \
        if rank == 0 else None
    dist.gather(t, bufs, dst=0)
    res = {"rank": rank, "halo": halo, "chunk": (start, length)}
    if rank == 0:
        stitched = np.concatenate([b.numpy() for b in bufs])
        ref = _oracle_chunk(O, stream, stages, freg, 0)
        res["err"] = float(np.abs(stitched - ref).max() / np.abs(ref).max())
        res["n"] = (stitched.size, ref.size)
    q.put(res)
    dist.destroy_process_group()


def test_time_chunk_sharding_stitches_to_the_single_stream(O):
    """Two ranks, one stream: each rank's [halo | chunk] from zero history, halo outputs
    dropped, gathered in rank order == the single-stream result (NCO phase from the
    absolute index, no hand-over)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_time_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r["rank"]: r for r in (q.get(timeout=180) for _ in range(world))}
    for p in procs:
        p.join(60)
    assert res[0]["halo"] % 320 == 0 and res[0]["halo"] >= 31 + 8 * 63 + 64 * 160
    assert res[1]["chunk"][0] == res[0]["chunk"][1]
    assert res[0]["n"][0] == res[0]["n"][1]
    assert res[0]["err"] < 1e-12


def test_time_chunk_helpers():
    shard = importlib.import_module("libperseus-sdr_amd.shard")
    ch = shard.time_chunks(8192 * 10, 4, 8192)
    assert ch == [(0, 16384), (16384, 16384), (32768, 16384), (49152, 32768)]
    assert sum(ln for _, ln in ch) == 81920
    with pytest.raises(ValueError):
        shard.time_chunks(1000, 2, 320)
    assert shard.cascade_halo([(8, np.zeros(127, np.float32))]) == 128
    with pytest.raises(ValueError):
        shard.cascade_halo([(25, np.zeros(100, np.float32), 12)])


# ---- RcclGroup when RCCL does not come up: every rank drops to the gloo control plane ---------------------------
class _FakePipe:
    def __init__(self, stages, device=0, mix=False, taps_fp16=False):
        self.stages, self.device, self.mix, self.freg = stages, device, mix, None

    def set_freg(self, f):
        self.freg = f


class _FakeComm:
    """Stands in for the ctypes binding of pddc_comm_* (no GPU here): rank 1 cannot make its
    communicator, so rank 0's first collective never returns -- what a half-up RCCL looks like."""
    closed = False

    @staticmethod
    def unique_id():
        return b"\x01" * 128

    @classmethod
    def init_rank(cls, world, rank, uid, local):
        if rank == 1:
            raise RuntimeError("ncclCommInitRank: unhandled system error (injected)")
        return cls()

    def barrier(self):
        import time
        time.sleep(3600)

    def close(self):
        _FakeComm.closed = True


class _FakePkg:
    Comm = _FakeComm
    Pipeline = _FakePipe
    PDDC_F_MIX, PDDC_F_TAPS_FP16 = 1, 2


def _fallback_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    shard = importlib.import_module("libperseus-sdr_amd.shard")
    grp = shard.RcclGroup(_FakePkg, rank, world, rank, init_timeout=2.0)
    h = load_taps("c320_s1_d8_32")
    pipe = grp.make_pipeline(_FakePkg, [(8, h)] if rank == 0 else None, 381178347 if rank == 0 else 0, True)
    grp.barrier()
    res = {"rank": rank, "comm_none": grp.comm is None, "err": grp.comm_error, "hard": grp.must_hard_exit,
           "tmax": grp.max_seconds(1.0 + rank), "names": grp.all_gather_object(f"r{rank}"),
           "plan_ok": pipe.freg == 381178347 and pipe.mix and len(pipe.stages) == 1 and pipe.stages[0][0] == 8
           and np.array_equal(pipe.stages[0][1], h), "device": pipe.device}
    grp.close()
    res["abandoned_comm_left_alone"] = not _FakeComm.closed
    q.put(res)
    q.close()
    q.join_thread()
    if grp.must_hard_exit:
        os._exit(0)


def test_rccl_group_falls_back_to_gloo_when_no_communicator():
    """bench.py --gpus N on a node whose RCCL cannot come up (or hangs coming up) still measures:
    the data path has no collective, only the plan / barrier / MAX cross ranks."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fallback_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda r: r["rank"])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert r["comm_none"]
        assert "injected" in r["err"] or "not up after" in r["err"]       # the first failing rank's reason, the same on all
        assert r["err"] == res[0]["err"]
        assert abs(r["tmax"] - 2.0) < 1e-12
        assert r["names"] == ["r0", "r1"]
        assert r["plan_ok"] and r["device"] == r["rank"]
        assert r["abandoned_comm_left_alone"]
    assert res[0]["hard"] and not res[1]["hard"]          # rank 0's helper thread is still inside the "collective"
