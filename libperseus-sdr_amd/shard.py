"""Multi-GPU sharding of the stream: one process per GPU, one independent
stream per rank (the reference models up to 8 receivers as 8 independent
descriptors, perseus-sdr.c:43-47) -- or ONE stream cut into
contiguous time chunks, chunk g on GPU g, each re-reading a halo of history
(8e (2): time_chunks / cascade_halo / Pipeline.seek).  The data path needs no
collective either way.  What crosses GPUs -- the configuration broadcast, the MAX
of the step time, the optional gather of the decimated output (BASELINE config 4)
-- is done by the C library's own RCCL calls (pddc_comm_*, csrc/ddc_multi.cpp)
through RcclGroup below; torch.distributed only carries the rendezvous (a gloo
group for the 128-byte RCCL id).  The module-level helpers (broadcast_config,
gather_to_root, ...) are the torch.distributed forms the CPU tests run on gloo,
where no GPU and hence no RCCL exists.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist


MAX_STAGES = 4                  # PDDC_MAX_STAGES (include/perseus_ddc.h)


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), \
        int(os.environ.get("LOCAL_RANK", "0"))


def is_dist() -> bool:
    """True once a process group exists (also for a 1-rank group, which the
    bench can force to exercise the RCCL calls on a single GPU)."""
    return dist.is_available() and dist.is_initialized()


def stream_seed(rank: int, base: int = 12345) -> int:
    """LCG seed of the stream owned by `rank` (BASELINE.md 3 / SURVEY.md 8d: seeds 12345+g)."""
    return (base + rank) & 0xFFFFFFFF


def cascade_halo(stages, align: int = 8) -> int:
    """Input samples of history a cascade of (D, taps[, L]) decimators needs in front of a
    chunk so that every output of the chunk is exact: h1 + D1*h2 + D1*D2*h3 ...,
    h_i = ntaps_i - 1, rounded up to a multiple of `align` and of the overall decimation."""
    span, dprod = 0, 1
    for st in stages:
        d, h = int(st[0]), np.asarray(st[1])
        if len(st) > 2 and st[2] and int(st[2]) > 1:
            raise ValueError("time-chunk sharding covers integer decimators only")
        span += dprod * (h.size - 1)
        dprod *= d
    unit = int(np.lcm(dprod, align))
    return ((span + unit - 1) // unit) * unit


def time_chunks(total_samples: int, world: int, unit: int):
    """Cut [0, total_samples) into `world` contiguous chunks whose boundaries are multiples
    of `unit` (the overall decimation, or the kernel tile to stay on the fused path).
    Returns [(start, length)] per rank; the last rank takes the remainder."""
    if total_samples % unit:
        raise ValueError("total_samples must be a multiple of unit")
    per = (total_samples // unit // world) * unit
    out = []
    for r in range(world):
        start = r * per
        out.append((start, per if r < world - 1 else total_samples - start))
    return out


def process_time_chunk(pipe, d_packed_with_halo, start: int, halo: int, total_decim: int):
    """Run one time chunk on this rank's pipeline: `d_packed_with_halo` holds the samples
    [start - halo', start + length) of the ONE stream, halo' = min(halo, start) (the stream's
    first chunk has nothing in front of it and starts from zero history like the stream itself).
    Returns the chunk's outputs (a view that drops the halo's)."""
    h = min(halo, start)
    pipe.seek(start - h)
    y = pipe.process(d_packed_with_halo)
    return y[h // total_decim:]


def broadcast_config(cfg: dict | None, device, src: int = 0) -> dict:
    """Rank `src` supplies {"freg": int, "stages": [(D, taps[, L]), ...]}; every rank
    returns the same dict with 3-tuples (D, taps, L), L = 1 for a plain decimator.  A few
    KB: one header + one tensor per stage.  This is the torch.distributed form (any backend;
    the CPU tests use gloo); on GPUs the C library broadcasts the plan itself over RCCL
    (RcclGroup.make_pipeline -> pddc_comm_bcast_pipeline)."""
    if not is_dist():
        return cfg
    rank = dist.get_rank()
    hdr = torch.zeros(2 + 3 * MAX_STAGES, dtype=torch.int64, device=device)
    if rank == src:
        if len(cfg["stages"]) > MAX_STAGES:
            raise ValueError(f"at most {MAX_STAGES} stages (PDDC_MAX_STAGES)")
        hdr[0] = int(cfg["freg"])
        hdr[1] = len(cfg["stages"])
        for i, st in enumerate(cfg["stages"]):
            hdr[2 + 3 * i] = int(st[0])
            hdr[3 + 3 * i] = int(np.asarray(st[1]).size)
            hdr[4 + 3 * i] = int(st[2]) if len(st) > 2 and st[2] and int(st[2]) > 1 else 1
    dist.broadcast(hdr, src)
    n = int(hdr[1])
    stages = []
    for i in range(n):
        d, nt, li = int(hdr[2 + 3 * i]), int(hdr[3 + 3 * i]), int(hdr[4 + 3 * i])
        t = torch.zeros(nt, dtype=torch.float32, device=device)
        if rank == src:
            t.copy_(torch.from_numpy(np.ascontiguousarray(cfg["stages"][i][1], dtype=np.float32)))
        dist.broadcast(t, src)
        stages.append((d, t.cpu().numpy(), li))
    return {"freg": int(hdr[0]), "stages": stages}


def max_over_ranks(seconds: float, device) -> float:
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if is_dist():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if is_dist():
        dist.barrier()


def gather_to_root_async(mine: torch.Tensor, bufs=None, dst: int = 0):
    """Start the gather and return a work handle (None without a process group):
    the transfer of batch k then overlaps the kernels of batch k+1, which write a
    different output buffer.  Call .wait() before `mine` is overwritten."""
    if not is_dist():
        return None
    if dist.get_rank() == dst:
        return dist.gather(mine, bufs, dst=dst, async_op=True)
    return dist.gather(mine, None, dst=dst, async_op=True)


def gather_to_root(mine: torch.Tensor, bufs=None, dst: int = 0):
    """Gather each rank's decimated output on rank `dst` (direct peer->root
    transfers: all 7 xGMI links of the root are used).  Returns the
    list of per-rank tensors on the root, None elsewhere."""
    if not is_dist():
        return [mine]
    if dist.get_rank() == dst:
        if bufs is None:
            bufs = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
        dist.gather(mine, bufs, dst=dst)
        return bufs
    dist.gather(mine, None, dst=dst)
    return None


# --------------------------------------------------------------------------
# The group of ranks bench.py runs in.  Control plane (rendezvous, the 128-byte
# RCCL id, small Python objects): a gloo process group over the launcher's store.
# GPU collectives: the C library's own RCCL calls (pddc_comm_*, ddc_multi.cpp) --
# torch's NCCL backend is not initialised at all.
# --------------------------------------------------------------------------
class Group:
    """A single rank (no launcher): every collective is the identity."""
    rank, world, local = 0, 1, 0
    comm = None
    comm_error = None
    must_hard_exit = False

    def make_pipeline(self, pkg, stages, freg, mix, taps_fp16=False):
        p = pkg.Pipeline(stages, device=self.local, mix=mix, taps_fp16=taps_fp16)
        if mix:
            p.set_freg(freg)
        return p

    def max_seconds(self, t: float) -> float:
        return t

    def comm_size(self) -> int:
        """ranks as the RCCL communicator counts them (pddc_comm_size) where one exists, else the launcher's"""
        return int(self.comm.size) if self.comm is not None else int(self.world)

    def barrier(self):
        pass

    def all_gather_object(self, obj):
        return [obj]

    def close(self):
        pass


import contextlib


@contextlib.contextmanager
def _c_stdout_to_stderr():
    """RCCL prints a version banner on C stdout when a communicator is made; a bench line must be
    the only thing on stdout, so file descriptor 1 points at stderr while that happens."""
    import ctypes
    import sys
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(saved, 1)
        os.close(saved)


class RcclGroup(Group):
    """One process per GPU.  With world == 1 this builds a 1-rank communicator so that a 1-GPU
    box still runs every RCCL call of the N>1 path.

    The communicator is made under a time limit (`init_timeout` seconds, env PDDC_COMM_INIT_TIMEOUT,
    default 180): the data path needs no collective, so a node whose RCCL cannot come up (or hangs
    doing so) must still be measured.  All ranks then agree over gloo whether everyone has a
    communicator; if any has none, ALL drop to the gloo control plane for the three things that
    cross ranks (plan broadcast, barrier, MAX of the step time) -- `comm` is None, `comm_error`
    says why, the gather legs (which are RCCL by definition) are skipped."""

    def __init__(self, pkg, rank, world, local, pg_backend="gloo", init_timeout=None):
        self.rank, self.world, self.local = rank, world, local
        self.pkg = pkg
        self._own_pg = False
        self.comm = None
        self.comm_error = None
        self.must_hard_exit = False               # a thread is stuck inside RCCL: leave through os._exit
        if init_timeout is None:
            init_timeout = float(os.environ.get("PDDC_COMM_INIT_TIMEOUT", "180"))
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            dist.init_process_group(pg_backend, rank=rank, world_size=world)
            self._own_pg = True
        err = None
        with _c_stdout_to_stderr():
            uid = [None]
            if rank == 0:
                try:
                    uid[0] = pkg.Comm.unique_id()
                except Exception as e:            # noqa: BLE001 -- whatever it is, the ranks must hear of it
                    err = f"{type(e).__name__}: {e}"
            if world > 1:
                dist.broadcast_object_list(uid, src=0)
            if uid[0] is None:
                err = err or "rank 0 could not make an RCCL id"
            else:
                err = self._init_comm(uid[0], init_timeout)
        if world > 1:
            errs = [None] * world
            dist.all_gather_object(errs, err)
            bad = [(r, e) for r, e in enumerate(errs) if e]
            if bad:
                err = "rank %d: %s" % bad[0]
        if err:
            self.comm_error = err
            self._abandoned = self.comm           # never destroyed: a peer of it may be hung
            self.comm = None

    def _init_comm(self, uid, timeout):
        """ncclCommInitRank + the first collective on a helper thread, joined with a time limit."""
        import threading
        box = {}

        def work():
            try:
                c = self.pkg.Comm.init_rank(self.world, self.rank, uid, self.local)
                box["comm"] = c
                c.barrier()                       # first collective: whatever RCCL still wants to say, it says now
                box["ok"] = True
            except Exception as e:                # noqa: BLE001
                box["err"] = f"{type(e).__name__}: {e}"

        t = threading.Thread(target=work, daemon=True)
        t.start()
        t.join(timeout)
        if t.is_alive():
            self.must_hard_exit = True
            self.comm = box.get("comm")
            return f"RCCL communicator not up after {timeout:.0f} s"
        self.comm = box.get("comm")
        return box.get("err")

    def make_pipeline(self, pkg, stages, freg, mix, taps_fp16=False):
        """Rank 0's plan reaches every rank through ncclBroadcast inside the C library."""
        if self.comm is None:                     # gloo control plane (see the class comment)
            cfg = {"freg": int(freg), "stages": stages} if self.rank == 0 else None
            if self.world > 1:
                cfg = broadcast_config(cfg, torch.device("cpu"))
            return Group.make_pipeline(self, pkg, cfg["stages"], cfg["freg"], mix, taps_fp16)
        flags = (pkg.PDDC_F_MIX if mix else 0) | (pkg.PDDC_F_TAPS_FP16 if taps_fp16 else 0)
        return self.comm.bcast_pipeline(stages if self.rank == 0 else None, freg if self.rank == 0 else 0,
                                        flags if self.rank == 0 else 0, root=0)

    def max_seconds(self, t: float) -> float:
        if self.comm is None:
            return max_over_ranks(t, torch.device("cpu")) if self.world > 1 else t
        return self.comm.max_f64(t)

    def barrier(self):
        if self.comm is None:
            if self.world > 1:
                dist.barrier()
            return
        self.comm.barrier()

    def all_gather_object(self, obj):
        if self.world == 1:
            return [obj]
        out = [None] * self.world
        dist.all_gather_object(out, obj)
        return out

    def close(self):
        if self.comm is not None:
            self.comm.close()
            self.comm = None
        if self._own_pg:
            dist.barrier()
            if not self.must_hard_exit:
                dist.destroy_process_group()
            self._own_pg = False


class TorchGroup(Group):
    """torch.distributed on any backend.  Used by the CPU plumbing tests (gloo), where no
    GPU -- hence no RCCL -- exists; never on a GPU box."""

    def __init__(self, rank, world, local, backend="gloo"):
        self.rank, self.world, self.local = rank, world, local
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend, rank=rank, world_size=world)
        self.device = torch.device("cpu")

    def max_seconds(self, t: float) -> float:
        return max_over_ranks(t, self.device)

    def barrier(self):
        dist.barrier()

    def all_gather_object(self, obj):
        out = [None] * self.world
        dist.all_gather_object(out, obj)
        return out

    def close(self):
        dist.barrier()
        dist.destroy_process_group()
