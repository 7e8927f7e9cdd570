"""Multi-GPU sharding of the stream: one process per GPU, one independent
stream per rank (the reference models up to 8 receivers as 8 independent
descriptors, perseus-sdr.c:43-47) -- or ONE stream cut into
contiguous time chunks, chunk g on GPU g, each re-reading a halo of history
(8e (2): time_chunks / cascade_halo / Pipeline.seek).  The data path needs no
collective either way; RCCL (torch.distributed backend "nccl") is used only to
  - broadcast the configuration (taps, NCO word, stage plan) from rank 0,
  - reduce the step time (MAX over ranks) for the benchmark,
  - optionally gather the decimated output to rank 0 (BASELINE config 4).
All functions work on any torch.distributed backend (tests use gloo on CPU).
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist


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
    """Rank `src` supplies {"freg": int, "stages": [(D, taps ndarray), ...]}; every
    rank returns the same dict.  A few KB: one small tensor broadcast per item."""
    if not is_dist():
        return cfg
    rank = dist.get_rank()
    hdr = torch.zeros(2 + 2 * 8, dtype=torch.int64, device=device)
    if rank == src:
        hdr[0] = int(cfg["freg"])
        hdr[1] = len(cfg["stages"])
        for i, (d, h) in enumerate(cfg["stages"]):
            hdr[2 + 2 * i] = int(d)
            hdr[3 + 2 * i] = int(np.asarray(h).size)
    dist.broadcast(hdr, src)
    n = int(hdr[1])
    stages = []
    for i in range(n):
        d, nt = int(hdr[2 + 2 * i]), int(hdr[3 + 2 * i])
        t = torch.zeros(nt, dtype=torch.float32, device=device)
        if rank == src:
            t.copy_(torch.from_numpy(np.ascontiguousarray(cfg["stages"][i][1], dtype=np.float32)))
        dist.broadcast(t, src)
        stages.append((d, t.cpu().numpy()))
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
