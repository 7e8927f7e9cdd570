"""The multi-GPU section of the C ABI (include/perseus_ddc.h, ddc_multi.cpp): RCCL called by
the C library itself.  CPU tests: plan (de)serialisation, loud failure without a device.
GPU tests (one GPU on the box): 1-rank communicators exercise every RCCL call of the N>1
path -- unique id, ncclCommInitRank / ncclCommInitAll, ncclBroadcast of the plan,
ncclAllReduce, the gather's send/recv group and its side-stream form."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_taps


def test_plan_pack_unpack_roundtrip(pkg):
    stages = [(8, load_taps("c320_s1_d8_32")), (10, load_taps("c320_s2_d8_64")),
              (25, np.linspace(-1, 1, 48, dtype=np.float32), 12)]
    raw = pkg.plan_pack(stages, freg=381178347, flags=pkg.PDDC_F_MIX | pkg.PDDC_F_OUT_PACKED24)
    assert len(raw) == 4 * (4 + 3 * 3 + 32 + 64 + 48)
    d = pkg.plan_unpack(raw)
    assert d["freg"] == 381178347 and d["flags"] == 9
    assert [(s[0], s[2]) for s in d["stages"]] == [(8, 1), (10, 1), (25, 12)]
    for a, b in zip(stages, d["stages"]):
        assert np.array_equal(np.asarray(a[1], np.float32), b[1])


def test_plan_unpack_rejects_garbage(pkg):
    with pytest.raises(pkg.PddcError):
        pkg.plan_unpack(b"\x00" * 64)
    raw = bytearray(pkg.plan_pack([(8, np.ones(16, np.float32))]))
    with pytest.raises(pkg.PddcError):
        pkg.plan_unpack(bytes(raw[:40]))                # truncated taps
    raw[12] = 9                                         # nstages out of range
    with pytest.raises(pkg.PddcError):
        pkg.plan_unpack(bytes(raw))
    with pytest.raises(pkg.PddcError):
        pkg.plan_pack([])                               # no stages


def test_comm_needs_a_device(pkg):
    L = pkg.ddc_lib()
    if L.pddc_device_count() > 0:
        pytest.skip("GPU present")
    h = C.c_void_p()
    uid = C.create_string_buffer(128)
    assert L.pddc_comm_init_rank(C.byref(h), 1, 0, uid, 0) == pkg.PDDC_ENODEV
    assert b"no CPU fallback" in L.pddc_last_error()
    hs = (C.c_void_p * 1)()
    assert L.pddc_comm_init_all(hs, 1, None) == pkg.PDDC_ENODEV
    assert L.pddc_comm_init_rank(C.byref(h), 2, 5, uid, 0) == pkg.PDDC_EINVAL     # rank out of range


@pytest.mark.gpu
def test_comm_single_rank_broadcast_pipeline_and_gather(pkg, O, dev):
    import torch
    uid = pkg.Comm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = pkg.Comm.init_rank(1, 0, uid, 0)
    assert (comm.rank, comm.size, comm.device) == (0, 1, 0)
    stages = [(8, load_taps("c320_s1_d8_32")), (8, load_taps("c320_s2_d8_64")), (5, load_taps("c320_s3_d5_161"))]
    freg = 381178347
    pipe = comm.bcast_pipeline(stages, freg=freg, flags=pkg.PDDC_F_MIX)        # plan went through ncclBroadcast
    assert pipe.freg == freg and pipe.decim == 320
    ns = 8192 * 40
    packed = O.lcg_bytes(6 * ns, 12345)
    d_in = torch.from_numpy(packed).to(dev)
    y = pipe.process(d_in)
    ref = O.ddc_chain(packed, stages, freg, True)
    assert O.rel_err(y.cpu().numpy().reshape(-1), ref) <= 1e-6
    direct = pkg.Pipeline(stages, mix=True)
    direct.set_freg(freg)
    assert torch.equal(direct.process(d_in), y)                                  # same plan, bit for bit
    # gather on the compute stream, then the side-stream form
    st = torch.cuda.current_stream(dev).cuda_stream
    recv = torch.zeros_like(y)
    comm.gather(y.data_ptr(), y.numel() * 4, recv.data_ptr(), 0, st)
    torch.cuda.synchronize()
    assert torch.equal(recv, y)
    recv.zero_()
    comm.gather_async(y.data_ptr(), y.numel() * 4, recv.data_ptr(), 0, st)
    comm.gather_fence(st)
    comm.gather_wait()
    assert torch.equal(recv, y)
    assert comm.max_f64(1.25) == 1.25
    comm.barrier()
    assert comm.bcast_bytes(b"taps+freg", 0) == b"taps+freg"
    buf = torch.arange(1024, dtype=torch.int32, device=dev)
    comm.bcast(buf.data_ptr(), 4096, 0, st)
    torch.cuda.synchronize()
    assert int(buf[1023]) == 1023
    pipe.close()
    direct.close()
    comm.close()


@pytest.mark.gpu
def test_comm_init_all_grouped_calls(pkg, O, dev):
    """One process, all its GPUs (here: the one visible): ncclCommInitAll + grouped per-communicator calls."""
    import torch
    L = pkg.ddc_lib()
    comms = pkg.Comm.init_all([0])
    assert len(comms) == 1 and comms[0].size == 1
    with pytest.raises(pkg.PddcError):
        pkg.Comm.init_all([0, 0])                       # one rank per GPU, said loudly
    x = torch.from_numpy(O.lcg_bytes(4096, 7)).to(dev)
    recv = torch.zeros_like(x)
    st = torch.cuda.current_stream(dev).cuda_stream
    pkg.check(L.pddc_comm_group_start())
    for c in comms:
        c.gather(x.data_ptr(), x.numel(), recv.data_ptr(), 0, st)
    pkg.check(L.pddc_comm_group_end())
    torch.cuda.synchronize()
    assert torch.equal(recv, x)
    for c in comms:
        c.close()


@pytest.mark.gpu
def test_bench_gather_leg_on_one_rank():
    """bench.py --gather: the N>1 code path (RcclGroup, plan broadcast, double-buffered gather on the side
    stream) on a 1-rank communicator."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--log2n", "24",
                        "--settle-ms", "20", "--no-cpu", "--gather"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, PDDC_BENCH_GATHER_C320="1"))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines[:5]                    # RCCL's banner went to stderr, not into the contract's stdout
    d = json.loads(lines[-1])
    assert d["ranks_seen"] == 1 and "RCCL called from the C library" in d["collectives"]
    g = d["gather"]
    assert "error" not in g, g
    assert g["this_workload"]["root_block_matches_own_output"] is True
    assert g["this_workload"]["all_blocks_match_their_ranks_checksums"] is True
    assert g["this_workload"]["out_bytes_per_rank_per_step"] == (1 << 24) // 8 * 8
    assert g["c320"]["root_block_matches_own_output"] is True and g["c320"]["value"] > 0
    assert d["verified"]["ok"] is True


def test_c_multi_gpu_bench_needs_a_gpu(pkg):
    exe = os.path.join(os.path.dirname(pkg.DDC_LIB), "perseus_multi_bench")
    assert os.path.exists(exe)
    if pkg.ddc_lib().pddc_device_count() > 0:
        pytest.skip("GPU present")
    p = subprocess.run([exe, "-g", "2"], capture_output=True, text=True, timeout=60)
    assert p.returncode != 0 and "no CPU path" in p.stderr


@pytest.mark.gpu
def test_c_multi_gpu_bench_with_gather(pkg, dev):
    """BASELINE config 4 from a plain C host (csrc/perseus_multi_bench.c): pddc_comm_init_all, one pipeline per
    GPU, grouped side-stream gathers -- here with the one GPU the box has (the same code drives eight)."""
    import re
    exe = os.path.join(os.path.dirname(pkg.DDC_LIB), "perseus_multi_bench")
    for extra in ([], ["-c"]):
        p = subprocess.run([exe, "-n", "24", "-s", "20", "-G"] + extra, capture_output=True, text=True, timeout=120)
        assert p.returncode == 0, p.stderr[-800:]
        m = re.search(r"(\d+) GPU\(s\).*: ([0-9.]+) MS/s aggregate.*with the gather to GPU 0: ([0-9.]+) MS/s", p.stdout)
        assert m, p.stdout
        assert int(m.group(1)) == 1 and float(m.group(2)) > 50000 and float(m.group(3)) > 20000
        # ... and the same figures as one JSON line with bench.py's keys
        d = json.loads(p.stdout.strip().splitlines()[-1])
        assert d["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
        assert d["n_gpus"] == 1 and d["steps"] == 20 and d["scaling"] == "weak" and d["unit"] == "MS/s"
        assert abs(d["value"] - float(m.group(2))) <= 0.2 and abs(d["gather"]["value"] - float(m.group(3))) <= 0.2
        assert d["config"]["samples_per_gpu_per_step"] == 1 << 24 and ("/320" in d["config"]["workload"]) == bool(extra)
        assert 0 <= d["gather"]["out_bytes_per_rank_per_step"] - (1 << 24) // (320 if extra else 8) * 8 <= 16
        assert d["rccl"]["running"] > 0
        print(p.stdout.strip().splitlines()[-2])


def test_the_rccl_in_use_matches_the_header_compiled_against(pkg, dev):
    """Two librccl live on these boxes (ROCm's and the torch wheel's); the library compares the running one's version
    with its header's before it makes a communicator and refuses a different major version."""
    import ctypes as C
    L = pkg.ddc_lib()
    run, hdr = C.c_int(), C.c_int()
    pkg.check(L.pddc_comm_rccl_version(C.byref(run), C.byref(hdr)))
    assert run.value > 0 and hdr.value > 0
    major = lambda v: v // 10000 if v >= 10000 else v // 1000
    assert major(run.value) == major(hdr.value), (run.value, hdr.value)
