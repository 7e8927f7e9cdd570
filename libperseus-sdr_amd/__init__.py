"""libperseus-sdr_amd -- MI355X-native I/Q ingest + decimation path behind the
libperseus-sdr API.

This package is only a loader: the product is two C-ABI shared libraries built
from csrc/ (hand-written HIP for gfx950 + plain C host code):

  libperseus_ddc.so   include/perseus_ddc.h   kernels + stream pipeline
  libperseus-sdr.so   include/perseus-sdr.h   drop-in perseus_* callback API

Python (ctypes) is used by tests/ and bench.py to drive them; torch only
supplies device memory, streams and torch.distributed.  The hyphen in the
package name follows the reference repo's name, so import it with
importlib.import_module("libperseus-sdr_amd").

There is no CPU fallback anywhere in this package: if the HIP library is not
built, loading raises; if no GPU is present, compute entry points return
PDDC_ENODEV and the wrappers raise.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
# (tools/ab.sh points the ctypes binding at an alternative build of the kernel library -- same-box A/B measurements -- through
# PDDC_DDC_LIB instead of overwriting the product; nothing else reads it)
DDC_LIB = os.environ.get("PDDC_DDC_LIB") or os.path.join(_HERE, "libperseus_ddc.so")
SDR_LIB = os.path.join(_HERE, "libperseus-sdr.so")

PDDC_OK, PDDC_EINVAL, PDDC_ENODEV, PDDC_EHIP, PDDC_ENOMEM, PDDC_ECAPACITY, PDDC_ESTATE, PDDC_ECOMM = 0, -1, -2, -3, -4, -5, -6, -7
PDDC_COMM_ID_BYTES = 128
PDDC_F_MIX, PDDC_F_TAPS_FP16, PDDC_F_NO_FAST, PDDC_F_OUT_PACKED24 = 1, 2, 4, 8


class GangItem(C.Structure):
    """pddc_gang_item (include/perseus_ddc.h)"""
    _fields_ = [("pipe", C.c_void_p), ("h_packed", C.c_void_p), ("seed", C.c_uint32), ("byte_offset", C.c_uint64),
                ("h_out", C.c_void_p), ("out_capacity", C.c_size_t), ("n_out", C.c_size_t), ("ticket", C.c_int)]


class PddcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"pddc error {code}: {msg}")
        self.code = code


def build(verbose: bool = False) -> None:
    """Compile csrc/ for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "all"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)


class StageDesc(C.Structure):
    _fields_ = [("decim", C.c_int), ("ntaps", C.c_int), ("taps", C.POINTER(C.c_float)),
                ("interp", C.c_int)]


_ddc = None


def ddc_lib() -> C.CDLL:
    """Load libperseus_ddc.so; raises if it has not been built."""
    global _ddc
    if _ddc is not None:
        return _ddc
    if not os.path.exists(DDC_LIB):
        raise FileNotFoundError(
            f"{DDC_LIB} is missing: run __graft_entry__.build() (there is no CPU fallback)")
    try:
        # torch bundles its own libamdhip64.so.7; load it first so that this
        # process holds ONE HIP runtime and torch streams/pointers are valid here
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(DDC_LIB)
    vp, sz = C.c_void_p, C.c_size_t
    L.pddc_version.restype = C.c_int
    L.pddc_last_error.restype = C.c_char_p
    L.pddc_device_count.restype = C.c_int
    L.pddc_nco_freg.argtypes = [C.c_double, C.c_double]
    L.pddc_nco_freg.restype = C.c_uint32
    L.pddc_unpack24_f32.argtypes = [vp, sz, vp, vp]
    L.pddc_unpack24_i32.argtypes = [vp, sz, vp, vp]
    L.pddc_pack24_f32.argtypes = [vp, sz, vp, vp]
    L.pddc_synth_lcg.argtypes = [vp, sz, C.c_uint32, C.c_uint64, vp]
    L.pddc_set_device.argtypes = [C.c_int]
    L.pddc_malloc.argtypes = [C.POINTER(vp), sz]
    L.pddc_free.argtypes = [vp]
    L.pddc_malloc_apart.argtypes = [C.POINTER(vp), sz, vp, sz, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.pddc_malloc_apart.restype = C.c_int
    L.pddc_memcpy_h2d.argtypes = [vp, vp, sz, vp]
    L.pddc_memcpy_d2h.argtypes = [vp, vp, sz, vp]
    L.pddc_stream_sync.argtypes = [vp]
    L.pddc_pipeline_create.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(StageDesc), C.c_int, C.c_uint32]
    L.pddc_pipeline_destroy.argtypes = [vp]
    L.pddc_pipeline_reset.argtypes = [vp]
    L.pddc_pipeline_seek.argtypes = [vp, C.c_uint64]
    L.pddc_pipeline_set_freg.argtypes = [vp, C.c_uint32]
    L.pddc_pipeline_set_center_freq.argtypes = [vp, C.c_double]
    L.pddc_pipeline_set_taps.argtypes = [vp, C.c_int, C.POINTER(C.c_float), C.c_int]
    L.pddc_pipeline_get_freg.argtypes = [vp]
    L.pddc_pipeline_get_freg.restype = C.c_uint32
    L.pddc_pipeline_get_phase_offset.argtypes = [vp]
    L.pddc_pipeline_get_phase_offset.restype = C.c_uint32
    L.pddc_pipeline_total_decim.argtypes = [vp]
    L.pddc_pipeline_max_output.argtypes = [vp, sz]
    L.pddc_pipeline_max_output.restype = sz
    L.pddc_pipeline_uses_fused.argtypes = [vp]
    L.pddc_pipeline_stage0_reads_packed.argtypes = [vp]
    L.pddc_pipeline_stage0_reads_packed.restype = C.c_int
    L.pddc_pipeline_uses_fused_pair.argtypes = [vp, sz]
    L.pddc_pipeline_uses_fused_pair.restype = C.c_int
    L.pddc_pipeline_uses_fused_cascade.argtypes = [vp, sz]
    L.pddc_pipeline_uses_fused_cascade.restype = C.c_int
    L.pddc_pipeline_stage0_on_i8.argtypes = [vp, sz]
    L.pddc_pipeline_stage0_on_i8.restype = C.c_int
    L.pddc_pipeline_arena_place.argtypes = [vp, vp, sz, sz, sz, sz, C.POINTER(sz), C.POINTER(C.c_float),
                                            C.POINTER(C.c_float), C.POINTER(C.c_int), vp]
    L.pddc_pipeline_arena_place.restype = C.c_int
    L.pddc_fir_i8_table.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, vp, sz, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.pddc_fir_i8_table.restype = C.c_int
    L.pddc_fir_i8_taps16.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, vp, sz, C.POINTER(C.c_double)]
    L.pddc_fir_i8_taps16.restype = C.c_int
    L.pddc_fir_i8x_tables.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.c_uint32, vp, sz,
                                      C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.pddc_fir_i8x_tables.restype = C.c_int
    L.pddc_fir_i8x_d10_tables.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_uint32, vp, sz,
                                          C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.pddc_fir_i8x_d10_tables.restype = C.c_int
    L.pddc_fir_i8x_taps2.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.c_uint32, C.POINTER(C.c_float), sz]
    L.pddc_fir_i8x_taps2.restype = C.c_int
    L.pddc_pipeline_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    L.pddc_pipeline_set_option.restype = C.c_int
    L.pddc_pipeline_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int)]
    L.pddc_pipeline_get_option.restype = C.c_int
    L.pddc_set_tunable.argtypes = [C.c_char_p, C.c_int]
    L.pddc_set_tunable.restype = C.c_int
    L.pddc_get_tunable.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
    L.pddc_get_tunable.restype = C.c_int
    L.pddc_pipeline_check.argtypes = [vp, vp]
    L.pddc_pipeline_check.restype = C.c_int
    L.pddc_comm_rccl_version.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.pddc_comm_rccl_version.restype = C.c_int
    L.pddc_pipeline_place_buffers.argtypes = [vp, vp, sz, vp]
    L.pddc_pipeline_place_buffers.restype = C.c_int
    L.pddc_pipeline_set_overlap.argtypes = [vp, C.c_int]
    L.pddc_pipeline_set_overlap.restype = C.c_int
    L.pddc_pipeline_fence.argtypes = [vp, vp]
    L.pddc_pipeline_fence.restype = C.c_int
    L.pddc_pipeline_process.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz), vp]
    L.pddc_pipeline_push_host.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz)]
    L.pddc_pipeline_push_host_async.argtypes = [vp, vp, sz, vp, sz, C.POINTER(sz), C.POINTER(C.c_int)]
    L.pddc_pipeline_push_synth_async.argtypes = [vp, C.c_uint32, C.c_uint64, sz, vp, sz, C.POINTER(sz),
                                                 C.POINTER(C.c_int)]
    L.pddc_pipeline_push_synth_async.restype = C.c_int
    L.pddc_pipeline_ticket_done.argtypes = [vp, C.c_int]
    L.pddc_pipeline_ticket_done.restype = C.c_int
    L.pddc_pipeline_next_output.argtypes = [vp, sz]
    L.pddc_pipeline_next_output.restype = sz
    L.pddc_pipeline_wait_ticket.argtypes = [vp, C.c_int]
    L.pddc_pipeline_wait.argtypes = [vp]
    L.pddc_host_alloc.argtypes = [C.POINTER(vp), sz]
    L.pddc_host_free.argtypes = [vp]
    L.pddc_gang_create.argtypes = [C.POINTER(vp), C.c_int]
    L.pddc_gang_create.restype = C.c_int
    L.pddc_gang_destroy.argtypes = [vp]
    L.pddc_gang_destroy.restype = C.c_int
    L.pddc_gang_push_async.argtypes = [vp, C.POINTER(GangItem), C.c_int, sz, C.POINTER(C.c_int)]
    L.pddc_gang_push_async.restype = C.c_int
    L.pddc_pipeline_time_stage0.argtypes = [vp, vp, sz, vp, C.c_int, vp, C.POINTER(C.c_float)]
    L.pddc_pipeline_time_stage0_inline.argtypes = [vp, C.c_int]
    L.pddc_pipeline_time_stage0_inline.restype = C.c_int
    L.pddc_pipeline_stage0_time.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.pddc_pipeline_stage0_time.restype = C.c_int
    L.pddc_pipeline_state_size.argtypes = [vp]
    L.pddc_pipeline_state_size.restype = sz
    L.pddc_pipeline_save_state.argtypes = [vp, vp, sz, C.POINTER(sz)]
    L.pddc_pipeline_save_state.restype = C.c_int
    L.pddc_pipeline_restore_state.argtypes = [vp, vp, sz]
    L.pddc_pipeline_restore_state.restype = C.c_int
    L.pddc_pipeline_workspace_size.argtypes = [vp, sz]
    L.pddc_pipeline_workspace_size.restype = sz
    L.pddc_pipeline_set_workspace.argtypes = [vp, vp, sz, sz]
    L.pddc_pipeline_set_workspace.restype = C.c_int
    L.pddc_pipeline_inject_failure.argtypes = [vp, C.c_int]
    L.pddc_pipeline_inject_failure.restype = C.c_int
    L.pddc_pipeline_schedule.argtypes = [vp, sz, C.POINTER(C.c_int)]
    L.pddc_pipeline_schedule.restype = C.c_int
    L.pddc_measure_copy.argtypes = [vp, vp, sz, C.c_int, vp, C.POINTER(C.c_float)]
    L.pddc_measure_copy.restype = C.c_int
    # multi-GPU section (RCCL, ddc_multi.cpp)
    L.pddc_comm_get_unique_id.argtypes = [vp]
    L.pddc_comm_init_rank.argtypes = [C.POINTER(vp), C.c_int, C.c_int, vp, C.c_int]
    L.pddc_comm_init_all.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(C.c_int)]
    L.pddc_comm_destroy.argtypes = [vp]
    L.pddc_comm_rank.argtypes = [vp]
    L.pddc_comm_size.argtypes = [vp]
    L.pddc_comm_device.argtypes = [vp]
    L.pddc_comm_bcast.argtypes = [vp, vp, sz, C.c_int, vp]
    L.pddc_comm_bcast_host.argtypes = [vp, vp, sz, C.c_int]
    L.pddc_comm_allreduce_max_f64.argtypes = [vp, C.POINTER(C.c_double)]
    L.pddc_comm_barrier.argtypes = [vp]
    L.pddc_comm_gather.argtypes = [vp, vp, sz, vp, C.c_int, vp]
    L.pddc_comm_gather_async.argtypes = [vp, vp, sz, vp, C.c_int, vp]
    L.pddc_comm_gather_fence.argtypes = [vp, vp]
    L.pddc_comm_gather_wait.argtypes = [vp]
    L.pddc_plan_pack.argtypes = [C.POINTER(StageDesc), C.c_int, C.c_uint32, C.c_uint32, vp, sz]
    L.pddc_plan_pack.restype = sz
    L.pddc_plan_unpack.argtypes = [vp, sz, C.POINTER(StageDesc), C.POINTER(C.c_int), C.POINTER(C.c_uint32),
                                   C.POINTER(C.c_uint32)]
    L.pddc_comm_bcast_pipeline.argtypes = [vp, C.c_int, C.POINTER(StageDesc), C.c_int, C.c_uint32, C.c_uint32,
                                           C.POINTER(vp)]
    for name in ("pddc_comm_get_unique_id", "pddc_comm_init_rank", "pddc_comm_init_all", "pddc_comm_destroy",
                 "pddc_comm_rank", "pddc_comm_size", "pddc_comm_device", "pddc_comm_group_start",
                 "pddc_comm_group_end", "pddc_comm_bcast", "pddc_comm_bcast_host", "pddc_comm_allreduce_max_f64",
                 "pddc_comm_barrier", "pddc_comm_gather", "pddc_comm_gather_async", "pddc_comm_gather_fence",
                 "pddc_comm_gather_wait", "pddc_plan_unpack", "pddc_comm_bcast_pipeline"):
        getattr(L, name).restype = C.c_int
    for name in ("pddc_unpack24_f32", "pddc_unpack24_i32", "pddc_pack24_f32", "pddc_synth_lcg", "pddc_set_device",
                 "pddc_malloc", "pddc_free", "pddc_memcpy_h2d", "pddc_memcpy_d2h", "pddc_stream_sync",
                 "pddc_pipeline_create", "pddc_pipeline_destroy", "pddc_pipeline_reset", "pddc_pipeline_seek",
                 "pddc_pipeline_set_freg", "pddc_pipeline_set_center_freq", "pddc_pipeline_set_taps",
                 "pddc_pipeline_total_decim", "pddc_pipeline_uses_fused", "pddc_pipeline_process",
                 "pddc_pipeline_push_host", "pddc_pipeline_time_stage0", "pddc_pipeline_push_host_async",
                 "pddc_pipeline_wait_ticket", "pddc_pipeline_wait", "pddc_host_alloc", "pddc_host_free"):
        getattr(L, name).restype = C.c_int
    _ddc = L
    return L


def set_tunable(name: str, value: int):
    """process-wide development knob of the launchers (pddc_set_tunable)"""
    check(ddc_lib().pddc_set_tunable(name.encode(), int(value)))


def get_tunable(name: str) -> int:
    v = C.c_int(0)
    check(ddc_lib().pddc_get_tunable(name.encode(), C.byref(v)))
    return int(v.value)


def check(rc: int) -> int:
    if rc < 0:
        raise PddcError(rc, ddc_lib().pddc_last_error().decode(errors="replace"))
    return rc


class Pipeline:
    """Thin handle over pddc_pipeline_* (include/perseus_ddc.h)."""

    def __init__(self, stages, device: int = 0, mix: bool = False, taps_fp16: bool = False,
                 no_fast: bool = False, out_packed: bool = False):
        import numpy as np
        L = ddc_lib()
        self._taps = [np.ascontiguousarray(st[1], dtype=np.float32) for st in stages]
        arr = (StageDesc * len(stages))()
        for i, st in enumerate(stages):
            arr[i].decim = int(st[0])
            arr[i].interp = int(st[2]) if len(st) > 2 and st[2] else 0
            arr[i].ntaps = int(self._taps[i].size)
            arr[i].taps = self._taps[i].ctypes.data_as(C.POINTER(C.c_float))
        flags = (PDDC_F_MIX if mix else 0) | (PDDC_F_TAPS_FP16 if taps_fp16 else 0) | \
                (PDDC_F_NO_FAST if no_fast else 0) | (PDDC_F_OUT_PACKED24 if out_packed else 0)
        self.out_packed = out_packed
        h = C.c_void_p()
        check(L.pddc_pipeline_create(C.byref(h), device, arr, len(stages), flags))
        self._h = h
        self.decim = L.pddc_pipeline_total_decim(h)

    @classmethod
    def from_handle(cls, handle, out_packed: bool = False):
        """Wrap a pddc_pipeline* made by the library itself (Comm.bcast_pipeline)."""
        self = cls.__new__(cls)
        self._taps = []
        self.out_packed = out_packed
        self._h = handle
        self.decim = ddc_lib().pddc_pipeline_total_decim(handle)
        return self

    def close(self):
        if getattr(self, "_h", None):
            ddc_lib().pddc_pipeline_destroy(self._h)
            self._h = None

    __del__ = close

    def reset(self):
        check(ddc_lib().pddc_pipeline_reset(self._h))

    def seek(self, abs_sample: int):
        """Zero history, stream positioned at absolute input sample `abs_sample` (NCO and
        decimation phases of a stream that started at 0): time-chunk sharding, shard.py."""
        check(ddc_lib().pddc_pipeline_seek(self._h, int(abs_sample)))

    def set_freg(self, freg: int):
        check(ddc_lib().pddc_pipeline_set_freg(self._h, freg & 0xFFFFFFFF))

    def set_center_freq(self, hz: float):
        check(ddc_lib().pddc_pipeline_set_center_freq(self._h, float(hz)))

    @property
    def freg(self) -> int:
        return int(ddc_lib().pddc_pipeline_get_freg(self._h))

    @property
    def phase_offset(self) -> int:
        """phase(n) = n*freg + phase_offset (mod 2^32); non-zero only after a retune mid-stream."""
        return int(ddc_lib().pddc_pipeline_get_phase_offset(self._h))

    @property
    def fused(self) -> bool:
        return bool(ddc_lib().pddc_pipeline_uses_fused(self._h))

    @property
    def stage0_reads_packed(self) -> bool:
        return bool(ddc_lib().pddc_pipeline_stage0_reads_packed(self._h))

    def fused_pair(self, nsamples: int) -> int:
        """0: stages 0 and 1 are separate kernels; 1: k_fir8's fused pair; 2: k_fir_i8x's (matrix cores, NCO in the taps)"""
        return int(ddc_lib().pddc_pipeline_uses_fused_pair(self._h, nsamples))

    def set_option(self, name: str, value: int):
        """kernel selection as API state: no_i8, i8x, i8x_pair, i8x_plain, i8x_blocks, no_fuse2, fuse3"""
        check(ddc_lib().pddc_pipeline_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_int(0)
        check(ddc_lib().pddc_pipeline_get_option(self._h, name.encode(), C.byref(v)))
        return int(v.value)

    def fused_cascade(self, nsamples: int) -> bool:
        """stages 0, 1 and 2 as one kernel for a batch of this size"""
        return bool(ddc_lib().pddc_pipeline_uses_fused_cascade(self._h, nsamples))

    def on_i8(self, nsamples: int) -> int:
        """stage 0 of a batch of nsamples runs on the int8 matrix cores: 0 no (k_fir8), 2 k_fir_i8x (plain, or the NCO folded
        into the taps; 1 was round 3's k_fir_i8, retired)"""
        return int(ddc_lib().pddc_pipeline_stage0_on_i8(self._h, nsamples))

    def check(self, stream: int = 0):
        """wait for `stream`; raises if a kernel of this pipeline flagged a failure"""
        check(ddc_lib().pddc_pipeline_check(self._h, stream))

    def place_buffers(self, d_in: int, max_nsamples: int, stream: int = 0):
        """allocate the pipeline's own inter-stage buffers away (in HBM extent class) from the batch they are made from"""
        check(ddc_lib().pddc_pipeline_place_buffers(self._h, d_in, max_nsamples, stream))

    def set_overlap(self, enable: bool = True):
        """the last stage (behind k_fir8's fused pair or first stage) is held back and rides along with the NEXT batch's
        first-stage launch as extra thread blocks; fence() launches what is still held back (see perseus_ddc.h)"""
        check(ddc_lib().pddc_pipeline_set_overlap(self._h, 1 if enable else 0))

    def fence(self, stream: int = 0):
        """launches the tail overlap mode still holds back on `stream` (call before reading the last output)"""
        check(ddc_lib().pddc_pipeline_fence(self._h, stream))

    def max_output(self, n: int) -> int:
        return int(ddc_lib().pddc_pipeline_max_output(self._h, n))

    def process_ptr(self, d_in: int, nsamples: int, d_out: int, out_cap: int, stream: int = 0) -> int:
        n = C.c_size_t(0)
        check(ddc_lib().pddc_pipeline_process(self._h, d_in, nsamples, d_out, out_cap, C.byref(n), stream))
        return n.value

    def process(self, packed_u8, out_f32=None, stream=None):
        """packed_u8: torch uint8 CUDA tensor; returns torch float32 tensor [n_out, 2]."""
        import torch
        ns = packed_u8.numel() // 6
        cap = self.max_output(ns) + 1
        st = stream if stream is not None else torch.cuda.current_stream(packed_u8.device).cuda_stream
        if self.out_packed:                       # 6 bytes per output sample
            out_u8 = torch.empty(6 * cap + 16, dtype=torch.uint8, device=packed_u8.device)
            n = self.process_ptr(packed_u8.data_ptr(), ns, out_u8.data_ptr(), cap, st)
            self.fence(st)
            return out_u8[:6 * n]
        if out_f32 is None:
            out_f32 = torch.empty((cap, 2), dtype=torch.float32, device=packed_u8.device)
        n = self.process_ptr(packed_u8.data_ptr(), ns, out_f32.data_ptr(), out_f32.numel() // 2, st)
        self.fence(st)                            # overlap mode: the caller is about to read this output
        return out_f32[:n]

    def push_host(self, packed_np):
        import numpy as np
        b = np.ascontiguousarray(packed_np, dtype=np.uint8)
        ns = b.size // 6
        cap = self.max_output(ns) + 1
        out = np.empty((cap, 2), dtype=np.float32)
        n = C.c_size_t(0)
        check(ddc_lib().pddc_pipeline_push_host(self._h, b.ctypes.data, ns, out.ctypes.data, cap, C.byref(n)))
        return out[:n.value]

    def push_host_async(self, h_in: int, nsamples: int, h_out: int, out_cap: int):
        """Raw-pointer form (pinned buffers from host_alloc): returns (n_out, ticket)."""
        n, t = C.c_size_t(0), C.c_int(-1)
        check(ddc_lib().pddc_pipeline_push_host_async(self._h, h_in, nsamples, h_out, out_cap, C.byref(n), C.byref(t)))
        return n.value, t.value

    def push_synth_async(self, seed: int, byte_offset: int, nsamples: int, h_out: int, out_cap: int):
        """Device-generated LCG batch (no H2D): returns (n_out, ticket)."""
        n, t = C.c_size_t(0), C.c_int(-1)
        check(ddc_lib().pddc_pipeline_push_synth_async(self._h, seed & 0xFFFFFFFF, byte_offset, nsamples, h_out,
                                                       out_cap, C.byref(n), C.byref(t)))
        return n.value, t.value

    def next_output(self, n: int) -> int:
        return int(ddc_lib().pddc_pipeline_next_output(self._h, n))

    def wait_ticket(self, ticket: int):
        check(ddc_lib().pddc_pipeline_wait_ticket(self._h, ticket))

    def wait(self):
        check(ddc_lib().pddc_pipeline_wait(self._h))

    def workspace_size(self, max_nsamples: int) -> int:
        """Bytes of caller-provided device memory the inter-stage buffers need for batches up to max_nsamples."""
        return int(ddc_lib().pddc_pipeline_workspace_size(self._h, max_nsamples))

    def set_workspace(self, d_ws: int | None, nbytes: int = 0, max_nsamples: int = 0):
        """Inter-stage buffers from the caller's memory (pddc_pipeline_set_workspace); None: own allocations again."""
        check(ddc_lib().pddc_pipeline_set_workspace(self._h, d_ws, nbytes, max_nsamples))

    def save_state(self) -> bytes:
        """The stream state as one blob (histories, decimation phases, sample counter, NCO)."""
        L = ddc_lib()
        n = L.pddc_pipeline_state_size(self._h)
        buf = C.create_string_buffer(n)
        used = C.c_size_t(0)
        check(L.pddc_pipeline_save_state(self._h, buf, n, C.byref(used)))
        return buf.raw[:used.value]

    def restore_state(self, blob: bytes):
        check(ddc_lib().pddc_pipeline_restore_state(self._h, C.create_string_buffer(blob, len(blob)), len(blob)))

    def schedule(self, nsamples: int) -> dict:
        """Tile schedule of the fused stage-0 kernel for a batch of nsamples."""
        o = (C.c_int * 5)()
        check(ddc_lib().pddc_pipeline_schedule(self._h, nsamples, o))
        return {"tile": o[0], "ntiles": o[1], "nblocks": o[2], "S": o[3], "K": o[4]}

    def time_stage0_inline(self, enable: bool):
        """Bracket the stage-0 kernel of every process() with HIP events (read with stage0_time)."""
        check(ddc_lib().pddc_pipeline_time_stage0_inline(self._h, 1 if enable else 0))

    def stage0_time(self):
        """-> (average ms of the stage-0 kernel over the process() calls since enabled, number of calls)."""
        ms, n = C.c_float(0), C.c_int(0)
        check(ddc_lib().pddc_pipeline_stage0_time(self._h, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def time_stage0(self, d_in: int, nsamples: int, d_out: int, iters: int, stream: int = 0) -> float:
        ms = C.c_float(0)
        check(ddc_lib().pddc_pipeline_time_stage0(self._h, d_in, nsamples, d_out, iters, stream, C.byref(ms)))
        return float(ms.value)


def _stage_array(stages):
    """[(D, taps[, L])] -> (StageDesc array, the numpy arrays that own the taps)."""
    import numpy as np
    keep = [np.ascontiguousarray(st[1], dtype=np.float32) for st in stages]
    arr = (StageDesc * max(len(stages), 1))()
    for i, st in enumerate(stages):
        arr[i].decim = int(st[0])
        arr[i].interp = int(st[2]) if len(st) > 2 and st[2] else 0
        arr[i].ntaps = int(keep[i].size)
        arr[i].taps = keep[i].ctypes.data_as(C.POINTER(C.c_float))
    return arr, keep


def plan_pack(stages, freg: int = 0, flags: int = 0) -> bytes:
    """pddc_plan_pack: the flat buffer a configuration broadcast carries."""
    L = ddc_lib()
    arr, _keep = _stage_array(stages)
    need = L.pddc_plan_pack(arr, len(stages), freg & 0xFFFFFFFF, flags, None, 0)
    if need == 0:
        raise PddcError(PDDC_EINVAL, L.pddc_last_error().decode(errors="replace"))
    buf = C.create_string_buffer(need)
    used = L.pddc_plan_pack(arr, len(stages), freg & 0xFFFFFFFF, flags, buf, need)
    assert used == need
    return buf.raw


def plan_unpack(raw: bytes):
    """pddc_plan_unpack -> {"freg", "flags", "stages": [(D, taps, L)]}."""
    import numpy as np
    L = ddc_lib()
    buf = C.create_string_buffer(raw, len(raw))
    arr = (StageDesc * 4)()
    n, freg, flags = C.c_int(0), C.c_uint32(0), C.c_uint32(0)
    check(L.pddc_plan_unpack(buf, len(raw), arr, C.byref(n), C.byref(freg), C.byref(flags)))
    stages = []
    for i in range(n.value):
        t = np.ctypeslib.as_array(arr[i].taps, shape=(arr[i].ntaps,)).copy()
        stages.append((arr[i].decim, t, arr[i].interp))
    return {"freg": freg.value, "flags": flags.value, "stages": stages}


class Comm:
    """One RCCL communicator rank on one GPU (pddc_comm_*, include/perseus_ddc.h): the
    C library calls librccl itself, torch is not involved."""

    def __init__(self, handle):
        self._h = handle
        L = ddc_lib()
        self.rank, self.size, self.device = L.pddc_comm_rank(handle), L.pddc_comm_size(handle), \
            L.pddc_comm_device(handle)

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(PDDC_COMM_ID_BYTES)
        check(ddc_lib().pddc_comm_get_unique_id(buf))
        return buf.raw

    @classmethod
    def init_rank(cls, nranks: int, rank: int, uid: bytes, device: int):
        h = C.c_void_p()
        check(ddc_lib().pddc_comm_init_rank(C.byref(h), nranks, rank, C.create_string_buffer(uid, len(uid)), device))
        return cls(h)

    @classmethod
    def init_all(cls, devices):
        n = len(devices)
        hs = (C.c_void_p * n)()
        check(ddc_lib().pddc_comm_init_all(hs, n, (C.c_int * n)(*devices)))
        return [cls(C.c_void_p(hs[i])) for i in range(n)]

    def close(self):
        if getattr(self, "_h", None):
            ddc_lib().pddc_comm_destroy(self._h)
            self._h = None

    def bcast(self, d_ptr: int, nbytes: int, root: int = 0, stream: int = 0):
        check(ddc_lib().pddc_comm_bcast(self._h, d_ptr, nbytes, root, stream))

    def bcast_bytes(self, raw, root: int = 0) -> bytes:
        """Host-level broadcast of a byte string whose length every rank knows."""
        buf = C.create_string_buffer(raw, len(raw))
        check(ddc_lib().pddc_comm_bcast_host(self._h, buf, len(raw), root))
        return buf.raw

    def bcast_pipeline(self, stages=None, freg: int = 0, flags: int = 0, root: int = 0) -> "Pipeline":
        """Root passes the plan; every rank returns a Pipeline built from the broadcast."""
        L = ddc_lib()
        h = C.c_void_p()
        if self.rank == root:
            arr, _keep = _stage_array(stages)
            check(L.pddc_comm_bcast_pipeline(self._h, root, arr, len(stages), freg & 0xFFFFFFFF, flags, C.byref(h)))
        else:
            check(L.pddc_comm_bcast_pipeline(self._h, root, None, 0, 0, 0, C.byref(h)))
        return Pipeline.from_handle(h, out_packed=bool(flags & PDDC_F_OUT_PACKED24))

    def max_f64(self, v: float) -> float:
        x = C.c_double(v)
        check(ddc_lib().pddc_comm_allreduce_max_f64(self._h, C.byref(x)))
        return x.value

    def barrier(self):
        check(ddc_lib().pddc_comm_barrier(self._h))

    def gather(self, d_send: int, nbytes: int, d_recv: int, root: int = 0, stream: int = 0):
        check(ddc_lib().pddc_comm_gather(self._h, d_send, nbytes, d_recv, root, stream))

    def gather_async(self, d_send: int, nbytes: int, d_recv: int, root: int = 0, after_stream: int = 0):
        check(ddc_lib().pddc_comm_gather_async(self._h, d_send, nbytes, d_recv, root, after_stream))

    def gather_fence(self, stream: int = 0):
        check(ddc_lib().pddc_comm_gather_fence(self._h, stream))

    def gather_wait(self):
        check(ddc_lib().pddc_comm_gather_wait(self._h))


class Gang:
    """pddc_gang: the pipelines of one GPU that stream together share one launch chain (include/perseus_ddc.h;
    the reference's eight descriptors behind one poll thread, perseus-sdr.c:43-47, 736-774)."""

    def __init__(self, device: int = 0):
        h = C.c_void_p()
        check(ddc_lib().pddc_gang_create(C.byref(h), device))
        self._h = h

    def push_async(self, items, nsamples: int):
        """items: dicts with pipe (Pipeline), h_out (pinned pointer), out_cap and either h_packed (pinned pointer) or
        seed + byte_offset (on-device source).  -> ([(n_out, ticket)], number of members that shared launches)."""
        arr = (GangItem * len(items))()
        for a, it in zip(arr, items):
            a.pipe = it["pipe"]._h
            a.h_packed = it.get("h_packed")
            a.seed = it.get("seed", 0) & 0xFFFFFFFF
            a.byte_offset = it.get("byte_offset", 0)
            a.h_out = it["h_out"]
            a.out_capacity = it["out_cap"]
        ng = C.c_int()
        check(ddc_lib().pddc_gang_push_async(self._h, arr, len(items), nsamples, C.byref(ng)))
        return [(a.n_out, a.ticket) for a in arr], ng.value

    def close(self):
        if getattr(self, "_h", None):
            ddc_lib().pddc_gang_destroy(self._h)
            self._h = None

    __del__ = close


class PinnedBuffer:
    """nbytes of pinned host memory (pddc_host_alloc) viewed as a numpy uint8 array."""

    def __init__(self, nbytes: int):
        import numpy as np
        p = C.c_void_p()
        check(ddc_lib().pddc_host_alloc(C.byref(p), nbytes))
        self.ptr, self.nbytes = p.value, nbytes
        self.array = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(self.ptr))

    def free(self):
        if getattr(self, "ptr", None):
            self.array = None
            ddc_lib().pddc_host_free(self.ptr)
            self.ptr = None

    __del__ = free


def unpack24_f32(packed_u8, stream=None):
    """torch uint8 CUDA tensor -> float32 [ns, 2] via the HIP kernel."""
    import torch
    ns = packed_u8.numel() // 6
    out = torch.empty((ns, 2), dtype=torch.float32, device=packed_u8.device)
    st = stream if stream is not None else torch.cuda.current_stream(packed_u8.device).cuda_stream
    check(ddc_lib().pddc_unpack24_f32(packed_u8.data_ptr(), ns, out.data_ptr(), st))
    return out


def unpack24_i32(packed_u8, stream=None):
    import torch
    ns = packed_u8.numel() // 6
    out = torch.empty((ns, 2), dtype=torch.int32, device=packed_u8.device)
    st = stream if stream is not None else torch.cuda.current_stream(packed_u8.device).cuda_stream
    check(ddc_lib().pddc_unpack24_i32(packed_u8.data_ptr(), ns, out.data_ptr(), st))
    return out


def pack24_f32(x_f32, stream=None):
    """torch float32 CUDA tensor [ns, 2] -> uint8 wire bytes [6*ns] via the HIP kernel."""
    import torch
    ns = x_f32.numel() // 2
    out = torch.empty(6 * ns + 16, dtype=torch.uint8, device=x_f32.device)
    st = stream if stream is not None else torch.cuda.current_stream(x_f32.device).cuda_stream
    check(ddc_lib().pddc_pack24_f32(x_f32.data_ptr(), ns, out.data_ptr(), st))
    return out[:6 * ns]


def measure_copy(d_dst: int, d_src: int, nbytes: int, iters: int = 20, stream: int = 0) -> float:
    """Average milliseconds of a device-to-device copy of nbytes (pddc_measure_copy)."""
    ms = C.c_float(0)
    check(ddc_lib().pddc_measure_copy(d_dst, d_src, nbytes, iters, stream, C.byref(ms)))
    return float(ms.value)


def synth_lcg(nbytes: int, seed: int = 12345, byte_offset: int = 0, device="cuda:0", stream=None):
    import torch
    out = torch.empty(nbytes, dtype=torch.uint8, device=device)
    st = stream if stream is not None else torch.cuda.current_stream(out.device).cuda_stream
    check(ddc_lib().pddc_synth_lcg(out.data_ptr(), nbytes, seed & 0xFFFFFFFF, byte_offset, st))
    return out


# --------------------------------------------------------------------------
# drop-in perseus_* API (include/perseus-sdr.h, include/perseus-amd-ext.h)
# --------------------------------------------------------------------------
PERSEUS_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p)


class EepromProdId(C.Structure):
    _pack_ = 1
    _fields_ = [("sn", C.c_uint16), ("prodcode", C.c_uint16), ("hwrel", C.c_uint8),
                ("hwver", C.c_uint8), ("signature", C.c_uint8 * 6)]


class AmdConfig(C.Structure):
    _fields_ = [("mode", C.c_int), ("source", C.c_int), ("lcg_seed", C.c_uint32),
                ("file_path", C.c_char_p), ("pace", C.c_int), ("gpu_device", C.c_int),
                ("batch_samples", C.c_uint32), ("drop_every", C.c_int), ("max_buffers", C.c_uint64),
                ("ep_packet_size", C.c_int), ("cpu_source", C.c_int), ("fault_script", C.c_char_p)]


class AmdStats(C.Structure):
    _fields_ = [("delivered", C.c_uint64), ("dropped", C.c_uint64), ("timeouts", C.c_uint64),
                ("dead_transfers", C.c_uint64), ("transfers", C.c_uint64), ("bytes_received", C.c_uint64),
                ("adc_samples", C.c_uint64), ("batches", C.c_uint64), ("gpu_device", C.c_int),
                ("gpu_source", C.c_int), ("peak_receivers_in_flight", C.c_int), ("ganged_batches", C.c_uint64),
                ("buffers_in_place", C.c_uint64), ("buffers_gathered", C.c_uint64)]


_sdr = None


def sdr_lib() -> C.CDLL:
    """Load libperseus-sdr.so (the perseus_* API); raises if not built."""
    global _sdr
    if _sdr is not None:
        return _sdr
    if not os.path.exists(SDR_LIB):
        raise FileNotFoundError(f"{SDR_LIB} is missing: run __graft_entry__.build()")
    ddc_lib()                      # same HIP runtime ordering as above
    L = C.CDLL(SDR_LIB)
    vp = C.c_void_p
    L.perseus_set_debug.argtypes = [C.c_int]
    L.perseus_set_debug.restype = None
    L.perseus_init.restype = C.c_int
    L.perseus_exit.restype = C.c_int
    L.perseus_open.argtypes = [C.c_int]
    L.perseus_open.restype = vp
    L.perseus_close.argtypes = [vp]
    L.perseus_firmware_download.argtypes = [vp, C.c_char_p]
    L.perseus_get_product_id.argtypes = [vp, C.POINTER(EepromProdId)]
    L.perseus_set_attenuator.argtypes = [vp, C.c_uint8]
    L.perseus_set_attenuator_in_db.argtypes = [vp, C.c_int]
    L.perseus_get_attenuator_values.argtypes = [vp, C.POINTER(C.c_int), C.c_uint]
    L.perseus_set_attenuator_n.argtypes = [vp, C.c_int]
    L.perseus_set_adc.argtypes = [vp, C.c_int, C.c_int]
    L.perseus_set_ddc_center_freq.argtypes = [vp, C.c_double, C.c_int]
    L.perseus_start_async_input.argtypes = [vp, C.c_uint32, PERSEUS_CALLBACK, vp]
    L.perseus_stop_async_input.argtypes = [vp]
    L.perseus_set_sampling_rate.argtypes = [vp, C.c_int]
    L.perseus_set_sampling_rate_n.argtypes = [vp, C.c_uint]
    L.perseus_get_sampling_rates.argtypes = [vp, C.POINTER(C.c_int), C.c_uint]
    L.perseus_is_preserie.argtypes = [vp, C.POINTER(C.c_int)]
    L.perseus_errorstr.restype = C.c_char_p
    L.perseus_amd_get_config.argtypes = [vp, C.POINTER(AmdConfig)]
    L.perseus_amd_set_config.argtypes = [vp, C.POINTER(AmdConfig)]
    L.perseus_amd_get_freg.argtypes = [vp]
    L.perseus_amd_get_freg.restype = C.c_uint32
    L.perseus_amd_get_sampling_rate.argtypes = [vp]
    L.perseus_amd_get_frontendctl.argtypes = [vp]
    L.perseus_amd_get_sioctl.argtypes = [vp]
    L.perseus_amd_buffers_delivered.argtypes = [vp]
    L.perseus_amd_buffers_delivered.restype = C.c_uint64
    L.perseus_amd_buffers_dropped.argtypes = [vp]
    L.perseus_amd_buffers_dropped.restype = C.c_uint64
    L.perseus_amd_source_running.argtypes = [vp]
    L.perseus_amd_get_stats.argtypes = [vp, C.POINTER(AmdStats)]
    L.perseus_amd_get_retune_log.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_int]
    L.perseus_amd_get_plan.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                       C.POINTER(C.POINTER(C.c_float))]
    L.perseus_amd_get_plan_interp.argtypes = [vp, C.POINTER(C.c_int)]
    L.perseus_amd_get_plan.restype = C.c_int
    L.perseus_amd_get_plan_interp.restype = C.c_int
    L.perseus_amd_plan_for_rate.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                            C.POINTER(C.POINTER(C.c_float))]
    L.perseus_amd_plan_for_rate.restype = C.c_int
    L.perseus_amd_effective_batch.argtypes = [vp]
    L.perseus_amd_effective_batch.restype = C.c_uint32
    L.perseus_amd_set_batch.argtypes = [vp, C.c_uint32]
    L.perseus_amd_set_batch.restype = C.c_int
    _sdr = L
    return L


def api_plan(rate: int):
    """The decimation plan the drop-in API builds for perseus_set_sampling_rate(rate) (perseus-sdr.c:776-892 selects the
    rate; the plan and its taps are this library's): [(decim, taps float32, interp)].  perseus_amd_plan_for_rate: no
    descriptor, no perseus_init / perseus_exit -- receivers the process has open are left alone."""
    import numpy as np
    L = sdr_lib()
    dec, nt, it = (C.c_int * 4)(), (C.c_int * 4)(), (C.c_int * 4)()
    n = L.perseus_amd_plan_for_rate(int(rate), None, dec, nt, it, None)
    if n < 0:
        raise RuntimeError(f"perseus_amd_plan_for_rate({rate}): {n}")
    taps = [np.zeros(nt[i], np.float32) for i in range(n)]
    arr = (C.POINTER(C.c_float) * 4)(*([t.ctypes.data_as(C.POINTER(C.c_float)) for t in taps] + [None] * (4 - n)))
    L.perseus_amd_plan_for_rate(int(rate), None, dec, nt, it, arr)
    return [(int(dec[i]), taps[i], int(it[i])) for i in range(n)]
