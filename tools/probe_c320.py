# NEEDS: git apply tools/ubench/fir8_probe_and_ablations.patch, build with -DPDDC_CLOCK_PROBE (revert afterwards)
"""Development: per-block start/end stamps of the fused pair kernel (build with -DPDDC_CLOCK_PROBE as
libperseus-sdr_amd/probe_ddc.so; run on the GPU box with PDDC_FIR8_BLOCKS=512|768)."""
import importlib, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "libperseus-sdr_amd", "libperseus_ddc.so")
shutil.copy(lib, "/tmp/keep_ddc.so")
shutil.copy(os.path.join(ROOT, "libperseus-sdr_amd", "probe_ddc.so"), lib)
try:
    import torch
    import bench
    pkg = importlib.import_module("libperseus-sdr_amd")
    wl = bench.workload_def(sys.argv[1] if len(sys.argv) > 1 else "c320")
    dev = torch.device("cuda", 0)
    ns = 1 << 28
    d_in = pkg.synth_lcg(6 * ns, 12345, 0, dev)
    pipe = pkg.Pipeline(wl["stages"], device=0, mix=wl["mix"])
    if wl["mix"]:
        pipe.set_freg(wl["freg"])
    out = torch.empty((pipe.max_output(ns) * 64 + 8, 2), dtype=torch.float32, device=dev)
    for _ in range(3):
        ms = pipe.time_stage0(d_in.data_ptr(), ns, out.data_ptr(), 50)
        print("stage0 ms", ms, flush=True)
finally:
    shutil.copy("/tmp/keep_ddc.so", lib)
