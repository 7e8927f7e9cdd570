"""Does the place of the INPUT buffer in HBM matter to a read-dominated kernel (the fused pair of the x320
cascade writes 1/48 of what it reads)?  Candidates 8 GiB apart, the same LCG bytes in each, stage 0 timed
back to back (300 launches after 100 untimed ones).  Usage on the GPU box: python tools/placement_probe8.py [workload]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
pkg = importlib.import_module("libperseus-sdr_amd")
wl = bench.workload_def(sys.argv[1] if len(sys.argv) > 1 else "c320")
ncand = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda", 0)
ns = 1 << 28
pipe = pkg.Pipeline(wl["stages"], device=0, mix=wl["mix"])
if wl["mix"]:
    pipe.set_freg(wl["freg"])
out = torch.empty((ns // 8 + 8, 2), dtype=torch.float32, device=dev)
keep = []
for k in range(ncand):
    d_in = pkg.synth_lcg(6 * ns, 12345, 0, dev)
    pipe.time_stage0(d_in.data_ptr(), ns, out.data_ptr(), 100)
    ms = pipe.time_stage0(d_in.data_ptr(), ns, out.data_ptr(), 300)
    print(f"input candidate {k} at 0x{d_in.data_ptr():x} (+{(d_in.data_ptr() - out.data_ptr()) / 2**30:.1f} GiB from the output): {ms:.4f} ms", flush=True)
    keep.append(d_in)
    try:
        keep.append(torch.empty(8 << 30, dtype=torch.uint8, device=dev))
    except RuntimeError:
        break
