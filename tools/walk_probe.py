"""Does k_fir8's WALK decide its placement sensitivity?  (round 5 review, item 1.)  k_fir_i8x hands its tiles round the
blocks (all CUs in one compact window of the batch) and hardly cares where its write stream lies (1-3 %); k_fir8 gives
every block one contiguous run of S tiles (512 read streams and 512 write streams megabytes apart) and cares 8-13 %.
This script times k_fir8's first-stage kernel at 2^28 samples under different walks -- the schedule tunables
fir8_dyn_pct / fir8_chunk / fir8_walk -- with the write side in the slot right behind the input (first come) and in the
slots +32 / +48 / +64 GiB of ONE arena.
usage (GPU box): python tools/walk_probe.py [case-substring ...]"""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

pkg = importlib.import_module("libperseus-sdr_amd")
NS = 1 << 28
dev = torch.device("cuda:0")


def main():
    sel = sys.argv[1:]
    L = pkg.ddc_lib()
    st = torch.cuda.current_stream(dev).cuda_stream
    slot = 8 << 30
    arena = torch.empty(72 << 30, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    time.sleep(3.0)
    pkg.check(L.pddc_synth_lcg(arena.data_ptr(), 6 * NS, 12345, 0, st))
    # (name, workload, options, tunables)
    walks = [("static+dyn (default)", {}),
             ("static+dyn, walk 0", {"fir8_walk": 0}),
             ("static only", {"fir8_walk": 0, "fir8_dyn_pct": 0}),
             ("round robin K=8", {"fir8_walk": 1, "fir8_chunk": 8}),
             ("round robin K=16", {"fir8_walk": 1, "fir8_chunk": 16}),
             ("round robin K=16 dyn 10", {"fir8_walk": 1, "fir8_chunk": 16, "fir8_dyn_pct": 10}),
             ("round robin K=16 dyn 25", {"fir8_walk": 1, "fir8_chunk": 16, "fir8_dyn_pct": 25}),
             ("round robin K=32", {"fir8_walk": 1, "fir8_chunk": 32}),
             ("round robin K=32 dyn 10", {"fir8_walk": 1, "fir8_chunk": 32, "fir8_dyn_pct": 10}),
             ("round robin K=64 dyn 10", {"fir8_walk": 1, "fir8_chunk": 64, "fir8_dyn_pct": 10})]
    cases = [("vector 127 (no_i8)", "d8_127", {"no_i8": 1}), ("pair c320 k_fir8", "c320", {})]
    slots = [1, 4, 6, 8]
    ref_out = {}
    for rnd in range(3):
        for cname, wlname, opts in cases:
            wl = bench.workload_def(wlname)
            for wname, tun in walks:
                name = f"{cname} | {wname}"
                if sel and not any(s in name for s in sel):
                    continue
                ok = True
                for k, v in tun.items():
                    try:
                        pkg.set_tunable(k, v)
                    except Exception:
                        ok = False
                if not ok:
                    for k in tun:
                        try:
                            pkg.set_tunable(k, -1 if k in ("fir8_dyn_pct", "fir8_walk") else 0)
                        except Exception:
                            pass
                    continue
                pipe = pkg.Pipeline(wl["stages"], mix=wl["mix"])
                if wl["mix"]:
                    pipe.set_freg(wl["freg"])
                for k, v in opts.items():
                    pipe.set_option(k, v)
                cascade = len(wl["stages"]) > 1
                ws = (pipe.workspace_size(NS) + 255) & ~255 if cascade else 0
                res = []
                for o in slots:
                    side = arena.data_ptr() + o * slot + (2 << 30)
                    if cascade:
                        pipe.set_workspace(side, ws, NS)
                    pipe.time_stage0(arena.data_ptr(), NS, side + ws, 30, st)
                    res.append(pipe.time_stage0(arena.data_ptr(), NS, side + ws, 100, st))
                sch = pipe.schedule(NS)
                # the result must not depend on the walk: one batch from zero history against the default walk's
                side = arena.data_ptr() + 1 * slot + (2 << 30)
                if cascade:
                    pipe.set_workspace(side, ws, NS)
                pipe.reset()
                rows = pipe.max_output(NS) + 8
                out = torch.zeros((rows, 2), dtype=torch.float32, device=dev)
                n = pipe.process_ptr(arena.data_ptr(), NS, out.data_ptr(), rows, st)
                pipe.fence(st)
                torch.cuda.synchronize()
                key = cname
                if key not in ref_out:
                    ref_out[key] = out[:n].clone()
                    same = "ref"
                else:
                    d = (out[:n] - ref_out[key]).abs().max().item() / ref_out[key].abs().max().item()
                    same = f"maxdiff/max {d:.2e} n={n}" + (" BAD" if d > 1e-6 or n != ref_out[key].shape[0] else "")
                del out
                print(f"round {rnd} {name:44s} " + " ".join(f"slot{o}: {r:.4f}" for o, r in zip(slots, res)) +
                      f"  S={sch['S']} K={sch['K']} nblk={sch['nblocks']} {same}", flush=True)
                pipe.close()
                for k in tun:
                    pkg.set_tunable(k, -1 if k in ("fir8_dyn_pct", "fir8_walk") else 0)


if __name__ == "__main__":
    main()
