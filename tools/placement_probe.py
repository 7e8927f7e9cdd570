#!/usr/bin/env python3
"""Does a kernel's speed depend on WHERE its buffers lie in HBM?  (It does: NOTEBOOK.md rounds 1-3 5(u), profiles/r02/i_placement_*.txt,
profiles/r03/f_placement_rule.txt.)  One script for the experiments that led to the placement rule; every mode times the
real pipeline (2^28 samples a launch, HIP events, untimed launches first) on explicit (input, output) addresses.

  --mode map     input at --in-gib inside ONE arena, output scanned over the arena in --step-mib steps (where is it slow?)
  --mode matrix  input slots x output slots, 8 GiB apart (the table pddc_arena_search ranks)
  --mode input   the output fixed, the input moved from slot to slot (a read-dominated kernel: does the input matter?)
  --mode allocs  no arena: --count separate allocations as outputs, in allocation order (what a plain hipMalloc gives)
  --mode small   one placed pair, --count pipeline objects created one after the other (do taps / histories matter?)

usage on the GPU box: python tools/placement_probe.py --mode map [--workload d8_127] [--arena-gib 48] [--in-gib 0,7,20]"""
import argparse
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

pkg = importlib.import_module("libperseus-sdr_amd")
G, M, NS = 1 << 30, 1 << 20, 1 << 28


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["map", "matrix", "input", "allocs", "small"], required=True)
    ap.add_argument("--workload", default="d8_127")
    ap.add_argument("--arena-gib", type=int, default=48)
    ap.add_argument("--in-gib", default="0", help="map: input offsets inside the arena, GiB, comma separated")
    ap.add_argument("--step-mib", type=int, default=256)
    ap.add_argument("--count", type=int, default=8)
    ap.add_argument("--launches", type=int, default=24)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev).cuda_stream
    wl = bench.workload_def(a.workload)

    def make_pipe():
        p = pkg.Pipeline(wl["stages"], mix=wl["mix"])
        if wl["mix"]:
            p.set_freg(wl["freg"])
        return p

    pipe = make_pipe()
    rows = pipe.max_output(NS) + 8
    out_bytes = rows * 8

    def fill(ptr):
        pkg.check(pkg.ddc_lib().pddc_synth_lcg(ptr, 6 * NS, 12345, 0, st))

    def ms(ip, op, p=None, warm=30):
        p = p or pipe
        for _ in range(warm):
            p.process_ptr(ip, NS, op, rows, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.launches):
            p.process_ptr(ip, NS, op, rows, st)
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / a.launches

    if a.mode == "allocs":
        src = torch.empty(6 * NS, dtype=torch.uint8, device=dev)
        fill(src.data_ptr())
        outs = [torch.empty(out_bytes, dtype=torch.uint8, device=dev) for _ in range(a.count)]
        ms(src.data_ptr(), outs[0].data_ptr(), warm=150)
        for k, o in enumerate(outs):
            print(f"output allocation {k} @ {o.data_ptr():#x} ({(o.data_ptr() - src.data_ptr()) / G:+.2f} GiB from the input): "
                  f"{ms(src.data_ptr(), o.data_ptr()):.4f} ms", flush=True)
        return
    arena = torch.empty(a.arena_gib * G, dtype=torch.uint8, device=dev)
    base, nslot = arena.data_ptr(), a.arena_gib // 8
    print(f"arena of {a.arena_gib} GiB @ {base:#x}, workload {a.workload}")
    fill(base)
    ms(base, base + 2 * G, warm=150)
    if a.mode == "map":
        for g in [int(v) for v in a.in_gib.split(",")]:
            ip = base + g * G
            fill(ip)
            line = []
            for k in range(a.arena_gib * 1024 // a.step_mib):
                off = k * a.step_mib * M
                if off + out_bytes > a.arena_gib * G or (off < g * G + 6 * NS and off + out_bytes > g * G):
                    line.append("  . ")
                    continue
                line.append(f"{ms(ip, base + off, warm=6):.3f}"[1:])
            print(f"input at +{g} GiB; output at +k*{a.step_mib} MiB, ms:\n  " + " ".join(line), flush=True)
    elif a.mode in ("matrix", "input"):
        in_slots = list(range(nslot)) if a.mode == "input" else sorted({0, nslot // 3, 2 * nslot // 3})
        out_slots = [nslot // 2] if a.mode == "input" else list(range(nslot))
        for i in in_slots:
            fill(base + i * 8 * G)
            row = [ms(base + i * 8 * G, base + o * 8 * G + 2 * G) for o in out_slots]
            print(f"input in slot {i:2d} (+{8 * i:3d} GiB); output slots {out_slots[0]}..{out_slots[-1]} (+2 GiB): "
                  + " ".join(f"{v:.3f}"[1:] for v in row), flush=True)
    else:                                   # small: the pipeline's own little buffers, allocated at different times
        table = {o: ms(base, base + o * 8 * G + 2 * G) for o in range(1, nslot)}
        best = min(table, key=table.get)
        print("output slots:", " ".join(f"{o}:{v:.4f}" for o, v in table.items()), "-> slot", best)
        keep = []
        for k in range(a.count):
            keep.append(torch.empty((17 + 5 * k) * M, dtype=torch.uint8, device=dev))      # something in between
            p = make_pipe()
            print(f"pipeline object {k}: {ms(base, base + best * 8 * G + 2 * G, p):.4f} ms", flush=True)
            p.close()


if __name__ == "__main__":
    main()
