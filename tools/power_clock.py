"""Power and shader clock while each first-stage form runs (rocm-smi polled from a side thread, 3 s of back-to-back launches of
2^28 samples per form): is the plateau of NOTEBOOK R5.7 the 1400 W cap?  usage: python tools/power_clock.py"""
import importlib, os, re, subprocess, sys, threading, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from i8x_time import taps, lowpass, pkg
dev = torch.device("cuda:0")
ns = 1 << 28
api = [(d, t) for d, t, _l in pkg.api_plan(250000)][:2]
cases = [("idle", None, False, {}),
         ("unpack only (k_unpack24)", "unpack", False, {}),
         ("plain 127 (72 matrix instr. per wave-tile)", [(8, taps("d8_127"))], False, {}),
         ("vector kernel, 127 taps (k_fir8)", [(8, taps("d8_127"))], False, {"no_i8": 1}),
         ("tuned 32 (72)", [api[0]], True, {}),
         ("tuned 127 (108)", [(8, taps("d8_127"))], True, {}),
         ("tuned 255 (216)", [(8, taps("d8_255"))], True, {}),
         ("pair 32/41 on k_fir_i8x", api, True, {"i8x_pair_max_log2": 28}),
         ("pair 32/41 on k_fir8", api, True, {})]
samples = []
stop = [False]


def poll():
    while not stop[0]:
        try:
            o = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            p = re.search(r"Power \(W\):\s*([0-9.]+)", o) or re.search(r"Socket Power.*?:\s*([0-9.]+)", o)
            c = re.search(r"sclk clock level.*?\((\d+)Mhz\)", o)
            m = re.search(r"mclk clock level.*?\((\d+)Mhz\)", o)
            f = re.search(r"fclk clock level.*?\((\d+)Mhz\)", o)
            samples.append((time.time(), float(p.group(1)) if p else None, int(c.group(1)) if c else None,
                            int(m.group(1)) if m else None, int(f.group(1)) if f else None))
        except Exception as e:                      # noqa: BLE001
            samples.append((time.time(), None, None, None, None))
        time.sleep(0.05)


d_in = pkg.synth_lcg(6 * ns, 12345, 0, dev)
st = torch.cuda.current_stream(dev).cuda_stream
th = threading.Thread(target=poll, daemon=True)
th.start()
for name, stages, mix, opts in cases:
    t0 = time.time()
    if stages is None:
        time.sleep(3.0)
        ms = 0.0
    else:
        if stages == "unpack":
            out = torch.empty((ns, 2), dtype=torch.float32, device=dev)
            run = lambda: pkg.check(pkg.ddc_lib().pddc_unpack24_f32(d_in.data_ptr(), ns, out.data_ptr(), st))
            pipe = None
        else:
            pipe = pkg.Pipeline(stages, mix=mix)
            for k, v in opts.items():
                pipe.set_option(k, v)
            if mix:
                pipe.set_center_freq(7.1e6)
            out = torch.empty((pipe.max_output(ns) + 8, 2), dtype=torch.float32, device=dev)
            run = lambda: pipe.process_ptr(d_in.data_ptr(), ns, out.data_ptr(), out.shape[0], st)
        for _ in range(300):
            run()
        torch.cuda.synchronize()
        t0 = time.time()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 0
        while time.time() - t0 < 3.0:
            for _ in range(200):
                run()
            n += 200
            torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        if pipe:
            pipe.close()
        del out
    t1 = time.time()
    sel = [s for s in samples if t0 + 0.5 <= s[0] <= t1 and s[1] is not None]
    pw = [s[1] for s in sel]
    ck = [s[2] for s in sel if s[2]]
    mk = [s[3] for s in sel if s[3]]
    fk = [s[4] for s in sel if s[4]]
    print(f"{name:44s} {ms:.4f} ms/launch   power {np.mean(pw) if pw else float('nan'):7.1f} W (max {max(pw) if pw else 0:.0f})   "
          f"sclk {np.mean(ck) if ck else float('nan'):6.0f} MHz (min {min(ck) if ck else 0})   mclk {np.mean(mk) if mk else float('nan'):5.0f}   fclk {np.mean(fk) if fk else float('nan'):5.0f}   ({len(sel)} readings)", flush=True)
stop[0] = True
