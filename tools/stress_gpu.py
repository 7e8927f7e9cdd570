# randomized stress of the stream state machine against the oracle: random plans, cuts, NCO words retuned
# between batches (phase-continuous), scheduler settings (PDDC_FIR8_BLOCKS / DYN_PCT / CHUNK / R), caller-provided
# workspaces, checkpoint/restore hops to a fresh pipeline in mid-stream, overlap mode (the last stage carried by the next
# launch), the int8 matrix-core first stage switched on and off between batches (options no_i8 / i8x_plain), and binary16
# tap storage (one case in four).
# Usage: python tools/stress_gpu.py [n]
import sys, os, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import load_taps
pkg = importlib.import_module("libperseus-sdr_amd")
from oracle import oracle as O
dev = torch.device("cuda:0")
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 40
h1, h2, h3 = load_taps("c320_s1_d8_32"), load_taps("c320_s2_d8_64"), load_taps("c320_s3_d5_161")
worst = 0.0
for it in range(n_iter):
    rng = np.random.default_rng(5000 + it)
    g = (rng.standard_normal(12 * 20) / 20).astype(np.float32)
    plans = [[(8, load_taps("d8_127"))], [(8, load_taps("d8_255"))], [(8, h1), (8, h2), (5, h3)], [(8, h1), (8, h2)],
             [(8, h1), (5, h3[:41]), (25, g, 12)], [(10, h3[:77])], [(8, h2), (4, h1)], [(5, h3), (8, h1)],
             [(10, h3[:51]), (5, h3)], [(10, h3[:40])], [(8, load_taps("d8_127")), (8, h2)]]
    stages = plans[int(rng.integers(0, len(plans)))]
    os.environ["PDDC_FIR8_BLOCKS"] = str(int(rng.choice([1, 2, 3, 5, 16, 512])))
    dyn = int(rng.choice([-1, -1, 0, 10, 20, 50, 100]))          # -1: the library's own default schedule
    chunk_k = int(rng.choice([1, 2, 3, 4, 8])) if dyn >= 0 else 0
    pkg.set_tunable("fir8_dyn_pct", dyn)                      # (process-wide launcher knobs: API state, not environment)
    pkg.set_tunable("fir8_chunk", chunk_k)
    # the walk (round 6), drawn from a generator of its own so that the other draws stay those of the earlier runs:
    # the launcher's default / static runs + dynamic tail / chunks handed round the blocks
    pkg.set_tunable("fir8_walk", int(np.random.default_rng(11000 + it).choice([-1, 0, 1, 1])))
    os.environ["PDDC_FIR8_DYN_PCT"], os.environ["PDDC_FIR8_CHUNK"] = str(dyn), str(chunk_k)   # (for the report line below)
    os.environ["PDDC_FIR8_R"] = str(int(rng.choice([4, 8])))
    mix = bool(rng.integers(0, 2))
    freg = int(rng.integers(0, 2**32))
    ns = 8 * int(rng.integers(1, 4096 * 12))
    if rng.integers(0, 2):
        ns = 8192 * int(rng.integers(1, 40))
    packed = O.lcg_bytes(6 * ns, 9000 + it)
    cuts = sorted(set([0, ns] + [8 * int(c) for c in rng.integers(1, max(2, ns // 8), size=4)] +
                      [4096 * int(c) for c in rng.integers(0, max(1, ns // 4096), size=3) if 0 < 4096 * int(c) < ns]))
    # a new tuning word at some of the batch boundaries (the retune path of the drop-in API)
    words = [freg]
    for _ in cuts[1:-1]:
        words.append(int(rng.integers(0, 2**32)) if (mix and rng.integers(0, 3) == 0) else words[-1])
    segs = [(a, w) for k, (a, w) in enumerate(zip(cuts[:-1], words)) if k == 0 or w != words[k - 1]]
    # one case in four with binary16 taps (PDDC_F_TAPS_FP16: k_fir_i8 then reads them as binary16 and quantises them itself);
    # drawn from a generator of its own so that the other cases stay the ones of the earlier runs
    f16 = bool(np.random.default_rng(7000 + it).integers(0, 4) == 0)
    if f16:
        stages = [(s[0], np.asarray(s[1], np.float32).astype(np.float16).astype(np.float32)) + tuple(s[2:]) for s in stages]
    ref = O.ddc_chain_retuned(packed, stages, segs) if mix else O.ddc_chain(packed, stages)
    pipe = pkg.Pipeline(stages, mix=mix, taps_fp16=f16)
    ov = len(stages) > 1 and bool(rng.integers(0, 2))
    if ov:
        pipe.set_overlap(True)
    # caller-provided workspace for the inter-stage buffers (sometimes dropped again in mid-stream), and sometimes a
    # hop to a fresh pipeline through save_state / restore_state between two batches
    use_ws = len(stages) > 1 and bool(rng.integers(0, 2))
    ws = None
    if use_ws:
        need = pipe.workspace_size(ns)
        ws = torch.empty(need + 256, dtype=torch.uint8, device=dev)
        pipe.set_workspace((ws.data_ptr() + 255) & ~255, need, ns)
    hops = 0
    parts = []
    for k, ((a, b), w) in enumerate(zip(zip(cuts[:-1], cuts[1:]), words)):
        pipe.set_freg(w)
        # the first stage's kernel, switched between batches through the pipeline's options (kernel selection is state, not
        # environment): the library's choice / the vector kernels only / the matrix-core kernel's plain form off (tuned only)
        i8 = int(rng.integers(0, 3))
        pipe.fence()
        pipe.set_option("no_i8", 1 if i8 == 1 else 0)
        pipe.set_option("i8x_plain", 0 if i8 == 2 else 1)
        parts.append(pipe.process(torch.from_numpy(packed[6 * a:6 * b].copy()).to(dev)).cpu().numpy().reshape(-1))
        act = int(rng.integers(0, 6))
        if act == 0 and use_ws:
            pipe.set_workspace(None)
            use_ws = False
        elif act == 1:
            blob = pipe.save_state()
            q = pkg.Pipeline(stages, mix=mix, taps_fp16=f16)
            if ov:
                q.set_overlap(True)
            q.restore_state(blob)
            pipe.close()
            pipe = q
            use_ws = False
            hops += 1
    y = np.concatenate(parts) if parts else np.zeros(0, np.float32)
    pipe.close()
    ok = y.size == ref.size
    err = O.rel_err(y, ref) if ok and ref.size else 0.0
    worst = max(worst, err)
    tag = "ok " if ok and err <= 1e-6 else "BAD"
    print(f"{tag} it {it} stages {[(s[0], len(s[1])) for s in stages]} mix {mix} ns {ns} cuts {len(cuts)-1} "
          f"blocks {os.environ['PDDC_FIR8_BLOCKS']} dyn {os.environ.get('PDDC_FIR8_DYN_PCT', 'default')} K {os.environ.get('PDDC_FIR8_CHUNK', 'default')} "
          f"R {os.environ['PDDC_FIR8_R']} f16 {f16} retunes {len(segs) - 1} ws {ws is not None} hops {hops} overlap {ov} err {err:.2e}", flush=True)
    if tag == "BAD":
        sys.exit(1)
print("worst", worst)
