"""debug helper: where does k_fir_i8x differ from the oracle?  python tools/i8x_debug.py <ntaps> <layout> [chunk]"""
import importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("libperseus-sdr_amd")
from oracle import oracle as O
dev = torch.device("cuda:0")
TILE = 8192
ntaps, layout = int(sys.argv[1]), int(sys.argv[2])
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 0
k = np.arange(ntaps) - (ntaps - 1) / 2.0
h = np.sinc(2 * 0.05 * k) * np.hamming(ntaps)
h = (h / h.sum()).astype(np.float32)
if ntaps == 127:
    h = np.fromfile(os.path.join(ROOT, "tests", "golden", "taps_d8_127.f32"), dtype=np.float32)
sizes = [TILE * 3, TILE + 8, 264, 8, 128, 256, TILE * 40 + 4096 + 16, TILE * 600, TILE * 2 - 8, TILE * 257]
cuts = np.concatenate([[0], np.cumsum(sizes)])
packed = O.lcg_bytes(6 * int(cuts[-1]), 2027)
FREG = 381178347
ref = O.ddc_chain(packed, [(8, h)], freg=FREG, mix=True).reshape(-1, 2)
for rep in range(15):
    pipe = pkg.Pipeline([(8, h)], mix=True)
    pipe.set_option("i8x_layout", layout)
    pipe.set_option("i8x_chunk", chunk)
    if os.environ.get("I8X_BLOCKS"):                     # a smaller persistent grid: fewer CUs busy, the chip off its power cap
        pipe.set_option("i8x_blocks", int(os.environ["I8X_BLOCKS"]))
    pipe.set_freg(FREG)
    pos = 0
    for a, b in zip(cuts[:-1], cuts[1:]):
        y = pipe.process(torch.from_numpy(packed[6 * a:6 * b]).to(dev)).cpu().numpy()
        r = ref[pos:pos + y.shape[0]]
        err = np.abs(y - r).max(axis=1)
        bad = np.nonzero(err > 1e-5)[0]
        if bad.size:
            ex = np.abs(y[bad, 0] - r[bad, 0]) > 1e-5
            ey = np.abs(y[bad, 1] - r[bad, 1]) > 1e-5
            if os.environ.get("I8X_DEBUG_WHICH"):
                # which operand did the wrong x take?  x = fma(-v, s, u c) with (u + j v) = ref / (c + j s), LO = c + j s = exp(-j theta)
                m_abs = (pos + bad).astype(np.uint64)
                ph = ((m_abs * np.uint64(8)) * np.uint64(FREG)) & np.uint64(0xFFFFFFFF)
                th = 2.0 * np.pi * ph.astype(np.float64) / 4294967296.0
                c, sn = np.cos(th), -np.sin(th)
                z = (r[bad, 0].astype(np.float64) + 1j * r[bad, 1]) / (c + 1j * sn)
                u, v = z.real, z.imag
                xb = y[bad, 0].astype(np.float64)
                cand = {"u*c (right)": -v * sn + u * c, "u*s": -v * sn + u * sn, "v*c": -v * sn + v * c, "v*s": -v * sn + v * sn,
                        "0": -v * sn, "-v*s only, product = previous output's u*c": None}
                # the same thread's previous output (o - 512 for NPT = 512) and next one
                for name, val in cand.items():
                    if val is not None:
                        print(f"      x_bad == -v*s + {name:12s}: max |diff| {np.abs(xb - val).max():.3e}  median {np.median(np.abs(xb - val)):.3e}")
                for d in (-512, 512, -1024, 1024):
                    j = bad + d
                    ok = (j >= 0) & (j < r.shape[0])
                    if ok.all():
                        mj = (pos + j).astype(np.uint64)
                        phj = ((mj * np.uint64(8)) * np.uint64(FREG)) & np.uint64(0xFFFFFFFF)
                        thj = 2.0 * np.pi * phj.astype(np.float64) / 4294967296.0
                        cj, sj = np.cos(thj), -np.sin(thj)
                        zj = (r[j, 0].astype(np.float64) + 1j * r[j, 1]) / (cj + 1j * sj)
                        for nm, val in ((f"u[{d:+d}]*c[{d:+d}]", zj.real * cj), (f"u*c[{d:+d}]", u * cj), (f"u[{d:+d}]*c", zj.real * c),
                                        (f"u*s[{d:+d}]", u * sj)):
                            print(f"      x_bad == -v*s + {nm:16s}: median |diff| {np.median(np.abs(xb - (-v * sn + val))):.3e}")
            print(f"   x only {int((ex & ~ey).sum())}, y only {int((~ex & ey).sum())}, both {int((ex & ey).sum())}; lanes {np.unique(bad % 64)[:70].tolist()}; o>>6 {np.unique((bad % 1024) >> 6).tolist()}")
            tiles = np.unique(bad // 1024)
            for tl in tiles[:0]:
                sel = bad[bad // 1024 == tl]
                # is the bad I a LATER value of the same post thread (o + 128 i), of this tile or of the block's next tile?
                cand = {}
                for dt in (0, 256, 512):
                    for i in range(0, 8):
                        j = sel + 128 * i + 1024 * dt
                        if j.max() < r.shape[0]:
                            for comp in (0, 1):
                                cand[(dt, i, comp)] = float(np.abs(y[sel, 0] - r[j, comp]).max())
                best = sorted(cand.items(), key=lambda kv: kv[1])[:3]
                print("   closest other values (tile offset, i, comp): ", best, " own |yI|max", float(np.abs(y[sel, 0]).max()))
            for tl in tiles[:0]:
                sel = bad[bad // 1024 == tl]
                for d in (256, 512, -256):
                    j = sel + 1024 * d
                    if j.min() >= 0 and j.max() < r.shape[0]:
                        print(f"   tile {tl}: max|y - ref(tile{d:+d})| = {np.abs(y[sel] - r[j]).max():.3e}; y[:2] {y[sel[:2]].tolist()} ref[:2] {r[sel[:2]].tolist()}")
            print(f"rep {rep} batch {b - a:8d} samples: {bad.size} bad outputs, tiles {tiles[:12]}{'...' if tiles.size > 12 else ''} "
                  f"within-tile idx range {bad.min() % 1024}..{bad.max() % 1024} first bad {bad[:6]} on_i8 {pipe.on_i8(b - a)}")
        pos += y.shape[0]
    pipe.close()
print("done")
