#!/bin/bash
# HBM traffic of the dominant kernel of every bench workload, from separate --pmc passes
# (FETCH_SIZE undercounts 16 B/lane streams by 2x on gfx950: MI355X_MICROARCH.md, HBM section).
# Usage on the GPU box: tools/pmc_traffic.sh [outdir]   -> <outdir>/pmc_traffic.json
OUT=${1:-gpurun_out/pmc_traffic}
mkdir -p $OUT
export TMPDIR=/tmp
# (d8_255_fp16: BASELINE config 5's binary16-stored leg -- the same workload with --taps-fp16)
for W in d8_127 d8_255 d8_255_fp16 c320 c320_fixture unpack; do
  WL=${W%_fp16}; EXTRA=""; [ "$W" != "$WL" ] && EXTRA="--taps-fp16"
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/${W}_$C -- python3 bench.py --no-cpu --no-verify --out-candidates 1 --workload $WL $EXTRA --steps 5 --warmup 1 > $OUT/${W}_$C.log 2>&1
  done
done
python3 - $OUT <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
kern = {"d8_127": "k_fir_i8x", "d8_255": "k_fir_i8x", "d8_255_fp16": "k_fir_i8x", "c320": "k_fir8", "c320_fixture": "k_fir8",
        "unpack": "k_unpack24"}
alg = {"d8_127": 7.0, "d8_255": 7.0, "d8_255_fp16": 7.0, "c320": 6.125, "c320_fixture": 6.125, "unpack": 14.0}      # bytes per input sample of that kernel
res = {}
for w, k in kern.items():
    v = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for f in glob.glob(f"{out}/{w}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if k in r["Kernel_Name"] and r["Counter_Name"] == c:
                    vals.append(float(r["Counter_Value"]))
        v[c] = sum(vals) / len(vals) if vals else None
    if v["FETCH_SIZE"] is not None and v["WRITE_SIZE"] is not None:
        res[w] = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
        print(f"{w:12s} {k:12s} corrected HBM bytes per launch {res[w]:.4e}   algorithmic {alg[w] * 2**28:.4e}")
import hashlib, subprocess
h = hashlib.sha256()
for f in ("ddc_kernels.hip", "ddc_kernels.h", "fir8_block.inc", "ddc_fir_i8.hip", "ddc_dev.h"):
    h.update(open("libperseus-sdr_amd/csrc/" + f, "rb").read())
try:
    commit = subprocess.check_output(["git", "rev-parse", "--short=12", "HEAD"], text=True, stderr=subprocess.DEVNULL).strip()
except Exception:
    commit = "unknown (no .git on the GPU box; tools/evidence.sh collect fills in the commit the pass is collected into)"
# bench.py reports roofline.traffic only while the kernel source and the launch shape are these
res["provenance"] = {"kernel_source_sha16": h.hexdigest()[:16], "log2n": 28, "commit": commit,
                     "kernels": kern, "tool": "tools/pmc_traffic.sh"}
res["note"] = ("HBM bytes per launch of the dominant kernel (2^28 samples): (2*FETCH_SIZE + WRITE_SIZE) KiB from separate "
               "rocprofv3 --pmc passes, FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section (tools/pmc_traffic.sh)")
json.dump(res, open(f"{out}/pmc_traffic.json", "w"), indent=1)
PY
