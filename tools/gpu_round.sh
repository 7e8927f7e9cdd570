#!/bin/bash
# One GPU-box pass: parity tests, smoke, default bench, rocprof kernel stats,
# PMC traffic.  Usage: tools/gpu_round.sh <tag>   (outputs under gpurun_out/<tag>/)
TAG=${1:-round}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
for W in d8_255 c320 unpack; do python bench.py --workload $W --no-cpu 2>/dev/null > $OUT/bench_$W.json; python -c "
import json,sys; d=json.load(open('$OUT/bench_$W.json')); print('$W', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --no-cpu > $OUT/bench_prof.log 2>&1
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null; cat $OUT/kernel_stats.csv
# PMC traffic, separate passes (FETCH_SIZE undercounts 16 B/lane streams by 2x on gfx950: MI355X_MICROARCH.md HBM)
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 bench.py --no-cpu --out-candidates 1 --steps 5 --warmup 1 > $OUT/pmc_$C.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, sys, json
out = sys.argv[1]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for f in glob.glob(f"{out}/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if ("k_fir8" in r["Kernel_Name"] or "k_fir_i8" in r["Kernel_Name"]) and r["Counter_Name"] == c:
                vals.append(float(r["Counter_Value"]))
    res[c] = sum(vals) / len(vals) if vals else None
print("PMC per first-stage kernel launch (k_fir8 or k_fir_i8; KiB units as reported):", res)
if res["FETCH_SIZE"] and res["WRITE_SIZE"]:
    # rocprofv3 reports KiB; FETCH_SIZE x2 correction for wide coalesced reads on gfx950
    traffic = (2 * res["FETCH_SIZE"] + res["WRITE_SIZE"]) * 1024
    print("corrected HBM bytes per launch:", traffic, " algorithmic:", 7 * 2**28)
    json.dump({"FETCH_SIZE_KiB": res["FETCH_SIZE"], "WRITE_SIZE_KiB": res["WRITE_SIZE"],
               "hbm_bytes_per_launch_corrected": traffic, "algorithmic_bytes": 7 * 2**28}, open(f"{out}/pmc_traffic_raw.json", "w"))
PY
