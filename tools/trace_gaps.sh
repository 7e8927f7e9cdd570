#!/bin/bash
# kernel trace of one bench workload: per-kernel stats and the gaps between consecutive kernels of the last steps
# usage on the GPU box: bash tools/trace_gaps.sh <workload> <outdir> [bench args]
WL=$1; OUT=$2; shift 2
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --workload $WL --no-cpu "$@" > $OUT/bench.json 2> $OUT/bench.err
cat $(find $OUT/prof -name "*kernel_stats.csv" | head -1) | cut -c1-200 | head -8
python3 - $OUT <<'PY'
import csv, glob, statistics, sys
f = glob.glob(sys.argv[1] + "/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_fir8" in r["Kernel_Name"] or "k_fir_i8" in r["Kernel_Name"] or "k_unpack24" in r["Kernel_Name"]]
seg = rows[idx[-40]:idx[-30] + 1] if len(idx) >= 40 else rows[-24:]           # inside the timed region
print("consecutive kernels inside the timed region (duration, then the gap to the next launch):")
for a, b in zip(seg, seg[1:]):
    print(f'{a["Kernel_Name"][:44]:44s} {int(a["End_Timestamp"]) - int(a["Start_Timestamp"]):8d} ns, then gap {int(b["Start_Timestamp"]) - int(a["End_Timestamp"]):7d} ns')
last = rows[idx[-200]:idx[-1] + 1] if len(idx) >= 200 else rows
by = {}
for r in last:
    by.setdefault(r["Kernel_Name"][:44], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in by.items():
    print(f"last 200 steps: {k:44s} n={len(v):4d} mean {statistics.mean(v) / 1e3:8.2f} us  min {min(v) / 1e3:8.2f}  max {max(v) / 1e3:8.2f}")
PY
head -c 200 $OUT/bench.json; echo
