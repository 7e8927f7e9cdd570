#!/bin/bash
# kernel trace of one bench workload: per-kernel stats and the gaps between consecutive kernels of the last steps
# usage on the GPU box: bash tools/trace_gaps.sh <workload> <outdir> [bench args]
WL=$1; OUT=$2; shift 2
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --workload $WL --no-cpu "$@" > $OUT/bench.json 2> $OUT/bench.err
cat $(find $OUT/prof -name "*kernel_stats.csv" | head -1) | cut -c1-200 | head -8
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
tail = rows[-24:]
for a, b in zip(tail, tail[1:]):
    print(f'{a["Kernel_Name"][:44]:44s} {int(a["End_Timestamp"]) - int(a["Start_Timestamp"]):8d} ns, then gap {int(b["Start_Timestamp"]) - int(a["End_Timestamp"]):7d} ns')
PY
head -c 200 $OUT/bench.json; echo
