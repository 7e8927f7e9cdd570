#!/bin/bash
# per-kernel durations and the gaps between consecutive kernels of one rate plan: tools/plan_trace.sh <rate> [plan_rates.py args]
R=${1:-1600000}; shift
export TMPDIR=/tmp
O=gpurun_out/plan_trace_$R
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/plan_rates.py --rates $R --log2n 28 --iters 20 "$@" > $O/log.txt 2>&1
grep rate $O/log.txt | cut -c1-140
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs head -6 | cut -c1-160
python3 - "$(find $O/kt -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))[-9:]
for a, b in zip(rows[:-1], rows[1:]):
    print(a["Kernel_Name"][:44], int(a["End_Timestamp"]) - int(a["Start_Timestamp"]), "ns, then gap", int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
PY
rm -rf $O/kt
