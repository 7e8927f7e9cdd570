#!/bin/bash
# kernel trace of the C client streaming 250 kS/s from the on-device source through the callback API (what bench.py
# --workload api250k times): per-kernel durations and the gaps between consecutive kernels.  usage: tools/api_trace.sh [log2 batch]
B=${1:-24}
export TMPDIR=/tmp PERSEUS_AMD_MODE=ddc PERSEUS_AMD_PACE=0 PERSEUS_AMD_BATCH=$((1 << B))
O=gpurun_out/api_trace_$B
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- libperseus-sdr_amd/perseus_plumbing -s 250000 -f 7100000 -n 2 -b 6144 -o none -a -d 0 -t 600 -B 10,40 > $O/log.txt 2>&1
grep api_bench $O/log.txt
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs head -6 | cut -c1-150
python3 - "$(find $O/kt -name '*kernel_trace.csv' | head -1)" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2: len(rows) // 2 + 13]
t0 = int(rows[0]["Start_Timestamp"])
for a, b in zip(rows[:-1], rows[1:]):
    print(f'{(int(a["Start_Timestamp"]) - t0) / 1e3:9.1f} us  {a["Kernel_Name"][:48]:48s} {(int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3:8.1f} us, next starts {(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3:7.1f} us after its end  (stream {a.get("Stream_Id", "?")})')
PY
rm -rf $O/kt
