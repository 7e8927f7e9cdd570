#!/bin/bash
# Timeline of the drop-in API with N virtual receivers on one GPU (C client, unpaced, on-device source): kernel trace,
# HIP API trace and copy trace of one run; prints GPU busy share, the kernels of one round and the host cost per HIP call.
# usage (GPU box): bash tools/api_trace.sh <N> <outdir> [rate]
N=$1; OUT=$2; RATE=${3:-250000}
mkdir -p $OUT
export TMPDIR=/tmp PERSEUS_AMD_PACE=0 PERSEUS_AMD_MODE=ddc
rocprofv3 --kernel-trace --hip-runtime-trace --memory-copy-trace --output-format csv -d $OUT/prof -- \
    libperseus-sdr_amd/perseus_plumbing -N $N -s $RATE -o none -t 1 -d 0 > $OUT/run.txt 2>&1
grep -E "receivers:" $OUT/run.txt
python3 - $OUT <<'PY'
import csv, glob, statistics, sys
def load(pat):
    f = glob.glob(sys.argv[1] + "/prof/**/*" + pat, recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
k = sorted(load("kernel_trace.csv"), key=lambda r: int(r["Start_Timestamp"]))
if k:
    mid = k[len(k) // 2: len(k) // 2 + 4000]
    t0, t1 = int(mid[0]["Start_Timestamp"]), int(mid[-1]["End_Timestamp"])
    ev = sorted([(int(r["Start_Timestamp"]), 1) for r in mid] + [(int(r["End_Timestamp"]), -1) for r in mid])
    busy = depth = 0; last = t0
    for t, d in ev:
        if depth > 0: busy += t - last
        depth += d; last = t
    print(f"{len(mid)} kernels over {(t1 - t0) / 1e6:.2f} ms: GPU busy (>=1 kernel resident) {100 * busy / (t1 - t0):.1f} %, "
          f"{(t1 - t0) / len(mid) / 1e3:.2f} us of wall per kernel")
    by = {}
    for r in mid:
        by.setdefault(r["Kernel_Name"][:60], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for n, v in sorted(by.items(), key=lambda x: -sum(x[1])):
        print(f"  {n:60s} n={len(v):5d} mean {statistics.mean(v) / 1e3:7.2f} us  sum {sum(v) / 1e6:7.2f} ms")
    print("  a stretch of the timeline (start offset us, duration us, queue, kernel):")
    for r in mid[2000:2024]:
        print(f'   {(int(r["Start_Timestamp"]) - t0) / 1e3:10.2f} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:7.2f}  q{r.get("Queue_Id", "?"):>3s}  {r["Kernel_Name"][:50]}')
h = load("hip_api_trace.csv")
by = {}
for r in h:
    by.setdefault(r["Function"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("host side, per HIP call:")
for n, v in sorted(by.items(), key=lambda x: -sum(x[1]))[:12]:
    print(f"  {n:36s} n={len(v):6d} mean {statistics.mean(v) / 1e3:7.2f} us  median {statistics.median(v) / 1e3:7.2f}  sum {sum(v) / 1e6:8.2f} ms")
c = load("memory_copy_trace.csv")
if c:
    v = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in c]
    print(f"copies: n={len(v)} mean {statistics.mean(v) / 1e3:.2f} us")
PY
