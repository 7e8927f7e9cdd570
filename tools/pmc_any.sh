#!/bin/bash
# PMC counters (separate passes, kernel trace only) for any python3 command:  tools/pmc_any.sh <outdir> <script.py> [args...]
set -u
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p "$OUT"
PASSES=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$OUT/pass$i" -- python3 "$@" > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed (see $OUT/pass$i.log)"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/pmc_summary.txt", "w") as fo:
    for k, d in acc.items():
        if "synth" in k or "rocclr" in k or "probe" in k:
            continue
        fo.write(k + "\n")
        for c, v in sorted(d.items()):
            fo.write(f"   {c:34s} n={len(v):3d} mean={sum(v)/len(v):.6g}\n")
print(open(out + "/pmc_summary.txt").read())
PY
