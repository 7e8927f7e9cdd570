#!/bin/bash
# counters of k_fir_i8x on one case of tools/i8x_time.py: issue mix, waits, LDS conflicts, matrix-pipe busy
# usage on the GPU box: bash tools/pmc_i8x.sh <outdir> <case> [opt=value ...]
set -u
OUT=$1; CASE=$2; shift 2
export TMPDIR=/tmp
mkdir -p "$OUT"
PASSES=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"
 "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU_MFMA_MOPS_I8"
 "SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$OUT/pass$i" -- python3 tools/i8x_time.py one $CASE "$@" > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed (see $OUT/pass$i.log)"
done
python3 - "$OUT" "$CASE $*" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_fir_i8" in r["Kernel_Name"] or "k_fir8" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:64]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/pmc_summary.txt", "w") as fo:
    fo.write("# tools/pmc_i8x.sh " + sys.argv[2] + ": chip-wide sums per launch, mean over the launches of the process\n")
    for k, d in acc.items():
        fo.write(k + "\n")
        for c, v in sorted(d.items()):
            fo.write(f"   {c:34s} n={len(v):3d} mean={sum(v)/len(v):.6g}\n")
print(open(out + "/pmc_summary.txt").read())
PY
rm -rf "$OUT"/pass*/
