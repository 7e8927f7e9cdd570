#!/bin/bash
# Round-3 evidence pass for the 255-tap /8 kernel: the SAME counters and a power / clock trace
# for the 127-tap and the 255-tap kernel side by side, plus two cheap experiments (R = 4 tiles with three and four
# blocks per CU).  Usage on the GPU box: bash tools/r03_evidence255.sh <outdir>
OUT=${1:-gpurun_out/r03_255}
mkdir -p $OUT
export TMPDIR=/tmp
one() { python bench.py --no-cpu --no-verify --steps 200 --warmup 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'])"; }
echo "== step ms / roofline fraction, same box =="
echo "d8_127 R=8 (default):        $(one --workload d8_127)"
echo "d8_255 R=8 512 blocks:       $(one --workload d8_255)"
echo "d8_255 R=4 512 blocks:       $(PDDC_FIR8_R=4 one --workload d8_255)"
echo "d8_255 R=4 768 blocks:       $(PDDC_FIR8_R=4 PDDC_FIR8_BLOCKS=768 one --workload d8_255)"
echo "d8_255 R=4 1024 blocks:      $(PDDC_FIR8_R=4 PDDC_FIR8_BLOCKS=1024 one --workload d8_255)"
for wl in d8_127 d8_255; do
  bash tools/pmc_collect.sh $OUT/pmc_$wl --workload $wl --steps 20 --warmup 3 > $OUT/pmc_$wl.log 2>&1
  echo "== PMC $wl =="
  grep -A28 "k_fir8" $OUT/pmc_$wl/pmc_summary.txt | grep -E "k_fir8|SQ_WAVE_CYCLES|SQ_WAIT_INST_ANY|SQ_WAIT_ANY|SQ_ACTIVE_INST_VALU|SQ_ACTIVE_INST_LDS|SQ_INSTS_VALU |SQ_BUSY_CYCLES|SQ_LDS_BANK|SQ_LDS_IDX|SQ_INST_CYCLES_SMEM|SQ_ACTIVE_INST_VMEM|GRBM_GUI|SQ_INSTS_LDS|SQ_ACTIVE_INST_ANY|SQ_WAIT_INST_LDS"
  (timeout 60 python bench.py --no-cpu --no-verify --workload $wl --steps 60000 --warmup 5 > $OUT/power_bench_$wl.txt 2>&1) &
  BP=$!
  sleep 12
  for i in $(seq 1 16); do
    kill -0 $BP 2>/dev/null || break
    rocm-smi --showclocks --showpower 2>&1 | grep -iE "sclk|power \(W\)" | sed 's/=//g' | tr '\n' ' '; echo
    sleep 0.5
  done > $OUT/power_trace_$wl.txt
  wait $BP
  echo "== power / sclk while $wl runs =="
  head -12 $OUT/power_trace_$wl.txt | cut -c1-160
done
