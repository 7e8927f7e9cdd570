#!/bin/bash
# Round-2 evidence pass for the 255-tap /8 kernel: PMC counters of
# k_fir8<32,8,...> (same counter set as profiles/r01/v8_pmc_summary.txt) + rocm-smi power/clock
# trace while that workload runs.  Usage on the GPU box: tools/r02_evidence255.sh <outdir>
OUT=${1:-gpurun_out/r02_255}
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --no-cpu --workload d8_255 > $OUT/bench_d8_255.json 2> $OUT/bench_d8_255.err
cat $OUT/bench_d8_255.json
bash tools/pmc_collect.sh $OUT/pmc --workload d8_255 --steps 20 --warmup 3 > $OUT/pmc_collect.log 2>&1
grep -A40 "k_fir8" $OUT/pmc/pmc_summary.txt | head -60
# power / clock trace
(timeout 100 python bench.py --no-cpu --workload d8_255 --steps 20000 --warmup 5 > $OUT/power_bench.txt 2>&1) &
BP=$!
for i in $(seq 1 50); do
  kill -0 $BP 2>/dev/null || break
  rocm-smi --showclocks --showpower 2>&1 | grep -iE "sclk|power \(W\)" | sed 's/=//g' | tr '\n' ' '; echo
  sleep 0.5
done > $OUT/power_trace_d8_255.txt
wait $BP
tail -1 $OUT/power_bench.txt | cut -c1-300
cat $OUT/power_trace_d8_255.txt | head -40
