#!/bin/bash
# tile-scheduler sweep of k_fir8 (127 taps /8, 2^28) under the favourable buffer placement (bench.py's rule);
# several combinations in one process class each; usage on the GPU box: bash tools/sched_sweep.sh [workload]
WL=${1:-d8_127}
for dyn in 20 60 100; do for k in 1 2 4; do
  echo -n "dyn=$dyn K=$k: "
  PDDC_FIR8_DYN_PCT=$dyn PDDC_FIR8_CHUNK=$k python bench.py --workload $WL --no-cpu --steps 100 --warmup 10 2>/dev/null | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['placement']['chosen']['ms'], d['placement']['first_come_ms'], d['verified']['ok'])"
done; done
echo -n "default: "; python bench.py --workload $WL --no-cpu --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['placement']['chosen']['ms'], d['placement']['first_come_ms'])"
