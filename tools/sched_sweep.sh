#!/bin/bash
# tile-scheduler sweep of k_fir8 (127 taps /8, 2^28) under the favourable buffer placement (bench.py's walk)
for dyn in 0 10 20 30 40; do for k in 1 2 4; do
  echo -n "dyn=$dyn K=$k: "
  PDDC_FIR8_DYN_PCT=$dyn PDDC_FIR8_CHUNK=$k python bench.py --no-cpu --no-verify --steps 100 --warmup 10 2>/dev/null | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], min(min(v) for v in d['placement']['step_ms_by_input_slot'].values()) if d['placement'] else None)"
done; done
