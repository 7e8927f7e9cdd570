#!/bin/bash
# tile-scheduler sweep of the fused pair (x320 cascade, 2^28) under the arena placement: share of the tiles handed
# out dynamically x chunk length (a dynamic chunk of the fused pair starts with one warm-up tile)
for rep in 1 2; do
for cfg in "0 8" "5 8" "10 8" "10 16" "20 16" "10 4" "20 32"; do
  set -- $cfg
  echo -n "dyn=$1 K=$2: "
  PDDC_FIR8_DYN_PCT=$1 PDDC_FIR8_CHUNK=$2 python bench.py --workload c320 --no-cpu --steps 100 --warmup 10 2>/dev/null | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'], d['verified']['ok'])"
done; done
