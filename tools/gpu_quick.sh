#!/bin/bash
# quick GPU check: parity suite + one bench line per workload (usage: gpurun -- bash tools/gpu_quick.sh)
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for wl in d8_127 d8_255 c320 unpack; do timeout 120 python bench.py --no-cpu --workload $wl --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms'])"; done
