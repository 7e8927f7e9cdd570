#!/bin/bash
# the API-side legs of the final pass again (perseus_api.c changed after it: zero-copy delivery); outputs under gpurun_out/r04_api/
OUT=gpurun_out/r04_api
mkdir -p $OUT
bash tools/api_receivers.sh > $OUT/api_receivers.txt 2>&1
for b in 0 22 24 26 28; do
  timeout 600 python bench.py --workload api250k --api-batch-log2 $b --steps 40 --warmup 10 2>$OUT/bench_api250k_$b.err | tail -1 > $OUT/bench_api250k_$b.json
  python - $OUT/bench_api250k_$b.json $b <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("api250k batch-log2", sys.argv[2], d["value"], "MS/s", d["ms_per_step"], "ms/step  batch", d["config"]["samples_per_gpu_per_step"])
PY
done > $OUT/api250k_legs.txt
bash tools/api_trace.sh 22 > $OUT/api_trace_2p22.txt 2>&1
bash tools/api_trace.sh 24 > $OUT/api_trace_2p24.txt 2>&1
cat $OUT/api250k_legs.txt; grep -v "receiver 0" $OUT/api_receivers.txt
