#!/bin/bash
# round 4: the bench legs (usage on the GPU box: bash tools/r04_bench.sh <outdir> [quick])
OUT=${1:-gpurun_out/r04_bench}
mkdir -p $OUT
for wl in d8_127 d8_255 c320 c320_fixture unpack; do
  timeout 300 python bench.py --no-cpu --workload $wl --steps 40 --warmup 10 2>$OUT/bench_$wl.err | tail -1 > $OUT/bench_$wl.json
  python - $OUT/bench_$wl.json $wl <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[2], d["value"], "MS/s", d["ms_per_step"], "ms  frac", d["roofline"]["frac"], "kernel_ms", d["roofline"].get("kernel_ms"),
          d["roofline"]["kernel"], "| verified", (d.get("verified") or {}).get("ok"), "| dtype", d["dtype"], "| ntaps", d["config"].get("ntaps"))
except Exception as e:
    print(sys.argv[2], "FAILED", e, open(sys.argv[1]).read()[-300:])
PY
done
for b in 0 22 24 26 28; do
  timeout 600 python bench.py --workload api250k --api-batch-log2 $b --steps 40 --warmup 10 2>$OUT/bench_api250k_$b.err | tail -1 > $OUT/bench_api250k_$b.json
  python - $OUT/bench_api250k_$b.json $b <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("api250k batch-log2", sys.argv[2], d["value"], "MS/s", d["ms_per_step"], "ms/step  frac", d["roofline"]["frac"], "batch", d["config"]["samples_per_gpu_per_step"], "ntaps", d["config"]["ntaps"])
except Exception as e:
    print("api250k", sys.argv[2], "FAILED", e, open(sys.argv[1]).read()[-300:], open(sys.argv[1].replace(".json", ".err")).read()[-500:])
PY
done
