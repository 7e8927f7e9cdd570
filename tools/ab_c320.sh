mkdir -p gpurun_out/r02_c320
for rep in 1 2; do
for cfg in "4 512" "4 768" "4 1024" "8 512" "4 256"; do
  set -- $cfg
  PDDC_FIR8_R=$1 PDDC_FIR8_BLOCKS=$2 python bench.py --workload c320 --steps 50 --warmup 5 --no-cpu > gpurun_out/r02_c320/r$1_b$2_$rep.json 2>/dev/null
  python - <<PY
import json
r=json.load(open("gpurun_out/r02_c320/r$1_b$2_$rep.json"))
print("R=$1 blocks=$2", r["value"], r["ms_per_step"], r["roofline"]["kernel_ms"], r["roofline"]["frac"], r["verified"]["ok"])
PY
done
done
