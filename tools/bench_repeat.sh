# several bench runs on one box: how reproducible is the placement search?   gpurun -- bash tools/bench_repeat.sh [n] [bench args]
N=${1:-6}; shift
mkdir -p gpurun_out/rep
for i in $(seq 1 $N); do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu "$@" > gpurun_out/rep/b$i.json 2>gpurun_out/rep/b$i.err; python - <<PY
import json
r=json.load(open("gpurun_out/rep/b$i.json"))
p=r["placement"]
allv=[v for row in p["step_ms_by_input_slot"].values() for v in (row.values() if isinstance(row, dict) else row)] if p else []
print(r["value"], r["ms_per_step"], r["roofline"]["kernel_ms"], r["roofline"]["frac"], r["roofline"]["frac_of_copy_ceiling"], p and p["chosen"], allv and (min(allv), max(allv), sum(v < 1.03 * min(allv) for v in allv), len(allv)), r["verified"]["ok"])
PY
done
