#!/bin/bash
# copy what tools/r0N_final.sh left under gpurun_out/r0N_final/ into profiles/ (run here, after the gpurun call)
# usage: bash tools/collect_final.sh [r03]
R=${1:-r03}
F=gpurun_out/${R}_final; P=profiles/$R
cp $F/bench_driver_args.json $P/m_final_bench.json; cp $F/bench.json $P/m_bench_default_args.json
for w in d8_255 c320 unpack; do cp $F/bench_$w.json $P/m_bench_$w.json; done
cp $F/bench_gather_1rank.json $P/m_bench_gather_1rank.json; cp $F/kernel_stats.csv $P/m_final_kernel_stats.csv
cp $F/pmc_traffic/pmc_traffic.json profiles/pmc_traffic.json
# the GPU box has no .git: the commit recorded is the HEAD this pass is collected on top of (the kernel source hash in
# the same record is what bench.py checks, and what ties the numbers to a source)
python3 - <<PY
import json, subprocess
d = json.load(open("profiles/pmc_traffic.json"))
head = subprocess.check_output(["git", "rev-parse", "--short=12", "HEAD"], text=True).strip()
dirty = bool(subprocess.check_output(["git", "status", "--porcelain", "--", "libperseus-sdr_amd/csrc"], text=True).strip())
d["provenance"]["commit"] = head + (" + uncommitted changes under csrc/ (committed next)" if dirty else "")
json.dump(d, open("profiles/pmc_traffic.json", "w"), indent=1)
PY
cp profiles/pmc_traffic.json $P/m_pmc_traffic_all_workloads.json
[ -f $F/pmc_127/pmc_summary.txt ] && cp $F/pmc_127/pmc_summary.txt $P/m_pmc_summary_d8_127.txt
[ -f $F/pmc_255/pmc_summary.txt ] && cp $F/pmc_255/pmc_summary.txt $P/m_pmc_summary_d8_255.txt
[ -f $F/sweep.json ] && cp $F/sweep.json $P/m_sweep_config5.json
grep -v amdgpu.ids $F/plan_rates.txt > $P/m_rate_plans.txt
[ -f $F/plan_rates_overlap.txt ] && grep -v amdgpu.ids $F/plan_rates_overlap.txt > $P/m_rate_plans_overlap.txt
[ -f $F/pcie_rate.txt ] && grep -v amdgpu.ids $F/pcie_rate.txt > $P/m_pcie_rate.txt
[ -f $F/plumbing_N8.txt ] && cp $F/plumbing_N8.txt $P/m_plumbing_N8_ddc.txt
[ -f $F/api_receivers.txt ] && cp $F/api_receivers.txt $P/m_api_receivers.txt
[ -f $F/arena_map.txt ] && grep -E "in@|best" $F/arena_map.txt > $P/k_arena_map.txt
[ -f $F/placement_input_c320.txt ] && grep -E "input candidate" $F/placement_input_c320.txt > $P/k_placement_input_c320.txt
(echo "rocprofv3 --kernel-trace of: python3 bench.py --workload c320 --no-cpu --steps 200 --warmup 5 (tools/trace_gaps.sh)"; grep -E "timed region|then gap|last 200" $F/trace_c320.txt) > $P/k_trace_c320.txt
(echo "# bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu, five processes on one box: value MS/s, ms/step, kernel ms, frac of 8 TB/s, frac of copy ceiling, chosen slots, (fastest pair, slowest pair, pairs within 3 % of the fastest, pairs tried), verified"; cat $F/bench_repeat_d8_127.txt; echo "# --workload c320, three processes"; cat $F/bench_repeat_c320.txt) > $P/m_bench_repeat.txt
[ -f $F/small_batches.txt ] && (echo "# bench.py --workload W --log2n n --steps 2000: value MS/s, ms/step, kernel ms, verified (tools/small_batch_default.sh)"; cat $F/small_batches.txt) > $P/m_small_batches.txt
[ -f $F/trace_d8_127.txt ] && (echo "rocprofv3 --kernel-trace --stats of: python3 bench.py --workload d8_127 --no-cpu --steps 200 --warmup 5 (tools/trace_gaps.sh); bench line of the same process:"; python3 -c "import json; d=json.load(open('$F/trace_d8_127/bench.json')); print('value', d['value'], 'MS/s, roofline.kernel_ms', d['roofline']['kernel_ms'], '(HIP events over the timed region)')"; echo "whole process (placement probes into slow pairs, settle phase and warm-up included):"; grep -E "k_fir8|k_fir_i8" $F/trace_d8_127.txt | head -1 | cut -c1-160; grep -E "timed region|then gap|last 200" $F/trace_d8_127.txt) > $P/m_trace_d8_127.txt
grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Lib" $F/multi_bench_c_host.txt > $P/m_multi_bench_c_host.txt
R=$R python3 - <<'PY'
import json
for f in ["m_final_bench", "m_bench_default_args", "m_bench_d8_255", "m_bench_c320", "m_bench_unpack", "m_bench_gather_1rank"]:
    r = json.load(open(f"profiles/" + __import__("os").environ.get("R", "r03") + f"/{f}.json"))
    print(f, r["value"], r["ms_per_step"], r["roofline"]["kernel_ms"], r["roofline"]["frac"], r["roofline"].get("frac_of_copy_ceiling"),
          "traffic" if r["roofline"]["traffic"] else "NO TRAFFIC", r["verified"]["ok"] if r.get("verified") else None)
print(json.load(open("profiles/pmc_traffic.json"))["provenance"])
PY
grep -E "passed|failed" $F/pytest_gpu.log | tail -1; tail -1 $F/smoke.log
