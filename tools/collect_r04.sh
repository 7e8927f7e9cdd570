#!/bin/bash
# copy what tools/r04_final.sh left under gpurun_out/r04_final/ into profiles/r04/ (run here, after the gpurun call)
F=gpurun_out/r04_final; P=profiles/r04
mkdir -p $P
cp $F/bench_driver_args.json $P/m_final_bench.json; cp $F/bench.json $P/m_bench_default_args.json
for w in d8_127 d8_255 c320 c320_fixture unpack; do [ -s $F/legs/bench_$w.json ] && cp $F/legs/bench_$w.json $P/m_bench_$w.json; done
for b in 0 22 24 26 28; do [ -s $F/legs/bench_api250k_$b.json ] && cp $F/legs/bench_api250k_$b.json $P/m_bench_api250k_batch2p$b.json; done
cp $F/bench_legs.txt $P/m_bench_legs.txt
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
for t in plan_rates plan_rates_overlap plan_rates_2p22 plan_rates_place_buffers i8x_sizes i8x_layouts i8x_chunks i8x_plain i8x_plainsizes api_receivers; do
  [ -f $F/$t.txt ] && grep -v amdgpu.ids $F/$t.txt > $P/m_$t.txt
done
(echo "# tools/plan_rates.py --log2n 28 --arena-gib 64: input, inter-stage workspace and output cut from ONE arena, the workspace/output side tried at every 2 GiB behind the input with the plan itself as the probe (probe_ms_every_2GiB in the lines); two runs on one box"; echo "## placed, run 1"; grep -v amdgpu $F/plan_rates_placed.txt; echo "## placed, run 2"; grep -v amdgpu $F/plan_rates_placed_b.txt) > $P/n_plan_rates_placed.txt
for t in pair 127nco 48nco; do [ -f $F/pmc_i8x_$t/pmc_summary.txt ] && cp $F/pmc_i8x_$t/pmc_summary.txt $P/m_pmc_summary_i8x_$t.txt; done
[ -f $F/pmc_i8x_plain127/pmc_summary.txt ] && cp $F/pmc_i8x_plain127/pmc_summary.txt $P/m_pmc_summary_i8x_plain127.txt
(echo "rocprofv3 --kernel-trace of: python3 bench.py --workload c320 --no-cpu --steps 200 --warmup 5 (tools/trace_gaps.sh)"; grep -E "timed region|then gap|last 200" $F/trace_c320.txt) > $P/m_trace_c320.txt
(echo "# bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu, processes on one box: value MS/s, ms/step, kernel ms, frac of 8 TB/s, frac of copy ceiling, chosen slots, (fastest pair, slowest pair, pairs within 3 % of the fastest, pairs tried), verified"; cat $F/bench_repeat_d8_127.txt; echo "# --workload c320, three processes"; cat $F/bench_repeat_c320.txt) > $P/m_bench_repeat.txt
[ -f $F/small_batches.txt ] && (echo "# bench.py --workload W --log2n n --steps 2000: value MS/s, ms/step, kernel ms, verified (tools/small_batch_default.sh)"; cat $F/small_batches.txt) > $P/m_small_batches.txt
[ -f $F/trace_d8_127.txt ] && (echo "rocprofv3 --kernel-trace --stats of: python3 bench.py --workload d8_127 --no-cpu --steps 200 --warmup 5 (tools/trace_gaps.sh); bench line of the same process:"; python3 -c "import json; d=json.load(open('$F/trace_d8_127/bench.json')); print('value', d['value'], 'MS/s, roofline.kernel_ms', d['roofline']['kernel_ms'], '(HIP events over the timed region)')"; echo "whole process (placement probes into slow pairs, settle phase and warm-up included):"; grep -E "k_fir8|k_fir_i8" $F/trace_d8_127.txt | head -1 | cut -c1-160; grep -E "timed region|then gap|last 200" $F/trace_d8_127.txt) > $P/m_trace_d8_127.txt
grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Lib" $F/multi_bench_c_host.txt > $P/m_multi_bench_c_host.txt
grep -E "passed|failed" $F/pytest_gpu.log | tail -1 > $P/m_pytest_gpu_and_smoke.txt; tail -1 $F/smoke.log >> $P/m_pytest_gpu_and_smoke.txt
python3 - <<'PY'
import json, os
for f in ["m_final_bench", "m_bench_default_args", "m_bench_d8_255", "m_bench_c320", "m_bench_c320_fixture", "m_bench_unpack", "m_bench_gather_1rank"]:
    p = f"profiles/r04/{f}.json"
    if not os.path.exists(p):
        print(f, "MISSING"); continue
    r = json.load(open(p))
    print(f, r["value"], r["ms_per_step"], r["roofline"]["kernel_ms"], r["roofline"]["frac"], r["roofline"].get("frac_of_copy_ceiling"),
          "traffic" if r["roofline"]["traffic"] else "NO TRAFFIC", r["verified"]["ok"] if r.get("verified") else None)
print(json.load(open("profiles/pmc_traffic.json"))["provenance"])
PY
cat $P/m_pytest_gpu_and_smoke.txt
