#!/bin/bash
# The measurement pass of a round, one script (rounds 2-4 each had their own rNN_final.sh / collect_rNN.sh):
#   on the GPU box:   gpurun --timeout 2400 -- bash tools/evidence.sh run r06      -> gpurun_out/r05_final/
#   here, afterwards: bash tools/evidence.sh collect r06                           -> profiles/r05/, profiles/pmc_traffic.json
# The kernels must not change after `run`: profiles/pmc_traffic.json records the SHA-256 of the kernel sources it was
# measured on and bench.py refuses the traffic figure for any other source.
set -u
MODE=${1:-run}; R=${2:-r06}
F=gpurun_out/${R}_final; P=profiles/$R
line() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    rf = d["roofline"]
    print(f'{sys.argv[2]:22s} {d["value"]:10.1f} MS/s (first come {d.get("value_first_come")})  {d["ms_per_step"]:.4f} ms/step  kernel {rf.get("kernel_ms")} ms  '
          f'frac {rf["frac"]}  of copy ceiling {rf.get("frac_of_copy_ceiling")}  traffic {"yes" if rf.get("traffic") else "none"}  '
          f'verified {(d.get("verified") or {}).get("ok")}  dtype {d["dtype"]}  | {rf["kernel"][:60]}')
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
if [ "$MODE" = run ]; then
  mkdir -p $F; export TMPDIR=/tmp
  # 1. HBM traffic of the dominant kernel of every workload (separate --pmc passes), first: bench.py then carries it
  bash tools/pmc_traffic.sh $F/pmc_traffic > $F/pmc_traffic.log 2>&1
  cp $F/pmc_traffic/pmc_traffic.json profiles/pmc_traffic.json
  # 2. the suite as the driver runs it, and smoke
  timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $F/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $F/pytest_gpu.log
  python -c "import __graft_entry__ as g; g.smoke()" > $F/smoke.log 2>&1
  # 3. the bench line with the driver's arguments, then every workload
  python bench.py --gpus 1 --steps 20 --warmup 5 > $F/bench_driver_args.json 2> $F/bench_driver_args.err
  for wl in d8_127 d8_255 c320 c320_fixture unpack; do
    timeout 300 python bench.py --no-cpu --workload $wl --steps 40 --warmup 10 2>$F/bench_$wl.err | tail -1 > $F/bench_$wl.json
  done
  timeout 300 python bench.py --no-cpu --workload d8_255 --taps-fp16 --steps 40 --warmup 10 2>$F/bench_d8_255_fp16.err | tail -1 > $F/bench_d8_255_fp16.json
  # 4. rocprofv3 kernel stats + the timed region's trace of the headline command
  bash tools/trace_gaps.sh d8_127 $F/trace_d8_127 --steps 200 --warmup 5 > $F/trace_d8_127.txt 2>&1
  cp $(find $F/trace_d8_127/prof -name "*kernel_stats.csv" | head -1) $F/kernel_stats_d8_127.csv 2>/dev/null
  bash tools/trace_gaps.sh c320 $F/trace_c320 --steps 200 --warmup 5 > $F/trace_c320.txt 2>&1
  # (round 6) the pair's kernel under its walks with the write side in four arena slots, the config-5 sweep on the shipped kernels
  python tools/walk_probe.py pair > $F/walk_probe_pair.txt 2>&1
  python tools/sweep_config5.py --out $F/sweep_config5.json > $F/sweep_config5.txt 2>&1
  # 5. every first-stage form on this box, the ten rate plans, the C hosts
  python tools/state_2p28.py > $F/state_2p28.txt 2>&1
  python tools/plan_rates.py --log2n 28 > $F/plan_rates.txt 2>&1
  python tools/plan_rates.py --log2n 28 --overlap > $F/plan_rates_overlap.txt 2>&1
  bash tools/api_receivers.sh > $F/api_receivers.txt 2>&1
  libperseus-sdr_amd/perseus_multi_bench -n 28 -s 200 > $F/multi_bench_c_host.txt 2>&1
  libperseus-sdr_amd/perseus_multi_bench -n 28 -s 100 -c -G >> $F/multi_bench_c_host.txt 2>&1
  # 6. issue mix / matrix-pipe counters of the headline kernel and the tuned forms
  for c in d8_127 d8_127+nco d8_255+nco; do bash tools/pmc_i8x.sh $F/pmc_i8x_${c/+/_} $c > /dev/null 2>&1; done
  rm -rf $F/trace_*/prof $F/pmc_traffic/*_SIZE
  tail -3 $F/pytest_gpu.log; tail -1 $F/smoke.log
  line $F/bench_driver_args.json "driver args"
  for wl in d8_127 d8_255 d8_255_fp16 c320 c320_fixture unpack; do line $F/bench_$wl.json $wl; done
  cat $F/pmc_traffic/pmc_traffic.json
elif [ "$MODE" = collect ]; then
  mkdir -p $P
  cp $F/bench_driver_args.json $P/z_final_bench.json
  for wl in d8_127 d8_255 d8_255_fp16 c320 c320_fixture unpack; do [ -s $F/bench_$wl.json ] && cp $F/bench_$wl.json $P/z_bench_$wl.json; done
  (line $F/bench_driver_args.json "driver args"; for wl in d8_127 d8_255 d8_255_fp16 c320 c320_fixture unpack; do line $F/bench_$wl.json $wl; done) > $P/z_bench_legs.txt
  cp $F/kernel_stats_d8_127.csv $P/z_final_kernel_stats.csv
  cp $F/pmc_traffic/pmc_traffic.json profiles/pmc_traffic.json
  python3 - <<'PY'
import json, subprocess
d = json.load(open("profiles/pmc_traffic.json"))
head = subprocess.check_output(["git", "rev-parse", "--short=12", "HEAD"], text=True).strip()
dirty = bool(subprocess.check_output(["git", "status", "--porcelain", "--", "libperseus-sdr_amd/csrc"], text=True).strip())
d["provenance"]["commit"] = head + (" + uncommitted changes under csrc/ (committed next)" if dirty else "")
json.dump(d, open("profiles/pmc_traffic.json", "w"), indent=1)
PY
  cp profiles/pmc_traffic.json $P/z_pmc_traffic_all_workloads.json
  [ -f $F/sweep_config5.json ] && cp $F/sweep_config5.json $P/sweep_config5.json && grep -v amdgpu.ids $F/sweep_config5.txt > $P/sweep_config5.txt
  [ -f $F/walk_probe_pair.txt ] && grep -v amdgpu.ids $F/walk_probe_pair.txt > $P/z_walk_probe_pair.txt
  for t in state_2p28 plan_rates plan_rates_overlap api_receivers; do [ -f $F/$t.txt ] && grep -v amdgpu.ids $F/$t.txt > $P/z_$t.txt; done
  for t in d8_127 d8_127_nco d8_255_nco; do [ -f $F/pmc_i8x_$t/pmc_summary.txt ] && cp $F/pmc_i8x_$t/pmc_summary.txt $P/z_pmc_summary_i8x_$t.txt; done
  for w in d8_127 c320; do
    (echo "rocprofv3 --kernel-trace --stats of: python3 bench.py --workload $w --no-cpu --steps 200 --warmup 5 (tools/trace_gaps.sh)"; grep -E "k_fir8|k_fir_i8|timed region|then gap|last 200" $F/trace_$w.txt | cut -c1-200) > $P/z_trace_$w.txt
  done
  grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Lib" $F/multi_bench_c_host.txt > $P/z_multi_bench_c_host.txt
  (grep -E "passed|failed|rc=" $F/pytest_gpu.log | tail -2; tail -1 $F/smoke.log) > $P/z_pytest_gpu_and_smoke.txt
  cat $P/z_bench_legs.txt $P/z_pytest_gpu_and_smoke.txt
fi
