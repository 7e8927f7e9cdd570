#!/bin/bash
# Round-2 final measurement pass on the GPU box (outputs under gpurun_out/r02_final/; copy what is to be
# judged into profiles/r02/).  The kernels must not change after this: profiles/pmc_traffic.json records the
# SHA-256 of the kernel source it was measured on and bench.py refuses the number for any other source.
OUT=gpurun_out/r02_final
mkdir -p $OUT
export TMPDIR=/tmp
# traffic first: the bench lines below print roofline.traffic only if profiles/pmc_traffic.json was measured on THIS kernel source
bash tools/pmc_traffic.sh $OUT/pmc_traffic > $OUT/pmc_traffic.log 2>&1
cp $OUT/pmc_traffic/pmc_traffic.json profiles/pmc_traffic.json
bash tools/gpu_round.sh r02_final > $OUT/gpu_round.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err
bash tools/pmc_collect.sh $OUT/pmc_127 --workload d8_127 --steps 20 --warmup 3 > $OUT/pmc_127.log 2>&1
bash tools/pmc_collect.sh $OUT/pmc_255 --workload d8_255 --steps 20 --warmup 3 > $OUT/pmc_255.log 2>&1
python tools/sweep.py > $OUT/sweep.log 2>&1; cp gpurun_out/sweep.json $OUT/sweep.json
python tools/plan_rates.py > $OUT/plan_rates.txt 2>&1
python tools/pcie_rate.py > $OUT/pcie_rate.txt 2>&1
PDDC_BENCH_GATHER_C320=1 python bench.py --no-cpu --gather --steps 20 --warmup 5 > $OUT/bench_gather_1rank.json 2>/dev/null
PERSEUS_AMD_PACE=0 PERSEUS_AMD_MODE=ddc libperseus-sdr_amd/perseus_plumbing -N 8 -s 250000 -o none -t 2 -d 0 > $OUT/plumbing_N8.txt 2>&1
PDDC_PLACEMENT=0 python tools/placement_probe.py --mode input --workload c320 --arena-gib 144 > $OUT/placement_input_c320.txt 2>&1
for i in 1 2 3; do python tools/placement_probe.py --mode matrix --arena-gib 192; done > $OUT/arena_map.txt 2>&1
bash tools/bench_repeat.sh 5 > $OUT/bench_repeat_d8_127.txt 2>&1
bash tools/bench_repeat.sh 3 --workload c320 > $OUT/bench_repeat_c320.txt 2>&1
libperseus-sdr_amd/perseus_multi_bench -n 28 -s 200 > $OUT/multi_bench_c_host.txt 2>&1
libperseus-sdr_amd/perseus_multi_bench -n 28 -s 200 -c >> $OUT/multi_bench_c_host.txt 2>&1
libperseus-sdr_amd/perseus_multi_bench -n 28 -s 100 -c -G >> $OUT/multi_bench_c_host.txt 2>&1
bash tools/small_batch_default.sh > $OUT/small_batches.txt 2>&1
bash tools/trace_gaps.sh d8_127 $OUT/trace_d8_127 --steps 200 --warmup 5 > $OUT/trace_d8_127.txt 2>&1
bash tools/trace_gaps.sh c320 $OUT/trace_c320 --steps 200 --warmup 5 > $OUT/trace_c320.txt 2>&1
cat $OUT/gpu_round.log | tail -25
cat $OUT/bench_driver_args.json
cat $OUT/pmc_traffic/pmc_traffic.json
tail -3 $OUT/plumbing_N8.txt
