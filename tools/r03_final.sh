#!/bin/bash
# Round-3 final measurement pass on the GPU box (outputs under gpurun_out/r03_final/; tools/collect_final.sh r03 copies
# what is to be judged into profiles/r03/).  The kernels must not change after this: profiles/pmc_traffic.json records
# the SHA-256 of the kernel source it was measured on and bench.py refuses the number for any other source.
OUT=gpurun_out/r03_final
mkdir -p $OUT
export TMPDIR=/tmp
# traffic first: the bench lines below print roofline.traffic only if profiles/pmc_traffic.json was measured on THIS kernel source
bash tools/pmc_traffic.sh $OUT/pmc_traffic > $OUT/pmc_traffic.log 2>&1
cp $OUT/pmc_traffic/pmc_traffic.json profiles/pmc_traffic.json
bash tools/gpu_round.sh r03_final > $OUT/gpu_round.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err
python tools/plan_rates.py > $OUT/plan_rates.txt 2>&1
python tools/plan_rates.py --overlap > $OUT/plan_rates_overlap.txt 2>&1
PDDC_BENCH_GATHER_C320=1 python bench.py --no-cpu --gather --steps 20 --warmup 5 > $OUT/bench_gather_1rank.json 2>/dev/null
bash tools/api_receivers.sh > $OUT/api_receivers.txt 2>&1
bash tools/bench_repeat.sh 4 > $OUT/bench_repeat_d8_127.txt 2>&1
bash tools/bench_repeat.sh 3 --workload c320 > $OUT/bench_repeat_c320.txt 2>&1
libperseus-sdr_amd/perseus_multi_bench -n 28 -s 200 > $OUT/multi_bench_c_host.txt 2>&1
libperseus-sdr_amd/perseus_multi_bench -n 28 -s 200 -c >> $OUT/multi_bench_c_host.txt 2>&1
libperseus-sdr_amd/perseus_multi_bench -n 28 -s 100 -c -G >> $OUT/multi_bench_c_host.txt 2>&1
bash tools/small_batch_default.sh > $OUT/small_batches.txt 2>&1
bash tools/trace_gaps.sh d8_127 $OUT/trace_d8_127 --steps 200 --warmup 5 > $OUT/trace_d8_127.txt 2>&1
bash tools/trace_gaps.sh c320 $OUT/trace_c320 --steps 200 --warmup 5 > $OUT/trace_c320.txt 2>&1
rm -rf $OUT/prof $OUT/trace_*/prof $OUT/pmc_traffic/*_SIZE $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
tail -25 $OUT/gpu_round.log
cat $OUT/bench_driver_args.json
cat $OUT/pmc_traffic/pmc_traffic.json
