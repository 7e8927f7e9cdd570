#!/bin/bash
# Round-4 final measurement pass on the GPU box (outputs under gpurun_out/r04_final/; tools/collect_r04.sh copies what is
# to be judged into profiles/r04/).  The kernels must not change after this: profiles/pmc_traffic.json records the SHA-256
# of the kernel sources it was measured on and bench.py refuses the number for any other source.
OUT=gpurun_out/r04_final
mkdir -p $OUT
export TMPDIR=/tmp
bash tools/pmc_traffic.sh $OUT/pmc_traffic > $OUT/pmc_traffic.log 2>&1
cp $OUT/pmc_traffic/pmc_traffic.json profiles/pmc_traffic.json
bash tools/gpu_round.sh r04_final > $OUT/gpu_round.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err
bash tools/r04_bench.sh $OUT/legs > $OUT/bench_legs.txt 2>&1
python tools/plan_rates.py --log2n 28 > $OUT/plan_rates.txt 2>&1
python tools/plan_rates.py --log2n 28 --overlap > $OUT/plan_rates_overlap.txt 2>&1
python tools/plan_rates.py --log2n 22 --warm-s 0.2 > $OUT/plan_rates_2p22.txt 2>&1
python tools/plan_rates.py --log2n 28 --place > $OUT/plan_rates_place_buffers.txt 2>&1
python tools/plan_rates.py --log2n 28 --arena-gib 64 > $OUT/plan_rates_placed.txt 2>&1
python tools/plan_rates.py --log2n 28 --arena-gib 64 > $OUT/plan_rates_placed_b.txt 2>&1
python tools/i8x_time.py sizes > $OUT/i8x_sizes.txt 2>&1
python tools/i8x_time.py layouts > $OUT/i8x_layouts.txt 2>&1
python tools/i8x_time.py chunks > $OUT/i8x_chunks.txt 2>&1
python tools/i8x_time.py plain > $OUT/i8x_plain.txt 2>&1
python tools/i8x_time.py plainsizes > $OUT/i8x_plainsizes.txt 2>&1
bash tools/pmc_i8x.sh $OUT/pmc_i8x_pair "c320api i8x_pair_max_log2=28" > /dev/null 2>&1
bash tools/pmc_i8x.sh $OUT/pmc_i8x_127nco d8_127+nco > /dev/null 2>&1
bash tools/pmc_i8x.sh $OUT/pmc_i8x_48nco d8_48+nco > /dev/null 2>&1
bash tools/pmc_i8x.sh $OUT/pmc_i8x_plain127 d8_127 > /dev/null 2>&1
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
cat $OUT/bench_legs.txt
cat $OUT/pmc_traffic/pmc_traffic.json
