#!/bin/bash
# the plan legs of tools/r04_final.sh again (the 2 MS/s plan changed to 10 * 4 after the third pass; no kernel source did)
OUT=gpurun_out/r04_final
mkdir -p $OUT
python tools/plan_rates.py --log2n 28 > $OUT/plan_rates.txt 2>&1
python tools/plan_rates.py --log2n 28 --overlap > $OUT/plan_rates_overlap.txt 2>&1
python tools/plan_rates.py --log2n 22 --warm-s 0.2 > $OUT/plan_rates_2p22.txt 2>&1
python tools/plan_rates.py --log2n 28 --arena-gib 64 > $OUT/plan_rates_placed.txt 2>&1
python tools/plan_rates.py --log2n 28 --arena-gib 64 > $OUT/plan_rates_placed_b.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -1 $OUT/pytest_gpu.log
cut -c1-110 $OUT/plan_rates_placed_b.txt | grep -v amdgpu
