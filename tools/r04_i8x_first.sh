#!/bin/bash
# round 4, first GPU pass of k_fir_i8x: its parity tests, then a quick timing of the tuned /8 stage and of the API's x320 plan
OUT=gpurun_out/r04a
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_i8x.py tests/test_gpu_i8.py -m gpu -x -q > $OUT/pytest_i8x.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_i8x.log
timeout 600 python tools/i8x_time.py ${1:-} > $OUT/i8x_time_${1:-all}.txt 2>&1; echo "time rc=$?"; cat $OUT/i8x_time_${1:-all}.txt
