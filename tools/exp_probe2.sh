for wl in c320 d8_255; do PDDC_PROBE_VERBOSE=1 python bench.py --no-cpu --workload $wl --steps 20 --warmup 5 2>&1 | grep -E "^\[blk\]" > gpurun_out/blk_$wl.txt; done
