run() { echo "== $1 dyn=$2 K=$3"; PDDC_FIR8_CHUNK=$3 PDDC_FIR8_DYN_PCT=$2 python bench.py --no-cpu --workload $1 --steps 20 --warmup 5 2>&1 | grep -E "probe\]|metric" | cut -c1-150 | sed 's/"unit.*ms_per_step/ ms_per_step/'; }
run d8_127 0 4; run d8_127 15 4; run d8_255 0 2; run d8_255 25 2; run c320 0 8; run c320 15 8
