for dyn in 10 20 30 40 50; do for k in 1 2 4; do
  echo -n "d8_255 dyn=$dyn K=$k: "
  PDDC_FIR8_DYN_PCT=$dyn PDDC_FIR8_CHUNK=$k python bench.py --workload d8_255 --no-cpu --no-verify --steps 100 --warmup 10 2>/dev/null | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])"
done; done
