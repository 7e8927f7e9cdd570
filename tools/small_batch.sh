# k_fir8 at small batches (BASELINE config 5's low end): persistent grid size.  gpurun -- bash tools/small_batch.sh [workload]
WL=${1:-d8_127}
for cfg in "20 64" "20 128" "20 256" "21 64" "21 128" "21 256" "21 512" "22 128" "22 170" "22 256" "22 512" "23 256" "23 340" "23 512" "24 512"; do
  set -- $cfg
  echo -n "$WL 2^$1 blocks=$2: "
  PDDC_FIR8_BLOCKS=$2 python bench.py --workload $WL --log2n $1 --no-cpu --steps 2000 --warmup 50 2>/dev/null | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['verified']['ok'])"
done
