# the library's own grid choice at small batches.  gpurun -- bash tools/small_batch_default.sh
for WL in d8_127 d8_255 c320; do for n in 20 21 22 23 24 26; do
  echo -n "$WL 2^$n: "
  python bench.py --workload $WL --log2n $n --no-cpu --steps 2000 --warmup 50 2>/dev/null | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['verified']['ok'])"
done; done
