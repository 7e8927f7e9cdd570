for wl in c320 d8_127; do
for R in 4 8; do for B in 512 768 1024; do
echo "== $wl R=$R B=$B"; PDDC_FIR8_R=$R PDDC_FIR8_BLOCKS=$B python bench.py --no-cpu --workload $wl --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
done; done; done
