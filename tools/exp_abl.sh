cd libperseus-sdr_amd
cp libperseus_ddc.so /tmp/keep.so
for f in abl_*.so; do
cp $f libperseus_ddc.so
for wl in d8_127 c320; do echo -n "$f $wl: "; (cd .. && python bench.py --no-cpu --workload $wl --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'])"); done
done
cp /tmp/keep.so libperseus_ddc.so
