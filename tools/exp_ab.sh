# A/B two builds of libperseus_ddc.so on ONE box (boxes differ by a few %): ab_old.so vs ab_new.so
cd libperseus-sdr_amd; cp libperseus_ddc.so /tmp/keep.so
for rep in 1 2 3; do for v in old new; do cp ab_$v.so libperseus_ddc.so
for wl in ${WLS:-d8_127 d8_255 c320}; do echo -n "$v $wl: "; (cd .. && timeout 120 python bench.py --no-cpu --workload $wl --steps ${STEPS:-200} --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'])"); done; done; done
cp /tmp/keep.so libperseus_ddc.so
