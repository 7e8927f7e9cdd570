#!/bin/bash
# same-box A/B of library builds (libperseus-sdr_amd/ab_<tag>.so made on the build machine): each is copied over
# libperseus_ddc.so in this scratch copy and bench.py --workload d8_255 is run; usage: bash tools/ab_i8.sh tag1 tag2 ...
cp libperseus-sdr_amd/libperseus_ddc.so libperseus-sdr_amd/ab_keep.so
for round in 1 2; do for t in "$@"; do
  cp libperseus-sdr_amd/ab_$t.so libperseus-sdr_amd/libperseus_ddc.so
  echo -n "$t: "
  python bench.py --workload d8_255 --no-cpu --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['frac'], d['placement']['chosen']['ms'], d['placement']['first_come_ms'], d['verified']['ok'])"
done; done
cp libperseus-sdr_amd/ab_keep.so libperseus-sdr_amd/libperseus_ddc.so
