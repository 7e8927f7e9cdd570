#!/bin/bash
# same-box A/B of whole library builds (libperseus-sdr_amd/ab_<tag>.so, made on the build machine): each is copied over
# libperseus_ddc.so in this scratch copy and bench.py runs the workload; two rounds.
# usage on the GPU box: bash tools/ab_libs.sh <workload> tag1 tag2 ...     (PDDC_* variables pass through)
WL=$1; shift
cp libperseus-sdr_amd/libperseus_ddc.so libperseus-sdr_amd/ab_keep.so
for round in 1 2; do for t in "$@"; do
  cp libperseus-sdr_amd/ab_$t.so libperseus-sdr_amd/libperseus_ddc.so
  echo -n "$WL $t: "
  python bench.py --workload $WL --no-cpu --steps 200 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['frac'], d['placement']['chosen']['ms'], d['placement']['first_come_ms'], d['verified']['ok'])"
done; done
cp libperseus-sdr_amd/ab_keep.so libperseus-sdr_amd/libperseus_ddc.so
