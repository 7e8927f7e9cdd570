#!/bin/bash
# A/B builds of libperseus_ddc.so that differ in -D flags of ddc_kernels.hip, measured on ONE box (boxes differ by
# a few %).   Build here (no GPU):  tools/ab.sh build base: prioU2F0:"-DPDDC_PRIO_U=2 -DPDDC_PRIO_F=0" ...
#            Run on the GPU box:    gpurun -- bash tools/ab.sh run      (WLS="d8_127 d8_255" REPS=3 STEPS=200)
set -u
cd "$(dirname "$0")/../libperseus-sdr_amd"
if [ "${1:-}" = build ]; then
  shift
  make -C csrc >/dev/null || exit 1
  rm -f ab_*.so
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $flags -c csrc/ddc_kernels.hip -o /tmp/ab_k.o 2>/tmp/ab_err.txt &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab_$name.so /tmp/ab_k.o csrc/ddc_fir_i8.o csrc/ddc_pipeline.o csrc/ddc_multi.o -L/opt/rocm/lib -lrccl &&
      echo "built ab_$name.so ($flags)" || { echo "FAILED $name"; tail -5 /tmp/ab_err.txt; }
  done
elif [ "${1:-}" = run ]; then
  cp libperseus_ddc.so /tmp/keep.so
  for rep in $(seq 1 ${REPS:-3}); do
    for f in ab_*.so; do
      cp $f libperseus_ddc.so
      for wl in ${WLS:-d8_127 d8_255}; do
        echo -n "$f $wl: "
        (cd .. && timeout 200 python bench.py --no-cpu --no-verify --workload $wl --steps ${STEPS:-200} --warmup 10 ${EXTRA:-} 2>/dev/null | tail -1 |
          python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'])")
      done
    done
  done
  cp /tmp/keep.so libperseus_ddc.so
else
  echo "usage: $0 build name:flags ... | run"; exit 2
fi
