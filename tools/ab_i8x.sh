#!/bin/bash
# A/B builds of libperseus_ddc.so that differ in -D flags of ddc_fir_i8.hip, measured on ONE box with tools/i8x_time.py.
#   build here (no GPU):  tools/ab_i8x.sh build base: nopost:"-DI8X_ABL_NOPOST" ...
#   run on the GPU box:   gpurun -- bash tools/ab_i8x.sh run "<case> [opt=val ..]" ["<case2> ..."]
set -u
cd "$(dirname "$0")/../libperseus-sdr_amd"
if [ "${1:-}" = build ]; then
  shift
  make -s -C csrc >/dev/null || exit 1
  rm -f ab_*.so
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-slp-vectorize $flags -c csrc/ddc_fir_i8.hip -o /tmp/ab_k.o 2>/tmp/ab_err.txt &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab_$name.so csrc/ddc_kernels.o /tmp/ab_k.o csrc/ddc_pipeline.o csrc/ddc_multi.o -L/opt/rocm/lib -lrccl &&
      echo "built ab_$name.so ($flags)" || { echo "FAILED $name"; tail -5 /tmp/ab_err.txt; }
  done
elif [ "${1:-}" = run ]; then
  shift
  cp libperseus_ddc.so /tmp/keep.so
  for rep in $(seq 1 ${REPS:-2}); do
    for f in ab_*.so; do
      cp $f libperseus_ddc.so
      for c in "$@"; do
        echo -n "$f: "
        (cd .. && timeout 200 python tools/i8x_time.py one $c 2>/dev/null | tail -1)
      done
    done
  done
  cp /tmp/keep.so libperseus_ddc.so
fi
