#!/bin/bash
# Same-box A/B of two SOURCE versions of ddc_fir_i8.hip: the committed one (HEAD) against the working tree.
#   build here:        tools/ab_src.sh build          -> libperseus-sdr_amd/ab_base.so, ab_new.so
#   run on the box:    gpurun -- bash tools/ab_src.sh run "<case substrings for tools/state_2p28.py>" [rounds]
set -u
cd "$(dirname "$0")/../libperseus-sdr_amd"
FLAGS="--offload-arch=gfx950 -O3 -fPIC -fvisibility=hidden -std=c++17 -fno-slp-vectorize -Icsrc"
if [ "${1:-}" = build ]; then
  make -s -C csrc >/dev/null || exit 1
  T=$(mktemp -d)
  git show HEAD:libperseus-sdr_amd/csrc/ddc_fir_i8.hip > $T/base.hip
  cp csrc/ddc_fir_i8.hip $T/new.hip
  for v in base new; do
    /opt/rocm/bin/hipcc $FLAGS -c $T/$v.hip -o $T/$v.o 2>$T/err.txt &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab_$v.so csrc/ddc_kernels.o $T/$v.o csrc/ddc_pipeline.o csrc/ddc_multi.o -L/opt/rocm/lib -lrccl &&
      echo "built ab_$v.so" || { echo "FAILED $v"; tail -5 $T/err.txt; }
  done
  rm -rf $T
elif [ "${1:-}" = run ]; then
  cp libperseus_ddc.so /tmp/keep.so
  for rep in $(seq 1 ${3:-2}); do
    for v in base new; do
      cp ab_$v.so libperseus_ddc.so
      (cd .. && timeout 300 python tools/state_2p28.py ${2:-plain tuned} 2>&1 | grep "round 1" | sed "s/^round 1/$v/")
    done
  done
  cp /tmp/keep.so libperseus_ddc.so
fi
