#!/bin/bash
# The two workarounds of ddc_fir_i8.hip judged on the PRODUCT kernel: libraries that differ only in the workaround, each run
# through tools/repeat_bits.py (the same batch N times per layout and form, every run compared bit for bit with the first).
#   build here (no GPU):   tools/hazard_ab.sh build      -> libperseus-sdr_amd/ab_{pad,nopad,slp}.so
#       pad    the product (asm store + s_nop 1; no SLP vectoriser)
#       nopad  the s_nop removed from the two asm stores
#       slp    the product's stores, but compiled WITH the SLP vectoriser (packed fp32 wherever hipcc forms it)
#   run on the GPU box:    gpurun -- bash tools/hazard_ab.sh run [reps]
set -u
cd "$(dirname "$0")/../libperseus-sdr_amd"
HIPFLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Icsrc"
if [ "${1:-}" = build ]; then
  make -s -C csrc >/dev/null || exit 1
  rm -f ab_*.so
  T=$(mktemp -d)
  cp csrc/ddc_fir_i8.hip $T/pad.hip
  sed 's/ off nt\\n\\ts_nop 1"/ off nt"/' csrc/ddc_fir_i8.hip > $T/nopad.hip
  cp csrc/ddc_fir_i8.hip $T/slp.hip
  grep -c 's_nop 1' $T/pad.hip $T/nopad.hip
  for v in pad nopad slp; do
    extra="-fno-slp-vectorize"; [ $v = slp ] && extra=""
    /opt/rocm/bin/hipcc $HIPFLAGS $extra -c $T/$v.hip -o $T/$v.o 2>$T/err.txt &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab_$v.so csrc/ddc_kernels.o $T/$v.o csrc/ddc_pipeline.o csrc/ddc_multi.o -L/opt/rocm/lib -lrccl &&
      echo "built ab_$v.so" || { echo "FAILED $v"; tail -5 $T/err.txt; }
  done
  rm -rf $T
elif [ "${1:-}" = run ]; then
  # (the alternatives are selected through PDDC_DDC_LIB: the product library is never overwritten -- the slp build is
  # KNOWN to deliver wrong outputs)
  for v in nopad slp pad; do
    echo "=== $v"
    (cd .. && PDDC_DDC_LIB="$PWD/libperseus-sdr_amd/ab_$v.so" timeout 600 python tools/repeat_bits.py ${2:-40} 2>&1 | tail -20)
  done
fi
