#!/bin/bash
# NEEDS the timing code that lives outside the product sources: git apply tools/ubench/fir8_probe_and_ablations.patch (revert afterwards)
# Ablation builds of k_fir8 (NOT valid outputs -- timing only): which part of the kernel costs what.
# Build here (no GPU needed):   tools/ablate.sh build
# Run on the GPU box:           gpurun -- bash tools/ablate.sh run
# Variants: global loads / FIR / global stores removed, alone and in pairs (profiles/r01/v7_power_and_ablation.txt).
# Do not combine BARRIERS with the dynamic schedule: without barriers the chunk hand-off races and a block can spin.
set -u
cd "$(dirname "$0")/../libperseus-sdr_amd"
VARIANTS=("LOADS" "FIR" "STORES" "LOADS -DPDDC_ABLATE_STORES" "FIR -DPDDC_ABLATE_STORES" "LOADS -DPDDC_ABLATE_FIR")
if [ "${1:-}" = build ]; then
  make -C csrc >/dev/null
  cp libperseus_ddc.so abl_NONE.so
  for v in "${VARIANTS[@]}"; do
    n=$(echo $v | tr -d ' -' | sed 's/DPDDC_ABLATE_/_/g')
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DPDDC_ABLATE_$v -c csrc/ddc_kernels.hip -o /tmp/abl_k.o 2>/dev/null &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abl_$n.so /tmp/abl_k.o csrc/ddc_pipeline.o && echo built abl_$n.so
  done
elif [ "${1:-}" = run ]; then
  cp libperseus_ddc.so /tmp/keep.so
  for f in abl_*.so; do
    cp $f libperseus_ddc.so
    for wl in d8_127 d8_255 c320; do
      echo -n "$f $wl: "
      (cd .. && timeout 120 python bench.py --no-cpu --workload $wl --steps 30 --warmup 5 2>/dev/null | tail -1 |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'])")
    done
  done
  cp /tmp/keep.so libperseus_ddc.so
  rm -f abl_*.so
else
  echo "usage: $0 build|run"; exit 2
fi
