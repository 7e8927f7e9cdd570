#!/bin/bash
# The drop-in API with N virtual receivers on one GPU (C client, unpaced, on-device source, 250 kS/s plan): aggregate
# ADC-rate throughput with the submit pass on the delivery thread alone (PERSEUS_AMD_SUBMIT_THREADS=0) and farmed out
# to helper threads (default).  usage (GPU box): bash tools/api_receivers.sh
EXE=libperseus-sdr_amd/perseus_plumbing
for th in 0 7; do
  for n in 1 8; do
    out=$(PERSEUS_AMD_SUBMIT_THREADS=$th PERSEUS_AMD_PACE=0 PERSEUS_AMD_MODE=ddc $EXE -N $n -s 250000 -o none -t 3 -d 0 2>&1 | grep -E "receivers:|Rate:" | tail -1)
    echo "submit helpers $th, N=$n: $out"
  done
done
