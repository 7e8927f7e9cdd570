#!/bin/bash
# The drop-in API with N virtual receivers on one GPU (C client, unpaced, on-device source, 250 kS/s plan): aggregate
# ADC-rate throughput with gang submission (default: the receivers of a GPU share one launch chain), without it
# (PERSEUS_AMD_GANG=0: a chain per receiver, farmed out to the submit helpers) and with neither (one thread, a chain
# per receiver).  usage (GPU box): bash tools/api_receivers.sh [rate]
EXE=libperseus-sdr_amd/perseus_plumbing
RATE=${1:-250000}
run() {   # label, N, env...
  local label=$1 n=$2; shift 2
  out=$(env "$@" PERSEUS_AMD_PACE=0 PERSEUS_AMD_MODE=ddc $EXE -N $n -s $RATE -o none -t 3 -d 0 2>&1)
  echo "$label, N=$n: $(echo "$out" | grep -E "receivers:" | tail -1)"
  echo "$out" | grep -E "^receiver 0 " | sed 's/^/      /'
}
run "gang" 1 A=1
run "gang" 2 A=1
run "gang" 4 A=1
run "gang" 8 A=1
run "no gang, submit helpers" 8 PERSEUS_AMD_GANG=0
run "no gang, one thread" 8 PERSEUS_AMD_GANG=0 PERSEUS_AMD_SUBMIT_THREADS=0
