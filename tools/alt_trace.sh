#!/bin/bash
# kernel durations of alternative stage orders: tools/alt_trace.sh <rate> <order> [<order> ...]
export TMPDIR=/tmp
O=gpurun_out/alt_trace
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/plan_alt.py "$@" > $O/log.txt 2>&1
grep rate $O/log.txt
find $O/kt -name "*kernel_stats.csv" | head -1 | xargs head -8 | cut -c1-140
rm -rf $O/kt
