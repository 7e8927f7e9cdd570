cd /tmp && export TMPDIR=/tmp
for i in 1 2 3 4 5; do
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_$i -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --workload c320 --steps 100 --warmup 10 2>/dev/null | tail -1 | cut -c1-120
done
