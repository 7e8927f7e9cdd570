timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for wl in ${WLS:-c320}; do
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_$wl -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --workload $wl --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/trace_$wl.log 2>&1
done
