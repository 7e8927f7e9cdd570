cd /tmp && export TMPDIR=/tmp
for sh in 256,4 128,4 64,4 256,2 128,2 64,2 256,1 64,1; do
export PDDC_GEN_SHAPE=$sh
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/shape_$sh -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --workload c320 --steps 20 --warmup 5 > /dev/null 2>&1
echo -n "$sh: "; grep -h "k_fir_generic" $GRAFT_REPO_ROOT/gpurun_out/shape_$sh/*/*kernel_stats.csv | cut -d, -f1-6 | cut -c1-30,100-
done
