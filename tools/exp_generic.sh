cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gen -- python3 tools/exp_generic.py
