mkdir -p gpurun_out/prof_pool
cd /tmp && export TMPDIR=/tmp
AB_PAGES=256 timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_pool/trace -o trace -- python $GRAFT_REPO_ROOT/tools/ab_pooling.py > $GRAFT_REPO_ROOT/gpurun_out/prof_pool/run.log 2>&1
grep "msim::pool" $GRAFT_REPO_ROOT/gpurun_out/prof_pool/trace/trace_kernel_stats.csv | cut -c1-60,150-260
tail -4 $GRAFT_REPO_ROOT/gpurun_out/prof_pool/run.log
