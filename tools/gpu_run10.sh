mkdir -p gpurun_out/prof_r01_extra
for mt in 8 4; do echo "MSIM_STREAM_MAX_TILES=$mt"; MSIM_STREAM_MAX_TILES=$mt AB_DOCS=65536 timeout 200 python tools/ab_regimes.py 5,6,8,12,16,24,32 2>&1 | grep -v amdgpu; done > gpurun_out/ab_crossover.log; cat gpurun_out/ab_crossover.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01_extra/trace -o trace -- python $GRAFT_REPO_ROOT/tools/prof_extra.py > $GRAFT_REPO_ROOT/gpurun_out/prof_r01_extra/run.log 2>&1
head -30 $GRAFT_REPO_ROOT/gpurun_out/prof_r01_extra/trace/trace_kernel_stats.csv | cut -c1-150
