#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_r01_extra
rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_r01_extra -o x -- python $R/tools/prof_extra.py > $R/gpurun_out/prof_r01_extra.log 2>&1
find $R/gpurun_out/prof_r01_extra -name "*kernel_trace.csv" -delete
f=$(find $R/gpurun_out/prof_r01_extra -name "*kernel_stats.csv" | head -1)
head -25 "$f" | cut -c1-200
