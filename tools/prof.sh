#!/bin/bash
# Profile bench.py on the GPU box: kernel-trace stats + separate PMC passes. Usage: tools_prof.sh <tag> [bench args...]
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity $*"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o trace -- $B > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -o pmc -- $B > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -o pmc -- $B > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_sq -o pmc -- $B > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.err
find $OUT -name "*.csv" | head -30
# drop the bulky per-dispatch traces of torch's init kernels: keep stats + counter csv
ls -la $OUT/*/* | head -40
