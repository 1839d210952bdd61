#!/bin/bash
# Profile a workload on the GPU box: kernel-trace stats + separate PMC passes (never combined with other trace domains).
# Usage: tools/prof.sh <tag> [bench args...]         -> profiles bench.py
#        PROF_CMD="python tools/prof_head.py" tools/prof.sh <tag>   -> profiles another workload
set -u
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ -n "${PROF_CMD:-}" ]; then B="$PROF_CMD"; cd $GRAFT_REPO_ROOT; else B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity $*"; fi
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o trace -- $B > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -o pmc -- $B > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -o pmc -- $B > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_sq -o pmc -- $B > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.err
if [ -n "${PROF_WAIT:-}" ]; then
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -f csv -d $OUT/pmc_wait -o pmc -- $B > $OUT/bench_pmc_wait.json 2> $OUT/pmc_wait.err
fi
find $OUT -name "*.csv" | head -30
ls -la $OUT/*/* | head -40
