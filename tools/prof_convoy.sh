#!/bin/bash
# HBM traffic of K1b at 1000 queries with and without the convoy: one FETCH_SIZE pass each (separate --pmc runs).
# the MSIM_* knobs exist in the measurement build only: `make -C colpali_amd/csrc ab` first
export COLPALI_AMD_LIB=${COLPALI_AMD_LIB:-tools/_ab/libmaxsim_ab.so}
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_convoy
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-parity --regimes 1000 --steps 3 --warmup 1"
for mode in 1 0; do
  MSIM_BATCH_CONVOY=$mode rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/convoy$mode -o pmc -- $B > $OUT/bench_convoy$mode.json 2> $OUT/convoy$mode.err
  python - <<PY
import csv, glob
for p in glob.glob("$OUT/convoy$mode/**/pmc_counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(p)) if "maxsim_batch_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
    big = [r for r in rows if float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) > 2e8]
    for r in big:
        print("MSIM_BATCH_CONVOY=$mode  %.1f ms  HBM read %.1f GB (FETCH_SIZE x 2)" % ((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6, float(r["Counter_Value"]) * 1024 * 2 / 1e9))
PY
done
