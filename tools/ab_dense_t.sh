#!/bin/bash
# per-kernel times of the dense backward with parts switched off (measurement build): rocprofv3 kernel stats per MSIM_DENSE_T_DBG value
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export COLPALI_AMD_LIB=$R/tools/_ab/libmaxsim_ab.so
for dbg in ${@:-0 1 2 4 8 3 6 15}; do
  MSIM_DENSE_T_DBG=$dbg rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/ab_dense_t_$dbg -o p -- python $R/tools/ab_dense_t.py > $R/gpurun_out/ab_dense_t_$dbg.log 2>&1
  f=$(find $R/gpurun_out/ab_dense_t_$dbg -name "*kernel_stats.csv" | head -1)
  echo "== dbg $dbg: $(grep Ld= $R/gpurun_out/ab_dense_t_$dbg.log)"
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("dense_t_", "batch_t_kernel")):
        print(f"   {r['Name'][:64]:64s} avg {float(r['AverageNs']) / 1e3:8.1f} us")
PY
  find $R/gpurun_out/ab_dense_t_$dbg -name "*kernel_trace.csv" -delete
done
