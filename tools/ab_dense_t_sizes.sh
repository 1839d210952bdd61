#!/bin/bash
# per-kernel times of the dense backward against the number of documents / pages (real build): slope = cost per stage, intercept = fixed cost
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "32 16 32" "32 64 32" "32 128 32" "32 256 32" "32 512 32" "32 256 8" "32 256 64"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/ab_sizes -o p -- python $R/tools/ab_dense_t.py $1 $2 $3 > $R/gpurun_out/ab_sizes.log 2>&1
  f=$(find $R/gpurun_out/ab_sizes -name "*kernel_stats.csv" | head -1)
  echo "== Ld $1 n_d $2 n_q $3"
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("dense_t_bwd", "batch_t_kernel")):
        print(f"   {r['Name'][:64]:64s} avg {float(r['AverageNs']) / 1e3:8.1f} us")
PY
  rm -rf $R/gpurun_out/ab_sizes
done
