#!/bin/bash
# per-kernel time of each loss step (separate traces so the shared kernels are attributed)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in ${@:-pairwise infonce smooth}; do
  rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_loss_$w -o p -- python $R/tools/prof_loss.py $w > $R/gpurun_out/prof_loss_$w.log 2>&1
  f=$(find $R/gpurun_out/prof_loss_$w -name "*kernel_stats.csv" | head -1)
  echo "== $w"; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e3:9.1f} us")
PY
  find $R/gpurun_out/prof_loss_$w -name "*kernel_trace.csv" -delete
done
