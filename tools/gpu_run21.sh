mkdir -p gpurun_out/prof_loss
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_loss/trace -o trace -- python $GRAFT_REPO_ROOT/tools/ab_loss.py > $GRAFT_REPO_ROOT/gpurun_out/prof_loss/run.log 2>&1
python - <<'PY'
import csv, os
p=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/prof_loss/trace/trace_kernel_stats.csv'
for r in csv.DictReader(open(p)):
    if 'msim::' in r['Name']:
        print(f"{r['Name'][:70]:70s} calls={r['Calls']:>3s} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
