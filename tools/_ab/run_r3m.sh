set -u
mkdir -p gpurun_out/r3m
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
AB_ONLY=ColbertLoss timeout 300 rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/r3m/prof -o sym -- python tools/ab_loss_sym.py > gpurun_out/r3m/prof_run.log 2>&1
f=$(find gpurun_out/r3m/prof -name "*kernel_stats.csv" | head -1); echo $f; head -14 "$f" | cut -c1-160
