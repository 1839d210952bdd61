set -u
mkdir -p gpurun_out/r3e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r3e/pytest_gpu.log
AB_ONLY=ColbertLoss timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r3e/prof -o sym -- python tools/ab_loss_sym.py > gpurun_out/r3e/prof_run.log 2>&1
cat gpurun_out/r3e/pytest_gpu.log; tail -3 gpurun_out/r3e/prof_run.log
f=$(find gpurun_out/r3e/prof -name "*kernel_stats.csv" | head -1); echo $f; head -25 "$f" | cut -c1-220
