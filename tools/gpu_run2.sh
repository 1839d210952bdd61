mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/t2.log
tail -30 gpurun_out/t2.log
if grep -q " passed" gpurun_out/t2.log && ! grep -q "failed" gpurun_out/t2.log; then
  timeout 1200 bash tools/prof.sh r01_nt > gpurun_out/prof_r01_nt.log 2>&1
  tail -3 gpurun_out/prof_r01_nt/bench_trace.json
fi
