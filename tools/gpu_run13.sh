mkdir -p gpurun_out
timeout 900 python tools/ab_pooling.py 2>&1 | grep -v amdgpu > gpurun_out/ab_pooling.log; cat gpurun_out/ab_pooling.log
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
