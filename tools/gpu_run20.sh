mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_loss.py -x -q 2>&1 | tail -3
timeout 300 python tools/ab_loss.py 2>&1 | grep -v amdgpu > gpurun_out/ab_loss.log; cat gpurun_out/ab_loss.log
