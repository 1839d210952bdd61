mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pooling.py -x -q 2>&1 | tail -4
timeout 900 python tools/ab_pooling.py 2>&1 | grep -v amdgpu > gpurun_out/ab_pooling2.log; cat gpurun_out/ab_pooling2.log
