mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pooling.py -x -q 2>&1 | tail -40 > gpurun_out/t12.log; tail -40 gpurun_out/t12.log
