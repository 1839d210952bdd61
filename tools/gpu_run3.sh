mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_head.py -x -q 2>&1 | tail -30 > gpurun_out/t3_head.log
tail -30 gpurun_out/t3_head.log
timeout 300 python tools/ab_head.py > gpurun_out/ab_head.log 2>&1; cat gpurun_out/ab_head.log
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/t3.log; tail -8 gpurun_out/t3.log
