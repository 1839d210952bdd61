mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_sim.py tests/test_gpu_head.py -x -q 2>&1 | tail -15 > gpurun_out/t7.log; tail -15 gpurun_out/t7.log
timeout 600 python tools/ab_dropin.py 2>&1 | grep -v amdgpu.ids > gpurun_out/ab_dropin2.log; cat gpurun_out/ab_dropin2.log
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
