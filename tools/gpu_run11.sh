mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_loss.py -x -q 2>&1 | tail -12 > gpurun_out/t11.log; tail -12 gpurun_out/t11.log
timeout 300 python tools/ab_generic.py 2>&1 | grep -v amdgpu > gpurun_out/ab_generic2.log; cat gpurun_out/ab_generic2.log
