set -u
mkdir -p gpurun_out/r3j
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_loss.py tests/test_gpu_abi.py -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r3j/pytest.log
timeout 300 python tools/ab_long_queries.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3j/ab_long_queries.log
timeout 200 python tools/ab_loss_sym.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3j/ab_loss_sym.log
cat gpurun_out/r3j/pytest.log gpurun_out/r3j/ab_long_queries.log gpurun_out/r3j/ab_loss_sym.log
