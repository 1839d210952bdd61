set -u
mkdir -p gpurun_out/r3f
timeout 900 python -m pytest tests/test_gpu_loss.py tests/test_gpu_abi.py tests/test_gpu_topk_e2e.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r3f/pytest_gpu.log
timeout 200 python tools/ab_loss_sym.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3f/ab_loss_sym.log
AB_B=64 timeout 200 python tools/ab_loss_sym.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r3f/ab_loss_sym.log
cat gpurun_out/r3f/pytest_gpu.log gpurun_out/r3f/ab_loss_sym.log
