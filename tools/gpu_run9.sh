mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_abi.py -x -q 2>&1 | tail -6
for i in 1 2; do for r in 4 5; do echo "RING=$r"; MSIM_STREAM_RING=$r AB_DOCS=65536 timeout 120 python tools/ab_regimes.py 1,2,4,8 2>&1 | grep -v amdgpu; done; done > gpurun_out/ab_ring.log; cat gpurun_out/ab_ring.log
