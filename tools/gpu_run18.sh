mkdir -p gpurun_out
MSIM_STREAM_IL=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_abi.py tests/test_gpu_topk.py -x -q 2>&1 | tail -4
for i in 1 2; do for il in 0 1; do echo "IL=$il"; MSIM_STREAM_IL=$il AB_DOCS=65536 timeout 120 python tools/ab_regimes.py 1,2,4,6,8 2>&1 | grep -v amdgpu; done; done > gpurun_out/ab_il.log; cat gpurun_out/ab_il.log
