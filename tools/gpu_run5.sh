mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_head.py -x -q 2>&1 | tail -8 > gpurun_out/t5_head.log; tail -8 gpurun_out/t5_head.log
for i in 1 2; do timeout 200 python tools/ab_head.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/ab_head2.log; cat gpurun_out/ab_head2.log
