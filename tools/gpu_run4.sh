mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_head.py -x -q 2>&1 | tail -8 > gpurun_out/t4_head.log; tail -8 gpurun_out/t4_head.log
MSIM_HEAD_PF=4 timeout 600 python -m pytest tests/test_gpu_head.py -x -q 2>&1 | tail -4
for i in 1 2; do for pf in 0 2 4; do echo "PF=$pf"; MSIM_HEAD_PF=$pf timeout 200 python tools/ab_head.py 2>&1 | grep K3; done; done > gpurun_out/ab_head_pf.log; cat gpurun_out/ab_head_pf.log
