mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/t1.log
for i in 1 2; do for nt in 0 1; do echo "NT=$nt"; MSIM_STREAM_NT=$nt AB_DOCS=65536 timeout 120 python tools/ab_regimes.py 1,2,4,8; done; done > gpurun_out/ab_nt.log 2>&1
timeout 300 python tools/ab_generic.py > gpurun_out/ab_generic.log 2>&1
tail -5 gpurun_out/t1.log; cat gpurun_out/ab_nt.log gpurun_out/ab_generic.log
