mkdir -p gpurun_out
timeout 600 python tools/ab_dropin.py 2>&1 | grep -v amdgpu.ids > gpurun_out/ab_dropin.log; cat gpurun_out/ab_dropin.log
