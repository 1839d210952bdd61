mkdir -p gpurun_out
timeout 600 python tools/ab_dropin.py 2>&1 | grep -v amdgpu.ids > gpurun_out/ab_dropin3.log; cat gpurun_out/ab_dropin3.log
timeout 900 python bench.py > gpurun_out/bench3.json 2> gpurun_out/bench3.err; tail -c 3000 gpurun_out/bench3.json; tail -3 gpurun_out/bench3.err
