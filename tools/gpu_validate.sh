#!/bin/bash
# What the driver runs at round end, in one gpurun call:  gpurun --timeout 1500 -- 'bash tools/gpu_validate.sh'
python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
mkdir -p gpurun_out
python bench.py > gpurun_out/bench_validate.json 2> gpurun_out/bench_validate.err; tail -c 3500 gpurun_out/bench_validate.json
