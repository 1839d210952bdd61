#!/bin/bash
python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/bench_r24.json 2> gpurun_out/bench_r24.err; tail -c 3000 gpurun_out/bench_r24.json
