#!/bin/bash
timeout 300 python -m pytest tests/test_gpu_loss.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3
timeout 200 python tools/ab_loss.py 2>&1 | grep -v amdgpu
timeout 200 bash tools/prof_loss.sh smooth 2>&1 | head -5
