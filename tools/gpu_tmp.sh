#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_loss.py tests/test_gpu_fuzz.py tests/test_gpu_abi.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/ab_loss.py 2>&1 | grep -v amdgpu
bash tools/prof_loss.sh smooth 2>&1 | head -6
