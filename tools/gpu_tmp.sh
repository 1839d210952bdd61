#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_loss.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | grep -v "^$" | tail -5
timeout 300 python tools/ab_loss.py 2>&1 | grep -v amdgpu
