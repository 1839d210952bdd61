#!/bin/bash
timeout 300 python -m pytest tests/test_gpu_head.py -x -q -m gpu 2>&1 | tail -2
timeout 200 python tools/ab_head.py 2>&1 | grep -v amdgpu
