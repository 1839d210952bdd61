#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -2
AB_DOCS=65536 timeout 600 python tools/ab_regimes.py 9,12,16,24,32,64 2>&1 | grep -v amdgpu
