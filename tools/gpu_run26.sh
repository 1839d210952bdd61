#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_abi.py tests/test_gpu_topk.py -x -q -m gpu 2>&1 | tail -4
AB_DOCS=65536 python tools/ab_regimes.py 9,10,12,14,16,20,24,28,32,40,48,64,100,128 2>&1 | grep -v amdgpu
