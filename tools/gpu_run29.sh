#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_topk.py tests/test_gpu_parity.py tests/test_gpu_abi.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python bench.py --no-cpu-baseline --no-parity --regimes "1,32" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value',d['value'],'ms_per_step',d['ms_per_step'],'kernel_ms',d['roofline']['kernel_ms'])
for r in d['regimes']: print(r['n_queries'], r['pairs_per_s'], r['ms_per_step'], r['kernel_ms'])
"
