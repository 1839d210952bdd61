set -u
mkdir -p gpurun_out/r3d
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r3d/pytest_gpu.log
timeout 200 python tools/ab_loss_sym.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r3d/ab_loss_sym.log
timeout 600 python bench.py > gpurun_out/r3d/bench_line.json 2> gpurun_out/r3d/bench_err.log
cat gpurun_out/r3d/pytest_gpu.log gpurun_out/r3d/ab_loss_sym.log; tail -5 gpurun_out/r3d/bench_err.log; python - <<'PY'
import json
try:
    j=json.loads(open('gpurun_out/r3d/bench_line.json').read().strip().splitlines()[-1])
    print({k:j[k] for k in ('value','ms_per_step')}, j['roofline']['frac'])
    for r in j['regimes']: print(r['n_queries'], round(r['kernel_ms'],3), r['bound'], round(r['frac'],3))
    print(j.get('forced_collective_1rank'))
    print(j['embed_head'])
except Exception as e: print('bench parse failed', e)
PY
