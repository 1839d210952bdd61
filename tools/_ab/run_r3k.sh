set -u
mkdir -p gpurun_out/r3k
timeout 900 python bench.py > gpurun_out/r3k/bench_line.json 2> gpurun_out/r3k/bench_err.log
wc -l gpurun_out/r3k/bench_line.json
BENCH_SHARE_GPU=1 BENCH_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --docs 20000 --steps 5 --warmup 2 --regimes 8 > gpurun_out/r3k/bench_gpus2_shared_gpu_gloo.json 2> gpurun_out/r3k/bench_gpus2_err.log
wc -l gpurun_out/r3k/bench_gpus2_shared_gpu_gloo.json; tail -3 gpurun_out/r3k/bench_gpus2_err.log
python - <<'PY'
import json
for f in ('gpurun_out/r3k/bench_line.json','gpurun_out/r3k/bench_gpus2_shared_gpu_gloo.json'):
    try:
        j=json.loads(open(f).read().strip())
        print(f, j['n_gpus'], j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline'].get('power'), j.get('rccl_ranks'))
        for r in j['regimes']: print('  ', r['n_queries'], round(r['kernel_ms'],3), r['bound'], round(r['frac'],3), r.get('power'))
    except Exception as e: print(f, 'parse failed', e)
PY
