mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/bench4.json 2> gpurun_out/bench4.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench4.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['roofline']['traffic'])
print([ (r['n_queries'], round(r['frac'],3), round(r['pairs_per_s']/1e6,1)) for r in d['regimes']])
print(d['embed_head']['fused_head'], d['dropin_from_host_lists'])
print(d['cpu_baseline']['value'], d['reference_on_this_gpu']['value'])
PY
