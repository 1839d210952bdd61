#!/usr/bin/env python
"""Within-process sweep of the MaxSim kernel over query-batch sizes (tuning aid; prints one line per size)."""
import os, sys, json, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import bench, colpali_amd as amd

docs = int(os.environ.get("AB_DOCS", "32768"))
dev = torch.device("cuda:0")
corpus = bench.make_shard(docs, 1024, dev, 1234)
for nq in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4,5,6,7,8,12,16,24,32,64").split(",")]:
    q = bench.make_queries(nq, 32, dev, 3)
    out = torch.empty((nq, docs), dtype=torch.float32, device=dev)
    for _ in range(3): amd.maxsim_scores(q, corpus, out=out)
    reps = max(3, 60 // nq)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); amd.maxsim_scores(q, corpus, out=out); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]
    r = bench.regime_numbers(nq, 32, docs, 1024, ms)
    print(f"nq={nq:4d} {ms:8.3f} ms  {nq*docs/ms/1e3:8.1f} Mpairs/s  {r['hbm_gbs']:7.0f} GB/s  {r['mfma_tflops']:7.0f} TF  {r['bound']} {r['frac']:.3f}", flush=True)
