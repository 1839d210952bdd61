#!/usr/bin/env python
"""One K1 variant (chosen through the MSIM_* tuning knobs in the environment) over a list of query-batch sizes: time per
launch on a resident shard and bitwise comparison of its scores with a reference set written by the default build
(AB_REF=write|check, file gpurun_out/ab_ref_scores.pt).  Tuning aid behind profiles/r02_logs/ab_ridge.log."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import bench, colpali_amd as amd

docs = int(os.environ.get("AB_DOCS", "65536"))
dev = torch.device("cuda:0")
corpus = bench.make_shard(docs, 1024, dev, 1234)
if os.environ.get("AB_ZERO") == "1":      # DVFS check (MI355X_MICROARCH.md): same binary, same traffic, operands that toggle nothing
    corpus.blob.zero_()
sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "5,6,7,8").split(",")]
tag = os.environ.get("AB_TAG", "default")
ref_mode = os.environ.get("AB_REF", "")
ref_path = os.path.join(os.path.dirname(__file__), "..", "gpurun_out", "ab_ref_scores.pt")
ref = torch.load(ref_path) if ref_mode == "check" and os.path.exists(ref_path) else {}
keep = {}
for nq in sizes:
    q = bench.make_queries(nq, 32, dev, 3)
    out = torch.empty((nq, docs), dtype=torch.float32, device=dev)
    for _ in range(3):
        amd.maxsim_scores(q, corpus, out=out)
    reps = max(5, 80 // nq)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); amd.maxsim_scores(q, corpus, out=out); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]
    r = bench.regime_numbers(nq, 32, docs, 1024, ms)
    sample = out[:, :: max(1, docs // 4096)].cpu()
    same = ""
    if ref_mode == "write":
        keep[nq] = sample
    elif nq in ref:
        same = " bitwise==default" if torch.equal(sample, ref[nq]) else f" DIFFERS max {float((sample - ref[nq]).abs().max()):.3e}"
    print(f"{tag:28s} nq={nq:4d} {ms:8.3f} ms  {nq*docs/ms/1e3:8.1f} Mpairs/s  {r['hbm_gbs']:7.0f} GB/s  {r['mfma_tflops']:7.0f} TF  {r['bound']} {r['frac']:.3f}{same}", flush=True)
if ref_mode == "write":
    old = torch.load(ref_path) if os.path.exists(ref_path) else {}
    old.update(keep)
    os.makedirs(os.path.dirname(ref_path), exist_ok=True)
    torch.save(old, ref_path)
