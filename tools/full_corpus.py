#!/usr/bin/env python
"""BASELINE config 4 WHOLE on one MI355X: 1 000 000 pre-embedded pages x 1024 patches x 128 bf16 = 256 GiB resident in the 288 GB of
HBM3E of a single GPU ("if HBM allows, the full 1 M").  Scores 1 / 4 / 32 queries against it, per-shard top-10, and
re-scores the returned ids + random pages with the CPU oracle (bench.topk_parity).  Refuses to start unless the corpus + 6 GiB fit."""
import json, os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import bench, colpali_amd as amd

dev = torch.device("cuda:0")
free, total = torch.cuda.mem_get_info(dev)
want_docs = int(os.environ.get("FULL_DOCS", "1000000"))
need = want_docs * 1024 * 256 + (6 << 30)
print(f"HBM: {total / 2**30:.1f} GiB total, {free / 2**30:.1f} GiB free; corpus of {want_docs} pages = {want_docs * 1024 * 256 / 2**30:.1f} GiB", flush=True)
if free < need:
    want_docs = int((free - (6 << 30)) // (1024 * 256) // 1000 * 1000)
    print(f"does not fit with 6 GiB of headroom: using {want_docs} pages = {want_docs * 1024 * 256 / 2**30:.1f} GiB", flush=True)
t0 = time.perf_counter()
corpus = bench.make_shard(want_docs, 1024, dev, 1234)
torch.cuda.synchronize()
print(f"generated on the device in {time.perf_counter() - t0:.1f} s", flush=True)
out = {"pages": want_docs, "corpus_gib": want_docs * 1024 * 256 / 2**30, "hbm_total_gib": total / 2**30, "regimes": []}
for nq in (1, 4, 32, 1000):
    q = bench.make_queries(nq, 32, dev, seed=99)
    scores = torch.empty((nq, want_docs), dtype=torch.float32, device=dev)
    for _ in range(2 if nq < 100 else 1):
        amd.maxsim_scores(q, corpus, out=scores)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5 if nq < 100 else 2)]
    for a, b in evs:
        a.record(); amd.maxsim_scores(q, corpus, out=scores); b.record()
    t1 = torch.cuda.Event(enable_timing=True); t2 = torch.cuda.Event(enable_timing=True)
    t1.record(); top_s, top_i = amd.topk(scores, 10, 0); t2.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]
    r = bench.regime_numbers(nq, 32, want_docs, 1024, ms)
    par = bench.topk_parity(amd, q, corpus, scores, top_s, top_i, 10, n_queries=min(nq, 2), n_random=200)
    out["regimes"].append({"n_queries": nq, "kernel_ms": ms, "pairs_per_s": nq * want_docs / ms * 1e3, "hbm_gbs": r["hbm_gbs"],
                           "mfma_tflops": r["mfma_tflops"], "bound": r["bound"], "frac": r["frac"], "topk10_ms": t1.elapsed_time(t2),
                           "topk_ids_equal_oracle": par["ids_equal"], "max_rel_err": par["max_rel_err"]})
    print(json.dumps(out["regimes"][-1]), flush=True)
print(json.dumps(out))
