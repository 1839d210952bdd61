#!/usr/bin/env python
"""K1s (the HBM-bound regime: 1 and 4 queries x 32 tokens) over 8 GiB of 64-row / 343-row / 1024-row documents for the build in
COLPALI_AMD_LIB: device time and fraction of 8 TB/s.  One process per variant; run variants alternately on ONE box (boxes differ by
several per cent)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import bench, colpali_amd as amd

dev = torch.device("cuda:0")
tag = os.environ.get("AB_TAG", "default")
for doc_len in (64, 343, 1024):
    n_docs = (8 << 30) // (doc_len * 256)
    corpus = bench.make_shard(n_docs, doc_len, dev, seed=5)
    for nq in (1, 4):
        q = amd.pack_queries(bench.make_query_list([32] * nq, seed=nq + doc_len), dev)
        scores = torch.empty((nq, n_docs), dtype=torch.float32, device=dev)
        for _ in range(3):
            amd.maxsim_scores(q, corpus, out=scores)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(9)]
        for a, b in evs:
            a.record(); amd.maxsim_scores(q, corpus, out=scores); b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in evs)[4]
        gbs = (n_docs * doc_len * 256 + nq * n_docs * 4) / ms / 1e6
        print(f"{tag:20s} doc_len {doc_len:5d}  {nq} x 32  {ms:8.3f} ms  {gbs:7.0f} GB/s  frac {gbs / 8000:.3f}  checksum {float(scores[:, ::1013].double().sum()):.6f}", flush=True)
        del scores
    del corpus
