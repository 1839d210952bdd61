#!/usr/bin/env python
"""Short documents on K1b: device time of 1000 queries (32 tokens, and ragged 12..48) over 8 GiB of 64-row / 343-row documents for the
build / knob setting in the environment (COLPALI_AMD_LIB, MSIM_BATCH_NW): the shipped eight-wave multi-block plan against the four-wave
form (two workgroups per CU, 64-row chunks).  One process per variant (the knobs are read once)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import bench, colpali_amd as amd

dev = torch.device("cuda:0")
tag = os.environ.get("AB_TAG", "default")
for doc_len in (64, 343, 1024):
    n_docs = (4 << 30) // (doc_len * 256)
    corpus = bench.make_shard(n_docs, doc_len, dev, seed=5)
    qsets = [("1000 x 32", [32] * 1000), ("1000 x U{12..48}", bench.parse_regime("1000xr12-48", 32)[1])]
    if os.environ.get("AB_MORE") == "1":       # the ten-unit form, and the pair / four-wave / one-block forms (they share K1b's producer and consumer)
        qsets += [("1000 x 40", [40] * 1000)] + ([(f"{n} x 32", [32] * n) for n in (10, 16, 32, 64)] if doc_len != 343 else [])
    for qname, lens in qsets:
        q = amd.pack_queries(bench.make_query_list(lens, seed=sum(lens) + doc_len), dev)
        scores = torch.empty((len(lens), n_docs), dtype=torch.float32, device=dev)
        amd.maxsim_scores(q, corpus, out=scores)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
        for a, b in evs:
            a.record(); amd.maxsim_scores(q, corpus, out=scores); b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in evs)[1]
        r = bench.regime_numbers(len(lens), 32, n_docs, doc_len, ms, q_tokens=sum(lens))
        print(f"{tag:34s} doc_len {doc_len:5d}  {qname:18s} {ms:9.2f} ms  {r['mfma_tflops']:7.0f} TFLOP/s  frac {r['frac']:.3f}  checksum {float(scores[::97, ::1013].double().sum()):.6f}", flush=True)
        del scores
    del corpus
