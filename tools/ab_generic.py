#!/usr/bin/env python
"""Throughput of the generic kernels (K1g: fp32 / widths other than 128 / long queries) -- tuning aid, one line per case."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import colpali_amd as amd
from colpali_amd.corpus import PackedCorpus

dev = torch.device("cuda:0")
docs, doc_len = int(os.environ.get("AB_DOCS", "8192")), 1024
cases = [(torch.bfloat16, 320, 32), (torch.float32, 128, 32), (torch.float16, 64, 32), (torch.bfloat16, 128, 160)]
for dtype, dim, q_len in cases:
    g = torch.Generator(device=dev).manual_seed(1)
    blob = torch.nn.functional.normalize(torch.randn((docs * doc_len, dim), generator=g, device=dev), dim=-1).to(dtype)
    offsets = (torch.arange(docs + 1, dtype=torch.int64) * doc_len).to(torch.int32).to(dev)
    corpus = PackedCorpus(blob=blob, offsets=offsets, clamp0=None, lengths=torch.full((docs,), doc_len, dtype=torch.int64))
    es = blob.element_size()
    for nq in (1, 4, 32, 128):
        q = torch.nn.functional.normalize(torch.randn((nq, q_len, dim), generator=g, device=dev), dim=-1).to(dtype)
        out = torch.empty((nq, docs), dtype=torch.float32, device=dev)
        for _ in range(2):
            amd.maxsim_scores(q, corpus, out=out)
        reps = 5
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in evs:
            a.record(); amd.maxsim_scores(q, corpus, out=out); b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in evs)[reps // 2]
        flops = 2.0 * nq * q_len * docs * doc_len * dim
        print(f"{str(dtype):15s} dim={dim:4d} Lq={q_len:3d} nq={nq:4d} {ms:9.3f} ms  {nq*docs/ms/1e3:8.2f} Mpairs/s  "
              f"{docs*doc_len*dim*es/ms/1e6:7.0f} GB/s(corpus once)  {flops/ms/1e9:7.1f} TF", flush=True)
    del blob, corpus
