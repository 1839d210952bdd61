#!/usr/bin/env python
"""Width 320 (ColQwen3) at the query counts between the HBM-bound and the 1000-query regime -- the plan ladder the round-5 review asked
about: 4 096 pages x 1 024 rows x 320 bf16 resident; roofline fractions per real token next to the width-128 fractions of the same counts."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import colpali_amd as amd
from colpali_amd.corpus import PackedCorpus

dev = torch.device("cuda:0")
import os as _os
for dim, n_docs in (((320, 4096),) if _os.environ.get('AB_ONLY_320') else ((320, 4096), (128, 10240))):
    g = torch.Generator(device=dev).manual_seed(11)
    blob = torch.nn.functional.normalize(torch.randn((n_docs * 1024, dim), generator=g, device=dev), dim=-1).to(torch.bfloat16)
    corpus = PackedCorpus(blob=blob, offsets=(torch.arange(n_docs + 1, dtype=torch.int64) * 1024).to(torch.int32).to(dev), clamp0=None,
                          lengths=torch.full((n_docs,), 1024, dtype=torch.int64))
    for nq in ((4, 5, 8, 10, 12, 16) if _os.environ.get('AB_ONLY_320') else (4, 8, 10, 12, 16, 20, 32, 40, 64, 1000)):
        for L in ((25, 32, 40) if _os.environ.get('AB_ONLY_320') else (32, 40)):
            tok = torch.nn.functional.normalize(torch.randn((nq * L, dim), generator=g, device=dev), dim=-1).to(torch.bfloat16)
            q = amd.pack_queries(list(tok.split([L] * nq)), dev)
            scores = torch.empty((nq, n_docs), dtype=torch.float32, device=dev)
            amd.maxsim_scores(q, corpus, out=scores)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
            for a, b in evs:
                a.record(); amd.maxsim_scores(q, corpus, out=scores); b.record()
            torch.cuda.synchronize()
            ms = sorted(a.elapsed_time(b) for a, b in evs)[2]
            alg = blob.numel() * 2 + nq * L * dim * 2 + nq * n_docs * 4
            flops = 2.0 * nq * L * n_docs * 1024 * dim
            hb, mf = alg / ms / 1e6 / 8000.0, flops / ms / 1e9 / 2500.0
            bound = "hbm" if alg / 8e12 >= flops / 2.5e15 else "mfma"
            print(f"dim {dim} nq {nq:4d} x {L}: {ms:8.3f} ms  {bound} frac {hb if bound == 'hbm' else mf:.3f}", flush=True)
    del blob, corpus
