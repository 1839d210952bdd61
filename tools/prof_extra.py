#!/usr/bin/env python
"""Workload for the rocprofv3 evidence of the kernels bench.py does not time as its headline: K3 (embedding head),
K1g (generic scorer: fp32 and dim 320), the smooth-max forward/backward and the pair-list backward."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import colpali_amd as amd
from colpali_amd.corpus import PackedCorpus

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
# K3: 500 pages x 1030 x 2048
hidden = torch.randn((500, 1030, 2048), generator=g, device=dev).to(torch.bfloat16)
W = (torch.randn((128, 2048), generator=g, device=dev) / 45).to(torch.bfloat16)
b = torch.zeros(128, dtype=torch.bfloat16, device=dev)
mask = torch.ones((500, 1030), dtype=torch.long, device=dev)
for _ in range(5):
    amd.embedding_head(hidden, W, b, mask)
del hidden
# K1g: 8192 docs x 1024 rows, fp32 dim 128 and bf16 dim 320, 4 queries
for dtype, dim in ((torch.float32, 128), (torch.bfloat16, 320)):
    docs, L = 8192, 1024
    blob = torch.nn.functional.normalize(torch.randn((docs * L, dim), generator=g, device=dev), dim=-1).to(dtype)
    off = (torch.arange(docs + 1, dtype=torch.int64) * L).to(torch.int32).to(dev)
    corpus = PackedCorpus(blob=blob, offsets=off, clamp0=None, lengths=torch.full((docs,), L, dtype=torch.int64))
    q = torch.nn.functional.normalize(torch.randn((4, 32, dim), generator=g, device=dev), dim=-1).to(dtype)
    for _ in range(5):
        amd.maxsim_scores(q, corpus)
    del blob, corpus
# training losses at BASELINE config 5 per-rank shapes: B=32 queries, C=256 docs, Lq=32, Ld=780
Q = torch.nn.functional.normalize(torch.randn((32, 32, 128), generator=g, device=dev), dim=-1).to(torch.bfloat16).requires_grad_(True)
D = torch.nn.functional.normalize(torch.randn((256, 780, 128), generator=g, device=dev), dim=-1).to(torch.bfloat16).requires_grad_(True)
for cls, kw in ((amd.ColbertPairwiseCELoss, {}), (amd.ColbertLoss, {}), (amd.ColbertLoss, {"use_smooth_max": True})):
    for _ in range(3):
        loss = cls(**kw)(Q, D, offset=0)
        loss.backward()
torch.cuda.synchronize()
print("done")
