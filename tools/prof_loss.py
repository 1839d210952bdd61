#!/usr/bin/env python
"""Workload for a rocprofv3 kernel trace of the three training-loss steps at BASELINE config 5's per-rank shapes."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import colpali_amd as amd

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
Q = torch.nn.functional.normalize(torch.randn((32, 32, 128), generator=g, device=dev), dim=-1).to(torch.bfloat16).requires_grad_(True)
D = torch.nn.functional.normalize(torch.randn((256, 780, 128), generator=g, device=dev), dim=-1).to(torch.bfloat16).requires_grad_(True)
for name, cls, kw in (("pairwise", amd.ColbertPairwiseCELoss, {}), ("infonce", amd.ColbertLoss, {}), ("smooth", amd.ColbertLoss, {"use_smooth_max": True})):
    if which not in ("all", name):
        continue
    for _ in range(5):
        Q.grad = D.grad = None
        cls(**kw)(Q, D, offset=0).backward()
torch.cuda.synchronize()
print("done")
