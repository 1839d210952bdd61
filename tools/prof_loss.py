#!/usr/bin/env python
"""Workload for a rocprofv3 kernel trace of the training-loss steps at BASELINE config 5's per-rank shapes.
  pairwise | infonce | smooth          forward direction: queries [32, 32, 128] against the gathered pages [256, 780, 128]
  pairwise_sym | infonce_sym           the trainer's symmetric direction (contrastive_trainer.py:202-206): pages [32, 780, 128] as
                                       query_embeddings against the gathered queries [256, 32, 128]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import colpali_amd as amd

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
unit = lambda *s: torch.nn.functional.normalize(torch.randn(s, generator=g, device=dev), dim=-1).to(torch.bfloat16).requires_grad_(True)  # noqa: E731
Q, D = unit(32, 32, 128), unit(256, 780, 128)
P, Qg = unit(32, 780, 128), unit(256, 32, 128)
for name, cls, kw in (("pairwise", amd.ColbertPairwiseCELoss, {}), ("infonce", amd.ColbertLoss, {}), ("smooth", amd.ColbertLoss, {"use_smooth_max": True})):
    for suffix, (a, b) in (("", (Q, D)), ("_sym", (P, Qg))):
        if which not in ("all", name + suffix) or (suffix and name == "smooth"):
            continue
        for _ in range(5):
            a.grad = b.grad = None
            cls(**kw)(a, b, offset=96).backward()
torch.cuda.synchronize()
print("done")
