#!/usr/bin/env python
"""Workload for the rocprofv3 counters of K3 (embed_head_kernel) alone: 500 ColPali pages x 1030 tokens x hidden 2048 bf16
(2.1 GB of hidden states, well past the 256 MiB Infinity Cache), 12 launches through colpali_amd.embedding_head."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import colpali_amd as amd

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
B, S, H = 500, 1030, int(os.environ.get("HEAD_H", "2048"))
hidden = torch.randn((B, S, H), generator=g, device=dev).to(torch.bfloat16)
W = (torch.randn((128, H), generator=g, device=dev) / H**0.5).to(torch.bfloat16)
b = (torch.randn((128,), generator=g, device=dev) * 0.1).to(torch.bfloat16)
mask = torch.ones((B, S), dtype=torch.long, device=dev)
mask[:, S - 6:] = 0
for _ in range(12):
    amd.embedding_head(hidden, W, b, mask)
torch.cuda.synchronize()
print("done")
