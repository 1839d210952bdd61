#!/usr/bin/env python
"""Embedding-head throughput (K3) vs the reference's three torch lines on the same GPU -- tuning aid."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import colpali_amd as amd

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for B, S, H in ((1000, 1030, 2048), (1000, 779, 1536), (256, 1030, 3584)):
    hidden = torch.randn((B, S, H), generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    weight = (torch.randn((128, H), generator=g, device=dev) / H**0.5).to(torch.bfloat16)
    bias = (torch.randn((128,), generator=g, device=dev) * 0.1).to(torch.bfloat16)
    mask = torch.ones((B, S), dtype=torch.long, device=dev)
    mask[:, S - 40:] = 0

    def ours():
        return amd.embedding_head(hidden, weight, bias, mask)

    def ref():
        proj = torch.nn.functional.linear(hidden, weight, bias)
        proj = proj / proj.norm(dim=-1, keepdim=True)
        return proj * mask.unsqueeze(-1)

    for name, fn in (("K3 fused head", ours), ("torch 3 lines ", ref)):
        for _ in range(2):
            fn()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
        for a, b in evs:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in evs)[3]
        byts = B * S * H * 2 + B * S * 256
        print(f"{name} B={B} S={S} H={H}: {ms:8.3f} ms  {byts/ms/1e6:7.0f} GB/s (hidden read + out write)  "
              f"{B*S/ms/1e3:8.2f} Mrows/s  {2.0*B*S*H*128/ms/1e9:7.1f} TF", flush=True)
    del hidden
