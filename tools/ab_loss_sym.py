#!/usr/bin/env python
"""Forward + backward time of the in-batch losses in the SYMMETRIC direction of the reference trainer (compute_symetric_loss:
pages as query_embeddings [B, 780, 128], queries as doc_embeddings [B, 32, 128]) next to the forward direction, and the reference
modules on the same GPU."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import colpali_amd as amd
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
B = int(os.environ.get("AB_B", "32"))
pages = torch.nn.functional.normalize(torch.randn((B, 780, 128), generator=g, device=dev), dim=-1).to(torch.bfloat16)
queries = torch.nn.functional.normalize(torch.randn((B, 32, 128), generator=g, device=dev), dim=-1).to(torch.bfloat16)
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
for cls in ("ColbertPairwiseCELoss", "ColbertLoss"):
    mod = getattr(amd, cls)()
    def step(a, b):
        a = a.detach().requires_grad_(True); b = b.detach().requires_grad_(True)
        mod(a, b).backward()
    print(f"{cls}: forward direction {timed(lambda: step(queries, pages)):.3f} ms, symmetric direction {timed(lambda: step(pages, queries)):.3f} ms", flush=True)
