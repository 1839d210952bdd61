#!/usr/bin/env python
"""Hierarchical token pooling: GPU batch (msim_pool_*) vs the reference's per-page torch.mm + SciPy on the host cores,
and what pooling buys the scorer (bytes streamed per page / 3)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
import torch
import colpali_amd as amd

warnings.simplefilter("ignore")
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
n_pages, n, dim, pf = int(os.environ.get("AB_PAGES", "512")), 1030, 128, 3
proto = torch.nn.functional.normalize(torch.randn(60, dim, generator=g), dim=-1)
pages = [torch.nn.functional.normalize(proto[torch.randint(0, 60, (n,), generator=g)] + 0.2 * torch.randn(n, dim, generator=g), dim=-1).to(torch.bfloat16)
         for _ in range(n_pages)]
gpu_pages = [p.to(dev) for p in pages]
pooler = amd.HierarchicalTokenPooler()
pooler.pool_embeddings(gpu_pages[:8], pool_factor=pf)
torch.cuda.synchronize(); t0 = time.perf_counter()
pooled = pooler.pool_embeddings(gpu_pages, pool_factor=pf)
torch.cuda.synchronize(); t_gpu = time.perf_counter() - t0
print(f"GPU: {n_pages} pages x {n} x {dim} bf16, pool_factor {pf}: {t_gpu*1e3:.1f} ms = {t_gpu/n_pages*1e3:.3f} ms/page "
      f"({n_pages/t_gpu:.0f} pages/s); pooled lengths {pooled[0].shape[0]}", flush=True)

from scipy.cluster.hierarchy import fcluster, linkage
def ref_one(e):   # hierarchical_token_pooling.py:112-140 on one page
    ef = e.to(torch.float32).cpu()
    Z = linkage(1 - torch.mm(ef, ef.t()).numpy(), metric="euclidean", method="ward")
    lab = fcluster(Z, t=max(ef.shape[0] // pf, 1), criterion="maxclust") - 1
    return torch.stack([torch.nn.functional.normalize(ef[torch.from_numpy(lab == c)].mean(dim=0), p=2, dim=-1) for c in range(int(lab.max()) + 1)])
ref_one(pages[0])
t0 = time.perf_counter()
for p in pages[:6]:
    ref_one(p)
t_cpu = (time.perf_counter() - t0) / 6
print(f"reference (torch.mm + scipy, one page at a time on the host): {t_cpu*1e3:.1f} ms/page -> GPU batch is {t_cpu/(t_gpu/n_pages):.0f}x per page", flush=True)

# what it buys the scorer
full = amd.pack_passages(gpu_pages, dev, batch_size=None)
small = amd.pack_passages(pooled, dev, batch_size=None)
q = torch.nn.functional.normalize(torch.randn(4, 32, dim, generator=g), dim=-1).to(torch.bfloat16).to(dev)
for name, corpus in (("full pages", full), ("pooled pages", small)):
    for _ in range(3): amd.maxsim_scores(q, corpus)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in evs:
        a.record(); amd.maxsim_scores(q, corpus); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)[5]
    print(f"scoring 4 queries x {n_pages} {name}: {ms*1e3:.0f} us ({corpus.nbytes/1e6:.0f} MB streamed)", flush=True)
