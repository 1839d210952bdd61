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
def ref_loss(kind):
    """colpali_engine/loss/late_interaction_losses.py:296-313 / :152-164 with the reference's own torch calls (class defaults:
    normalize_scores=True, temperature 1.0 / 0.02), on this GPU -- the module being replaced, timed beside ours."""
    def f(q, d):
        lengths = (q[:, :, 0] != 0).sum(dim=1)
        raw = torch.einsum("bnd,csd->bcns", q, d)
        scores = raw.amax(dim=3).sum(dim=2) / lengths.unsqueeze(1)
        if kind == "ColbertPairwiseCELoss":
            pos = scores.diagonal()
            top2 = scores.topk(2, dim=1).values
            neg = torch.where(top2[:, 0] == pos, top2[:, 1], top2[:, 0])
            return torch.nn.functional.softplus(neg - pos).mean()
        return torch.nn.functional.cross_entropy(scores / 0.02, torch.arange(q.shape[0], device=dev))
    return f


for cls in ("ColbertPairwiseCELoss", "ColbertLoss"):
    if os.environ.get("AB_ONLY") and os.environ["AB_ONLY"] != cls:
        continue
    mod = getattr(amd, cls)()
    ref = ref_loss(cls)

    def step(fn, a, b):
        a = a.detach().requires_grad_(True); b = b.detach().requires_grad_(True)
        fn(a, b).backward()

    ours_f, ours_s = timed(lambda: step(mod, queries, pages)), timed(lambda: step(mod, pages, queries))
    ref_f, ref_s = timed(lambda: step(ref, queries, pages)), timed(lambda: step(ref, pages, queries))
    print(f"{cls:22s} B={B}: forward direction ours {ours_f:.3f} ms / reference module on this GPU {ref_f:.3f} ms;   "
          f"symmetric direction (pages as queries) ours {ours_s:.3f} ms / reference {ref_s:.3f} ms", flush=True)
