#!/usr/bin/env python
"""Loss forward+backward time at BASELINE config 5's per-rank shapes, ours vs the reference module on the same GPU."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import colpali_amd as amd

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
B, C, Lq, Ld = 32, 256, 32, 780
Q = torch.nn.functional.normalize(torch.randn((B, Lq, 128), generator=g, device=dev), dim=-1).to(torch.bfloat16).requires_grad_(True)
D = torch.nn.functional.normalize(torch.randn((C, Ld, 128), generator=g, device=dev), dim=-1).to(torch.bfloat16).requires_grad_(True)


def ref_loss(kind, smooth):
    """colpali_engine/loss/late_interaction_losses.py:296-313 / :152-164 with the reference's own torch calls."""
    def f(q, d):
        raw = torch.einsum("bnd,csd->bcns", q, d)
        scores = (0.1 * torch.logsumexp(raw / 0.1, dim=3)).sum(2) if smooth else raw.amax(dim=3).sum(dim=2)
        if kind == "pairwise":
            pos = scores.diagonal()
            top2 = scores.topk(2, dim=1).values
            neg = torch.where(top2[:, 0] == pos, top2[:, 1], top2[:, 0])
            return torch.nn.functional.softplus(neg - pos).mean()
        return torch.nn.functional.cross_entropy(scores / 0.02, torch.arange(B, device=dev))
    return f


def timed(fn, reps=5):
    for _ in range(2):
        Q.grad = D.grad = None
        fn().backward()
    ts = []
    for _ in range(reps):
        Q.grad = D.grad = None
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn().backward()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3


def timed_graph(loss_fn, reps=20):
    """The same step captured once as a hipGraph and replayed: what a trainer that graphs its step pays."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            Q.grad = D.grad = None
            loss_fn(Q, D).backward()
    torch.cuda.current_stream().wait_stream(s)
    Q.grad = D.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        loss_fn(Q, D).backward()
    graph.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        graph.replay()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3


for name, ours, kind, smooth in (("ColbertPairwiseCELoss", amd.ColbertPairwiseCELoss(normalize_scores=False), "pairwise", False),
                                 ("ColbertPairwiseCELoss norm", amd.ColbertPairwiseCELoss(), "pairwise", False),
                                 ("ColbertLoss", amd.ColbertLoss(normalize_scores=False), "infonce", False),
                                 ("ColbertLoss smooth-max", amd.ColbertLoss(normalize_scores=False, use_smooth_max=True), "infonce", True)):
    t_ours = timed(lambda: ours(Q, D))
    torch.cuda.reset_peak_memory_stats()
    ours(Q, D).backward(); torch.cuda.synchronize(); m_ours = torch.cuda.max_memory_allocated() / 2**20
    rf = ref_loss(kind, smooth)
    t_ref = timed(lambda: rf(Q, D))
    torch.cuda.reset_peak_memory_stats()
    rf(Q, D).backward(); torch.cuda.synchronize(); m_ref = torch.cuda.max_memory_allocated() / 2**20
    t_graph = timed_graph(ours)
    print(f"{name:26s} B={B} C={C} Lq={Lq} Ld={Ld}: ours {t_ours:7.3f} ms eager, {t_graph:7.3f} ms as one hipGraph / peak {m_ours:7.0f} MiB   reference on this GPU {t_ref:7.3f} ms / peak {m_ref:7.0f} MiB", flush=True)
