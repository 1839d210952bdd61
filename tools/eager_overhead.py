#!/usr/bin/env python
"""Where the HOST time of an eager loss step goes (the reference trainer does not capture graphs): cProfile of 200 eager
forward + backward steps of ColbertPairwiseCELoss / ColbertLoss at BASELINE config 5's per-rank shape, both trainer directions."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import colpali_amd as amd

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
unit = lambda *s: torch.nn.functional.normalize(torch.randn(s, generator=g, device=dev), dim=-1).to(torch.bfloat16).requires_grad_(True)  # noqa: E731
Q, D, P, Qg = unit(32, 32, 128), unit(256, 780, 128), unit(32, 780, 128), unit(256, 32, 128)
which = sys.argv[1] if len(sys.argv) > 1 else "ColbertPairwiseCELoss"
mod = getattr(amd, which)()


def step():
    for t in (Q, D, P, Qg):
        t.grad = None
    l1 = mod(query_embeddings=Q, doc_embeddings=D, offset=96)
    l2 = mod(query_embeddings=P, doc_embeddings=Qg, offset=96)
    ((l1 + l2) / 2).backward()


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    step()
torch.cuda.synchronize()
print(f"{which}: {1e3 * (time.perf_counter() - t0) / 200:.3f} ms per eager step (both directions)")
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:6000])
