#!/usr/bin/env python
"""End-to-end drop-in call from host lists (BASELINE configs 2/3 geometry): ours vs the reference's scorer on the same GPU."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import bench
import colpali_amd as amd

g = torch.Generator().manual_seed(1)
def unit(n): return torch.nn.functional.normalize(torch.randn(n, 128, generator=g), dim=-1).to(torch.bfloat16)
for name, lens in (("C2 ColPali 1000 x 1030", [1030] * 1000),
                   ("C3 ColQwen2 1000 x U{267..779}", torch.randint(267, 780, (1000,), generator=g).tolist())):
    qs = [unit(32) for _ in range(100)]
    ps = [unit(n) for n in lens]
    def timed(fn, reps=5):
        fn(); ts = []
        for _ in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2] * 1e3
    ours = timed(lambda: amd.score_multi_vector(qs, ps, device="cuda:0"))
    ref = timed(lambda: bench.reference_scorer(qs, ps, "cuda:0"), reps=3)
    dev = torch.device("cuda:0")
    t_pq = timed(lambda: amd.pack_queries(qs, dev))
    t_pp = timed(lambda: amd.pack_passages(ps, dev))
    corpus = amd.pack_passages(ps, dev); q = amd.pack_queries(qs, dev)
    t_k = timed(lambda: amd.maxsim_scores(q, corpus))
    t_cat = timed(lambda: torch.cat([p.reshape(-1, 128) for p in ps], dim=0))
    print(f"{name}: ours {ours:8.2f} ms ({100*len(ps)/ours/1e3:6.2f} Mpairs/s)  reference on cuda:0 {ref:8.2f} ms  speedup {ref/ours:5.1f}x | "
          f"pack_queries {t_pq:.2f}  pack_passages {t_pp:.2f} (host cat {t_cat:.2f})  kernel {t_k:.3f} ms", flush=True)

# where the drop-in's time goes (C2 geometry): cProfile of one call
import cProfile, pstats, io
qs = [unit(32) for _ in range(100)]
ps = [unit(1030) for _ in range(1000)]
amd.score_multi_vector(qs, ps, device="cuda:0")
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    amd.score_multi_vector(qs, ps, device="cuda:0")
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
