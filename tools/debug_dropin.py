import sys, time, cProfile, pstats, io
sys.path.insert(0, "/root/repo")
import torch, colpali_amd as amd
g = torch.Generator().manual_seed(21)
def unit(n): return torch.nn.functional.normalize(torch.randn(n, 128, generator=g), dim=-1).to(torch.bfloat16)
big = torch.empty((30 << 30,), dtype=torch.uint8, device="cuda:0")
for name, lens in (("c2", [1030] * 1000), ("c3", torch.randint(267, 780, (1000,), generator=g).tolist())):
    qs, ps = [unit(32) for _ in range(100)], [unit(n) for n in lens]
    amd.score_multi_vector(qs, ps, device="cuda:0")
    ts = []
    for _ in range(7):
        torch.cuda.synchronize(); t0 = time.perf_counter(); amd.score_multi_vector(qs, ps, device="cuda:0"); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(name, ["%.1f" % t for t in ts])
    if name == "c3":
        pr = cProfile.Profile(); pr.enable()
        for _ in range(3): amd.score_multi_vector(qs, ps, device="cuda:0")
        pr.disable(); s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14); print(s.getvalue()[:3000])
