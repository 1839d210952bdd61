#!/usr/bin/env python
"""The drop-in call runs 7.7 ms by itself and 10.5 ms inside bench.py: which part of bench.py's process state costs the 3 ms?
The same 100 x 1000 x 1030 call: (a) fresh process, (b) with bench.py's 30 + 30 GiB of resident shards allocated, (c) after bench.py's
cpu_baseline (torch CPU work), with the phases of each."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import bench, colpali_amd as amd
from colpali_amd import scoring as S

torch.set_num_threads(min(torch.get_num_threads(), amd._lib.effective_cpus()))
g = torch.Generator().manual_seed(21)
unit = lambda n: torch.nn.functional.normalize(torch.randn(n, 128, generator=g), dim=-1).to(torch.bfloat16)  # noqa: E731
qs, ps = [unit(32) for _ in range(100)], [unit(1030) for _ in range(1000)]


def run(tag, n=40):
    for _ in range(3):
        amd.score_multi_vector(qs, ps, device="cuda:0")
    rows = []
    for _ in range(n):
        S.TIMELINE = []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        amd.score_multi_vector(qs, ps, device="cuda:0")
        torch.cuda.synchronize(); t1 = time.perf_counter()
        tl = dict(S.TIMELINE)
        rows.append(((t1 - t0) * 1e3, (tl["checked"] - t0) * 1e3, (tl["issued"] - tl["checked"]) * 1e3, (tl["done"] - tl["issued"]) * 1e3))
    S.TIMELINE = None
    med = lambda k: sorted(r[k] for r in rows)[len(rows) // 2]   # noqa: E731
    print(f"{tag:60s}: median {med(0):.2f} ms  p95 {sorted(r[0] for r in rows)[int(n*0.95)]:.2f} | queries + checks {med(1):.2f} | gather + H2D loop {med(2):.2f} | GPU tail {med(3):.2f}", flush=True)


run("fresh process")
dev = torch.device("cuda:0")
corpus = bench.make_shard(125000, 1024, dev, seed=1234)
zero = torch.zeros_like(corpus.blob)
run("with 30.5 + 30.5 GiB of resident shards allocated")
q = bench.make_queries(4, 32, dev, 99)
for _ in range(20):
    amd.maxsim_scores(q, corpus)
torch.cuda.synchronize()
run("after 20 launches over the resident shard")
bench.cpu_baseline(32, 1024)
run("after cpu_baseline (torch CPU scorer, 16 threads)")
bench.torch_gpu_reference(32, 1024)
run("after torch_gpu_reference")
bench.embed_head_numbers(amd, dev)
run("after embed_head_numbers")
