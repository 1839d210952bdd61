#!/usr/bin/env python
"""In-process A/B of the drop-in pipeline's shape at BASELINE configs 2 / 3 (process-to-process placement noise is larger than the
effects): sub-range size (scoring._PIPE_RANGE_BYTES), first / last chunk of the upload (corpus._EDGE_CHUNK_BYTES / _EDGE_LAST_BYTES).
Every setting twice, interleaved; median / p95 of 15 calls."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import colpali_amd as amd
from colpali_amd import corpus as C, scoring as S

g = torch.Generator().manual_seed(1)
def unit(n): return torch.nn.functional.normalize(torch.randn(n, 128, generator=g), dim=-1).to(torch.bfloat16)
MB = 1 << 20
SETTINGS = (("48 MB ranges, edges 16/16 (round 5)", 48, 16, -1), ("16 MB ranges, edges 16/16", 16, 16, -1), ("16 MB ranges, edges 16/8", 16, 16, 8),
            ("16 MB ranges, edges 8/8", 16, 8, 8), ("16 MB ranges, edges 16/4", 16, 16, 4), ("24 MB ranges, edges 16/8", 24, 16, 8))
for geom, lens in (("config 2", [1030] * 1000), ("config 3", torch.randint(267, 780, (1000,), generator=g).tolist())):
    qs = [unit(32) for _ in range(100)]
    ps = [unit(n) for n in lens]
    for rep in range(2):
        for name, rng_mb, e0, e1 in SETTINGS:
            S._PIPE_RANGE_BYTES = rng_mb * MB
            C._EDGE_CHUNK_BYTES = e0 * MB
            C._EDGE_LAST_BYTES = -1 if e1 < 0 else e1 * MB
            for _ in range(3):
                amd.score_multi_vector(qs, ps, device="cuda:0")
            ts = []
            for _ in range(15):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                out = amd.score_multi_vector(qs, ps, device="cuda:0")
                ts.append((time.perf_counter() - t0) * 1e3)
            ts.sort()
            print(f"{geom} {name:38s} median {ts[7]:6.2f} ms  p95 {ts[13]:6.2f}  min {ts[0]:6.2f}  checksum {float(out.double().sum()):.4f}", flush=True)
