#!/usr/bin/env python
"""First / last chunk size of the drop-in's pipelined upload (colpali_amd/corpus.py: _EDGE_CHUNK_BYTES), toggled INSIDE one process
and interleaved -- process placement on the host moves the call by more than any knob does (tools/ab_dropin_knobs.sh: 6.2 - 8.5 ms
between processes of the same setting)."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import colpali_amd as amd
from colpali_amd import corpus as C

torch.set_num_threads(max(1, min(torch.get_num_threads(), amd._lib.effective_cpus())))
g = torch.Generator().manual_seed(21)
tok = torch.nn.functional.normalize(torch.randn(100 * 32 + 1000 * 1030, 128, generator=g), dim=-1).to(torch.bfloat16)
qs = [t.clone() for t in tok[:3200].split(32)]
ps = [t.clone() for t in tok[3200:].split(1030)]
del tok
time.sleep(0.5)
for _ in range(4):
    ref = amd.score_multi_vector(qs, ps, device="cuda:0")
settings = [0, 2, 4, 8, 16]
res = {e: [] for e in settings}
for rnd in range(6):
    for e in settings:
        C._EDGE_CHUNK_BYTES = e << 20
        amd.score_multi_vector(qs, ps, device="cuda:0")
        for _ in range(9):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = amd.score_multi_vector(qs, ps, device="cuda:0")
            torch.cuda.synchronize(); res[e].append((time.perf_counter() - t0) * 1e3)
        assert torch.equal(out, ref)
for e in settings:
    ts = sorted(res[e])
    print(f"edge chunk {e:2d} MiB: median {ts[len(ts) // 2]:6.2f} ms  p95 {ts[int(len(ts) * 0.95)]:6.2f}  min {ts[0]:6.2f}  ({len(ts)} calls, interleaved)", flush=True)
