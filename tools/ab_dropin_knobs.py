#!/usr/bin/env python
"""The drop-in call from host lists (BASELINE config 2: 100 queries x 1000 pages of 1030 rows) under the upload knobs of
colpali_amd/corpus.py -- COLPALI_AMD_EDGE_CHUNK_MB (first / last chunk of the pipelined upload), COLPALI_AMD_COPY_THREADS,
COLPALI_AMD_STAGING_MB: one line per process, 31 calls."""
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
time.sleep(0.5)                      # the container's CPU quota recovers from the generator's threads
for _ in range(4):
    ref = amd.score_multi_vector(qs, ps, device="cuda:0")
ts = []
for _ in range(31):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = amd.score_multi_vector(qs, ps, device="cuda:0")
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
assert torch.equal(out, ref)
ts.sort()
knobs = " ".join(f"{k[12:]}={os.environ[k]}" for k in sorted(os.environ) if k.startswith("COLPALI_AMD_") and k != "COLPALI_AMD_LIB")
cpus = amd._lib.gpu_local_cpus(torch.device("cuda:0"))
knobs += f" | gpu-local cpus: {'-' if cpus is None else len(cpus)}, main thread now on cpu {__import__('ctypes').CDLL(None).sched_getcpu()}"
print(f"{knobs or 'defaults':58s} threads {C._COPY_THREADS:2d}  median {ts[15]:6.2f} ms  p95 {ts[29]:6.2f}  min {ts[0]:6.2f}  max {ts[-1]:6.2f}   "
      f"checksum {float(out.double().sum()):.6f}", flush=True)
