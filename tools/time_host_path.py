#!/usr/bin/env python
"""score_multi_vector(device="cpu") -- the library's host-core scorer -- over thread counts on this host (128 queries x 1024 docs)."""
import os, sys, time, subprocess
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import colpali_amd as amd

print(subprocess.run("lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Core|Socket|avx512f' | cut -c1-200; grep -o -m1 'avx512[a-z_0-9]*' /proc/cpuinfo | sort -u | tr '\\n' ' '",
                     shell=True, capture_output=True, text=True).stdout)
g = torch.Generator().manual_seed(0)
unit = lambda n: torch.nn.functional.normalize(torch.randn(n, 128, generator=g), dim=-1).to(torch.bfloat16)   # noqa: E731
qs, ps = [unit(32) for _ in range(128)], [unit(1024) for _ in range(1024)]
amd.score_multi_vector(qs[:4], ps[:16], device="cpu")
for nt in (1, 8, 32, 64, 128, 256):
    os.environ["COLPALI_AMD_HOST_THREADS"] = str(nt)
    n_d = 64 if nt == 1 else 1024
    amd.score_multi_vector(qs, ps[:n_d], device="cpu")
    t0 = time.perf_counter()
    amd.score_multi_vector(qs, ps[:n_d], device="cpu")
    dt = time.perf_counter() - t0
    print(f"threads {nt:4d}: {128 * n_d / dt:10.0f} pairs/s  {128 * n_d * 8.39e6 / dt / 1e9:8.0f} GFLOP/s", flush=True)
