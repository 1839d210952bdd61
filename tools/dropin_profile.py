#!/usr/bin/env python
"""Where the drop-in call from host lists spends its time (BASELINE config 2 geometry: 100 queries x 1000 pages of 1030 rows):
cProfile of score_multi_vector over a few calls, next to the raw pinned H2D bandwidth and the raw gather (memcpy) rate of this host."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import colpali_amd as amd
from colpali_amd import corpus as C

g = torch.Generator().manual_seed(21)
unit = lambda n: torch.nn.functional.normalize(torch.randn(n, 128, generator=g), dim=-1).to(torch.bfloat16)  # noqa: E731
qs, ps = [unit(32) for _ in range(100)], [unit(1030) for _ in range(1000)]
nbytes = sum(p.numel() * 2 for p in ps)
dev = torch.device("cuda:0")
for _ in range(3):
    amd.score_multi_vector(qs, ps, device="cuda:0")
ts = []
for _ in range(21):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    amd.score_multi_vector(qs, ps, device="cuda:0")
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
ts.sort()
print(f"score_multi_vector 100 x 1000 x 1030 from host lists: median {ts[10]:.2f} ms, p95 {ts[19]:.2f}, min {ts[0]:.2f}, max {ts[-1]:.2f}")
# raw H2D: one pinned buffer of the corpus size, and 32 MiB pieces
pin = torch.empty((nbytes,), dtype=torch.uint8, pin_memory=True)
devb = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
for piece in (nbytes, 32 << 20, 8 << 20):
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for o in range(0, nbytes, piece):
            devb[o:o + piece].copy_(pin[o:o + piece], non_blocking=True)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(f"pinned H2D of {nbytes / 1e6:.0f} MB in pieces of {piece / 2**20:.0f} MiB: {best * 1e3:.2f} ms = {nbytes / best / 1e9:.1f} GB/s")
# raw gather: msim_host_gather of the 1000 pages into the pinned buffer, by thread count
import numpy as np
L = amd._lib.lib()
srcs = np.asarray([p.data_ptr() for p in ps], dtype=np.uint64)
sizes = np.asarray([p.numel() * 2 for p in ps], dtype=np.int64)
offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
for th in (1, 2, 4, 8, 16, 32):
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        L.msim_host_gather(pin.data_ptr(), srcs.ctypes.data, offs.ctypes.data, sizes.ctypes.data, len(ps), th)
        best = min(best, time.perf_counter() - t0)
    print(f"msim_host_gather of 1000 pages into pinned memory, {th:2d} threads: {best * 1e3:.2f} ms = {nbytes / best / 1e9:.1f} GB/s")
print("copy threads the product uses:", C._COPY_THREADS, " staging bytes:", C.STAGING_BYTES)
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    amd.score_multi_vector(qs, ps, device="cuda:0")
pr.disable()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats("cumtime").print_stats(28)
