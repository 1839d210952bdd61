#!/usr/bin/env python
"""The ridge probe the round-5 review asked for (one GPU-minute): the body a barrier-free band for 10 queries x 32 tokens would
run -- ONE 512-register wave per SIMD holding all 20 units (10 tiles = 320 B-operand registers, 192 of them AGPRs), operand
fragments from a wave-private LDS slab, folds next to the MFMAs -- with NO DMA, NO barrier and NO HBM traffic, i.e. an upper bound
of what such a kernel could reach, next to the shipped two-waves-per-SIMD body (variant 12) and the 8-tile forms of round 3.
Decision rule: build the band only if this reaches 0.80 of 2.5 PFLOP/s on zeros."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from tools import probe

dev = torch.device("cuda:0")
L = probe.lib()
rows = 256 * 16 * 4 * 32
g = torch.Generator(device=dev).manual_seed(1)
X = torch.nn.functional.normalize(torch.randn((rows, 128), generator=g, device=dev), dim=-1).to(torch.bfloat16)
Z = torch.zeros_like(X)
sink = torch.zeros(4, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream()
VARIANTS = ((12, "shipped K1b body: 8 waves x 4 tiles", 8, 4), (24, "8 waves x 5 tiles (4 in AGPRs)", 8, 5),
            (13, "4 waves x 8 tiles", 4, 8), (20, "4 x 8 + prefetch + deferred fold", 4, 8),
            (25, "RIDGE 4 waves x 10 tiles", 4, 10), (26, "RIDGE 4 x 10 + prefetch", 4, 10),
            (27, "RIDGE 4 x 10 + prefetch + deferred fold", 4, 10), (28, "RIDGE 4 x 10, A in registers, deferred fold", 4, 10))
best = {}
for data, name in ((X, "random unit rows"), (Z, "zeros")):
    for variant, what, waves, nt in VARIANTS:
        iters = 20000
        ms = []
        for i in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st)
            rc = L.msim_probe_mfma(variant, data.data_ptr(), rows, iters, sink.data_ptr(), st.cuda_stream)
            b.record(st)
            torch.cuda.synchronize()
            assert rc == 0, L.msim_probe_last_error()
            if i >= 1:
                ms.append(a.elapsed_time(b))
        t = sorted(ms)[len(ms) // 2]
        flop = 256 * waves * iters * nt * 16 * 16384
        frac = flop / t / 1e9 / 2500
        if nt == 10:
            best[name] = max(best.get(name, 0.0), frac)
        print(f"{name:18s} variant {variant:2d} ({what:44s}): {t:8.3f} ms  {flop / t / 1e9:7.0f} TFLOP/s = {frac:.3f} of 2.5 PF", flush=True)
print(f"ridge body, best of the 10-tile forms: zeros {best['zeros']:.3f}, random unit rows {best['random unit rows']:.3f} of 2.5 PF "
      f"-> {'BUILD (>= 0.80 on zeros)' if best['zeros'] >= 0.80 else 'do not build (< 0.80 on zeros, before any DMA or HBM traffic)'}")
