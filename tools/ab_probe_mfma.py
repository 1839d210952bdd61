#!/usr/bin/env python
"""msim_probe_mfma: the matrix-core ceiling of this MI355X under its power budget on random unit-row bf16 operands
(and on zeros, where the chip clocks higher), next to the 2.5 PFLOP/s spec figure."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import bench, colpali_amd as amd

dev = torch.device("cuda:0")
from tools import probe
L = probe.lib()          # tools/probe/libmaxsim_probe.so (include/maxsim_probe.h): make -C tools/probe
rows = 256 * 16 * 4 * 32          # >= 256 x 8 x (5 + 1) x 32 for the five-tile plan
g = torch.Generator(device=dev).manual_seed(1)
X = torch.nn.functional.normalize(torch.randn((rows, 128), generator=g, device=dev), dim=-1).to(torch.bfloat16)
Z = torch.zeros_like(X)
sink = torch.zeros(4, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream()
for data, name in ((X, "random unit rows"), (Z, "zeros")):
    for variant, what in ((0, "A in registers"), (1, "A from LDS"), (2, "A in registers + max folds"), (3, "A from LDS + max folds"),
                          (4, "16x16x32 tiles, registers"), (5, "16x16x32, A from LDS"), (6, "16x16x32, registers + folds"),
                          (7, "16x16x32, A from LDS + folds"), (8, "mix, 12 waves x 3 tiles"), (9, "mix, 16 waves x 2 tiles"),
                          (10, "registers, 12 waves x 3"), (11, "registers, 16 waves x 2"),
                          (12, "K1b body: 8 waves x 4 tiles"), (16, "K1b body, 8x4, A in registers"), (17, "K1b body: 4 waves x 4 tiles"),
                          (13, "K1b body: 4 waves x 8 tiles"), (14, "K1b body, 4x8, A in registers"), (15, "K1b body: 4 waves x 6 tiles"),
                          (18, "K1b body 4x8 + prefetch"), (19, "K1b body 4x6 + prefetch"),
                          (20, "4x8 + prefetch + deferred fold"), (21, "4x8 + pf + deferred, pinned"), (22, "4x6 + pf + deferred fold"),
                          (23, "4x8 A in regs, deferred fold"), (24, "K1b body: 8 waves x 5 tiles"))[int(os.environ.get("PROBE_FROM", "0")):]:
        for iters in (20000,):
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
            mix = {12: (8, 4), 16: (8, 4), 17: (4, 4), 13: (4, 8), 14: (4, 8), 15: (4, 6), 18: (4, 8), 19: (4, 6), 20: (4, 8), 21: (4, 8), 22: (4, 6), 23: (4, 8), 24: (8, 5)}
            if variant in mix:
                flop = 256 * mix[variant][0] * iters * mix[variant][1] * 16 * 16384
            else:
                flop = 256 * 8 * iters * 32 * 32768 if variant < 8 else 256 * 12 * iters * 48 * 16384 if variant in (8, 10) else 256 * 16 * iters * 32 * 16384
            print(f"{name:18s} variant {variant} ({what:28s}) iters {iters:6d}: {t:8.3f} ms  {flop / t / 1e9:7.0f} TFLOP/s = {flop / t / 1e9 / 2500:.3f} of 2.5 PF", flush=True)
