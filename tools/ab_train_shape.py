#!/usr/bin/env python
"""The scores launch of the training step alone (BASELINE config 5's per-rank shape: 32 queries x 32 tokens against 256 pages x 780
rows, and the symmetric direction) under the MSIM_* plan knobs of the measurement build: device time per launch from a replayed
hipGraph of 20 launches, and a bitwise check against the first configuration run in this process' parent (AB_REF file)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import colpali_amd as amd

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
B, C, Lq, Ld = 32, 256, 32, 780
Q = torch.nn.functional.normalize(torch.randn((B, Lq, 128), generator=g, device=dev), dim=-1).to(torch.bfloat16)
D = torch.nn.functional.normalize(torch.randn((C, Ld, 128), generator=g, device=dev), dim=-1).to(torch.bfloat16)
tag = os.environ.get("AB_TAG", "default")
for name, q, d in (("forward direction  32x32 vs 256x780", Q, D), ("symmetric 32x780 vs 256x32", D[:B].contiguous(), Q.repeat(8, 1, 1).contiguous())):
    corpus = amd.pack_passages(d, dev, batch_size=1 << 30)
    out = torch.empty((q.shape[0], d.shape[0]), dtype=torch.float32, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            amd.maxsim_scores(q, corpus, out=out)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            for _ in range(20):
                amd.maxsim_scores(q, corpus, out=out)
        ts = []
        for _ in range(7):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s); graph.replay(); b.record(s)
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / 20 * 1e3)
    us = sorted(ts)[len(ts) // 2]
    flop = 2.0 * q.shape[0] * q.shape[1] * d.shape[0] * d.shape[1] * 128
    print(f"{tag:24s} {name:38s}: {us:7.1f} us per launch  {flop / us / 1e6:7.0f} TFLOP/s = {flop / us / 1e6 / 2500:.3f} of 2.5 PF   checksum {float(out.double().sum()):.6f}", flush=True)
