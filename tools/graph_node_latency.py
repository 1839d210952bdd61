#!/usr/bin/env python
"""How long is one dependent kernel boundary inside a hipGraph on this box?  N tiny dependent launches (one element) captured as one
graph: device time per node.  The floor under every small kernel of the loss step."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch

x = torch.zeros(1, device="cuda:0")
big = torch.zeros(64 << 20, device="cuda:0", dtype=torch.uint8)
for n in (1, 10, 50, 200):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        x.add_(1)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            x.add_(1)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(f"graph of {n:4d} dependent one-element kernels: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us per replay = {e0.elapsed_time(e1) / 20 / n * 1e3:6.2f} us per node", flush=True)
# eager back-to-back (stream order) for comparison
for _ in range(10):
    x.add_(1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    x.add_(1)
e1.record()
torch.cuda.synchronize()
print(f"eager, 200 launches on one stream: {e0.elapsed_time(e1) / 200 * 1e3:6.2f} us per launch", flush=True)
