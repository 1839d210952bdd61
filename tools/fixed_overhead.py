#!/usr/bin/env python
"""Fixed cost of the two big kernels of the loss step at BASELINE config 5's shape: K1b (forward direction: 32 queries x 32 tokens
against 256 pages) and K1t (symmetric direction: 32 pages against 256 queries of 32 tokens), one page per workgroup -- device time
as a function of the page length (slope = steady state per row, intercept = launch + prologue + epilogue)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import colpali_amd as amd
from colpali_amd import _lib, loss as L_

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
unit = lambda *s: torch.nn.functional.normalize(torch.randn(s, generator=g, device=dev), dim=-1).to(torch.bfloat16)  # noqa: E731


def dev_us(fn, reps=30):
    for _ in range(5):
        fn()
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):
        for _ in range(reps):
            fn()
    graph.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); graph.replay(); graph.replay(); graph.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * reps) * 1e3


lib = _lib.lib()
rows = []
for Ld in (130, 390, 780, 1560, 3120):
    Q, D = unit(32, 32, 128), unit(256, Ld, 128)           # forward direction
    corpus = L_._dense_corpus(D)
    out = torch.empty((32, 256), dtype=torch.float32, device=dev)
    t_b = dev_us(lambda: amd.maxsim_scores(Q, corpus, out=out))
    P, Qg = unit(32, Ld, 128), unit(256, 32, 128)           # symmetric direction
    out2 = torch.empty((32, 256), dtype=torch.float32, device=dev)

    def k1t():
        rc = lib.msim_fwd_transposed(0, _lib.ptr(P), 32, Ld, _lib.ptr(Qg), 256, 32, 128, _lib.ptr(out2), 256, None, _lib.current_stream_handle(dev))
        assert rc == 0
    t_t = dev_us(k1t) if Ld > 128 else float("nan")
    flop = 2.0 * 32 * 256 * 32 * Ld * 128
    rows.append((Ld, t_b, t_t))
    print(f"page rows {Ld:5d}: K1b {t_b:7.1f} us ({flop / t_b / 1e6 / 1e3:5.0f} TFLOP/s)   K1t {t_t:7.1f} us ({flop / t_t / 1e6 / 1e3:5.0f} TFLOP/s)   MFMA peak time {flop / 2.5e15 * 1e6:5.1f} us", flush=True)
(l0, b0, t0), (l1, b1, t1) = rows[2], rows[4]
print(f"slope between 780 and 3120 rows: K1b {(b1 - b0) / (l1 - l0) * 780:.1f} us per 780 rows, intercept {b0 - (b1 - b0) / (l1 - l0) * l0:.1f} us;   "
      f"K1t {(t1 - t0) / (l1 - l0) * 780:.1f} us per 780 rows, intercept {t0 - (t1 - t0) / (l1 - l0) * l0:.1f} us")
