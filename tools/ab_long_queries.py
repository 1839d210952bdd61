#!/usr/bin/env python
"""Long queries (pages as queries: 780 tokens = 25 token tiles) against a resident shard: msim_fwd WITH its workspace (128-token
segments on K1b, partial sums added in segment order) and WITHOUT it (the generic kernels), same inputs, scores compared."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import bench, colpali_amd as amd
from colpali_amd import _lib

docs = int(os.environ.get("AB_DOCS", "16384"))
dev = torch.device("cuda:0")
corpus = bench.make_shard(docs, 1024, dev, 1234)
L = _lib.lib()
st = torch.cuda.current_stream()
for n_q, Lq in ((1, 780), (4, 780), (16, 780), (4, 256), (64, 200)):
    q = bench.make_queries(n_q, Lq, dev, 3)
    res = {}
    for name, use_ws in (("segments on K1b", True), ("generic kernels", False)):
        nbytes = L.msim_fwd_workspace_bytes(0, n_q, Lq, docs, 128)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev) if use_ws else None
        out = torch.empty((n_q, docs), dtype=torch.float32, device=dev)
        ms = []
        for i in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st)
            rc = L.msim_fwd(0, _lib.ptr(q), n_q, Lq, _lib.ptr(corpus.blob), _lib.ptr(corpus.offsets), None, docs, 128, _lib.ptr(out), docs, 0,
                            _lib.ptr(ws), st.cuda_stream)
            b.record(st)
            torch.cuda.synchronize()
            assert rc == 0, L.msim_last_error()
            if i >= 2:
                ms.append(a.elapsed_time(b))
        t = sorted(ms)[len(ms) // 2]
        res[name] = (t, out)
        flop = 2.0 * n_q * Lq * docs * 1024 * 128
        print(f"{n_q:3d} queries x {Lq} tokens vs {docs} docs x 1024: {name}: {t:8.3f} ms  {flop / t / 1e9:7.0f} TFLOP/s", flush=True)
    d = (res["segments on K1b"][1] - res["generic kernels"][1]).abs().max() / res["generic kernels"][1].abs().max()
    print(f"    max relative difference between the two: {float(d):.2e}", flush=True)
