#!/usr/bin/env python
"""What the 'GPU tail' of the drop-in call (1.7 ms behind the last uploaded byte) is made of: device time of the MaxSim launch over the
whole 1000-page corpus and over sub-ranges of it (the pipeline's launches: the whole blob + a slice of the absolute offsets), 100 queries x
32 tokens, and the D2H of the result."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import colpali_amd as amd
from colpali_amd.corpus import PackedCorpus

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(21)
unit = lambda n: torch.nn.functional.normalize(torch.randn(n, 128, generator=g), dim=-1).to(torch.bfloat16)  # noqa: E731
qs, ps = [unit(32) for _ in range(100)], [unit(1030) for _ in range(1000)]
q = amd.pack_queries(qs, dev)
corpus = amd.pack_passages(ps, dev, batch_size=128)
out = torch.empty((100, 1000), dtype=torch.float32, device=dev)


def dev_ms(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2], ts[-1]


print("whole corpus, one launch: %.3f ms (max %.3f)" % dev_ms(lambda: amd.maxsim_scores(q, corpus, out=out)))
for lo, hi in ((0, 256), (768, 1000), (0, 128), (0, 512)):
    part = PackedCorpus(blob=corpus.blob, offsets=corpus.offsets[lo:hi + 1], clamp0=None, lengths=corpus.lengths[lo:hi])
    print(f"passages {lo:4d}..{hi:4d} (whole blob, offsets slice): %.3f ms (max %.3f)" % dev_ms(lambda: amd.maxsim_scores(q, part, out=out[:, lo:hi])))
ts = []
for _ in range(20):
    torch.cuda.synchronize(); t0 = time.perf_counter(); out.cpu(); ts.append((time.perf_counter() - t0) * 1e3)
print("out.cpu() (pageable): median %.3f ms" % sorted(ts)[10])
import ctypes
L = amd._lib.lib()
plan = (ctypes.c_int32 * 5)()
L.msim_fwd_plan(q.offsets_host.data_ptr(), 100, 0, plan)
print("plan for the 100 queries (kernel, waves, units per wave, blocks, heaviest wave):", list(plan))
