#!/usr/bin/env python
"""Where the cycles of K1b (maxsim_batch_kernel) go: per wave of workgroup 0, s_memtime ticks per phase of the chunk loop
(MSIM_BATCH_TRACE_PTR debug knob): vmcnt (own share of the chunk landed), convoy, barrier, DMA issue (4 LDS-DMA pieces of the refill),
slabs (operand reads + MFMAs + folds of the chunk), document epilogue."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
os.environ.setdefault("COLPALI_AMD_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ab", "libmaxsim_trace.so"))   # `make -C colpali_amd/csrc trace`
dev = torch.device("cuda:0")
trace = torch.zeros(8 * 8, dtype=torch.int64, device=dev)
os.environ["MSIM_BATCH_TRACE_PTR"] = str(trace.data_ptr())
import bench, colpali_amd as amd
docs = int(os.environ.get("AB_DOCS", "32768"))
doc_len = int(os.environ.get("AB_DOC_LEN", "1024"))
corpus = bench.make_shard(docs, doc_len, dev, 1234)
if os.environ.get("AB_ZERO") == "1":
    corpus.blob.zero_()
for nq in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "8,16,32,256").split(",")]:
    q = bench.make_queries(nq, 32, dev, 3)
    out = torch.empty((nq, docs), dtype=torch.float32, device=dev)
    for _ in range(2):
        amd.maxsim_scores(q, corpus, out=out)
    trace.zero_()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); amd.maxsim_scores(q, corpus, out=out); e.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(e)
    t = trace.view(8, 8).cpu()
    print(f"nq={nq}: {ms:.3f} ms  {2*nq*32*docs*doc_len*128/ms/1e9:.0f} TF; ticks per chunk and wave of workgroup 0 (100 MHz s_memtime? compare with ms):")
    for w in range(8):
        n = int(t[w, 6])
        if n == 0:
            continue
        tot = sum(int(t[w, i]) for i in range(6))
        names = (("vmcnt+convoy", "barrier", "issue", "token sums", "slabs", "doc ends") if os.environ.get("MSIM_BATCH_PACKED") == "1"
                 else ("vmcnt", "convoy", "barrier", "issue", "slabs", "epilogue"))       # K1bK (several documents per chunk) / K1b
        print(f"  wave {w}: chunks {n:6d}  " + "  ".join(f"{nm} {int(t[w, i]) / n:7.1f}" for i, nm in enumerate(names)) + f"  total {tot / n:8.1f}  (sum {tot} ticks)")
