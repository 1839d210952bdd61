#!/usr/bin/env python
"""Where the cycles of K3 (embed_head_kernel, barrier variant) go: per wave of workgroup 0, cycles per phase of the chunk loop, summed by
s_memtime stamps inside the kernel (MSIM_HEAD_TRACE_PTR debug knob).  Phases of a compute wave per K chunk: issue (4 LDS-DMA pieces of
its own rows), vmcnt (wait until the chunk's hidden states have landed), barrier (weight chunk landed + everyone done with the previous
one), compute (20 operand reads + 16 MFMAs, stamped after the accumulators are readable); plus the per-tile epilogue.  Loader wave: issue
(16 weight pieces), vmcnt (weight chunk landed), barrier."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
os.environ.setdefault("COLPALI_AMD_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ab", "libmaxsim_trace.so"))   # `make -C colpali_amd/csrc trace`
dev = torch.device("cuda:0")
trace = torch.zeros(9 * 8, dtype=torch.int64, device=dev)
os.environ["MSIM_HEAD_TRACE_PTR"] = str(trace.data_ptr())
import colpali_amd as amd
g = torch.Generator(device=dev).manual_seed(0)
for B, S, H in ((1000, 1030, 2048), (1000, 779, 1536), (256, 1030, 3584)):
    hidden = torch.randn((B, S, H), generator=g, device=dev).to(torch.bfloat16)
    W = (torch.randn((128, H), generator=g, device=dev) / H**0.5).to(torch.bfloat16)
    b = torch.zeros(128, dtype=torch.bfloat16, device=dev)
    mask = torch.ones((B, S), dtype=torch.long, device=dev)
    for _ in range(3):
        amd.embedding_head(hidden, W, b, mask)
    trace.zero_()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); amd.embedding_head(hidden, W, b, mask); e.record(); torch.cuda.synchronize()
    t = trace.view(9, 8).cpu()
    ms = a.elapsed_time(e)
    print(f"H={H}: {ms:.3f} ms  {B*S*H*2/ms/1e6:.0f} GB/s; chunks per wave {int(t[0,4])}; cycles per chunk (s_memtime ticks = shader cycles):")
    for w in range(8):
        n = max(int(t[w, 4]), 1)
        print(f"  compute wave {w}: issue {int(t[w,0])/n:7.0f}  vmcnt {int(t[w,1])/n:7.0f}  barrier {int(t[w,2])/n:7.0f}  compute {int(t[w,3])/n:7.0f}  epilogue/chunk {int(t[w,5])/n:6.0f}  total {(int(t[w,0])+int(t[w,1])+int(t[w,2])+int(t[w,3])+int(t[w,5]))/n:7.0f}")
    n = max(int(t[8, 4]), 1)
    print(f"  loader wave   : issue {int(t[8,0])/n:7.0f}  vmcnt {int(t[8,1])/n:7.0f}  barrier {int(t[8,2])/n:7.0f}")
    del hidden
