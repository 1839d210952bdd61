#!/usr/bin/env python
"""Phase stamps (s_memtime, 100 MHz) of one workgroup of maxsim_bwd_dd_rows_kernel in the measurement build, ColbertLoss at config 5's shape."""
import ctypes, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import colpali_amd as amd
from colpali_amd import _lib

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
B, C, Lq, Ld = 32, 256, 32, 780
Q = torch.nn.functional.normalize(torch.randn((B, Lq, 128), generator=g, device=dev), dim=-1).to(torch.bfloat16).requires_grad_(True)
D = torch.nn.functional.normalize(torch.randn((C, Ld, 128), generator=g, device=dev), dim=-1).to(torch.bfloat16).requires_grad_(True)
loss = amd.ColbertLoss(normalize_scores=False)
L = _lib.lib()
L.msim_ab_rows_trace.argtypes = [ctypes.c_void_p]
for it in range(4):
    Q.grad = D.grad = None
    loss(Q, D).backward()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    assert L.msim_ab_rows_trace(buf) == 0
    st = list(buf)[:9]
    names = ["range + pairs", "zero counts", "entries + histogram", "scan", "scatter", "rows: first 256", "rows: the rest", "end"]
    d = [(st[i + 1] - st[i]) / 100.0 for i in range(8)]
    print("  ".join(f"{n} {x:.2f} us" for n, x in zip(names, d)), f"| total {(st[8] - st[0]) / 100.0:.2f} us")
