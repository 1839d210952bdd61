#!/usr/bin/env python
"""Times of the dense hard-max backward's kernels at BASELINE config 5's symmetric shape (pages [32, 780, 128] x queries [256, Lq, 128])
through HIP events around msim_fwd_transposed_route and msim_dense_t_bwd (the whole backward: images + dP + dR + sum).  With a
measurement build (make -C colpali_amd/csrc ab; COLPALI_AMD_LIB=tools/_ab/libmaxsim_ab.so) MSIM_DENSE_T_DBG switches parts of the
kernels off (1 no LDS-DMA in the loop, 2 no MFMAs, 4 no W build, 8 no fragment reads): where the time goes."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import colpali_amd as amd
from colpali_amd import loss as L_

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
Ld = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n_d = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n_q = int(sys.argv[3]) if len(sys.argv) > 3 else 32
unit = lambda *s: torch.nn.functional.normalize(torch.randn(s, generator=g, device=dev), dim=-1).to(torch.bfloat16)  # noqa: E731
P, R = unit(n_q, 780, 128), unit(n_d, Ld, 128)
G = torch.randn(n_q, n_d, generator=g, device=dev) * 0.01
scores, _, route = L_._dense_t_forward(P, R)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in evs)[reps // 2] * 1e3


print(f"Ld={Ld} n_d={n_d} n_q={n_q} dbg={os.environ.get('MSIM_DENSE_T_DBG', '0')}: forward+route {timed(lambda: L_._dense_t_forward(P, R)):.1f} us, "
      f"backward (images + dP + dR + sum) {timed(lambda: L_._dense_t_backward(P, R, G, route)):.1f} us", flush=True)

if os.environ.get("COLPALI_AMD_LIB") and os.environ.get("MSIM_DENSE_T_TRACE"):
    # s_memtime sums per wave of dP's phases (measurement build): loop | wait vmcnt | wait lgkmcnt | barrier | stage switch block
    buf = torch.zeros((7 * n_q * 8, 8), dtype=torch.int64, device=dev)
    os.environ["MSIM_DENSE_T_DBG_OUT"] = str(buf.data_ptr())
    L_._dense_t_backward(P, R, G, route)
    torch.cuda.synchronize()
    del os.environ["MSIM_DENSE_T_DBG_OUT"]
    b = buf.cpu().double()
    live = b[:, 0] > 0
    m = b[live].mean(dim=0)
    print(f"dP phases, mean over {int(live.sum())} live waves, s_memtime ticks (100 MHz => 10 ns): loop {m[0]:.0f}  vmcnt {m[1]:.0f}  lgkmcnt {m[2]:.0f}  "
          f"barrier {m[3]:.0f}  switch-block {m[4]:.0f}  stages {m[5]:.0f}", flush=True)
