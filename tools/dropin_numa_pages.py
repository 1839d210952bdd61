#!/usr/bin/env python
"""Where should the drop-in's gather threads sit when the CALLER'S page tensors live on the other socket than the GPU?
For pages first-touched on (the GPU's node | the other node) x gather threads on (the GPU's node | the pages' node): median / p95 of
15 score_multi_vector calls at BASELINE config 2's geometry (1000 x 1030 x 128 bf16 from a host list, 100 queries) and config 3's."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import colpali_amd as amd
from colpali_amd import _lib

dev = torch.device("cuda:0")
gpu_cpus = _lib.gpu_local_cpus(dev)
all_cpus = os.sched_getaffinity(0)
if gpu_cpus is None:
    print("one NUMA node (or sysfs silent): nothing to measure")
    sys.exit(0)
other_cpus = frozenset(all_cpus - gpu_cpus)
g = torch.Generator().manual_seed(1)
def unit(n): return torch.nn.functional.normalize(torch.randn(n, 128, generator=g), dim=-1).to(torch.bfloat16)
import ctypes
_libc = ctypes.CDLL(None, use_errno=True)
_libc.syscall.restype = ctypes.c_long


def migrate(t, node):
    """move_pages(2): every page of the tensor's storage to `node` (first touch does not decide: malloc recycles pages)."""
    a0 = t.data_ptr() & ~4095
    n = (t.data_ptr() + t.numel() * t.element_size() - a0 + 4095) // 4096
    pages = (ctypes.c_void_p * n)(*[a0 + 4096 * i for i in range(n)])
    nodes = (ctypes.c_int * n)(*([node] * n))
    status = (ctypes.c_int * n)()
    _libc.syscall(ctypes.c_long(279), ctypes.c_int(0), ctypes.c_ulong(n), pages, nodes, status, ctypes.c_int(2))   # MPOL_MF_MOVE


def node_of_cpus(cpus):
    for node in range(8):
        c = _lib.node_cpus(node)
        if c and c & cpus:
            return node
    return 0
for geom, lens in (("config 2 (1000 x 1030)", [1030] * 1000), ("config 3 (1000 x U{267..779})", torch.randint(267, 780, (1000,), generator=g).tolist())):
    for where, cpus in (("GPU's node", gpu_cpus), ("other node", other_cpus)):
        qs = [unit(32) for _ in range(100)]
        ps = [unit(n) for n in lens]
        for t in qs + ps:
            migrate(t, node_of_cpus(cpus))
        nodes = _lib.nodes_of_addresses([p.data_ptr() for p in ps[::125]])
        from colpali_amd import corpus as _C
        runs = [("gpu", 8), ("pages", 8), ("gpu", 8), ("pages", 8)] if os.environ.get("NUMA_THREADS_SWEEP") != "1" else \
               [("gpu", 8), ("gpu", 12), ("gpu", 16), ("gpu", 6), ("gpu", 8), ("gpu", 12), ("gpu", 16), ("gpu", 6)]
        for policy, threads in runs:
            _lib.GATHER_NODE_POLICY = policy
            _C._COPY_THREADS = threads
            for _ in range(3):
                amd.score_multi_vector(qs, ps, device="cuda:0")
            ts = []
            for _ in range(15):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                out = amd.score_multi_vector(qs, ps, device="cuda:0")
                ts.append((time.perf_counter() - t0) * 1e3)
            ts.sort()
            print(f"{geom:30s} pages on the {where} (move_pages says nodes {sorted(set(nodes))})  gather threads: {policy:5s} x {threads:2d}  median {ts[7]:6.2f} ms  p95 {ts[13]:6.2f}  min {ts[0]:6.2f}  checksum {float(out.double().sum()):.4f}", flush=True)
        del qs, ps
