#!/usr/bin/env python
"""Why does one drop-in call in N take 70-90 ms instead of 7?  Per call: wall time next to the deltas of /proc/vmstat (NUMA hinting
faults, migrated pages, minor faults, THP events) and of Python's GC counters, for a few variants of the same 100 x 1000 x 1030 call."""
import gc, os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import colpali_amd as amd
from colpali_amd import corpus as C

KEYS = ("numa_hint_faults", "numa_hint_faults_local", "numa_pages_migrated", "pgmigrate_success", "pgfault", "thp_fault_alloc", "numa_pte_updates")


def vmstat():
    out = {}
    for line in open("/proc/vmstat"):
        k, v = line.split()
        if k in KEYS:
            out[k] = int(v)
    return out


def rd(path):
    try:
        return open(path).read().strip()
    except Exception as e:
        return f"<{type(e).__name__}>"


print("numa_balancing:", rd("/proc/sys/kernel/numa_balancing"), " nodes:", rd("/sys/devices/system/node/online"), " cpus:", os.cpu_count(),
      " THP:", rd("/sys/kernel/mm/transparent_hugepage/enabled"), " affinity:", len(os.sched_getaffinity(0)), flush=True)
g = torch.Generator().manual_seed(21)
unit = lambda n: torch.nn.functional.normalize(torch.randn(n, 128, generator=g), dim=-1).to(torch.bfloat16)  # noqa: E731
qs, ps = [unit(32) for _ in range(100)], [unit(1030) for _ in range(1000)]


def run(tag, n=40, pre=None):
    if pre:
        pre()
    for _ in range(3):
        amd.score_multi_vector(qs, ps, device="cuda:0")
    rows = []
    for _ in range(n):
        v0, g0 = vmstat(), gc.get_count()
        gs0 = [s["collections"] for s in gc.get_stats()]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        amd.score_multi_vector(qs, ps, device="cuda:0")
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        v1 = vmstat()
        gs1 = [s["collections"] for s in gc.get_stats()]
        rows.append((dt, {k: v1[k] - v0[k] for k in v0}, [b - a for a, b in zip(gs0, gs1)]))
    ts = sorted(r[0] for r in rows)
    print(f"## {tag}: median {ts[len(ts)//2]:.2f} ms  p95 {ts[int(len(ts)*0.95)]:.2f}  max {ts[-1]:.2f}", flush=True)
    for dt, dv, dg in rows:
        if dt > 2 * ts[len(ts) // 2]:
            print(f"   slow call {dt:7.2f} ms: vmstat deltas {dv}  gc collections {dg}", flush=True)
    fast = [r for r in rows if r[0] <= 1.2 * ts[len(ts) // 2]]
    if fast:
        avg = {k: sum(r[1][k] for r in fast) / len(fast) for k in fast[0][1]}
        print(f"   typical fast call: vmstat deltas {avg}", flush=True)


run("default")
run("gc disabled", pre=gc.disable)
gc.enable()
run("one copy thread", pre=lambda: setattr(C, "_COPY_THREADS", 1))
C._COPY_THREADS = 8
run("pages re-touched by the main thread first (x = p + 0)", pre=lambda: [p.add_(0) for p in ps])
pinned = [p.pin_memory() for p in ps]
ps = pinned
run("pages in pinned memory (the caller's choice)")
