#!/usr/bin/env python
"""Second look at the 70-90 ms stalls of the drop-in call: (a) the raw pinned H2D copy of the same 264 MB by itself, 200 times;
(b) the product's own phase stamps per call (checks | gather + H2D issue loop | GPU tail); (c) the same call with the upload on the
caller's stream instead of the copy stream."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import colpali_amd as amd
from colpali_amd import scoring as S

def rd(path):
    try:
        return open(path).read().strip().replace("\n", " | ")
    except Exception as e:
        return f"<{type(e).__name__}>"


def throttle():
    """(nr_throttled, throttled time) of this process' cgroup: v2 (cpu.stat in the unified hierarchy) or v1 (cpu/cpu.stat)."""
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            d = dict(line.split() for line in open(path))
            return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", d.get("throttled_time", 0)))
        except Exception:
            continue
    return (-1, -1)


print("cgroup cpu.max:", rd("/sys/fs/cgroup/cpu.max"), " v1 quota/period:", rd("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), rd("/sys/fs/cgroup/cpu/cpu.cfs_period_us"),
      " cpu.stat:", rd("/sys/fs/cgroup/cpu.stat")[:200], rd("/sys/fs/cgroup/cpu/cpu.stat")[:200], " cpus:", os.cpu_count(),
      " affinity:", len(os.sched_getaffinity(0)), " loadavg:", rd("/proc/loadavg"), flush=True)
dev = torch.device("cuda:0")
nbytes = 1000 * 1030 * 256
pin = torch.empty((nbytes,), dtype=torch.uint8, pin_memory=True)
devb = torch.empty((nbytes,), dtype=torch.uint8, device=dev)


def dist(ts, tag):
    ts = sorted(ts)
    slow = [t for t in ts if t > 2 * ts[len(ts) // 2]]
    print(f"{tag}: n {len(ts)}  median {ts[len(ts)//2]:.2f} ms  p95 {ts[int(len(ts)*0.95)]:.2f}  max {ts[-1]:.2f}  slow (> 2 x median): {len(slow)}  {[round(t,1) for t in slow[:12]]}", flush=True)


for piece in (32 << 20, nbytes):
    ts = []
    for _ in range(200):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for o in range(0, nbytes, piece):
            devb[o:o + piece].copy_(pin[o:o + piece], non_blocking=True)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    dist(ts, f"raw pinned H2D of {nbytes/1e6:.0f} MB in pieces of {piece/2**20:.0f} MiB")
ts = []
for _ in range(200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x = torch.empty((nbytes,), dtype=torch.uint8, device=dev); x[:16].zero_(); del x
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
dist(ts, "torch.empty(264 MB) + free")
g = torch.Generator().manual_seed(21)
unit = lambda n: torch.nn.functional.normalize(torch.randn(n, 128, generator=g), dim=-1).to(torch.bfloat16)  # noqa: E731
qs, ps = [unit(32) for _ in range(100)], [unit(1030) for _ in range(1000)]
for _ in range(3):
    amd.score_multi_vector(qs, ps, device="cuda:0")
rows = []
import resource
for _ in range(80):
    S.TIMELINE = []
    th0, ru0 = throttle(), resource.getrusage(resource.RUSAGE_SELF)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    amd.score_multi_vector(qs, ps, device="cuda:0")
    torch.cuda.synchronize(); t1 = time.perf_counter()
    th1, ru1 = throttle(), resource.getrusage(resource.RUSAGE_SELF)
    tl = dict(S.TIMELINE)
    rows.append(((t1 - t0) * 1e3, (tl["begin"] - t0) * 1e3, (tl["checked"] - tl["begin"]) * 1e3, (tl["issued"] - tl["checked"]) * 1e3,
                 (tl["done"] - tl["issued"]) * 1e3, (t1 - tl["done"]) * 1e3, th1[0] - th0[0], (th1[1] - th0[1]) / 1e3,
                 (ru1.ru_utime - ru0.ru_utime + ru1.ru_stime - ru0.ru_stime) * 1e3, ru1.ru_nivcsw - ru0.ru_nivcsw, ru1.ru_nvcsw - ru0.ru_nvcsw))
S.TIMELINE = None
dist([r[0] for r in rows], "score_multi_vector 100 x 1000 x 1030")
med = sorted(r[0] for r in rows)[len(rows) // 2]
for r in rows:
    if r[0] > 2 * med:
        print("   slow call %.1f ms: pack_queries %.2f | checks %.2f | gather + H2D issue loop %.2f | GPU tail %.2f | after %.2f || cgroup throttled periods +%d (%.1f ms) | process CPU %.1f ms | ctx switches invol %d vol %d" % r, flush=True)
fast = [r for r in rows if r[0] <= 1.2 * med]
print("   typical call: total %.2f | pack_queries %.2f | checks %.2f | gather + H2D issue loop %.2f | GPU tail %.2f | after %.2f || throttled +%.2f (%.2f ms) | process CPU %.1f ms | ctx switches invol %.1f vol %.1f" %
      tuple(sum(r[i] for r in fast) / len(fast) for i in range(11)), flush=True)
