#!/usr/bin/env python
"""Which threads burn the 540 ms of CPU time one 9 ms drop-in call costs (and thereby exhaust the container's CPU quota: cgroup
cpu.max = 16 CPUs per 100 ms period, one throttled period per slow call)?  Per-thread utime + stime from /proc/self/task/*/stat
around a batch of calls, by thread name, for a few settings."""
import collections, os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import colpali_amd as amd

TICK = os.sysconf("SC_CLK_TCK")


def threads():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            st = open(f"/proc/self/task/{tid}/stat").read()
            name = st[st.index("(") + 1: st.rindex(")")]
            f = st[st.rindex(")") + 2:].split()
            out[int(tid)] = (name, (int(f[11]) + int(f[12])) / TICK * 1e3)     # utime + stime, ms
        except Exception:
            pass
    return out


def throttled():
    try:
        d = dict(line.split() for line in open("/sys/fs/cgroup/cpu.stat"))
        return int(d.get("nr_throttled", 0))
    except Exception:
        return -1


g = torch.Generator().manual_seed(21)
unit = lambda n: torch.nn.functional.normalize(torch.randn(n, 128, generator=g), dim=-1).to(torch.bfloat16)  # noqa: E731
qs, ps = [unit(32) for _ in range(100)], [unit(1030) for _ in range(1000)]
print("torch threads:", torch.get_num_threads(), " OMP_WAIT_POLICY:", os.environ.get("OMP_WAIT_POLICY"), " GOMP_SPINCOUNT:", os.environ.get("GOMP_SPINCOUNT"),
      " cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None, flush=True)


def batch(tag, n=30):
    for _ in range(3):
        amd.score_multi_vector(qs, ps, device="cuda:0")
    time.sleep(0.3)
    t_before, th0 = threads(), throttled()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        amd.score_multi_vector(qs, ps, device="cuda:0")
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    t_after = threads()
    by_name = collections.Counter()
    n_by_name = collections.Counter()
    for tid, (name, ms) in t_after.items():
        d = ms - t_before.get(tid, (name, 0.0))[1]
        if d > 0:
            by_name[name] += d
            n_by_name[name] += 1
    ts.sort()
    total = sum(by_name.values())
    print(f"## {tag}: median {ts[len(ts)//2]:.2f} ms  p95 {ts[int(len(ts)*0.95)]:.2f}  slow {sum(t > 2*ts[len(ts)//2] for t in ts)}/{n}  throttled periods +{throttled()-th0}  "
          f"CPU {total/n:.1f} ms per call over {len(t_after)} threads", flush=True)
    for name, ms in by_name.most_common(6):
        print(f"      {name:20s} {n_by_name[name]:4d} threads  {ms/n:8.1f} ms CPU per call", flush=True)


batch("as imported")
torch.set_num_threads(8)
batch("torch.set_num_threads(8)")
torch.set_num_threads(1)
batch("torch.set_num_threads(1)")
