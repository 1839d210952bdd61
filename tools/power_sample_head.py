#!/usr/bin/env python
"""Socket power and shader clock (rocm-smi) while K3 (colpali_amd.embedding_head) runs back to back, per hidden size, next to its
bare access pattern (msim_probe_stream, 128-byte pieces): is the embedding head power-capped like the scorer's ridge?"""
import json, os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import colpali_amd as amd

dev = torch.device("cuda:0")
SMI = "/opt/rocm/bin/rocm-smi"
L = amd._lib.lib()
from tools import probe
P = probe.lib()          # tools/probe/libmaxsim_probe.so (include/maxsim_probe.h)


def sample():
    try:
        card = json.loads(subprocess.run([SMI, "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout)
        card = card[sorted(card.keys())[0]]
        pw = next((float(v) for k, v in card.items() if "ower" in k and re.match(r"^[0-9.]+$", str(v))), None)
        m = re.search(r"(\d+)\s*Mhz", next((str(v) for k, v in card.items() if k.lower().startswith("sclk")), ""), re.I)
        return pw, int(m.group(1)) if m else None
    except Exception:
        return None, None


def run(name, fn, byts, secs=2.5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    got, stop = [], threading.Event()

    def sampler():
        time.sleep(0.4)
        while not stop.is_set():
            got.append(sample())

    th = threading.Thread(target=sampler)
    th.start()
    n, t0 = 0, time.perf_counter()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    while time.perf_counter() - t0 < secs:
        for _ in range(20):
            fn()
            n += 1
        torch.cuda.synchronize()
    b.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    ms = a.elapsed_time(b) / n
    pw = [p for p, _ in got if p is not None]
    ck = [c for _, c in got if c is not None]
    print(f"{name:58s} {ms:7.3f} ms {byts / ms / 1e6:6.0f} GB/s   power W avg {sum(pw) / max(len(pw), 1):7.1f} max {max(pw) if pw else 0:7.1f}   "
          f"sclk MHz avg {sum(ck) / max(len(ck), 1):6.0f}", flush=True)


g = torch.Generator(device=dev).manual_seed(0)
sink = torch.zeros(4, dtype=torch.float32, device=dev)
for B, S, H in ((1000, 1030, 2048), (1000, 779, 1536), (256, 1030, 3584)):
    hidden = torch.randn((B, S, H), generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    weight = (torch.randn((128, H), generator=g, device=dev) / H**0.5).to(torch.bfloat16)
    bias = (torch.randn((128,), generator=g, device=dev) * 0.1).to(torch.bfloat16)
    mask = torch.ones((B, S), dtype=torch.long, device=dev)
    rows = B * S // 256 * 256
    st = torch.cuda.current_stream().cuda_stream
    run(f"K3 fused head, hidden {H} ({B} x {S} rows)", lambda: amd.embedding_head(hidden, weight, bias, mask), B * S * H * 2 + B * S * 256)
    run(f"  its bare access pattern (probe, 128-B pieces), hidden {H}", lambda: P.msim_probe_stream(1, hidden.data_ptr(), rows, H, sink.data_ptr(), st),
        rows * H * 2)
    zh = torch.zeros_like(hidden)
    run(f"  K3 on ZERO hidden states, hidden {H}", lambda: amd.embedding_head(zh, weight, bias, mask), B * S * H * 2 + B * S * 256)
    del hidden, zh
