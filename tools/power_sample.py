#!/usr/bin/env python
"""What the chip's power management does in every regime of the scorer: msim_fwd is launched back to back for a few seconds per
query-batch size on a resident shard (random unit rows, and the same shard zero-filled) while a thread samples
`rocm-smi --showpower --showclocks` -- average socket power and the shader clock next to the achieved HBM GB/s and MFMA TFLOP/s.
Evidence behind DESIGN.md's statement that everything from 4 queries up is power-bound (the MI355X clocks to its power budget:
MI355X_MICROARCH.md, DVFS)."""
import json, os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
import bench, colpali_amd as amd

docs = int(os.environ.get("AB_DOCS", "65536"))
secs = float(os.environ.get("PS_SECS", "2.5"))
sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,4,8,10,16,32,256").split(",")]
dev = torch.device("cuda:0")
SMI = "/opt/rocm/bin/rocm-smi"


def sample():
    try:
        out = subprocess.run([SMI, "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
        j = json.loads(out)
        card = j[sorted(j.keys())[0]]
        power = next((float(v) for k, v in card.items() if "ower" in k and re.match(r"^[0-9.]+$", str(v))), None)
        sclk = next((str(v) for k, v in card.items() if k.lower().startswith("sclk")), None)
        m = re.search(r"(\d+)\s*Mhz", sclk or "", re.I)
        return power, (int(m.group(1)) if m else None), None
    except Exception as e:  # keep whatever came back
        return None, None, f"{type(e).__name__}: {e}"


first = subprocess.run([SMI, "--showpower", "--showclocks", "--json"], capture_output=True, text=True).stdout
print("# raw rocm-smi sample (idle):", first.strip()[:600], flush=True)
for fill in ("random unit rows", "zeros"):
    corpus = bench.make_shard(docs, 1024, dev, 1234)
    if fill == "zeros":
        corpus.blob.zero_()
    for nq in sizes:
        q = bench.make_queries(nq, 32, dev, 3)
        out = torch.empty((nq, docs), dtype=torch.float32, device=dev)
        for _ in range(3):
            amd.maxsim_scores(q, corpus, out=out)
        torch.cuda.synchronize()
        stop = threading.Event()
        got = []

        def sampler():
            time.sleep(0.4)                      # let the clocks settle under load
            while not stop.is_set():
                got.append(sample())

        th = threading.Thread(target=sampler)
        th.start()
        t0 = time.perf_counter()
        n = 0
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        while time.perf_counter() - t0 < secs:
            for _ in range(max(1, 64 // nq)):
                amd.maxsim_scores(q, corpus, out=out)
                n += 1
            torch.cuda.synchronize()
        b.record()
        torch.cuda.synchronize()
        stop.set()
        th.join()
        ms = a.elapsed_time(b) / n
        r = bench.regime_numbers(nq, 32, docs, 1024, ms)
        pw = [p for p, _, _ in got if p is not None]
        ck = [c for _, c, _ in got if c is not None]
        err = next((e for _, _, e in got if e), "")
        print(f"{fill:17s} nq={nq:4d} {ms:8.3f} ms {r['hbm_gbs']:6.0f} GB/s {r['mfma_tflops']:6.0f} TF  {r['bound']} {r['frac']:.3f}  "
              f"power W avg {sum(pw) / len(pw) if pw else float('nan'):7.1f} max {max(pw) if pw else float('nan'):7.1f}  "
              f"sclk MHz avg {sum(ck) / len(ck) if ck else float('nan'):6.0f} min {min(ck) if ck else float('nan'):6.0f}  ({len(got)} samples) {err}", flush=True)
    del corpus
