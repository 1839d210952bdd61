#!/usr/bin/env python
"""Energy A/B of two builds of the library on the SAME box, interleaved (round-4 review, weak 3): the round-3 library (commit 01dd5cb:
32-token tiles, per-tile butterflies) against today's (16-token flat units, per-token maxima through an LDS table) at the Lq = 32 batch
regimes the driver's line showed 3-5 % slower -- ms per launch, socket watts and shader clock sampled from rocm-smi while the launch
repeats, and JOULES PER PAIR.  Both libraries are loaded into one process through ctypes and driven through the one entry point whose
signature did not change, msim_fwd.
  usage (GPU box): python tools/ab_energy.py [sizes=20,32,40,1000] > gpurun_out/ab_energy.log"""
import ctypes, json, os, re, subprocess, sys, threading, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch
import bench

docs = int(os.environ.get("AB_DOCS", "65536"))
secs = float(os.environ.get("PS_SECS", "2.5"))
sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "20,32,40,1000").split(",")]
dev = torch.device("cuda:0")
SMI = "/opt/rocm/bin/rocm-smi"
vp, i32, i64, u32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_uint32


def load(path):
    L = ctypes.CDLL(path)
    L.msim_fwd.argtypes = [i32, vp, i32, i32, vp, vp, vp, i32, i32, vp, i64, u32, vp, vp]
    L.msim_fwd.restype = i32
    L.msim_fwd_workspace_bytes.argtypes = [i32, i32, i32, i32, i32]
    L.msim_fwd_workspace_bytes.restype = ctypes.c_size_t
    L.msim_abi_version.restype = i32
    return L


LIBS = [("round 3 (01dd5cb)", load(os.path.join(ROOT, "tools/_ab/libmaxsim_r3_01dd5cb.so"))),
        ("today", load(os.path.join(ROOT, "colpali_amd/csrc/libmaxsim_gfx950.so")))]


def sample():
    try:
        out = subprocess.run([SMI, "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
        card = json.loads(out)
        card = card[sorted(card.keys())[0]]
        power = next((float(v) for k, v in card.items() if "ower" in k and re.match(r"^[0-9.]+$", str(v))), None)
        sclk = next((str(v) for k, v in card.items() if k.lower().startswith("sclk")), None)
        m = re.search(r"(\d+)\s*Mhz", sclk or "", re.I)
        return power, (int(m.group(1)) if m else None)
    except Exception:
        return None, None


corpus = bench.make_shard(docs, 1024, dev, 1234)
stream = torch.cuda.current_stream().cuda_stream
print(f"# tools/ab_energy.py: {docs} docs x 1024 x 128 bf16 resident ({docs * 1024 * 256 / 2**30:.1f} GiB), Lq = 32, {secs} s of back-to-back launches per "
      f"(build, size), builds interleaved twice; ABI versions {[L.msim_abi_version() for _, L in LIBS]}", flush=True)
for nq in sizes:
    q = bench.make_queries(nq, 32, dev, 3)
    outs, wss = {}, {}
    for name, L in LIBS:
        outs[name] = torch.empty((nq, docs), dtype=torch.float32, device=dev)
        n = L.msim_fwd_workspace_bytes(0, nq, 32, docs, 128)
        wss[name] = torch.empty((max(n, 16),), dtype=torch.uint8, device=dev)

    def launch(name, L):
        rc = L.msim_fwd(0, q.data_ptr(), nq, 32, corpus.blob.data_ptr(), corpus.offsets.data_ptr(), None, docs, 128,
                        outs[name].data_ptr(), docs, 0, wss[name].data_ptr(), stream)
        assert rc == 0, (name, rc)

    for name, L in LIBS:
        for _ in range(2):
            launch(name, L)
    torch.cuda.synchronize()
    d = (outs[LIBS[0][0]] - outs[LIBS[1][0]]).abs().max().item()
    print(f"## {nq} queries x 32 tokens: max |score difference| between the builds {d:.3e}", flush=True)
    for rep in (1, 2):
        for name, L in LIBS:
            stop, got = threading.Event(), []

            def sampler():
                time.sleep(0.4)
                while not stop.is_set():
                    got.append(sample())

            th = threading.Thread(target=sampler)
            th.start()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0, n = time.perf_counter(), 0
            a.record()
            while time.perf_counter() - t0 < secs:
                for _ in range(max(1, 64 // nq)):
                    launch(name, L)
                    n += 1
                torch.cuda.synchronize()
            b.record()
            torch.cuda.synchronize()
            stop.set()
            th.join()
            ms = a.elapsed_time(b) / n
            pw = [p for p, _ in got if p]
            ck = [c for _, c in got if c]
            w = sum(pw) / len(pw) if pw else float("nan")
            mhz = sum(ck) / len(ck) if ck else float("nan")
            r = bench.regime_numbers(nq, 32, docs, 1024, ms)
            print(f"{name:18s} run {rep}: {ms:9.3f} ms/launch  {r['mfma_tflops']:7.0f} TFLOP/s  frac {r['frac']:.3f}   {w:7.1f} W  {mhz:6.0f} MHz   "
                  f"{w * ms * 1e-3 / (nq * docs) * 1e9:8.3f} nJ per pair   ({len(pw)} samples)", flush=True)
