#!/usr/bin/env python
"""bench.py -- MaxSim (query, doc) pairs scored per second on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` (N>1 is launched through
torch.distributed.run, one rank per GPU, RCCL).  Rank 0 prints ONE JSON line.

Workload (BASELINE.json: "MaxSim (query,doc) pairs scored/sec; achieved HBM GB/s vs peak",
synthetic 32-token-query x 1024-patch-doc x d=128): every rank holds a resident shard of a
pre-embedded corpus (packed bf16 blob in HBM, generated on the device); one step scores a query
batch against the whole shard with the fused gfx950 MaxSim kernel, selects the per-shard top-k
and (N>1) merges the shards' top-k with one RCCL all-gather.  Weak scaling: the shard per GPU is
fixed, the corpus grows with N.

The JSON line carries
  roofline     -- of the dominant kernel (the fused MaxSim kernel), from HIP events recorded on the
                  launch stream inside the timed region; algorithmic bytes = docs streamed once
                  per query block (262144/Bq + 4 B per pair)
  cpu_baseline -- the reference's CPU scorer (oracle/torch_port.py restates
                  processing_utils.py:163-186 with the same torch calls) timed on this box's
                  host cores on a bounded sample of the same workload, rank 0, N=1 only
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md): 8.0 TB/s; 6.29 TB/s is the measured copy ceiling
MFMA_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA peak


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--docs", type=int, default=125000,
                    help="documents per GPU shard (1024 patches x 128 bf16 = 256 KiB each); 125000 = the 1M-doc corpus of "
                         "BASELINE config 4 over 8 GPUs")
    ap.add_argument("--doc-len", type=int, default=1024)
    ap.add_argument("--nq", type=int, default=4, help="queries per step (32 tokens each); 4 = BASELINE config 1's query batch")
    ap.add_argument("--regimes", type=str, default="1,32,1000",
                    help="other query-batch sizes measured after the headline and reported under 'regimes' ('' = none)")
    ap.add_argument("--q-len", type=int, default=32)
    ap.add_argument("--topk", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    return ap.parse_args()


def make_shard(n_docs, doc_len, device, seed):
    """Unit-norm bf16 rows, generated on the device in chunks (no host copy of the corpus exists)."""
    from colpali_amd.corpus import PackedCorpus

    g = torch.Generator(device=device).manual_seed(seed)
    blob = torch.empty((n_docs * doc_len, 128), dtype=torch.bfloat16, device=device)
    chunk = 512
    for d0 in range(0, n_docs, chunk):
        n = min(chunk, n_docs - d0)
        x = torch.randn((n * doc_len, 128), generator=g, device=device, dtype=torch.float32)
        blob[d0 * doc_len : (d0 + n) * doc_len] = torch.nn.functional.normalize(x, dim=-1).to(torch.bfloat16)
    lengths = torch.full((n_docs,), doc_len, dtype=torch.int64)
    offsets = (torch.arange(n_docs + 1, dtype=torch.int64) * doc_len).to(torch.int32).to(device)
    return PackedCorpus(blob=blob, offsets=offsets, clamp0=None, lengths=lengths)


def make_queries(n_q, q_len, device, seed):
    g = torch.Generator().manual_seed(seed)
    q = torch.nn.functional.normalize(torch.randn(n_q, q_len, 128, generator=g), dim=-1).to(torch.bfloat16)
    return q.to(device)


def parity_sample(q, corpus, scores, n_sample=48):
    """Re-score a random sample of the shard's documents with the CPU oracle."""
    from oracle import maxsim_oracle as mo

    n = len(corpus)
    idx = torch.randperm(n, generator=torch.Generator().manual_seed(7))[:n_sample].tolist()
    L = int(corpus.lengths[0])
    docs = [corpus.blob[i * L : (i + 1) * L].float().cpu().numpy() for i in idx]
    want = mo.score_multi_vector([x.float().cpu().numpy() for x in q], docs, batch_size=10**9, mode="f32")
    got = scores[:, idx].float().cpu().numpy()
    import numpy as np

    return float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1.0)))


def cpu_baseline(q_len, doc_len):
    """Reference CPU scorer (torch port) on a bounded sample: 32 queries x 512 docs, bf16 and fp32."""
    from oracle import torch_port

    g = torch.Generator().manual_seed(11)
    n_q, n_d = 32, 512
    qs = [torch.nn.functional.normalize(torch.randn(q_len, 128, generator=g), dim=-1).to(torch.bfloat16) for _ in range(n_q)]
    ps = [torch.nn.functional.normalize(torch.randn(doc_len, 128, generator=g), dim=-1).to(torch.bfloat16) for _ in range(n_d)]
    best = {}
    for name, cast in (("bf16", lambda t: t), ("fp32", lambda t: t.float())):
        a, b = [cast(t) for t in qs], [cast(t) for t in ps]
        torch_port.score_multi_vector_cpu(a[:4], b[:16])
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            torch_port.score_multi_vector_cpu(a, b)
            ts.append(time.perf_counter() - t0)
        best[name] = n_q * n_d / min(ts)
    kind = max(best, key=best.get)
    return {
        "value": best[kind], "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": f"{n_q} queries x {n_d} docs ({q_len}x128 vs {doc_len}x128), reference blocking batch_size=128, "
                  f"best of 3, torch CPU einsum/max/sum; bf16 inputs {best['bf16']:.0f} pairs/s, fp32 inputs {best['fp32']:.0f} pairs/s",
        "host_cpus": os.cpu_count(),
    }


def reference_scorer(qs, ps, device):
    """The reference's blocked scorer (oracle/torch_port.py restates processing_utils.py:132-187 with its own torch calls) on
    `device`: the baseline legs of this file and of tools/ab_dropin.py go through here, nothing else does."""
    from oracle import torch_port

    return torch_port.score_multi_vector_cpu(qs, ps, device=device)


def torch_gpu_reference(q_len, doc_len):
    """What the unmodified reference does on this same GPU (its torch einsum/max/sum with host-side padding and
    H2D per block, processing_utils.py:170-180): informational, not the optimisation target."""
    from oracle import torch_port

    g = torch.Generator().manual_seed(12)
    n_q, n_d = 128, 1024
    qs = [torch.nn.functional.normalize(torch.randn(q_len, 128, generator=g), dim=-1).to(torch.bfloat16) for _ in range(n_q)]
    ps = [torch.nn.functional.normalize(torch.randn(doc_len, 128, generator=g), dim=-1).to(torch.bfloat16) for _ in range(n_d)]
    torch_port.score_multi_vector_cpu(qs[:8], ps[:128], device="cuda:0")
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        torch_port.score_multi_vector_cpu(qs, ps, device="cuda:0")
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return {"value": n_q * n_d / min(ts), "unit": "pairs/s",
            "sample": f"{n_q} queries x {n_d} docs from host lists through the reference's blocked einsum on cuda:0 "
                      f"(includes its per-block pad_sequence + H2D), best of 3"}


def dropin_numbers(amd):
    """BASELINE configs 2/3 geometry through the drop-in entry point itself: 100 queries x 1000 pages handed over as HOST
    lists (what README.md:121-126 leaves the user with), end to end including packing, PCIe upload and the D2H of the
    result -- never the headline `value`, which is measured with the corpus resident."""
    from oracle import torch_port

    g = torch.Generator().manual_seed(21)

    def unit(n):
        return torch.nn.functional.normalize(torch.randn(n, 128, generator=g), dim=-1).to(torch.bfloat16)

    out = {}
    for name, lens in (("config2_colpali_1000x1030", [1030] * 1000),
                       ("config3_colqwen2_1000x267-779", torch.randint(267, 780, (1000,), generator=g).tolist())):
        qs, ps = [unit(32) for _ in range(100)], [unit(n) for n in lens]

        def timed(fn, reps):
            fn()
            ts = []
            for _ in range(reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            return sorted(ts)[len(ts) // 2]

        ours = timed(lambda: amd.score_multi_vector(qs, ps, device="cuda:0"), 5)
        ref = timed(lambda: torch_port.score_multi_vector_cpu(qs, ps, device="cuda:0"), 3)
        out[name] = {"pairs": 100 * len(ps), "ms": ours * 1e3, "pairs_per_s": 100 * len(ps) / ours,
                     "reference_on_this_gpu_ms": ref * 1e3, "speedup_vs_reference_on_this_gpu": ref / ours}
    return out


def embed_head_numbers(amd, dev):
    """SURVEY 8(f) N1, the step before the path: hidden states of 500 ColPali pages (1030 x 2048 bf16, 2.1 GB) ->
    projection + L2 norm + mask, written as the scorer's corpus rows.  HBM-bound (128 FLOP per streamed byte)."""
    B, S, H = 500, 1030, 2048
    g = torch.Generator(device=dev).manual_seed(3)
    hidden = torch.randn((B, S, H), generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    weight = (torch.randn((128, H), generator=g, device=dev) / H**0.5).to(torch.bfloat16)
    bias = (torch.randn((128,), generator=g, device=dev) * 0.1).to(torch.bfloat16)
    mask = torch.ones((B, S), dtype=torch.long, device=dev)
    mask[:, S - 6:] = 0

    def ref():
        proj = torch.nn.functional.linear(hidden, weight, bias)
        proj = proj / proj.norm(dim=-1, keepdim=True)
        return proj * mask.unsqueeze(-1)

    out = {}
    for name, fn in (("fused_head", lambda: amd.embedding_head(hidden, weight, bias, mask)), ("reference_lines_on_this_gpu", ref)):
        for _ in range(2):
            fn()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
        for a, b in evs:
            a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in evs)[3]
        byts = B * S * H * 2 + B * S * 256
        out[name] = {"ms": ms, "rows_per_s": B * S / ms * 1e3, "hbm_gbs": byts / ms / 1e6, "frac_of_8TBs": byts / ms / 1e6 / HBM_PEAK_GBS}
    out["workload"] = f"{B} pages x {S} tokens x hidden {H} bf16 -> [rows, 128] unit rows (algorithmic bytes = hidden read + rows written)"
    del hidden
    return out


def run_regime(amd, q, corpus, steps, warmup, topk, world, rank, dist):
    """Time `steps` full steps; returns (seconds for the K steps [max over ranks], kernel ms/launch list)."""
    dev = q.device
    scores = torch.empty((q.shape[0], len(corpus)), dtype=torch.float32, device=dev)

    def step(ev=None):
        if ev is not None:
            ev[0].record()
        amd.maxsim_scores(q, corpus, out=scores)
        if ev is not None:
            ev[1].record()
        return amd.shard_topk(scores, topk, corpus.id_base, world, dist) if hasattr(amd, "shard_topk") else None

    for _ in range(warmup):
        step()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(evs[i])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kern_ms = [a.elapsed_time(b) for a, b in evs]
    return dt, kern_ms, scores


def pmc_traffic(n_q, n_docs, doc_len):
    """HBM bytes per launch measured with rocprofv3 PMC counters for this exact workload (committed under
    profiles/ by tools/summarize_profile.py; FETCH_SIZE doubled per MI355X_MICROARCH.md, HBM section), else None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    try:
        table = json.load(open(path))
    except Exception:
        return None
    return table.get(f"nq{n_q}_docs{n_docs}_len{doc_len}")


def regime_numbers(n_q, q_len, n_docs, doc_len, kern_ms_avg):
    pairs = n_q * n_docs
    alg_bytes = n_docs * doc_len * 256 + n_q * q_len * 256 + pairs * 4   # docs streamed once per launch
    flops = 2.0 * n_q * q_len * n_docs * doc_len * 128
    sec = kern_ms_avg * 1e-3
    gbs, tf = alg_bytes / sec / 1e9, flops / sec / 1e12
    hbm_bound_s, mfma_bound_s = alg_bytes / (HBM_PEAK_GBS * 1e9), flops / (MFMA_PEAK_TFLOPS * 1e12)
    if hbm_bound_s >= mfma_bound_s:
        roof = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}
    else:
        roof = {"bound": "mfma", "achieved": tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_PEAK_TFLOPS}
    roof.update({"traffic": pmc_traffic(n_q, n_docs, doc_len), "kernel": "maxsim fused forward", "kernel_ms": kern_ms_avg,
                 "algorithmic_bytes_per_launch": alg_bytes, "flops_per_launch": flops,
                 "hbm_gbs": gbs, "mfma_tflops": tf})
    return roof


def stream_ceiling(amd, corpus):
    """The machine's own ceiling for K1s' document stream: the same LDS-DMA loads of the same resident shard with no MFMA, no
    max/sum and no output (msim_probe_stream, include/maxsim.h).  GB/s of the shard bytes; HIP events on the launch stream."""
    L = amd._lib.lib()
    rows = int(corpus.blob.shape[0]) // 256 * 256
    sink = torch.zeros(4, dtype=torch.float32, device=corpus.blob.device)
    st = torch.cuda.current_stream()
    ms = []
    for i in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        rc = L.msim_probe_stream(0, corpus.blob.data_ptr(), rows, 128, sink.data_ptr(), st.cuda_stream)
        b.record(st)
        torch.cuda.synchronize()
        if rc != 0:
            raise RuntimeError(f"msim_probe_stream failed: {L.msim_last_error().decode()}")
        if i >= 2:
            ms.append(a.elapsed_time(b))
    t = sorted(ms)[len(ms) // 2]
    return {"gbs": rows * 256 / t / 1e6, "ms": t, "what": "msim_probe_stream(MSIM_PROBE_ROWS256B) over the same shard: "
            "K1s' loads without its arithmetic"}


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # BENCH_SHARE_GPU=1 + BENCH_DIST_BACKEND=gloo: plumbing test of the N>1 path on a 1-GPU box (not a measurement)
    share_gpu = os.environ.get("BENCH_SHARE_GPU") == "1"
    dev_index = local_rank % torch.cuda.device_count() if share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist  # "nccl" is RCCL on ROCm
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    import colpali_amd as amd

    amd._lib.lib()  # fail loudly if the HIP library is missing
    corpus = make_shard(args.docs, args.doc_len, dev, seed=1234 + rank)
    corpus.id_base = rank * args.docs
    q = make_queries(args.nq, args.q_len, dev, seed=99)
    torch.cuda.synchronize()

    dt, kern_ms, scores = run_regime(amd, q, corpus, args.steps, args.warmup, args.topk, world, rank, dist)
    kern_avg = sum(kern_ms) / len(kern_ms)
    pairs_per_step = args.nq * args.docs * world
    out = {
        "metric": "MaxSim (query,doc) pairs scored/sec",
        "value": pairs_per_step * args.steps / dt,
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic",
        "config": {
            "workload": f"{args.nq} query x {args.q_len} tokens vs resident pre-embedded shard of {args.docs} docs x "
                        f"{args.doc_len} patches x d=128 bf16 per GPU ({args.docs * args.doc_len * 256 / 2**30:.1f} GiB/GPU), "
                        f"fused MaxSim + per-shard top-{args.topk}" + (" + RCCL all-gather merge" if world > 1 else ""),
            "docs_per_gpu": args.docs, "doc_len": args.doc_len, "n_queries": args.nq, "q_len": args.q_len,
            "top_k": args.topk, "parallelism": f"corpus-sharded x{world}",
        },
        "roofline": regime_numbers(args.nq, args.q_len, args.docs, args.doc_len, kern_avg),
    }
    if corpus.blob.shape[1] == 128:
        ceil_ = stream_ceiling(amd, corpus)
        out["roofline"]["stream_ceiling_gbs"] = ceil_["gbs"]
        out["roofline"]["stream_ceiling_what"] = ceil_["what"]
        if out["roofline"]["bound"] == "hbm":
            out["roofline"]["frac_of_stream_ceiling"] = out["roofline"]["achieved"] / ceil_["gbs"]
    if rank == 0 and not args.no_parity:
        out["parity_max_rel_err_vs_oracle_sample"] = parity_sample(q, corpus, scores)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.q_len, args.doc_len)
        out["reference_on_this_gpu"] = torch_gpu_reference(args.q_len, args.doc_len)
        out["embed_head"] = embed_head_numbers(amd, dev)
        out["dropin_from_host_lists"] = dropin_numbers(amd)

    # other regimes of the same step on the same resident shard (every rank takes part: collectives inside)
    regimes = []
    for nq in [int(x) for x in args.regimes.split(",") if x]:
        qq = make_queries(nq, args.q_len, dev, seed=5 + nq)
        steps = max(3, min(args.steps, 2000 // max(nq, 1)))
        d, km, _ = run_regime(amd, qq, corpus, steps, 2, args.topk, world, rank, dist)
        r = regime_numbers(nq, args.q_len, args.docs, args.doc_len, sum(km) / len(km))
        regimes.append({"n_queries": nq, "steps": steps, "pairs_per_s": nq * args.docs * world * steps / d,
                        "ms_per_step": d / steps * 1e3, "kernel_ms": r["kernel_ms"], "bound": r["bound"],
                        "frac": r["frac"], "hbm_gbs_per_gpu": r["hbm_gbs"], "mfma_tflops_per_gpu": r["mfma_tflops"]})
        del qq
    out["regimes"] = regimes

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
