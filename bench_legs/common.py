"""bench.py's shared pieces: the machine's peaks, synthetic shards and query batches, the roofline arithmetic of one launch."""
from __future__ import annotations

import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md): 8.0 TB/s; 6.29 TB/s is the measured copy ceiling
MFMA_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA peak


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


def make_ragged_shard(n_docs, lo, hi, device, seed):
    """BASELINE config 3's page geometry on the resident path: ColQwen2 pages of U{lo..hi} patch rows each (dynamic resolution), unit-norm
    bf16 rows generated on the device."""
    from colpali_amd.corpus import PackedCorpus

    gl = torch.Generator().manual_seed(seed)
    lengths = torch.randint(lo, hi + 1, (n_docs,), generator=gl)
    offsets = torch.zeros(n_docs + 1, dtype=torch.int64)
    torch.cumsum(lengths, 0, out=offsets[1:])
    rows = int(offsets[-1])
    g = torch.Generator(device=device).manual_seed(seed)
    blob = torch.empty((rows, 128), dtype=torch.bfloat16, device=device)
    step = 1 << 19
    for r0 in range(0, rows, step):
        n = min(step, rows - r0)
        x = torch.randn((n, 128), generator=g, device=device, dtype=torch.float32)
        blob[r0:r0 + n] = torch.nn.functional.normalize(x, dim=-1).to(torch.bfloat16)
    return PackedCorpus(blob=blob, offsets=offsets.to(torch.int32).to(device), clamp0=None, lengths=lengths.to(torch.int64))


def make_queries(n_q, q_len, device, seed):
    g = torch.Generator().manual_seed(seed)
    q = torch.nn.functional.normalize(torch.randn(n_q, q_len, 128, generator=g), dim=-1).to(torch.bfloat16)
    return q.to(device)


def parse_regime(spec, default_len):
    """'N' | 'NxL' | 'NxrA-B' -> (n_queries, [length of every query], label)."""
    if "x" not in spec:
        n = int(spec)
        return n, [default_len] * n, str(default_len)
    n, ln = spec.split("x", 1)
    n = int(n)
    if ln.startswith("r"):
        lo, hi = (int(v) for v in ln[1:].split("-"))
        g = torch.Generator().manual_seed(1000 + n + lo * 7 + hi)
        return n, torch.randint(lo, hi + 1, (n,), generator=g).tolist(), f"U{{{lo}..{hi}}}"
    return n, [int(ln)] * n, ln


def make_query_list(lens, seed):
    """Host list of [len_i, 128] unit-row bf16 queries -- the drop-in's own input form (ragged lengths are the normal case)."""
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.functional.normalize(torch.randn(n, 128, generator=g), dim=-1).to(torch.bfloat16) for n in lens]


def pmc_traffic(n_q, n_docs, doc_len, q_tokens=None):
    """HBM bytes per launch measured with rocprofv3 PMC counters for this exact workload (committed under
    profiles/ by tools/summarize_profile.py; FETCH_SIZE doubled per MI355X_MICROARCH.md, HBM section), else None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    try:
        table = json.load(open(path))
    except Exception:
        return None
    q_tokens = n_q * 32 if q_tokens is None else q_tokens
    hit = table.get(f"nq{n_q}_tok{q_tokens}_docs{n_docs}_len{doc_len}")
    if hit is None and q_tokens == n_q * 32:
        hit = table.get(f"nq{n_q}_docs{n_docs}_len{doc_len}")
    return hit


def regime_numbers(n_q, q_len, n_docs, doc_len, kern_ms_avg, q_tokens=None):
    """`q_tokens`: REAL query tokens in the batch (ragged batches); FLOP and bytes count real tokens only -- padding an
    implementation adds is never credited."""
    pairs = n_q * n_docs
    q_tokens = n_q * q_len if q_tokens is None else q_tokens
    alg_bytes = n_docs * doc_len * 256 + q_tokens * 256 + pairs * 4   # docs streamed once per launch
    flops = 2.0 * q_tokens * n_docs * doc_len * 128
    sec = kern_ms_avg * 1e-3
    gbs, tf = alg_bytes / sec / 1e9, flops / sec / 1e12
    hbm_bound_s, mfma_bound_s = alg_bytes / (HBM_PEAK_GBS * 1e9), flops / (MFMA_PEAK_TFLOPS * 1e12)
    if hbm_bound_s >= mfma_bound_s:
        roof = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}
    else:
        roof = {"bound": "mfma", "achieved": tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_PEAK_TFLOPS}
    roof.update({"traffic": pmc_traffic(n_q, n_docs, doc_len, q_tokens),
                 "traffic_source": "profiles/pmc_traffic.json (committed rocprofv3 --pmc pass of this workload; not re-measured in this run)",
                 "kernel": "maxsim fused forward", "kernel_ms": kern_ms_avg,
                 "algorithmic_bytes_per_launch": alg_bytes, "flops_per_launch": flops,
                 "hbm_gbs": gbs, "mfma_tflops": tf})
    return roof
