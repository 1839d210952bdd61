"""The reference timed beside the product: its CPU scorer on this box's host cores (cpu_baseline) and its torch calls on this GPU."""
from __future__ import annotations

import json
import os
import sys
import time

import torch

from .common import HBM_PEAK_GBS, MFMA_PEAK_TFLOPS, ROOT, make_queries, make_query_list, make_ragged_shard, make_shard, parse_regime, regime_numbers  # noqa: F401

def cpu_baseline(q_len, doc_len):
    """The reference's CPU scorer on the slice SURVEY 8(d) names: 128 queries x 1024 docs (131 072 pairs, one 128 x 128 block row of the
    reference's blocking x 8), bf16 and fp32 inputs, best of 2.  kind = "reference": the VERBATIM
    colpali_engine/utils/processing_utils.py (BaseVisualRetrieverProcessor.score_multi_vector, :132-187) -- the live checkout where it
    exists, else the byte-for-byte git-ignored copy oracle/fetch_reference_tests.py leaves under tests/_reference_pkg/ (it travels to the
    GPU box with the working tree); kind = "port": oracle/torch_port.py, the restatement with the same torch calls, only where neither
    file is present."""
    from oracle import refimport, torch_port

    try:
        proc, _, where = refimport.load_hot_path()
        scorer, kind = (lambda a, b: proc.score_multi_vector(a, b, batch_size=128, device="cpu")), "reference"
        what = f"the reference's own processing_utils.py ({where} copy), BaseVisualRetrieverProcessor.score_multi_vector(device='cpu')"
    except Exception:
        scorer, kind = torch_port.score_multi_vector_cpu, "port"
        what = "oracle/torch_port.py (restatement of processing_utils.py:163-186 with the same torch calls)"
    g = torch.Generator().manual_seed(11)
    n_q, n_d = 128, 1024
    qs = [torch.nn.functional.normalize(torch.randn(q_len, 128, generator=g), dim=-1).to(torch.bfloat16) for _ in range(n_q)]
    ps = [torch.nn.functional.normalize(torch.randn(doc_len, 128, generator=g), dim=-1).to(torch.bfloat16) for _ in range(n_d)]
    best = {}
    for name, cast in (("bf16", lambda t: t), ("fp32", lambda t: t.float())):
        a, b = [cast(t) for t in qs], [cast(t) for t in ps]
        scorer(a[:4], b[:16])
        ts = []
        for _ in range(2):
            t0 = time.perf_counter()
            scorer(a, b)
            ts.append(time.perf_counter() - t0)
        best[name] = n_q * n_d / min(ts)
    top = max(best, key=best.get)
    # the product's own host-core path (score_multi_vector(device="cpu") -> msim_fwd_host) on the same sample and cores: context
    import colpali_amd as amd

    amd.score_multi_vector(qs[:4], ps[:16], device="cpu")
    t0 = time.perf_counter()
    amd.score_multi_vector(qs, ps, device="cpu")
    host_path = n_q * n_d / (time.perf_counter() - t0)
    return {
        "value": best[top], "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": kind, "what": what,
        "colpali_amd_host_path_pairs_per_s": host_path,
        "sample": f"{n_q} queries x {n_d} docs ({q_len}x128 vs {doc_len}x128), reference blocking batch_size=128, "
                  f"best of 2, torch CPU einsum/max/sum; bf16 inputs {best['bf16']:.0f} pairs/s, fp32 inputs {best['fp32']:.0f} pairs/s",
        "host_cpus": os.cpu_count(), "torch_num_threads": torch.get_num_threads(),
        "cpus_the_container_grants": amd._lib.effective_cpus(),       # affinity and cgroup CPU quota (cpu.max): what `cores` can really use
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
