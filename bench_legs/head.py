"""The embedding head against its HBM roof."""
from __future__ import annotations

import json
import os
import sys
import time

import torch

from .common import HBM_PEAK_GBS, MFMA_PEAK_TFLOPS, ROOT, make_queries, make_query_list, make_ragged_shard, make_shard, parse_regime, regime_numbers  # noqa: F401

def embed_head_numbers(amd, dev):
    """SURVEY 8(f) N1, the step before the path: hidden states of 1000 ColPali pages (1030 x 2048 bf16, 4.2 GB) ->
    projection + L2 norm + mask, written as the scorer's corpus rows.  HBM-bound (128 FLOP per streamed byte)."""
    out = _embed_head_shape(amd, dev, 1000, 1030, 2048)                     # BASELINE config 2: 1k ColPali pages (PaliGemma-3B, hidden 2048)
    out["colqwen2_1000x779x1536"] = _embed_head_shape(amd, dev, 1000, 779, 1536)   # config 3: ColQwen2 (Qwen2-VL-2B, hidden 1536)
    return out


def _embed_head_shape(amd, dev, B, S, H):
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
        # the same call 20 times back to back (no gap between launches: what an indexing loop over batches sees)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            fn()
        b.record()
        torch.cuda.synchronize()
        out[name]["ms_back_to_back"] = a.elapsed_time(b) / 20
        out[name]["hbm_gbs_back_to_back"] = byts / out[name]["ms_back_to_back"] / 1e6
        out[name]["frac_of_8TBs_back_to_back"] = out[name]["hbm_gbs_back_to_back"] / HBM_PEAK_GBS   # `ms` above also holds the host's launch latency
    out["workload"] = f"{B} pages x {S} tokens x hidden {H} bf16 -> [rows, 128] unit rows (algorithmic bytes = hidden read + rows written)"
    del hidden
    return out
