"""The ONE stdout line of bench.py, kept small enough for any reader of it.

Round 5's line had grown to 29 KB (every leg's full report inline) and the driver stopped parsing it: that round has no
driver-measured headline.  The contract since round 6: stdout carries `compact_line(detail)` -- the contract keys, `roofline`,
`cpu_baseline`, `topk_parity` and a flat dict of scalar summaries, at most MAX_LINE_BYTES of strict JSON (no NaN / Infinity
tokens) -- and the full report of every leg goes to `bench_detail.json` next to bench.py (and to gpurun_out/ when that
directory exists).  tests/test_bench_line.py holds the size and strictness bound on a stub of the full report.
"""
from __future__ import annotations

import json
import math

MAX_LINE_BYTES = 8192          # hard cap
TARGET_LINE_BYTES = 4096       # what the line is built for

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data")
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "algorithmic_bytes_per_launch",
                 "stream_ceiling_gbs", "frac_of_stream_ceiling")
CPU_BASELINE_KEYS = ("value", "unit", "cores", "kind", "sample")
CONFIG_KEYS = ("workload", "docs_per_gpu", "doc_len", "n_queries", "q_len", "top_k", "parallelism")
MULTI_GPU_KEYS = ("rccl_ranks", "dist_backend", "per_rank_kernel_ms", "shared_gpu_plumbing_run", "launched_by")


def strict(obj):
    """`obj` with every non-finite float replaced by None and every number rounded to 6 significant digits (recursively):
    what json.dumps(..., allow_nan=False) accepts, and short."""
    if isinstance(obj, bool) or obj is None or isinstance(obj, (int, str)):
        return obj
    if isinstance(obj, float):
        if not math.isfinite(obj):
            return None
        return float(f"{obj:.6g}")
    if isinstance(obj, dict):
        return {str(k): strict(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [strict(v) for v in obj]
    if hasattr(obj, "item"):               # numpy / torch scalars
        return strict(obj.item())
    return str(obj)


def _dig(d, *path, default=None):
    for p in path:
        if not isinstance(d, dict) or p not in d:
            return default
        d = d[p]
    return d


def _short(text, n):
    text = str(text)
    return text if len(text) <= n else text[: n - 3] + "..."


def summaries(detail: dict) -> dict:
    """The scalar summaries of the legs (name -> number / bool), in the order they are dropped LAST-first when the line is too long."""
    s = {}

    def put(name, value):
        if value is not None:
            s[name] = value

    loss = detail.get("loss_step_config5") or {}
    for cls, tag in (("ColbertPairwiseCELoss", "pairwise"), ("ColbertLoss", "infonce"), ("ColbertSigmoidLoss", "sigmoid")):
        both = _dig(loss, "Lq32", cls, "ours", "both_directions") or {}
        put(f"loss_{tag}_both_dir_device_ms", both.get("one_hipgraph_device_ms"))
        put(f"loss_{tag}_both_dir_frac", both.get("frac_of_roof"))
        fwd = _dig(loss, "Lq32", cls, "ours", "forward_direction") or {}
        put(f"loss_{tag}_fwd_dir_device_ms", fwd.get("one_hipgraph_device_ms"))
        put(f"loss_{tag}_speedup_vs_reference_eager", _dig(loss, "Lq32", cls, "speedup_vs_reference_both_directions_eager"))
    put("loss_pairwise_dQ_err", _dig(loss, "Lq32", "ColbertPairwiseCELoss", "parity_vs_float64_oracle", "dQ_max_err_over_max_abs"))
    put("loss_infonce_dD_err", _dig(loss, "Lq32", "ColbertLoss", "parity_vs_float64_oracle", "dD_max_err_over_max_abs"))
    for cfg, tag in (("config2_colpali_1000x1030", "dropin_config2"), ("config3_colqwen2_1000x267-779", "dropin_config3")):
        leg = _dig(detail, "dropin_from_host_lists", cfg) or {}
        put(f"{tag}_ms", leg.get("ms"))
        put(f"{tag}_frac_of_h2d", _dig(leg, "breakdown", "frac_of_h2d_roof"))
        put(f"{tag}_max_rel_err_vs_reference_fp32", leg.get("max_rel_err_vs_reference_fp32_on_this_gpu"))
    put("config1_cpu_ms", _dig(detail, "dropin_from_host_lists", "config1_4x16", "ours_on_this_host_cpu_ms"))
    put("config1_reference_cpu_ms", _dig(detail, "dropin_from_host_lists", "config1_4x16", "reference_on_this_host_cpu_ms"))
    for r in detail.get("regimes") or []:
        if r.get("q_len") == "32" and r.get("n_queries") in (8, 10, 16, 32, 1000):
            put(f"nq{r['n_queries']}_frac", r.get("frac"))
            put(f"nq{r['n_queries']}_bound", r.get("bound"))
        if r.get("n_queries") == 1000 and r.get("q_len") not in ("32", "20", "48"):
            put(f"nq1000_len{r['q_len']}_frac".replace("{", "").replace("}", "").replace("..", "-"), r.get("frac"))
    put("reference_on_this_gpu_pairs_per_s", _dig(detail, "reference_on_this_gpu", "value"))
    put("embed_head_frac_of_hbm", _dig(detail, "embed_head", "fused_head", "frac_of_8TBs_back_to_back"))
    put("short_docs_64_rows_1000q_frac", _dig(detail, "resident_short_documents", "64_rows", "1000_queries_x_32", "frac"))
    put("short_docs_64_rows_ragged_frac", _dig(detail, "resident_short_documents", "64_rows", "1000_queries_ragged_12_48", "frac"))
    put("pooled_343_rows_1000q_frac", _dig(detail, "resident_short_documents", "pooled_343_rows", "1000_queries_x_32", "frac"))
    put("width320_1000q_frac", _dig(detail, "resident_width_320", "1000_queries_x_32", "frac"))
    put("width320_4q_frac", _dig(detail, "resident_width_320", "4_queries_x_32", "frac"))
    put("colqwen2_pages_4q_frac", _dig(detail, "resident_colqwen2_page_geometry", "4_queries_x_32", "frac"))
    put("forced_collective_equal", _dig(detail, "forced_collective_1rank", "ids_and_scores_equal_to_non_collective"))
    put("vlm_colpali_pages_per_s", _dig(detail, "embed_and_score_1k_pages_vlm_in_the_loop", "pages_per_s"))
    put("vlm_colqwen2_pages_per_s", _dig(detail, "embed_and_score_1k_pages_vlm_in_the_loop_colqwen2", "pages_per_s"))
    put("frac_on_zeros", _dig(detail, "roofline", "power", "frac_on_zeros"))
    put("socket_power_w_avg", _dig(detail, "roofline", "power", "socket_power_w_avg"))
    return s


def compact_line(detail: dict, detail_path: str | None = None) -> dict:
    """The dict bench.py prints: contract keys + config + roofline + cpu_baseline + topk_parity + scalar summaries."""
    line = {k: detail.get(k) for k in CONTRACT_KEYS if k in detail}
    cfg = detail.get("config") or {}
    line["config"] = {k: (_short(cfg[k], 200) if isinstance(cfg[k], str) else cfg[k]) for k in CONFIG_KEYS if k in cfg}
    roof = detail.get("roofline") or {}
    line["roofline"] = {k: roof.get(k) for k in ROOFLINE_KEYS if k in roof}
    cb = detail.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: (_short(cb[k], 260) if isinstance(cb.get(k), str) else cb.get(k)) for k in CPU_BASELINE_KEYS if k in cb}
    par = detail.get("topk_parity")
    if par:
        p = {k: par.get(k) for k in ("k", "ids_equal", "ids_exact_equal", "max_rel_err", "merge_equals_oracle_merge_on_all_ranks",
                                     "merged_topk_equals_unsharded_topk", "ranks_checked") if k in par}
        for k in ("ids_equal", "ids_exact_equal"):
            if _dig(par, "k100", k) is not None:
                p[f"k100_{k}"] = par["k100"][k]
        vr = par.get("vs_reference_fp32") or {}
        for k in ("k10_ids_exact_equal", "k100_ids_exact_equal", "k100_differing_positions", "k100_max_swap_gap_ulps",
                  "reference_cpu_vs_reference_gpu_k100_ids_equal"):
            if k in vr:
                p[f"vs_reference_fp32_{k}"] = vr[k]
        line["topk_parity"] = p
    for k in MULTI_GPU_KEYS:
        if k in detail:
            line[k] = detail[k]
    line["summary"] = summaries(detail)
    if detail_path:
        line["detail"] = detail_path
    return strict(line)


def dumps_line(detail: dict, detail_path: str | None = None) -> str:
    """Strict JSON of the compact line, at most MAX_LINE_BYTES: summaries are dropped from the end until it fits (the contract keys,
    roofline and cpu_baseline never are)."""
    line = compact_line(detail, detail_path)
    text = json.dumps(line, allow_nan=False, separators=(", ", ": "))
    while len(text.encode()) > TARGET_LINE_BYTES and line.get("summary"):
        line["summary"].pop(next(reversed(line["summary"])))
        text = json.dumps(line, allow_nan=False, separators=(", ", ": "))
    if len(text.encode()) > MAX_LINE_BYTES:
        raise ValueError(f"bench line is {len(text.encode())} bytes even without summaries (cap {MAX_LINE_BYTES})")
    return text


def dumps_detail(detail: dict) -> str:
    """The full report as strict JSON (non-finite numbers -> null)."""
    def clean(o):
        if isinstance(o, float):
            return o if math.isfinite(o) else None
        if isinstance(o, dict):
            return {str(k): clean(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [clean(v) for v in o]
        if isinstance(o, (bool, int, str)) or o is None:
            return o
        if hasattr(o, "item"):
            return clean(o.item())
        return str(o)

    return json.dumps(clean(detail), allow_nan=False, indent=1)
