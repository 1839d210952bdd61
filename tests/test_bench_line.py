"""bench.py's stdout contract: ONE compact strict-JSON line (bench_legs/line.py).  Round 5's line was 29 KB with every leg's report
inline and the driver could not parse it -- these tests hold the size and strictness bound on the full report of that very run and on
a hostile stub (NaN / inf, very long strings, hundreds of regimes)."""
import json
import math
import os

import pytest

from bench_legs import line

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _parse_strict(text):
    def refuse(tok):
        raise AssertionError(f"non-strict JSON token {tok}")

    assert "\n" not in text
    return json.loads(text, parse_constant=refuse)


def _check(text):
    assert len(text.encode()) <= line.MAX_LINE_BYTES
    d = _parse_strict(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data", "config"):
        assert k in d, k
    assert "vs_baseline" in d and d["vs_baseline"] is None
    assert isinstance(d["roofline"]["frac"], float) and d["roofline"]["bound"] in ("hbm", "mfma")
    for k in ("achieved", "peak", "unit", "traffic"):
        assert k in d["roofline"], k
    assert "workload" in d["config"] and "model" not in d["config"]
    return d


def test_round5_full_report_becomes_a_small_strict_line():
    path = os.path.join(ROOT, "profiles", "r05_logs", "bench_line_final.json")
    detail = json.load(open(path))
    assert len(json.dumps(detail)) > 20000                       # the thing that broke the parse
    text = line.dumps_line(detail, "bench_detail.json")
    d = _check(text)
    assert len(text.encode()) <= line.TARGET_LINE_BYTES
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] in ("reference", "port") and d["cpu_baseline"]["cores"] >= 1
    assert d["value"] == pytest.approx(detail["value"], rel=1e-5)
    assert d["roofline"]["frac"] == pytest.approx(detail["roofline"]["frac"], rel=1e-5)
    assert d["topk_parity"]["ids_equal"] is True
    assert d["detail"] == "bench_detail.json"
    assert all(not isinstance(v, (dict, list)) for v in d["summary"].values())      # summaries are scalars


def test_hostile_report_still_fits_and_is_strict():
    detail = {
        "metric": "MaxSim (query,doc) pairs scored/sec", "value": 1.0e8, "unit": "pairs/s", "n_gpus": 8, "steps": 20, "warmup": 3,
        "ms_per_step": 4.9, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "w" * 5000, "docs_per_gpu": 125000, "doc_len": 1024, "n_queries": 4, "q_len": 32, "top_k": 10, "parallelism": "corpus-sharded x8"},
        "roofline": {"bound": "hbm", "achieved": 6700.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.84, "traffic": None, "kernel": "k",
                     "kernel_ms": float("nan"), "algorithmic_bytes_per_launch": 32770032768, "power": {"frac_on_zeros": float("inf")},
                     "traffic_source": "x" * 3000},
        "cpu_baseline": {"value": 65000.0, "unit": "pairs/s", "cores": 16, "kind": "reference", "sample": "s" * 4000, "what": "y" * 4000},
        "topk_parity": {"k": 10, "ids_equal": True, "ids_exact_equal": True, "max_rel_err": float("nan"), "what": "z" * 3000,
                        "k100": {"ids_equal": True, "ids_exact_equal": False}},
        "rccl_ranks": 8, "per_rank_kernel_ms": [4.8] * 8,
        "regimes": [{"n_queries": n, "q_len": "32", "frac": 0.5, "bound": "mfma", "blob": "r" * 500} for n in range(1, 400)],
        "loss_step_config5": {"Lq32": {"ColbertLoss": {"ours": {"both_directions": {"one_hipgraph_device_ms": float("nan"), "frac_of_roof": 0.1}}}}},
        "embed_and_score_1k_pages_vlm_in_the_loop": {"error": "e" * 9000},
    }
    text = line.dumps_line(detail, None)
    d = _check(text)
    assert d["roofline"]["kernel_ms"] is None and d["topk_parity"]["max_rel_err"] is None      # NaN -> null, never a bare NaN token
    assert len(d["config"]["workload"]) <= 200 and len(d["cpu_baseline"]["sample"]) <= 260
    assert d["rccl_ranks"] == 8 and len(d["per_rank_kernel_ms"]) == 8
    full = line.dumps_detail(detail)
    back = json.loads(full, parse_constant=lambda tok: (_ for _ in ()).throw(AssertionError(tok)))
    assert back["roofline"]["kernel_ms"] is None and len(back["regimes"]) == 399


def test_strict_rounds_and_nulls():
    assert line.strict({"a": float("nan"), "b": [1.23456789012, float("-inf")], "c": True, "d": None}) == {"a": None, "b": [1.23457, None], "c": True, "d": None}
    assert math.isclose(line.strict(32770032768), 32770032768)        # integers are kept exactly


def test_bench_py_prints_through_the_compact_line():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "bench_line.dumps_line(out" in src and "json.dumps(out)" not in src
    assert "allow_nan=False" in open(os.path.join(ROOT, "bench_legs", "line.py")).read()
