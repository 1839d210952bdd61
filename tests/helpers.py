"""Shared helpers for the parity tests (CPU side only; no product imports)."""
from __future__ import annotations

import numpy as np

from oracle import maxsim_oracle as mo


def split_rows(flat: np.ndarray, lens) -> list:
    out, o = [], 0
    for n in lens:
        out.append(flat[o : o + int(n)])
        o += int(n)
    return out


def ragged_from_golden(z):
    """score_ragged_d128.npz -> (list of uint16 [L,dim] queries, list of docs)."""
    dim = int(z["dim"])
    q = z["q_bits"].reshape(-1, dim)
    p = z["p_bits"].reshape(-1, dim)
    return split_rows(q, z["q_lens"]), split_rows(p, z["p_lens"])


def config1_inputs(z):
    """Regenerate BASELINE config-1 inputs from the seed and check their sha256."""
    import hashlib

    import torch
    import torch.nn.functional as F

    g = torch.Generator().manual_seed(int(z["seed"]))

    def unit(n):
        return F.normalize(torch.randn(n, int(z["dim"]), generator=g), dim=-1).to(torch.bfloat16)

    qs = [unit(int(z["Lq"])) for _ in range(int(z["n_q"]))]
    ps = [unit(int(z["Ld"])) for _ in range(int(z["n_d"]))]
    h = hashlib.sha256()
    for t in qs + ps:
        h.update(t.contiguous().view(torch.int16).numpy().tobytes())
    assert h.digest() == z["sha256"].tobytes(), "torch RNG stream changed: regenerate tests/golden"
    return qs, ps


def bits_list_to_f32(lst):
    return [mo.bf16_bits_to_f32(x) for x in lst]


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-6)))


def topk_tie_aware_equal(got_idx, truth_scores, k, rtol=1e-6):
    """Top-k index parity that tolerates permutations only inside exact/near ties.

    `truth_scores` is the oracle's score row; `got_idx` the candidate top-k.
    Accept iff for every rank r the truth score of got_idx[r] equals the r-th
    best truth score within rtol (so a differing index is only allowed when the
    two documents are indistinguishable at the oracle's own precision).
    """
    order = np.argsort(-truth_scores, kind="stable")[:k]
    want = truth_scores[order]
    got = truth_scores[np.asarray(got_idx[:k], dtype=np.int64)]
    return bool(np.all(np.abs(got - want) <= rtol * np.maximum(np.abs(want), 1e-6)))
