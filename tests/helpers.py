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
    return bool(np.all(np.abs(got - want) <= rtol * np.maximum(np.abs(want), 1.0)))


def planted_inputs(z, tag):
    """Regenerate the planted-retrieval inputs of topk_planted.npz (tests/golden/make_golden.py:planted_inputs, restated
    here because that file imports the live reference) and check their sha256.  Returns (queries, docs) as bf16 tensors."""
    import hashlib

    import torch
    import torch.nn.functional as F

    seed, n_q, Lq, n_d, lo, hi, n_pl = (int(v) for v in z[f"{tag}_params"])
    dim = 128
    g = torch.Generator().manual_seed(seed)

    def unit(n):
        return F.normalize(torch.randn(n, dim, generator=g), dim=-1).to(torch.bfloat16)

    qs = [unit(Lq) for _ in range(n_q)]
    lens = [lo] * n_d if lo == hi else torch.randint(lo, hi + 1, (n_d,), generator=g).tolist()
    ps = [unit(n).float() for n in lens]
    slots = torch.randperm(n_d, generator=g)[: n_q * n_pl].view(n_q, n_pl)
    for qi in range(n_q):
        for j in range(n_pl):
            d = int(slots[qi, j])
            sigma = 0.3 + 0.05 * j
            noisy = F.normalize(qs[qi].float() + sigma * torch.randn(Lq, dim, generator=g) / dim**0.5, dim=-1)
            rows = torch.randperm(lens[d], generator=g)[:Lq]
            ps[d][rows] = noisy
    ps = [p.to(torch.bfloat16) for p in ps]
    h = hashlib.sha256()
    for t in qs + ps:
        h.update(t.contiguous().view(torch.int16).numpy().tobytes())
    assert h.digest() == z[f"{tag}_sha256"].tobytes(), "torch RNG stream changed: regenerate tests/golden/topk_planted.npz"
    return qs, ps


def ranking_tolerance(got_scores, truth_scores, cap=2e-6):
    """Tolerance for the tie-aware ranking comparison: two documents may swap ranks only if their truth scores are closer
    than twice the largest relative score error actually measured (two computations that agree to e cannot disagree on the
    order of scores further apart than 2e).  The measured error itself must stay under cap / 2."""
    got = np.asarray(got_scores, dtype=np.float64)
    truth = np.asarray(truth_scores, dtype=np.float64)
    e = float(np.max(np.abs(got - truth) / np.maximum(np.abs(truth), 1.0)))     # same error measure as every score test
    assert 2 * e <= cap, f"score error {e:.3e} too large for a meaningful ranking comparison"
    return 2 * e + 1e-9
