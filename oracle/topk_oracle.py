"""numpy restatement of the ranking used for sharded retrieval -- TEST INFRASTRUCTURE ONLY.

The reference ranks by sorting score_multi_vector's output (callers use torch.topk/argsort;
processing_utils.py:189-219 exposes k=10 through the experimental PLAID path).  The product
defines the total order (score descending, id ascending); this file states it with a stable
lexicographic sort so the HIP selection kernel and the multi-shard merge can be checked
bit-exactly, ties included.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


def topk(scores: np.ndarray, k: int, id_base: int = 0, ids: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
    scores = np.asarray(scores, dtype=np.float32)
    n_q, n = scores.shape
    out_s = np.full((n_q, k), -np.inf, dtype=np.float32)
    out_i = np.full((n_q, k), -1, dtype=np.int64)
    for r in range(n_q):
        row_ids = (np.arange(n, dtype=np.int64) + id_base) if ids is None else np.asarray(ids[r], dtype=np.int64)
        valid = row_ids >= 0
        s, i = scores[r][valid] + np.float32(0.0), row_ids[valid]     # +0.0 folds -0.0 onto +0.0
        order = np.lexsort((i, -s.astype(np.float64)))               # primary: score desc, secondary: id asc
        order = order[:k]
        out_s[r, : len(order)] = s[order]
        out_i[r, : len(order)] = i[order]
    return out_s, out_i


def torch_select(scores, k, id_base=0, ids=None):
    """Same contract as colpali_amd.retrieval.topk, for injecting into host-logic tests on CPU."""
    import torch

    s, i = topk(scores.cpu().numpy(), k, id_base, None if ids is None else ids.cpu().numpy())
    return torch.from_numpy(s), torch.from_numpy(i)
