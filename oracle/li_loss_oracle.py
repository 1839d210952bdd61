"""float64 torch restatement of the reference's in-batch late-interaction losses -- TEST INFRASTRUCTURE ONLY.

Restates colpali_engine/loss/late_interaction_losses.py
    :296-313  ColbertPairwiseCELoss.forward
    :152-164  ColbertLoss.forward
    :444-465  ColbertSigmoidLoss.forward
    :40-107   helper semantics (normalisation, pos-aware filtering)
with the materialised einsum exactly as the reference writes it, in float64 so that it can serve as
truth for both fp32 (golden vectors) and bf16-valued inputs.  Gradients come from torch autograd on
this restatement.  Pinned by tests/test_loss_oracle_golden.py against tests/golden/loss_small.npz
(outputs of the live reference modules).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F  # noqa: N812


def _aggregate(raw, use_smooth_max, tau, dim_max, dim_sum):
    """:72-91 -- amax, or the smooth max tau * logsumexp(raw / tau) (:40-44), then the token sum."""
    if use_smooth_max:
        return (tau * torch.logsumexp(raw / tau, dim=dim_max)).sum(dim=dim_sum)
    return raw.amax(dim=dim_max).sum(dim=dim_sum)


def _scores(q, d, offset, normalize_scores, pos_aware_negative_filtering, filter_threshold, filter_factor,
            use_smooth_max=False, tau=0.1):
    lengths = (q[:, :, 0] != 0).sum(dim=1)                                   # :296
    raw = torch.einsum("bnd,csd->bcns", q, d)                                # :297
    scores = _aggregate(raw, use_smooth_max, tau, 3, 2)                      # :298 -> :88-91
    if normalize_scores:
        scores = scores / lengths.unsqueeze(1)                               # :58
    B = scores.size(0)
    idx = torch.arange(B)
    pos_idx = idx + offset                                                   # :37
    if pos_aware_negative_filtering:                                         # :100-107
        pos = scores[idx, pos_idx]
        mask = scores > filter_threshold * pos.unsqueeze(1)
        mask[idx, pos_idx] = False
        scores = torch.where(mask, scores * filter_factor, scores)
    return scores, pos_idx


def loss_and_grads(kind: str, Q: torch.Tensor, D: torch.Tensor, offset: int = 0, temperature=None,
                   normalize_scores: bool = True, pos_aware_negative_filtering: bool = False,
                   filter_threshold: float = 0.95, filter_factor: float = 0.5,
                   use_smooth_max: bool = False, tau: float = 0.1):
    """kind in {"pairwise", "infonce", "sigmoid"} -> (loss, dQ, dD) as float64 tensors."""
    q = Q.detach().double().requires_grad_(True)
    d = D.detach().double().requires_grad_(True)
    scores, pos_idx = _scores(q, d, offset, normalize_scores, pos_aware_negative_filtering, filter_threshold, filter_factor,
                              use_smooth_max, tau)
    if kind == "pairwise":
        T = 1.0 if temperature is None else temperature
        pos = scores.diagonal(offset=offset)                                 # :309
        top2 = scores.topk(2, dim=1).values                                  # :310
        neg = torch.where(top2[:, 0] == pos, top2[:, 1], top2[:, 0])         # :311
        loss = F.softplus((neg - pos) / T).mean()                            # :313
    elif kind == "infonce":
        T = 0.02 if temperature is None else temperature
        loss = F.cross_entropy(scores / T, pos_idx)                          # :164
    elif kind == "sigmoid":
        T = 0.02 if temperature is None else temperature
        n = scores.size(0)
        sign = -torch.ones(n * n, dtype=scores.dtype)
        sign[pos_idx * (n + 1)] = 1.0                                        # :456-459
        loss = F.softplus(-(scores.reshape(-1) / T) * sign).mean()           # :462-465
    else:
        raise ValueError(kind)
    loss.backward()
    return loss.detach(), q.grad, d.grad


def negatives_loss_and_grads(kind: str, Q, D, N, offset: int = 0, temperature: float = 0.02,
                             normalize_scores: bool = True, in_batch_term_weight: float = 0.5,
                             use_smooth_max: bool = False, tau: float = 0.1):
    """Explicit-negative variants (:215-252 "negative_ce", :361-398 "pairwise_negative_ce").
    N: [B, n_neg, Lneg, dim].  Returns (loss, dQ, dD, dN) in float64."""
    q = Q.detach().double().requires_grad_(True)
    d = D.detach().double().requires_grad_(True)
    n = N.detach().double().requires_grad_(True)
    B = q.size(0)
    lengths = (q[:, :, 0] != 0).sum(dim=1)
    pos_raw = torch.einsum("bnd,bsd->bns", q, d[offset : offset + B])          # :235-237 / :381-383
    neg_raw = torch.einsum("bnd,blsd->blns", q, n)                             # :238 / :384
    pos = _aggregate(pos_raw, use_smooth_max, tau, 2, 1)
    neg = _aggregate(neg_raw, use_smooth_max, tau, 3, 2)
    if normalize_scores:
        pos = pos / lengths
        neg = neg / lengths.unsqueeze(1)
    loss = F.softplus((neg - pos.unsqueeze(1)) / temperature).mean()           # :246 / :392
    if in_batch_term_weight > 0:
        scores, pos_idx = _scores(q, d, offset, normalize_scores, False, 0.95, 0.5, use_smooth_max, tau)
        if kind == "negative_ce":
            ib = F.cross_entropy(scores / temperature, pos_idx)
        else:
            p = scores.diagonal(offset=offset)
            top2 = scores.topk(2, dim=1).values
            ng = torch.where(top2[:, 0] == p, top2[:, 1], top2[:, 0])
            ib = F.softplus((ng - p) / temperature).mean()
        loss = loss * (1 - in_batch_term_weight) + ib * in_batch_term_weight
    loss.backward()
    return loss.detach(), q.grad, d.grad, n.grad
