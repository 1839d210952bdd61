"""Property-based differential pinning of the C oracle against the LIVE reference scorer (build container only).

SURVEY.md 8(c): the reference's own tests fix shapes, not values, so what pins the restatement is the reference
function itself on many small random cases: ragged lengths, every reference block size, all-negative similarities with
and without longer block-mates (the zero-padding row joins the max), zero query rows, duplicated documents, widths
other than 128, 3-D tensor inputs.  Truth tier: fp32 inputs, 2e-6 relative.  Literal tier: bf16 inputs, bit-equal
(the reference's bf16 CPU path is deterministic: bf16(fp32 dot) -> max -> fp32 sum -> bf16).
"""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from oracle import maxsim_oracle as mo
from oracle import refimport

pytestmark = pytest.mark.skipif(not refimport.available(), reason="reference checkout not present")

case = st.fixed_dictionaries({
    "seed": st.integers(0, 2**31 - 1),
    "dim": st.sampled_from([4, 32, 128]),
    "q_lens": st.lists(st.integers(1, 6), min_size=1, max_size=5),
    "d_lens": st.lists(st.integers(1, 9), min_size=1, max_size=9),
    "batch_size": st.sampled_from([1, 2, 3, 128]),
    "negative": st.booleans(),          # documents on the far side of the queries: every similarity < 0
    "zero_query_row": st.booleans(),
    "duplicate_doc": st.booleans(),
})


def _build(c, dtype):
    g = torch.Generator().manual_seed(c["seed"])
    dim = c["dim"]
    base = torch.nn.functional.normalize(torch.randn(dim, generator=g), dim=-1)

    def rows(n, sign):
        x = torch.nn.functional.normalize(sign * base + 0.3 * torch.randn(n, dim, generator=g), dim=-1)
        return x.to(dtype)

    qs = [rows(n, 1.0) for n in c["q_lens"]]
    ps = [rows(n, -1.0 if c["negative"] else 1.0) for n in c["d_lens"]]
    if c["zero_query_row"]:
        qs[0] = qs[0].clone()
        qs[0][-1] = 0
    if c["duplicate_doc"] and len(ps) > 1:
        ps[-1] = ps[0].clone()
    return qs, ps


@settings(max_examples=40, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow])
@given(case)
def test_truth_tier_equals_the_live_reference_on_fp32_inputs(c):
    P, _ = refimport.load()
    qs, ps = _build(c, torch.float32)
    want = P.score_multi_vector(qs, ps, batch_size=c["batch_size"], device="cpu").numpy()
    got = mo.score_multi_vector([q.numpy() for q in qs], [p.numpy() for p in ps], batch_size=c["batch_size"], mode="f32")
    assert got.shape == want.shape
    assert np.all(np.abs(got - want) <= 2e-6 * np.maximum(np.abs(want), 1.0))


@settings(max_examples=40, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow])
@given(case)
def test_literal_tier_is_bit_equal_to_the_live_reference_on_bf16_inputs(c):
    P, _ = refimport.load()
    qs, ps = _build(c, torch.bfloat16)
    want = P.score_multi_vector(qs, ps, batch_size=c["batch_size"], device="cpu").numpy()
    got = mo.score_multi_vector([q.float().numpy() for q in qs], [p.float().numpy() for p in ps],
                                batch_size=c["batch_size"], mode="bf16ref")
    np.testing.assert_array_equal(got, want)


@settings(max_examples=15, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow])
@given(st.integers(0, 2**31 - 1), st.integers(1, 4), st.integers(1, 5), st.integers(1, 6), st.integers(1, 7))
def test_tensor_inputs_equal_the_live_reference(seed, n_q, n_d, lq, ld):
    # 3-D inputs: the reference re-stacks them block by block; physically present zero rows take part in the max
    P, _ = refimport.load()
    g = torch.Generator().manual_seed(seed)
    Q = torch.randn(n_q, lq, 16, generator=g)
    D = -torch.randn(n_d, ld, 16, generator=g).abs() * torch.sign(Q[0, 0]).abs()     # mixed signs
    D[0, -1] = 0
    want = P.score_multi_vector(Q, D, device="cpu").numpy()
    got = mo.score_multi_vector([q.numpy() for q in Q], [d.numpy() for d in D], mode="f32")
    assert np.all(np.abs(got - want) <= 2e-6 * np.maximum(np.abs(want), 1.0))


# ---------------------------------------------------------------------------------------------------------------------
# The loss oracle (oracle/li_loss_oracle.py, float64) against the live reference modules on fp32 inputs: value and
# autograd gradients, random constructor flags, offsets (C = world * B), ragged queries (zero rows inside the tensors).
from oracle import li_loss_oracle as lo  # noqa: E402

loss_case = st.fixed_dictionaries({
    "seed": st.integers(0, 2**31 - 1),
    "kind": st.sampled_from(["pairwise", "infonce", "sigmoid"]),
    "B": st.integers(2, 5),
    "world": st.integers(1, 3),
    "rank": st.integers(0, 2),
    "Lq": st.integers(1, 5),
    "Ld": st.integers(1, 7),
    "normalize_scores": st.booleans(),
    "pos_aware_negative_filtering": st.booleans(),
    "use_smooth_max": st.booleans(),
    "pad_queries": st.booleans(),
})
_CLS = {"pairwise": "ColbertPairwiseCELoss", "infonce": "ColbertLoss", "sigmoid": "ColbertSigmoidLoss"}


@settings(max_examples=60, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow])
@given(loss_case)
def test_loss_oracle_equals_the_live_reference_modules(c):
    _, L = refimport.load()
    g = torch.Generator().manual_seed(c["seed"])
    B, world = c["B"], c["world"]
    rank = c["rank"] % world
    if c["kind"] == "sigmoid":
        world, rank = 1, 0                                  # :447: the sigmoid loss flattens a square score matrix
    C, dim = world * B, 8
    Q = torch.nn.functional.normalize(torch.randn(B, c["Lq"], dim, generator=g), dim=-1)
    D = torch.nn.functional.normalize(torch.randn(C, c["Ld"], dim, generator=g), dim=-1)
    if c["pad_queries"] and c["Lq"] > 1:
        Q[0, -1] = 0                                        # a padded query position: shortens `lengths` (:296)
    kw = dict(normalize_scores=c["normalize_scores"], pos_aware_negative_filtering=c["pos_aware_negative_filtering"],
              use_smooth_max=c["use_smooth_max"])
    offset = rank * B
    q = Q.clone().requires_grad_(True)
    d = D.clone().requires_grad_(True)
    want = getattr(L, _CLS[c["kind"]])(**kw)(q, d, offset=offset)
    want.backward()
    loss, dq, dd = lo.loss_and_grads(c["kind"], Q, D, offset=offset, **kw)
    w = float(want.detach())
    assert abs(float(loss) - w) <= 2e-5 * max(1.0, abs(w))
    # the hard max routes a gradient to ONE arg-max; exact ties (only the zero query row produces them) may be split
    # differently by the two implementations, so that row is excluded, like everywhere else in the test suite
    mask = torch.ones_like(Q)
    if c["pad_queries"] and c["Lq"] > 1:
        mask[0, -1] = 0
    # the reference runs in fp32: its gradients carry ~1e-7 of the LARGEST gradient as absolute noise (softmax at T = 0.02)
    tol_q = 2e-4 * q.grad.abs() + 2e-6 * max(1.0, float(q.grad.abs().max()))
    tol_d = 2e-4 * d.grad.abs() + 2e-6 * max(1.0, float(d.grad.abs().max()))
    assert torch.all(((dq.float() - q.grad) * mask).abs() <= tol_q)
    if not (c["pad_queries"] and c["Lq"] > 1 and not c["use_smooth_max"]):
        assert torch.all((dd.float() - d.grad).abs() <= tol_d)


neg_case = st.fixed_dictionaries({
    "seed": st.integers(0, 2**31 - 1),
    "kind": st.sampled_from(["negative_ce", "pairwise_negative_ce"]),
    "B": st.integers(2, 4),
    "n_neg": st.integers(1, 3),
    "Lq": st.integers(1, 4),
    "Ld": st.integers(1, 6),
    "Ln": st.integers(1, 5),
    "normalize_scores": st.booleans(),
    "use_smooth_max": st.booleans(),
    "in_batch_term_weight": st.sampled_from([0.0, 0.3, 0.5]),
})
_NEG_CLS = {"negative_ce": "ColbertNegativeCELoss", "pairwise_negative_ce": "ColbertPairwiseNegativeCELoss"}


@settings(max_examples=40, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow])
@given(neg_case)
def test_explicit_negative_loss_oracle_equals_the_live_reference_modules(c):
    """late_interaction_losses.py:167-252 / :316-398 (paired contractions "bnd,bsd->bns" and "bnd,blsd->blns")."""
    _, L = refimport.load()
    g = torch.Generator().manual_seed(c["seed"])
    B, dim = c["B"], 8
    Q = torch.nn.functional.normalize(torch.randn(B, c["Lq"], dim, generator=g), dim=-1)
    D = torch.nn.functional.normalize(torch.randn(B, c["Ld"], dim, generator=g), dim=-1)
    N = torch.nn.functional.normalize(torch.randn(B, c["n_neg"], c["Ln"], dim, generator=g), dim=-1)
    kw = dict(normalize_scores=c["normalize_scores"], use_smooth_max=c["use_smooth_max"],
              in_batch_term_weight=c["in_batch_term_weight"])
    q, d, n = (t.clone().requires_grad_(True) for t in (Q, D, N))
    want = getattr(L, _NEG_CLS[c["kind"]])(**kw)(q, d, n)
    want.backward()
    loss, dq, dd, dn = lo.negatives_loss_and_grads(c["kind"], Q, D, N, **kw)
    w = float(want.detach())
    assert abs(float(loss) - w) <= 2e-5 * max(1.0, abs(w))
    for got, ref in ((dq, q.grad), (dd, d.grad), (dn, n.grad)):
        ref = ref if ref is not None else torch.zeros_like(got, dtype=torch.float32)
        tol = 2e-4 * ref.abs() + 2e-6 * max(1.0, float(ref.abs().max()))
        assert torch.all((got.float() - ref).abs() <= tol)


# ---------------------------------------------------------------------------------------------------------------------
# Token pooling: oracle/pooling_oracle.py (a restatement of SciPy's Ward NN-chain / fcluster) against the live reference pooler
from oracle import pooling_oracle as po  # noqa: E402


@settings(max_examples=25, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow])
@given(st.integers(0, 2**31 - 1), st.integers(2, 40), st.sampled_from([8, 32]), st.sampled_from([1, 2, 3, 5]), st.booleans())
def test_pooling_oracle_equals_the_live_reference_pooler(seed, n, dim, pool_factor, duplicates):
    """hierarchical_token_pooling.py:83-146 on random pages, including pages with repeated rows (exact distance ties)."""
    Pooler = refimport.load_token_pooler()
    g = torch.Generator().manual_seed(seed)
    e = torch.nn.functional.normalize(torch.randn(n, dim, generator=g), dim=-1)
    if duplicates and n >= 4:
        e[n // 2] = e[0]
        e[-1] = e[1]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # SciPy's "looks like an uncondensed distance matrix": the reference passes one on purpose
        want_pooled, want_map = Pooler()._pool_single_embedding(e, pool_factor)
    got_pooled, got_map = po.pool_single_embedding(e.numpy(), pool_factor)
    assert sorted(got_map) == sorted(want_map)
    for c in want_map:
        np.testing.assert_array_equal(np.sort(np.asarray(got_map[c])), np.sort(want_map[c][0].numpy()), err_msg=f"cluster {c}")
    np.testing.assert_allclose(got_pooled, want_pooled.numpy(), rtol=2e-6, atol=1e-7)
