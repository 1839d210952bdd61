"""CPU: the oracle's end-to-end ranking (C scorer -> topk_oracle) against the LIVE reference's ranking committed in
tests/golden/topk_planted.npz (processing_utils.py:132-187 on fp32 copies, then torch.topk as
scripts/compute_hardnegs.py:92-94 does).  This pins the checker the GPU ranking tests and bench.py's `topk_parity` use."""
import numpy as np
import pytest

from oracle import maxsim_oracle as mo
from oracle import topk_oracle

from .conftest import load_golden
from .helpers import planted_inputs, ranking_tolerance, topk_tie_aware_equal


@pytest.mark.parametrize("tag", ["dense", "ragged"])
def test_oracle_ranking_equals_live_reference_ranking(tag):
    z = load_golden("topk_planted.npz")
    qs, ps = planted_inputs(z, tag)
    truth = mo.score_multi_vector([q.float().numpy() for q in qs], [p.float().numpy() for p in ps], batch_size=10**9)
    ref = z[f"{tag}_scores"]
    np.testing.assert_allclose(truth, ref, rtol=2e-6, atol=0)
    # planted top-10: ids identical to torch.topk of the reference's scores, order included
    _, ids10 = topk_oracle.topk(truth, 10)
    np.testing.assert_array_equal(ids10, z[f"{tag}_top10"])
    for r in range(truth.shape[0]):
        assert sorted(ids10[r].tolist()) == sorted(z[f"{tag}_planted"][r].tolist())
    # k = 100 reaches into the densely packed random documents: identical up to the precision of the two score sets
    _, ids100 = topk_oracle.topk(truth, 100)
    tol = ranking_tolerance(truth, ref)
    for r in range(truth.shape[0]):
        assert topk_tie_aware_equal(ids100[r], ref[r], 100, rtol=tol)
        assert topk_tie_aware_equal(z[f"{tag}_top100"][r], truth[r], 100, rtol=tol)


def test_default_blocking_changes_no_planted_rank():
    # the reference's default batch_size=128 zero-pads ragged blocks (clamp0); scores here are positive, so the ranking of the
    # golden is the same with and without blocking -- the oracle reproduces the blocked scores too
    z = load_golden("topk_planted.npz")
    qs, ps = planted_inputs(z, "ragged")
    blocked = mo.score_multi_vector([q.float().numpy() for q in qs], [p.float().numpy() for p in ps], batch_size=128)
    np.testing.assert_allclose(blocked, z["ragged_scores_bs128"], rtol=2e-6, atol=0)
    np.testing.assert_array_equal(topk_oracle.topk(blocked, 10)[1], z["ragged_top10_bs128"])
