"""Checks that only run where the reference checkout exists (the build container).

They re-validate, against the LIVE reference functions, (a) the committed golden
fixtures and (b) oracle/torch_port.py, the restatement bench.py times as cpu_baseline.
"""
import numpy as np
import pytest
import torch

from oracle import refimport, torch_port
from tests.conftest import load_golden
from tests.helpers import config1_inputs

pytestmark = pytest.mark.skipif(not refimport.available(), reason="reference checkout not present")


def test_golden_config1_is_what_the_live_reference_returns():
    P, _ = refimport.load()
    z = load_golden("score_config1.npz")
    qs, ps = config1_inputs(z)
    np.testing.assert_array_equal(P.score_multi_vector(qs, ps, device="cpu").numpy(), z["literal"])
    np.testing.assert_array_equal(
        P.score_multi_vector([q.float() for q in qs], [p.float() for p in ps], device="cpu").numpy(), z["truth"])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_torch_port_equals_live_reference(dtype):
    P, _ = refimport.load()
    g = torch.Generator().manual_seed(3)
    qs = [torch.randn(n, 128, generator=g).to(dtype) for n in (5, 32, 17)]
    ps = [torch.randn(n, 128, generator=g).to(dtype) for n in (64, 33, 100, 1, 47)]
    for bs in (128, 2):
        want = P.score_multi_vector(qs, ps, batch_size=bs, device="cpu")
        got = torch_port.score_multi_vector_cpu(qs, ps, batch_size=bs)
        assert torch.equal(got, want)


def test_reference_own_scorer_unit_tests_pass_here():
    """tests/utils/test_processing_utils.py:15-35 restated against the live reference."""
    P, _ = refimport.load()
    qs = [torch.randn(2, 32), torch.randn(4, 32)]
    ps = [torch.randn(8, 32), torch.randn(4, 32), torch.randn(16, 32)]
    a = P.score_multi_vector(qs, ps, device="cpu")
    b = P.score_multi_vector(torch.nn.utils.rnn.pad_sequence(qs, batch_first=True),
                             torch.nn.utils.rnn.pad_sequence(ps, batch_first=True), device="cpu")
    assert a.shape == (2, 3) and torch.allclose(a, b)


def test_patch_routes_the_reference_entry_points_to_the_native_implementations():
    import colpali_amd

    P, L = refimport.load()
    orig = P.score_multi_vector
    colpali_amd.patch_colpali_engine()
    try:
        assert P.score_multi_vector is colpali_amd.score_multi_vector
        assert L.ColbertPairwiseCELoss is colpali_amd.ColbertPairwiseCELoss
        import colpali_engine.loss as pkg

        assert pkg.ColbertPairwiseCELoss is colpali_amd.ColbertPairwiseCELoss
        # same constructor surface as the reference class (late_interaction_losses.py:266-277)
        import inspect

        colpali_amd.unpatch_colpali_engine()
        def surface(fn):   # parameter names, order and defaults (annotations may be strings on one side)
            return [(p.name, p.default, p.kind) for p in inspect.signature(fn).parameters.values()]

        assert surface(colpali_amd.ColbertPairwiseCELoss.__init__) == surface(L.ColbertPairwiseCELoss.__init__)
        assert surface(colpali_amd.ColbertPairwiseCELoss.forward) == surface(L.ColbertPairwiseCELoss.forward)
        assert surface(colpali_amd.ColbertLoss.__init__) == surface(L.ColbertLoss.__init__)
        assert surface(colpali_amd.ColbertSigmoidLoss.__init__) == surface(L.ColbertSigmoidLoss.__init__)
        assert surface(colpali_amd.score_multi_vector) == surface(orig)
        assert surface(colpali_amd.score_single_vector) == surface(P.score_single_vector)
        sm = refimport.load_similarity_map_utils()
        assert surface(colpali_amd.get_similarity_maps_from_embeddings) == surface(sm.get_similarity_maps_from_embeddings)
    finally:
        colpali_amd.unpatch_colpali_engine()
    assert P.score_multi_vector is orig


def test_helper_known_answers_of_the_reference_hold_for_the_native_module():
    """tests/loss/test_li_losses.py:15-73 restated against colpali_amd.ColbertModule (CPU-sized helpers)."""
    import colpali_amd

    m = colpali_amd.ColbertModule(max_batch_size=5)
    idx, pos = m._get_idx(batch_size=3, offset=2, device=torch.device("cpu"))
    assert idx.tolist() == [0, 1, 2] and pos.tolist() == [2, 3, 4]
    m = colpali_amd.ColbertModule(tau=2.0)
    assert torch.allclose(m._smooth_max(torch.tensor([[0.0, 2.0]]), dim=1), 2.0 * torch.log(1 + torch.exp(torch.tensor(1.0))))
    m = colpali_amd.ColbertModule()
    raw = torch.tensor([[[1.0, 2.0], [3.0, 4.0]], [[5.0, 6.0], [7.0, 8.0]]])
    assert torch.allclose(m._aggregate(raw, use_smooth_max=False, dim_max=2, dim_sum=1), torch.tensor([6.0, 14.0]))
    s = torch.tensor([[1.0, 0.96], [0.5, 1.0]])
    m._filter_high_negatives(s, torch.tensor([0, 1]))
    assert abs(float(s[0, 1]) - 0.48) < 1e-6 and float(s[1, 0]) == 0.5
    n = m._apply_normalization(torch.tensor([[0.5, 1.0], [0.2, 0.8]]), torch.tensor([2.0, 4.0]))
    assert torch.allclose(n, torch.tensor([[0.25, 0.5], [0.05, 0.2]]))
