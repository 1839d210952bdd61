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
