"""Pin oracle/li_loss_oracle.py against the golden outputs of the live reference loss modules (CPU)."""
import numpy as np
import pytest
import torch

from oracle import li_loss_oracle as lo
from tests.conftest import load_golden

VARIANTS = {
    "default": dict(),
    "nonorm": dict(normalize_scores=False),
    "nonorm_T05": dict(normalize_scores=False, temperature=0.5),
    "filter": dict(normalize_scores=False, pos_aware_negative_filtering=True),
}


@pytest.mark.parametrize("cls,kind", [("ColbertPairwiseCELoss", "pairwise"), ("ColbertLoss", "infonce")])
@pytest.mark.parametrize("offset", [0, 6])
def test_loss_value_and_gradients_match_the_reference(cls, kind, offset):
    z = load_golden("loss_small.npz")
    Q, D = torch.from_numpy(z["Q"]), torch.from_numpy(z["D"])
    for vname, kw in VARIANTS.items():
        key = f"{cls}_{vname}_off{offset}"
        if key + "_loss" not in z.files:
            continue
        loss, dQ, dD = lo.loss_and_grads(kind, Q, D, offset=offset, **kw)
        assert abs(float(loss) - float(z[key + "_loss"])) < 2e-6 * max(1.0, abs(float(loss))), key
        np.testing.assert_allclose(dQ.numpy(), z[key + "_dQ"], rtol=2e-4, atol=2e-6, err_msg=key)
        np.testing.assert_allclose(dD.numpy(), z[key + "_dD"], rtol=2e-4, atol=2e-6, err_msg=key)


def test_known_answers_from_the_reference_unit_tests():
    # tests/loss/test_li_losses.py:137-147 (pairwise, zeros -> ln 2) and :76-88 (InfoNCE, zeros -> ln B)
    z = load_golden("loss_kat.npz")
    q = torch.zeros(2, 1, 3)
    loss, _, _ = lo.loss_and_grads("pairwise", q, q.clone(), normalize_scores=False, temperature=1.0)
    assert abs(float(loss) - float(z["ln2"])) < 1e-7 and abs(float(z["pairwise_zero"]) - float(z["ln2"])) < 1e-7
    q3 = torch.zeros(3, 1, 4)
    loss, _, _ = lo.loss_and_grads("infonce", q3, q3.clone(), normalize_scores=False, temperature=1.0)
    assert abs(float(loss) - np.log(3.0)) < 1e-7


@pytest.mark.parametrize("cls,kind", [("ColbertNegativeCELoss", "negative_ce"), ("ColbertPairwiseNegativeCELoss", "pairwise_negative_ce")])
def test_explicit_negative_variants_match_the_reference(cls, kind):
    z = load_golden("loss_negatives.npz")
    Q, D, N = (torch.from_numpy(z[k]) for k in ("Q", "D", "N"))
    variants = {"default": dict(), "nonorm_w0": dict(normalize_scores=False, in_batch_term_weight=0.0),
                "T1_w03": dict(temperature=1.0, in_batch_term_weight=0.3)}
    for vname, kw in variants.items():
        for offset in (0, 6):
            key = f"{cls}_{vname}_off{offset}"
            loss, dQ, dD, dN = lo.negatives_loss_and_grads(kind, Q, D, N, offset=offset, **kw)
            assert abs(float(loss) - float(z[key + "_loss"])) < 5e-6 * max(1.0, abs(float(loss))), key
            np.testing.assert_allclose(dQ.numpy(), z[key + "_dQ"], rtol=5e-4, atol=1e-5, err_msg=key)
            np.testing.assert_allclose(dN.numpy(), z[key + "_dN"], rtol=5e-4, atol=1e-5, err_msg=key)


SMOOTH = {"tau01": dict(use_smooth_max=True),
          "tau002_nonorm_T1": dict(use_smooth_max=True, tau=0.02, normalize_scores=False, temperature=1.0)}
SMOOTH_NEG = {"tau01": dict(use_smooth_max=True),
              "tau05_T1_w03": dict(use_smooth_max=True, tau=0.5, temperature=1.0, in_batch_term_weight=0.3)}


@pytest.mark.parametrize("cls,kind", [("ColbertPairwiseCELoss", "pairwise"), ("ColbertLoss", "infonce")])
def test_smooth_max_loss_and_gradients_match_the_reference(cls, kind):
    """use_smooth_max=True (late_interaction_losses.py:40-44, :88-90) against the live reference's outputs."""
    z = load_golden("loss_smooth.npz")
    zs = load_golden("loss_small.npz")
    Q, D = torch.from_numpy(zs["Q"]), torch.from_numpy(zs["D"])
    for vname, kw in SMOOTH.items():
        for offset in (0, 6):
            key = f"{cls}_{vname}_off{offset}"
            loss, dQ, dD = lo.loss_and_grads(kind, Q, D, offset=offset, **kw)
            assert abs(float(loss) - float(z[key + "_loss"])) < 5e-6 * max(1.0, abs(float(loss))), key
            np.testing.assert_allclose(dQ.numpy(), z[key + "_dQ"], rtol=5e-4, atol=5e-6, err_msg=key)
            if key + "_dD" in z.files:
                np.testing.assert_allclose(dD.numpy(), z[key + "_dD"], rtol=5e-4, atol=5e-6, err_msg=key)


@pytest.mark.parametrize("cls,kind", [("ColbertNegativeCELoss", "negative_ce"), ("ColbertPairwiseNegativeCELoss", "pairwise_negative_ce")])
def test_smooth_max_explicit_negative_variants_match_the_reference(cls, kind):
    z = load_golden("loss_smooth.npz")
    zn = load_golden("loss_negatives.npz")
    Q, D, N = (torch.from_numpy(zn[k]) for k in ("Q", "D", "N"))
    for vname, kw in SMOOTH_NEG.items():
        for offset in (0, 6):
            key = f"{cls}_{vname}_off{offset}"
            loss, dQ, dD, dN = lo.negatives_loss_and_grads(kind, Q, D, N, offset=offset, **kw)
            assert abs(float(loss) - float(z[key + "_loss"])) < 5e-6 * max(1.0, abs(float(loss))), key
            np.testing.assert_allclose(dQ.numpy(), z[key + "_dQ"], rtol=5e-4, atol=1e-5, err_msg=key)
            if key + "_dN" in z.files:
                np.testing.assert_allclose(dN.numpy(), z[key + "_dN"], rtol=5e-4, atol=1e-5, err_msg=key)


SIGMOID_VARIANTS = {"default": dict(), "nonorm_T1": dict(normalize_scores=False, temperature=1.0),
                    "filter_T05": dict(pos_aware_negative_filtering=True, temperature=0.5),
                    "smooth_T1": dict(use_smooth_max=True, temperature=1.0)}


def test_sigmoid_loss_matches_the_reference():
    """ColbertSigmoidLoss (late_interaction_losses.py:401-465), square in-batch case."""
    z = load_golden("loss_sigmoid.npz")
    zs = load_golden("loss_small.npz")
    Q, D = torch.from_numpy(zs["Q"]), torch.from_numpy(zs["D"])[:6]
    for vname, kw in SIGMOID_VARIANTS.items():
        loss, dQ, dD = lo.loss_and_grads("sigmoid", Q, D, **kw)
        assert abs(float(loss) - float(z[vname + "_loss"])) < 5e-6 * max(1.0, abs(float(loss))), vname
        np.testing.assert_allclose(dQ.numpy(), z[vname + "_dQ"], rtol=5e-4, atol=5e-6, err_msg=vname)
        np.testing.assert_allclose(dD.numpy(), z[vname + "_dD"], rtol=5e-4, atol=5e-6, err_msg=vname)
