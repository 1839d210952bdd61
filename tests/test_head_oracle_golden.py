"""Pin oracle/head_oracle.py against the live reference's ColPali.forward (tests/golden/head_colpali_tiny.npz) -- CPU."""
import numpy as np
import torch

from oracle import head_oracle as ho
from tests.conftest import load_golden


def _bf16(bits: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(bits.view(np.int16).copy()).view(torch.bfloat16)


def test_fp32_head_matches_the_reference_forward():
    z = load_golden("head_colpali_tiny.npz")
    h, w, b = (torch.from_numpy(z[k]) for k in ("hidden_f32", "weight_f32", "bias_f32"))
    mask = torch.from_numpy(z["attention_mask"])
    got = ho.head_literal(h, w, b, mask)
    np.testing.assert_allclose(got.numpy(), z["out_f32"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(ho.head_truth(h, w, b, mask).numpy(), z["out_f32"], rtol=1e-5, atol=1e-6)
    assert np.all(z["out_f32"][1, 30:] == 0) and np.all(z["out_f32"][3, 11:] == 0)     # masked positions are exactly zero


def test_bf16_head_matches_the_reference_forward_bit_for_bit():
    z = load_golden("head_colpali_tiny.npz")
    h, w, b = (_bf16(z[k]) for k in ("hidden_bf16", "weight_bf16", "bias_bf16"))
    mask = torch.from_numpy(z["attention_mask"])
    got = ho.head_literal(h, w, b, mask)
    assert got.dtype == torch.bfloat16
    want = _bf16(z["out_bf16"])
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))
