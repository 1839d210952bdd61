"""Pin oracle/ against the golden vectors produced by the live reference.

CPU only.  Tolerances: truth tier 2e-6 relative (fp32 summation order is the
only freedom); literal tier: bit-equal to the reference's bf16 CPU output.
"""
import numpy as np
import pytest

from oracle import maxsim_oracle as mo
from tests.conftest import load_golden
from tests.helpers import bits_list_to_f32, config1_inputs, ragged_from_golden, rel_err

TRUTH_RTOL = 2e-6


@pytest.mark.parametrize("name", ["score_ragged_d128.npz", "score_ragged_d320.npz"])      # d320: ColQwen3's width (round 5)
def test_ragged_lists_all_block_sizes(name):
    z = load_golden(name)
    qs, ps = ragged_from_golden(z)
    qs, ps = bits_list_to_f32(qs), bits_list_to_f32(ps)
    for bs in z["batch_sizes"]:
        got = mo.score_multi_vector(qs, ps, batch_size=int(bs), mode="f32")
        assert got.shape == (len(qs), len(ps))
        assert rel_err(got, z[f"truth_bs{bs}"]) < TRUTH_RTOL
        lit = mo.score_multi_vector(qs, ps, batch_size=int(bs), mode="bf16ref")
        np.testing.assert_array_equal(lit, z[f"literal_bs{bs}"])


def test_block_size_changes_scores_only_through_padding():
    z = load_golden("score_ragged_d128.npz")
    # the reference itself gives different answers for different block sizes (finding 4) ...
    assert not np.array_equal(z["truth_bs128"], z["truth_bs1"])
    # ... and never a smaller score when more zero rows are visible
    assert np.all(z["truth_bs128"] >= z["truth_bs1"] - 1e-6)


def test_config1_truth_and_literal():
    z = load_golden("score_config1.npz")
    qs, ps = config1_inputs(z)
    qf = [q.float().numpy() for q in qs]
    pf = [p.float().numpy() for p in ps]
    assert rel_err(mo.score_multi_vector(qf, pf, mode="f32"), z["truth"]) < TRUTH_RTOL
    np.testing.assert_array_equal(mo.score_multi_vector(qf, pf, mode="bf16ref"), z["literal"])


def test_negative_similarities_and_zero_padding():
    z = load_golden("score_negative_clamp.npz")
    q = mo.bf16_bits_to_f32(z["q_bits"])
    short = mo.bf16_bits_to_f32(z["short_bits"])
    long_ = mo.bf16_bits_to_f32(z["long_bits"])
    alone = mo.score_multi_vector([q], [short])
    block = mo.score_multi_vector([q], [short, long_])
    split = mo.score_multi_vector([q], [short, long_], batch_size=1)
    assert rel_err(alone, z["truth_alone"]) < TRUTH_RTOL
    assert rel_err(block, z["truth_block"]) < TRUTH_RTOL
    assert rel_err(split, z["truth_split"]) < TRUTH_RTOL
    assert alone[0, 0] < 0 <= block[0, 0]          # the padding flips the sign
    assert split[0, 0] == alone[0, 0]


def test_tensor3d_inputs_with_physical_zero_rows():
    z = load_golden("score_tensor3d.npz")
    q = mo.bf16_bits_to_f32(z["q_bits"]).reshape(z["q_shape"])
    p = mo.bf16_bits_to_f32(z["p_bits"]).reshape(z["p_shape"])
    got = mo.score_multi_vector(list(q), list(p), mode="f32")
    assert rel_err(got, z["truth"]) < TRUTH_RTOL
    np.testing.assert_array_equal(mo.score_multi_vector(list(q), list(p), mode="bf16ref"), z["literal"])


def test_reference_unit_test_shape_fp32_d32():
    z = load_golden("score_fp32_d32.npz")
    q = z["q"].reshape(-1, 32)
    p = z["p"].reshape(-1, 32)
    qs = [q[:2], q[2:6]]
    ps = [p[:8], p[8:12], p[12:28]]
    got = mo.score_multi_vector(qs, ps)
    assert np.allclose(got, z["scores_list"], rtol=1e-5, atol=1e-6)
    assert np.allclose(got, z["scores_tensor"], rtol=1e-5, atol=1e-6)   # list == padded tensor


def test_empty_inputs_raise_like_the_reference():
    with pytest.raises(ValueError, match="No queries provided"):
        mo.score_multi_vector([], [np.zeros((2, 4), np.float32)])
    with pytest.raises(ValueError, match="No passages provided"):
        mo.score_multi_vector([np.zeros((2, 4), np.float32)], [])


def test_argmax_variant_matches_scores():
    z = load_golden("score_ragged_d128.npz")
    qs, ps = ragged_from_golden(z)
    Q = mo.pad_queries(bits_list_to_f32(qs))
    blob, off = mo.pack_docs(bits_list_to_f32(ps))
    s, am = mo.maxsim_argmax_f32(Q, blob, off, None)
    np.testing.assert_array_equal(s, mo.maxsim_f32(Q, blob, off, None))
    assert am.min() >= 0 and np.all(am < np.diff(off)[None, :, None])


def test_bf16_roundtrip_helpers():
    x = np.random.default_rng(0).standard_normal(4096).astype(np.float32)
    b = mo.f32_to_bf16_bits(x)
    import torch

    want = torch.from_numpy(x).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    np.testing.assert_array_equal(b, want)
    np.testing.assert_array_equal(mo.f32_to_bf16_bits(mo.bf16_bits_to_f32(b)), b)
