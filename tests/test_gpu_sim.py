"""GPU parity of the plain similarity-matrix kernel (msim_sim_matrix) and its two drop-ins against the live reference's
golden outputs: score_single_vector (processing_utils.py:103-130) and get_similarity_maps_from_embeddings
(interpretability/similarity_map_utils.py:9-55).  fp32: 1e-5 relative to max(|x|, 1); bf16: fp32-accurate values
within 1e-5 of the reference evaluated on fp32 upcasts, REF rounding mode within one bf16 ulp of its bf16 output."""
import numpy as np
import pytest
import torch

from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import colpali_amd

    colpali_amd._lib.lib()
    return colpali_amd


def _bf16(bits: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(bits.view(np.int16).copy()).view(torch.bfloat16)


def close(got, want, rtol=1e-5):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return np.max(np.abs(got - want) / np.maximum(np.abs(want), 1.0)) <= rtol


def test_score_single_vector_reference_unit_test_shape_and_goldens(amd):
    z = load_golden("sim_matrix.npz")
    qs, ps = list(torch.from_numpy(z["sv_q_f32"])), list(torch.from_numpy(z["sv_p_f32"]))
    scores = amd.score_single_vector(qs, ps, device="cuda:0")        # tests/utils/test_processing_utils.py:10-14
    assert scores.shape == (len(qs), len(ps)) and scores.dtype == torch.float32 and scores.device.type == "cuda"
    assert close(scores.cpu().numpy(), z["sv_scores_f32"])
    q, p = _bf16(z["sv_q_bf16"]).view(37, 1024), _bf16(z["sv_p_bf16"]).view(101, 1024)
    got = amd.score_single_vector(q, p, device="cuda:0").cpu().numpy()            # tensor inputs, BiPali width
    assert close(got, z["sv_scores_bf16_truth"])
    lit = amd.similarity_matrix(q.cuda(), p.cuda(), ref_rounding=True).cpu().numpy()
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(z["sv_scores_bf16"]), 1e-3))) - 7)
    assert np.all(np.abs(lit - z["sv_scores_bf16"]) <= ulp) and np.mean(lit == z["sv_scores_bf16"]) > 0.9
    with pytest.raises(ValueError, match="No queries provided"):
        amd.score_single_vector([], ps, device="cuda:0")


@pytest.mark.parametrize("dtype,n_a,n_b,dim", [(torch.bfloat16, 300, 1000, 128), (torch.float32, 33, 70, 48),
                                               (torch.float16, 129, 31, 320), (torch.bfloat16, 1, 5000, 100)])
def test_similarity_matrix_against_float64(amd, dtype, n_a, n_b, dim):
    g = torch.Generator().manual_seed(n_a + n_b)
    a, b = torch.randn(n_a, dim, generator=g).to(dtype), torch.randn(n_b, dim, generator=g).to(dtype)
    got = amd.similarity_matrix(a.cuda(), b.cuda()).cpu().double()
    want = a.double() @ b.double().T
    assert torch.max((got - want).abs() / want.abs().clamp_min(1.0)) < 1e-5
    # transpose-detecting: the matrix is not symmetric in its roles
    assert got.shape == (n_a, n_b)


def test_similarity_maps_match_the_reference(amd):
    z = load_golden("sim_matrix.npz")
    img = _bf16(z["map_img_bf16"]).view(2, 40, 128)
    qry = _bf16(z["map_qry_bf16"]).view(2, 9, 128)
    mask = torch.from_numpy(z["map_mask"])
    n_patches = [tuple(int(v) for v in row) for row in z["map_n_patches"]]
    maps = amd.get_similarity_maps_from_embeddings(img.cuda(), qry.cuda(), n_patches, mask.cuda())
    assert [tuple(m.shape) for m in maps] == [(9, 6, 5), (9, 4, 7)] and maps[0].dtype == torch.bfloat16
    for i, m in enumerate(maps):
        want = z[f"map_bf16_{i}"]
        ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(want), 1e-3))) - 7)
        assert np.all(np.abs(m.float().cpu().numpy() - want) <= ulp)
    maps32 = amd.get_similarity_maps_from_embeddings(img.float().cuda(), qry.float().cuda(), n_patches, mask.cuda())
    for i, m in enumerate(maps32):
        assert m.dtype == torch.float32 and close(m.cpu().numpy(), z[f"map_f32_{i}"])
    with pytest.raises(ValueError, match="does not match the number of non-padded image tokens"):
        amd.get_similarity_maps_from_embeddings(img.cuda(), qry.cuda(), [(6, 6), (4, 7)], mask.cuda())
