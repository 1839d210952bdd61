"""GPU parity of hierarchical token pooling (msim_pool_cluster / msim_pool_reduce, colpali_amd.HierarchicalTokenPooler).

Cluster labels are index work: they must EQUAL the reference's (golden outputs of the live reference pooler, SciPy on the
same inputs).  The one step whose rounding cannot be made bit-equal to the CPU is the fp32 Gram matrix (GEMM accumulation
order); the exactness of everything after it is tested with inputs whose Gram matrix is exact in fp32 in any order (entries on
a coarse dyadic grid), ties included.  Pooled rows: fp32 within 1e-5 relative + 1e-6, 16-bit within one ulp of the dtype.
"""
import warnings

import numpy as np
import pytest
import torch

from oracle import pooling_oracle as po
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import colpali_amd

    colpali_amd._lib.lib()
    return colpali_amd


def _labels_from_mapping(mapping, n):
    lab = np.full(n, -1, np.int64)
    for c, idx in mapping.items():
        lab[idx[0].cpu().numpy()] = c
    return lab


def _pooled_close(got: torch.Tensor, want: np.ndarray, dtype):
    g = got.float().cpu().numpy()
    if dtype == torch.float32:
        return np.all(np.abs(g - want) <= 1e-5 * np.abs(want) + 1e-6)
    mant = 7 if dtype == torch.bfloat16 else 10
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(want), 1e-6))) - mant)
    return np.all(np.abs(g - want) <= ulp)


@pytest.mark.parametrize("tag,dtype", [("f32", torch.float32), ("bf16", torch.bfloat16)])
def test_goldens_of_the_live_reference_pooler(amd, tag, dtype):
    z = load_golden("token_pooling.npz")
    embs = [torch.from_numpy(e).to(dtype) for e in np.split(z["emb_f32"], np.cumsum(z["lens"])[:-1])]
    pooler = amd.HierarchicalTokenPooler()
    for pf in (2, 3, 4):
        res = pooler.pool_embeddings([e.cuda() for e in embs], pool_factor=pf, return_dict=True)
        assert isinstance(res, amd.TokenPoolingOutput) and len(res.pooled_embeddings) == len(embs)
        for i, e in enumerate(embs):
            want_labels = z[f"{tag}_pf{pf}_{i}_labels"]
            got_labels = _labels_from_mapping(res.cluster_id_to_indices[i], e.shape[0])
            np.testing.assert_array_equal(got_labels, want_labels, err_msg=f"{tag} pf={pf} page={i}")
            pe = res.pooled_embeddings[i]
            assert pe.dtype == dtype and pe.device.type == "cuda" and tuple(pe.shape) == z[f"{tag}_pf{pf}_{i}_pooled"].shape
            assert _pooled_close(pe, z[f"{tag}_pf{pf}_{i}_pooled"], dtype), (tag, pf, i)
            assert sorted(res.cluster_id_to_indices[i].keys()) == list(range(max(e.shape[0] // pf, 1)))


def test_exact_gram_inputs_reproduce_scipy_labels_including_ties(amd):
    """Entries in {-2..2}/4 and dim 16: every dot product is exact in fp32 in any summation order, so the distance matrix
    the GPU clusters is bit-identical to the CPU's and the labels must equal SciPy's -- with massive ties."""
    from scipy.cluster.hierarchy import fcluster, linkage

    warnings.simplefilter("ignore")
    rng = np.random.default_rng(3)
    pages = [(rng.integers(-2, 3, size=(n, 16)) / 4.0).astype(np.float32) for n in (5, 33, 64, 100, 257, 2, 3)]
    pooler = amd.HierarchicalTokenPooler()
    for pf in (2, 3, 5):
        res = pooler.pool_embeddings([torch.from_numpy(p).cuda() for p in pages], pool_factor=pf, return_dict=True)
        for i, p in enumerate(pages):
            n = p.shape[0]
            X = (np.float32(1) - (torch.from_numpy(p) @ torch.from_numpy(p).T).numpy())
            want = fcluster(linkage(X, metric="euclidean", method="ward"), t=max(n // pf, 1), criterion="maxclust") - 1
            got = _labels_from_mapping(res.cluster_id_to_indices[i], n)
            np.testing.assert_array_equal(got, want, err_msg=f"pf={pf} page={i} n={n}")
            np.testing.assert_array_equal(got, po.cluster_labels(X, max(n // pf, 1)))


@pytest.mark.parametrize("n,dim,dtype", [(1030, 128, torch.bfloat16), (779, 128, torch.float32), (300, 320, torch.bfloat16)])
def test_page_sized_inputs_against_scipy(amd, n, dim, dtype):
    """A ColPali / ColQwen2 sized page with cluster structure (noisy copies of prototypes, like real patches)."""
    from scipy.cluster.hierarchy import fcluster, linkage

    warnings.simplefilter("ignore")
    g = torch.Generator().manual_seed(n)
    proto = torch.nn.functional.normalize(torch.randn(40, dim, generator=g), dim=-1)
    e = torch.nn.functional.normalize(proto[torch.randint(0, 40, (n,), generator=g)] + 0.2 * torch.randn(n, dim, generator=g), dim=-1).to(dtype)
    res = amd.HierarchicalTokenPooler().pool_embeddings([e.cuda(), e[: n // 2].cuda()], pool_factor=3, return_dict=True)
    for i, page in enumerate((e, e[: n // 2])):
        ef = page.float()
        X = 1 - torch.mm(ef, ef.t()).numpy()
        m = page.shape[0]
        want = fcluster(linkage(X, metric="euclidean", method="ward"), t=max(m // 3, 1), criterion="maxclust") - 1
        got = _labels_from_mapping(res.cluster_id_to_indices[i], m)
        np.testing.assert_array_equal(got, want)
        k = int(want.max()) + 1
        pooled = torch.stack([torch.nn.functional.normalize(ef[torch.from_numpy(want == c)].mean(dim=0), p=2, dim=-1) for c in range(k)])
        assert _pooled_close(res.pooled_embeddings[i], pooled.to(dtype).float().numpy(), dtype)


def test_tensor_input_padding_and_api_contract(amd):
    g = torch.Generator().manual_seed(9)
    a, b = torch.nn.functional.normalize(torch.randn(10, 128, generator=g), dim=-1), torch.nn.functional.normalize(torch.randn(20, 128, generator=g), dim=-1)
    pooler = amd.HierarchicalTokenPooler()
    lst = pooler.pool_embeddings([a.cuda(), b.cuda()], pool_factor=2)
    assert isinstance(lst, list) and lst[0].shape == (5, 128) and lst[1].shape == (10, 128)
    for side in ("left", "right"):
        padded = torch.nn.utils.rnn.pad_sequence([a, b], batch_first=True, padding_value=0.0, padding_side=side).cuda()
        out = pooler.pool_embeddings(padded, pool_factor=2, padding=True, padding_side=side)
        assert isinstance(out, torch.Tensor) and out.shape == (2, 10, 128)
        rows = out[0][5:] if side == "left" else out[0][:5]
        assert torch.allclose(rows.cpu(), lst[0].cpu(), atol=1e-6)
        assert torch.count_nonzero(out[0][:5] if side == "left" else out[0][5:]) == 0
    # CPU tensors in -> CPU tensors out (the reference restores the original device, :143)
    cpu_out = pooler.pool_embeddings([a, b], pool_factor=2)
    assert cpu_out[0].device.type == "cpu" and torch.allclose(cpu_out[0], lst[0].cpu(), atol=1e-6)
    # pool_factor 1 is the identity with a single cluster map (:107-109)
    ident = pooler.pool_embeddings([a.cuda()], pool_factor=1, return_dict=True)
    assert torch.equal(ident.pooled_embeddings[0], a.cuda()) and torch.equal(ident.cluster_id_to_indices[0][0][0], torch.arange(10))
    assert pooler.pool_embeddings([], pool_factor=2).pooled_embeddings == []
    with pytest.raises(ValueError, match="more than one token"):
        pooler.pool_embeddings([a[:1].cuda()], pool_factor=2)
    with pytest.raises(ValueError, match="list of 2D tensors or a 3D tensor"):
        pooler.pool_embeddings(a.cuda(), pool_factor=2)


def test_pages_above_the_lds_capacity_keep_their_state_in_hbm(amd):
    """The reference has no page-size limit (hierarchical_token_pooling.py:83-146).  Pages above 2048 rows run the cluster kernel
    with its state in HBM; a short page of the same call still takes the LDS form.  Labels must equal SciPy's: once on a
    page with real cluster structure, once on exact-Gram inputs with massive ties."""
    from scipy.cluster.hierarchy import fcluster, linkage

    warnings.simplefilter("ignore")
    g = torch.Generator().manual_seed(2600)
    proto = torch.nn.functional.normalize(torch.randn(90, 128, generator=g), dim=-1)
    big = torch.nn.functional.normalize(proto[torch.randint(0, 90, (2600,), generator=g)] + 0.2 * torch.randn(2600, 128, generator=g), dim=-1)
    rng = np.random.default_rng(11)
    ties = torch.from_numpy((rng.integers(-2, 3, size=(2300, 16)) / 4.0).astype(np.float32))
    small = big[:300].clone()
    pages = [big, small, ties]
    pooler = amd.HierarchicalTokenPooler()
    res = pooler.pool_embeddings([big.cuda(), small.cuda()], pool_factor=3, return_dict=True)
    res2 = pooler.pool_embeddings([ties.cuda()], pool_factor=3, return_dict=True)              # (another width: a call of its own)
    res.pooled_embeddings.extend(res2.pooled_embeddings)
    res.cluster_id_to_indices.extend(res2.cluster_id_to_indices)
    for i, page in enumerate(pages):
        m = page.shape[0]
        X = np.float32(1) - torch.mm(page, page.t()).numpy()
        want = fcluster(linkage(X, metric="euclidean", method="ward"), t=max(m // 3, 1), criterion="maxclust") - 1
        got = _labels_from_mapping(res.cluster_id_to_indices[i], m)
        np.testing.assert_array_equal(got, want, err_msg=f"page {i} of {m} rows")
        k = int(want.max()) + 1
        assert res.pooled_embeddings[i].shape == (k, page.shape[1])
        if i < 2:
            pooled = torch.stack([torch.nn.functional.normalize(page[torch.from_numpy(want == c)].mean(dim=0), p=2, dim=-1) for c in range(k)])
            assert _pooled_close(res.pooled_embeddings[i], pooled.numpy(), torch.float32)
    with pytest.raises(NotImplementedError, match="at most 32768"):
        amd.pooling.cluster_pages(torch.zeros((1, 128), device="cuda"), torch.zeros(2, dtype=torch.int32, device="cuda"),
                                  torch.tensor([40000]), 3)
