"""Pin oracle/pooling_oracle.py (CPU): every clustering stage against SciPy's own compiled functions -- the algorithm lives in
that third-party dependency -- and the whole pooling function against golden outputs of the live reference pooler."""
import warnings

import numpy as np
import pytest

from oracle import pooling_oracle as po
from tests.conftest import load_golden

scipy_hier = pytest.importorskip("scipy.cluster.hierarchy")
from scipy.cluster import _hierarchy  # noqa: E402
from scipy.spatial import distance  # noqa: E402


def test_every_stage_equals_scipy_bit_for_bit_including_ties():
    warnings.simplefilter("ignore")
    rng = np.random.default_rng(0)
    for trial in range(60):
        n, dim = int(rng.integers(2, 60)), int(rng.integers(2, 40))
        E = rng.standard_normal((n, dim)).astype(np.float32)
        E /= np.linalg.norm(E, axis=1, keepdims=True)
        if trial % 4 == 0:
            E[rng.integers(0, n, size=n // 3)] = E[0]              # duplicated rows -> exact zero distances, ties everywhere
        X = (np.float32(1) - (E @ E.T)).astype(np.float32)
        y = distance.pdist(X.astype(np.float64), "euclidean")
        assert np.array_equal(y, po.pdist_rows(X))
        Zs = _hierarchy.nn_chain(y, n, scipy_hier._LINKAGE_METHODS["ward"])
        Zo = po.sort_and_label(po.ward_nn_chain(y, n), n)
        assert np.array_equal(Zs, Zo), trial
        for t in sorted({1, 2, max(n // 2, 1), max(n // 3, 1), max(n // 4, 1), max(n - 1, 1), n, n + 3}):
            assert np.array_equal(scipy_hier.fcluster(Zs, t=t, criterion="maxclust"), po.fcluster_maxclust(Zo, n, t)), (trial, t)


def test_pooling_matches_the_live_reference_goldens():
    z = load_golden("token_pooling.npz")
    embs = np.split(z["emb_f32"], np.cumsum(z["lens"])[:-1])
    for pf in (2, 3, 4):
        for i, e in enumerate(embs):
            pooled, mapping = po.pool_single_embedding(e, pf)
            labels = np.full(e.shape[0], -1, np.int32)
            for c, idx in mapping.items():
                labels[idx] = c
            np.testing.assert_array_equal(labels, z[f"f32_pf{pf}_{i}_labels"], err_msg=f"pf={pf} page={i}")
            np.testing.assert_allclose(pooled, z[f"f32_pf{pf}_{i}_pooled"], rtol=2e-6, atol=1e-7)
