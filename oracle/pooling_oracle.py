"""CPU restatement of the reference's hierarchical token pooling -- TEST INFRASTRUCTURE ONLY.

Reference: colpali_engine/compression/token_pooling/hierarchical_token_pooling.py:83-146 (`_pool_single_embedding`):
    similarities = torch.mm(embedding, embedding.t());  distances = 1 - similarities.numpy()           (:117-118)
    Z = linkage(distances, metric="euclidean", method="ward")                                          (:120)
    cluster_labels = fcluster(Z, t=max(token_length // pool_factor, 1), criterion="maxclust") - 1      (:121-122)
    pooled[c] = normalize(mean(embedding[labels == c]))   for c in range(max_clusters), skipping empty   (:127-140)
The clustering lives in a third-party dependency, SciPy (pyproject.toml pins `scipy`; this container has 1.15.3), whose
compiled Cython sources are not shipped.  Its published algorithm is restated here:
  * `linkage` on a 2-D array treats the ROWS of the [n, n] matrix as n observations and calls `pdist` (double,
    sequential sum of squared differences in column order, sqrt);
  * ward uses the nearest-neighbour-chain algorithm (`_hierarchy.nn_chain`): chain restarts at the first active cluster,
    the previous chain element is preferred among equals, merges are recorded as (min id, max id), the Lance-Williams
    update is sqrt(((ni+nx) d_xi^2 + (ni+ny) d_yi^2 - ni d_xy^2) / (nx+ny+ni)) evaluated as in `_ward`;
    the merges are then stably sorted by distance and relabelled with a union-find (`label`);
  * `fcluster(..., "maxclust")` = `cluster_maxclust_dist`: the maximum merge distance below each node is the monotone
    criterion; a bisection over the node indices finds the smallest threshold giving at most t clusters, then a
    depth-first traversal from the root numbers the flat clusters (`cluster_monocrit`).
Pinned by tests/test_pooling_oracle.py: every stage against SciPy's own compiled functions on random and tie-heavy
inputs (bit-equal Z, identical labels), and the whole function against golden outputs of the live reference pooler.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np


def pdist_rows(X: np.ndarray) -> np.ndarray:
    """Condensed euclidean distances between the rows of X, double precision, sequential accumulation over the columns
    (scipy.spatial.distance.pdist(X, "euclidean") on a float64 copy)."""
    X = np.asarray(X, dtype=np.float64)
    n = X.shape[0]
    out = np.empty(n * (n - 1) // 2, dtype=np.float64)
    k = 0
    for i in range(n - 1):
        diff = X[i + 1 :] - X[i]                       # [n-i-1, m]
        acc = np.zeros(diff.shape[0], dtype=np.float64)
        for c in range(diff.shape[1]):                 # column order, one rounding per product and per add
            acc = acc + diff[:, c] * diff[:, c]
        out[k : k + diff.shape[0]] = np.sqrt(acc)
        k += diff.shape[0]
    return out


def _square(y: np.ndarray, n: int) -> np.ndarray:
    D = np.zeros((n, n), dtype=np.float64)
    iu = np.triu_indices(n, 1)
    D[iu] = y
    D.T[iu] = y
    return D


def ward_nn_chain(y: np.ndarray, n: int) -> np.ndarray:
    """scipy.cluster._hierarchy.nn_chain(y, n, ward) before sorting/labelling: rows (x, y, dist, size)."""
    D = _square(y, n)
    size = np.ones(n, dtype=np.int64)
    Z = np.empty((n - 1, 4), dtype=np.float64)
    chain: List[int] = []
    idx = np.arange(n)
    for k in range(n - 1):
        if not chain:
            chain.append(int(np.argmax(size > 0)))     # first active cluster
        while True:
            x = chain[-1]
            if len(chain) > 1:
                yb = chain[-2]
                current_min = D[x, yb]
            else:
                yb = -1
                current_min = np.inf
            row = np.where((size > 0) & (idx != x), D[x], np.inf)
            i = int(np.argmin(row))                    # first index among equal minima
            if row[i] < current_min:                   # strictly smaller than the previous chain element's distance
                current_min = row[i]
                yb = i
            if len(chain) > 1 and yb == chain[-2]:
                break
            chain.append(yb)
        chain.pop()
        chain.pop()
        yy = yb
        if x > yy:
            x, yy = yy, x
        nx, ny = int(size[x]), int(size[yy])
        Z[k] = (x, yy, current_min, nx + ny)
        size[x] = 0
        size[yy] = nx + ny
        act = np.nonzero((size > 0) & (idx != yy))[0]
        ni = size[act].astype(np.float64)
        t = 1.0 / (nx + ny + ni)
        dxi, dyi = D[act, x], D[act, yy]
        new = np.sqrt((ni + nx) * t * dxi * dxi + (ni + ny) * t * dyi * dyi - ni * t * current_min * current_min)
        D[act, yy] = new
        D[yy, act] = new
    return Z


def sort_and_label(Z: np.ndarray, n: int) -> np.ndarray:
    """Stable sort by distance, then cluster ids by union-find (scipy `label`)."""
    Z = Z[np.argsort(Z[:, 2], kind="mergesort")].copy()
    parent = np.arange(2 * n - 1)
    sizes = np.ones(2 * n - 1, dtype=np.int64)
    nxt = n

    def find(a):
        r = a
        while parent[r] != r:
            r = parent[r]
        while parent[a] != r:
            parent[a], a = r, parent[a]
        return r

    for i in range(n - 1):
        xr, yr = find(int(Z[i, 0])), find(int(Z[i, 1]))
        Z[i, 0], Z[i, 1] = (xr, yr) if xr < yr else (yr, xr)
        parent[xr] = nxt
        parent[yr] = nxt
        sizes[nxt] = sizes[xr] + sizes[yr]
        Z[i, 3] = sizes[nxt]
        nxt += 1
    return Z


def max_dists(Z: np.ndarray, n: int) -> np.ndarray:
    """MD[i] = largest merge distance in the subtree of node n + i (get_max_dist_for_each_cluster)."""
    MD = np.empty(n - 1, dtype=np.float64)
    for i in range(n - 1):
        m = Z[i, 2]
        for c in (int(Z[i, 0]), int(Z[i, 1])):
            if c >= n:
                m = max(m, MD[c - n])
        MD[i] = m
    return MD


def _count_clusters(Z, MC, n, thresh, max_nc):
    """Number of flat clusters at `thresh` (stops early once it exceeds max_nc), as cluster_maxclust_monocrit counts."""
    visited = np.zeros(2 * n - 1, dtype=bool)
    stack = [2 * n - 2]
    nc = 0
    while stack:
        root = stack[-1] - n
        lc, rc = int(Z[root, 0]), int(Z[root, 1])
        if MC[root] <= thresh:
            nc += 1
            if nc > max_nc:
                break
            stack.pop()
            visited[lc] = visited[rc] = True
            continue
        if not visited[lc]:
            visited[lc] = True
            if lc >= n:
                stack.append(lc)
                continue
            nc += 1
            if nc > max_nc:
                break
        if not visited[rc]:
            visited[rc] = True
            if rc >= n:
                stack.append(rc)
                continue
            nc += 1
            if nc > max_nc:
                break
        stack.pop()
    return nc


def monocrit_labels(Z, MC, n, cutoff) -> np.ndarray:
    """Flat cluster numbers 1..k by depth-first traversal (cluster_monocrit)."""
    T = np.zeros(n, dtype=np.int32)
    visited = np.zeros(2 * n - 1, dtype=bool)
    stack = [2 * n - 2]
    n_cluster, leader = 0, -1
    while stack:
        root = stack[-1] - n
        lc, rc = int(Z[root, 0]), int(Z[root, 1])
        if leader == -1 and MC[root] <= cutoff:
            leader = root
            n_cluster += 1
        if lc >= n and not visited[lc]:
            visited[lc] = True
            stack.append(lc)
            continue
        if rc >= n and not visited[rc]:
            visited[rc] = True
            stack.append(rc)
            continue
        if lc < n:
            if leader == -1:
                n_cluster += 1
            T[lc] = n_cluster
        if rc < n:
            if leader == -1:
                n_cluster += 1
            T[rc] = n_cluster
        if leader == root:
            leader = -1
        stack.pop()
    return T


def fcluster_maxclust(Z: np.ndarray, n: int, t: int) -> np.ndarray:
    """fcluster(Z, t, "maxclust"): the smallest threshold among {-inf, MC[0], ..., MC[n-2]} that leaves at most t flat
    clusters (MC is non-decreasing along the sorted merges, so the count is monotone and the bisection SciPy runs over
    the node indices lands on this threshold value whatever its exact index arithmetic), then the depth-first numbering."""
    if t >= n:                                          # SciPy numbers the n singletons by index in this case
        return np.arange(1, n + 1, dtype=np.int32)      # (never requested by the pooler: t <= n // 2)
    MC = max_dists(Z, n)
    lo, hi = -1, n - 2                                  # invariant: count(thresh[hi]) <= t, thresh[-1] = -inf
    while hi - lo > 0:
        i = (lo + hi) >> 1                              # floor: -1 .. n-3
        thresh = -np.inf if i < 0 else MC[i]
        if _count_clusters(Z, MC, n, thresh, t) > t:
            lo = i + 1
        else:
            hi = i
    return monocrit_labels(Z, MC, n, -np.inf if hi < 0 else MC[hi])


def cluster_labels(distances: np.ndarray, max_clusters: int) -> np.ndarray:
    """labels (0-based) = fcluster(linkage(distances, "ward"), max_clusters, "maxclust") - 1 for an [n, n] matrix."""
    n = distances.shape[0]
    Z = sort_and_label(ward_nn_chain(pdist_rows(distances), n), n)
    return fcluster_maxclust(Z, n, max_clusters) - 1


def pool_single_embedding(embedding: np.ndarray, pool_factor: int) -> Tuple[np.ndarray, Dict[int, np.ndarray]]:
    """hierarchical_token_pooling.py:83-146 on an fp32 [n, dim] array -> (pooled [k, dim] fp32, cluster -> indices)."""
    e = np.asarray(embedding, dtype=np.float32)
    n = e.shape[0]
    if n == 1:
        raise ValueError("The input tensor must have more than one token.")
    if pool_factor == 1:
        return e, {0: np.arange(n)}
    sims = (e.astype(np.float32) @ e.astype(np.float32).T).astype(np.float32)
    distances = (np.float32(1) - sims).astype(np.float32)
    max_clusters = max(n // pool_factor, 1)
    labels = cluster_labels(distances, max_clusters)
    pooled, mapping = [], {}
    for c in range(max_clusters):
        members = np.nonzero(labels == c)[0]
        mapping[c] = members
        if members.size:
            v = e[members].astype(np.float32).mean(axis=0, dtype=np.float32)
            v = v / max(np.float32(np.sqrt(np.sum(v.astype(np.float32) ** 2, dtype=np.float32))), np.float32(1e-12))
            pooled.append(v.astype(np.float32))
    return np.stack(pooled, axis=0), mapping
