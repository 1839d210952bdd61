"""Sharded-corpus retrieval: fused MaxSim per shard, deterministic top-k, RCCL merge.

The reference has no sharded retrieval (its scorer is single-device and returns the full
[n_q, n_p] matrix on the CPU, colpali_engine/utils/processing_utils.py:180-186; its only top-k
API is the experimental get_topk_plaid, :189-219).  This module is the MI355X-native piece
BASELINE.json config 4 asks for: the pre-embedded corpus is sharded by contiguous id ranges, one
process per GPU; each rank scores its resident shard and keeps its best k per query; ONE
all-gather of [n_q, k] (score f32, id i64) over RCCL/xGMI and a k-way merge give every rank the
global top-k.  The order is total -- (score descending, id ascending) -- so the answer does not
depend on the number of shards.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch

from . import _lib
from .corpus import PackedCorpus
from .corpus import PackedQueries, pack_queries
from .scoring import maxsim_scores


def shard_range(n_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous id range [lo, hi) of `rank`; the first n_total % world ranks hold one more."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    q, r = divmod(n_total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def topk(scores: torch.Tensor, k: int, id_base: int = 0, ids: Optional[torch.Tensor] = None,
         out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Row-wise top-k on the GPU, ordered by (score desc, id asc); pads with (-inf, -1).

    scores: fp32 [n_q, n] (device). ids: optional int64 [n_q, n] candidate ids (default id_base + column).
    out: optional preallocated contiguous (fp32 [n_q, k], int64 [n_q, k]) to write into.
    """
    L = _lib.lib()
    if scores.dim() != 2 or scores.dtype != torch.float32 or scores.device.type != "cuda":
        raise ValueError("scores must be a 2-D fp32 tensor on the GPU")
    if scores.stride(1) != 1 and scores.shape[1] > 1:
        scores = scores.contiguous()
    n_q, n = scores.shape
    ld = scores.stride(0) if n_q > 1 else max(n, 1)
    if ids is not None:
        if ids.shape != scores.shape or ids.dtype != torch.int64 or ids.device != scores.device:
            raise ValueError("ids must be int64 with the shape/device of scores")
        if not ids.is_contiguous() or ld != max(n, 1):
            ids = ids.contiguous()
            scores = scores.contiguous()
            ld = max(n, 1)
    dev = scores.device
    if out is not None:
        out_s, out_i = out
        if (out_s.shape != (n_q, k) or out_i.shape != (n_q, k) or out_s.dtype != torch.float32 or out_i.dtype != torch.int64
                or not out_s.is_contiguous() or not out_i.is_contiguous() or out_s.device != dev or out_i.device != dev):
            raise ValueError("out must be contiguous (fp32 [n_q, k], int64 [n_q, k]) tensors on the scores' device")
    else:
        out_s = torch.empty((n_q, k), dtype=torch.float32, device=dev)
        out_i = torch.empty((n_q, k), dtype=torch.int64, device=dev)
    ws_bytes = L.msim_topk_workspace_bytes(n_q, n, k)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev) if ws_bytes else None
    with torch.cuda.device(dev):
        rc = L.msim_topk_f32(_lib.ptr(scores), _lib.ptr(ids), n_q, n, ld, k, id_base, _lib.ptr(out_s), _lib.ptr(out_i),
                             _lib.ptr(ws), _lib.current_stream_handle(dev))
    _lib.check(rc, "msim_topk_f32")
    return out_s, out_i


def merge_gathered(all_s: torch.Tensor, all_i: torch.Tensor, k: int,
                   select: Callable = topk) -> Tuple[torch.Tensor, torch.Tensor]:
    """[world, n_q, k] gathered candidates -> global [n_q, k]."""
    world, n_q, kk = all_s.shape
    cand_s = all_s.permute(1, 0, 2).reshape(n_q, world * kk).contiguous()
    cand_i = all_i.permute(1, 0, 2).reshape(n_q, world * kk).contiguous()
    return select(cand_s, k, 0, cand_i)


def shard_topk(scores: torch.Tensor, k: int, id_base: int, world: int = 1, dist=None, group=None,
               select: Callable = topk, force_collective: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-shard top-k of a local score matrix, then (world > 1) all-gather + merge.

    `scores` [n_q, n_local]: column j is document id_base + j.  Returns the same global
    (scores [n_q, k], ids [n_q, k]) on every rank.  `force_collective=True` sends a single rank through the
    message packing, the all-gather (RCCL under the `nccl` backend) and the strided-view merge as well: the
    multi-GPU code path, exercised on the one GPU a test box has (the result is the same by construction).
    """
    if world <= 1 and not force_collective:
        return select(scores, k, id_base, None)
    world = max(world, 1)
    if dist is None:
        import torch.distributed as dist  # noqa: PLW0642 - the default collective library
    # one message per rank: [scores fp32 n_q*k | pad to 8 | ids int64 n_q*k] -- ONE all-gather of 12 bytes per candidate
    n_q = scores.shape[0]
    sb = n_q * k * 4
    sbp = (sb + 7) // 8 * 8
    nbytes = sbp + n_q * k * 8
    mine = torch.empty((nbytes,), dtype=torch.uint8, device=scores.device)
    my_s = mine[:sb].view(torch.float32).view(n_q, k)
    my_i = mine[sbp:].view(torch.int64).view(n_q, k)
    if select is topk:
        topk(scores, k, id_base, None, out=(my_s, my_i))                    # the selection kernel writes the message in place
    else:
        loc_s, loc_i = select(scores, k, id_base, None)
        my_s.copy_(loc_s)
        my_i.copy_(loc_i)
    flat = torch.empty((world * nbytes,), dtype=torch.uint8, device=scores.device)     # rank-major concatenation
    dist.all_gather_into_tensor(flat, mine, group=group)                   # RCCL all-gather over xGMI (nccl backend)
    gathered = flat.view(world, nbytes)
    all_s = gathered[:, :sb].view(torch.float32).view(world, n_q, k)
    all_i = gathered[:, sbp:].view(torch.int64).view(world, n_q, k)
    return merge_gathered(all_s, all_i, k, select)


class ShardedRetriever:
    """One instance per process/GPU; holds this rank's resident shard of the corpus."""

    def __init__(self, shard: PackedCorpus, world: int = 1, rank: int = 0, dist=None, group=None,
                 score_fn: Callable = maxsim_scores, select: Callable = topk, force_collective: bool = False):
        self.shard, self.world, self.rank = shard, world, rank
        self.dist, self.group = dist, group
        self._score, self._select = score_fn, select
        self.force_collective = force_collective
        if world > 1 and dist is None:
            import torch.distributed as dist_mod

            self.dist = dist_mod

    def search(self, queries, k: int = 10, compact: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        """queries (replicated on every rank): a `PackedQueries`, a list of [len_i, 128] tensors, or a [n_q, Lq, 128] tensor.
        A host list is packed into the flat layout (ragged lengths, zero rows dropped on the way into the staging buffer).  A dense
        DEVICE tensor is scored as it stands unless `compact=True`: dropping its zero padding rows needs the per-query counts on the
        host (one small D2H + a synchronisation), which would make a call that is otherwise fully asynchronous and hipGraph-capturable
        block the host (round-4 advisor finding).  Callers with heavily padded query boxes pass `compact=True`, or pack once with
        `pack_queries` and hand the `PackedQueries` over; the scores are the same either way (a zero row adds exactly 0)."""
        if self._score is maxsim_scores and not isinstance(queries, PackedQueries):
            dense_on_device = isinstance(queries, torch.Tensor) and queries.device.type == "cuda"
            queries = pack_queries(queries, self.shard.device, compact=compact or not dense_on_device)
        scores = self._score(queries, self.shard)
        return shard_topk(scores, k, self.shard.id_base, self.world, self.dist, self.group, self._select,
                          force_collective=self.force_collective)


class ExactMaxSimIndex:
    """What `create_plaid_index` returns here: the resident packed corpus behind the interface the reference's
    `get_topk_plaid` talks to -- `index.search(queries_embeddings=[n, Lq, dim], top_k=k)` (processing_utils.py:217-220).
    The reference delegates that call to the third-party `fast_plaid.search.FastPlaid` (not vendored in the reference
    checkout, unpinned: README.md:108-111 `pip install --no-deps fast-plaid fastkmeans`), an APPROXIMATE centroid-pruned
    index; this one is exact: fused MaxSim over every page + the deterministic (score desc, id asc) top-k.  `search`
    returns FastPlaid's published result shape: per query a list of (document id, score) tuples, best first."""

    def __init__(self, corpus: PackedCorpus, world: int = 1, rank: int = 0, dist=None, group=None):
        self.retriever = ShardedRetriever(corpus, world, rank, dist, group)

    def search(self, queries_embeddings: torch.Tensor, top_k: int = 10):
        dev = self.retriever.shard.device
        q = queries_embeddings.to(device=dev, dtype=self.retriever.shard.blob.dtype).contiguous()
        if q.dim() != 3:
            raise ValueError("queries_embeddings must be [n_queries, query_length, dim]")
        top_s, top_i = self.retriever.search(q, k=top_k, compact=True)     # padded blocks from get_topk_plaid: the result is read back anyway
        top_s, top_i = top_s.cpu().tolist(), top_i.cpu().tolist()
        return [[(int(i), float(s)) for s, i in zip(row_s, row_i) if i >= 0] for row_s, row_i in zip(top_s, top_i)]


def create_plaid_index(ps, device=None) -> ExactMaxSimIndex:
    """Drop-in for `BaseVisualRetrieverProcessor.create_plaid_index` (processing_utils.py:226-244): same arguments; builds
    the resident packed corpus instead of a FastPlaid index (see ExactMaxSimIndex).  Like the reference -- which hands
    FastPlaid the unpadded pages -- no block zero-padding semantics apply here (`batch_size=None`)."""
    from .corpus import pack_passages
    from .scoring import _require_gpu, get_torch_device

    if len(ps) == 0:
        raise ValueError("No passages provided")
    dev = _require_gpu(device or get_torch_device("auto"))
    return ExactMaxSimIndex(pack_passages(list(ps) if not isinstance(ps, torch.Tensor) else ps, dev, batch_size=None))


def get_topk_plaid(qs, plaid_index, k: int = 10, batch_size: int = 128, device=None):
    """Drop-in for `BaseVisualRetrieverProcessor.get_topk_plaid` (processing_utils.py:189-223): the same loop over
    blocks of `batch_size` queries, `pad_sequence(padding_value=0)` per block, one `plaid_index.search(...)` per block,
    and the same return value: the list of per-block results."""
    from .scoring import get_torch_device

    device = device or get_torch_device("auto")
    if len(qs) == 0:
        raise ValueError("No queries provided")
    scores_list = []
    for i in range(0, len(qs), batch_size):
        block = qs[i: i + batch_size]
        qs_batch = torch.nn.utils.rnn.pad_sequence(list(block), batch_first=True, padding_value=0).to(device)
        scores_list.append(plaid_index.search(queries_embeddings=qs_batch, top_k=k))
    return scores_list
