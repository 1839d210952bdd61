"""Hierarchical token pooling on the MI355X: drop-in for the reference's CPU/SciPy pooler.

Mirrors colpali_engine/compression/token_pooling/base_token_pooling.py (`TokenPoolingOutput`, `pool_embeddings`
:106-167 with its validation / unbind / re-pad logic) and hierarchical_token_pooling.py:12-146
(`HierarchicalTokenPooler`): Ward clustering of a page's patch embeddings on the rows of 1 - E E^T, at most
`token_length // pool_factor` clusters, each replaced by the normalised mean of its members.  The reference does this
page by page with torch.mm + scipy.cluster.hierarchy on the host; here a whole batch of pages is clustered on the GPU
(colpali_amd/csrc/token_pooling.hip behind msim_pool_cluster / msim_pool_reduce).  Pooled pages are what shrinks the
bytes the MaxSim scorer has to stream (pool_factor 3: a third of the corpus).  No CPU fallback.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple, Union, cast

import torch

from . import _lib
from .scoring import get_torch_device

_MAX_ROWS = 32768                   # msim_pool_cluster's cap (a page above 2048 rows keeps its clustering state in HBM instead of LDS)
_WS_BUDGET_BYTES = 4 << 30          # fp32 + fp64 [n, n] workspaces of the pages clustered in one launch


@dataclass
class TokenPoolingOutput:
    """Same fields as the reference's dataclass (base_token_pooling.py:9-23)."""

    pooled_embeddings: Union[List[torch.Tensor], torch.Tensor]
    cluster_id_to_indices: Optional[List[Dict[int, Tuple[torch.Tensor]]]] = None


def _unbind_padded(embeddings: torch.Tensor, padding_value: float, padding_side: str) -> List[torch.Tensor]:
    """colpali_engine/utils/torch_utils.py:57-99."""
    out: List[torch.Tensor] = []
    for seq in embeddings:
        valid = (~torch.all(seq.eq(padding_value), dim=-1)).nonzero(as_tuple=False)
        if padding_side == "left":
            out.append(seq[:0] if valid.numel() == 0 else seq[int(valid[0].item()):])
        elif padding_side == "right":
            out.append(seq[:0] if valid.numel() == 0 else seq[: int(valid[-1].item()) + 1])
        else:
            raise ValueError("padding_side must be either 'left' or 'right'.")
    return out


def cluster_pages(blob: torch.Tensor, offsets: torch.Tensor, lengths: torch.Tensor, pool_factor: int):
    """Device-level entry.  blob [rows, width] (bf16 | f16 | f32, width = kernel width) packed pages on the GPU, offsets
    int32 [n+1] on the GPU, lengths int64 [n] on the host.  Returns (labels int32 [rows], n_clusters int32 [n]) on the GPU."""
    L = _lib.lib()
    dev = blob.device
    n_pages = int(lengths.numel())
    labels = torch.empty((max(int(blob.shape[0]), 1),), dtype=torch.int32, device=dev)
    n_clusters = torch.zeros((max(n_pages, 1),), dtype=torch.int32, device=dev)
    if n_pages == 0:
        return labels[:0], n_clusters[:0]
    if int(lengths.max()) > _MAX_ROWS:
        raise NotImplementedError(f"token pooling: a page of {int(lengths.max())} rows; at most {_MAX_ROWS} are supported "
                                  "(the [n, n] float64 distance matrix of such a page is above 8 GiB)")
    sq = lengths.to(torch.int64) ** 2
    start = 0
    offs_host = torch.zeros(n_pages + 1, dtype=torch.int64)
    torch.cumsum(lengths.to(torch.int64), 0, out=offs_host[1:])
    while start < n_pages:                                   # chunks of pages whose [n, n] workspaces fit the budget
        end, acc = start, 0
        while end < n_pages and (end == start or (acc + int(sq[end])) * 12 <= _WS_BUDGET_BYTES) and end - start < 65535:
            acc += int(sq[end])
            end += 1
        ws_off = torch.zeros(end - start + 1, dtype=torch.int64)
        torch.cumsum(sq[start:end], 0, out=ws_off[1:])
        x_ws = torch.empty((max(acc, 1),), dtype=torch.float32, device=dev)
        d_ws = torch.empty((max(acc, 1),), dtype=torch.float64, device=dev)
        ws_dev = ws_off.to(dev)
        with torch.cuda.device(dev):
            rc = L.msim_pool_cluster(_lib.dtype_code(blob.dtype), _lib.ptr(blob), offsets[start:].data_ptr(), end - start,
                                     blob.shape[1], int(lengths[start:end].max()), _lib.ptr(ws_dev), pool_factor,
                                     _lib.ptr(x_ws), _lib.ptr(d_ws), _lib.ptr(labels), n_clusters[start:].data_ptr(),
                                     _lib.current_stream_handle(dev))
        _lib.check(rc, "msim_pool_cluster")
        del x_ws, d_ws                                       # stream-ordered: the allocator reuses them after the kernels
        start = end
    return labels[: int(blob.shape[0])], n_clusters[:n_pages]


class HierarchicalTokenPooler:
    """Drop-in for colpali_engine.compression.token_pooling.HierarchicalTokenPooler (same methods, arguments, errors and
    return structures); `num_workers` is accepted and ignored -- the pages of a call are clustered concurrently on the GPU."""

    def __init__(self, device: Optional[Union[str, torch.device]] = None):
        self._device = device

    # ---- base_token_pooling.py:49-104
    def _validate_embeddings(self, embeddings) -> None:
        if isinstance(embeddings, list) and not embeddings:
            raise ValueError("Empty embeddings list provided")
        is_list_of_2d = isinstance(embeddings, list) and embeddings[0].dim() == 2
        is_3d = isinstance(embeddings, torch.Tensor) and embeddings.dim() == 3
        if not is_list_of_2d and not is_3d:
            raise ValueError("The input tensor must be a list of 2D tensors or a 3D tensor.")

    def _prepare_embeddings(self, embeddings, padding: bool = False, padding_side: str = "left") -> List[torch.Tensor]:
        if isinstance(embeddings, torch.Tensor) and embeddings.dim() == 3:
            if padding:
                return _unbind_padded(embeddings, 0.0, padding_side)
            return list(embeddings.unbind(dim=0))
        return cast(List[torch.Tensor], embeddings)

    # ---- hierarchical_token_pooling.py:38-146
    def _pool_embeddings_impl(self, embeddings: List[torch.Tensor], pool_factor: int, num_workers: Optional[int] = None,
                              want_maps: bool = True):
        if not (num_workers is None or num_workers >= 1):
            raise ValueError(f"Invalid number of workers: {num_workers}")
        for e in embeddings:
            if e.dim() != 2:
                raise ValueError("The input tensor must be a 2D tensor.")
            if e.size(0) == 1:
                raise ValueError("The input tensor must have more than one token.")
        if pool_factor == 1:
            return list(embeddings), [{0: (torch.arange(e.size(0)),)} for e in embeddings]
        dev = torch.device(self._device or get_torch_device("auto"))
        if dev.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError("colpali_amd.HierarchicalTokenPooler runs on an MI355X only (no CPU fallback)")
        dtype, dim = embeddings[0].dtype, embeddings[0].size(1)
        if dtype not in (torch.bfloat16, torch.float16, torch.float32):
            raise NotImplementedError(f"token pooling: dtype {dtype}; bf16 / fp16 / fp32 only")
        if any(e.dtype != dtype or e.size(1) != dim for e in embeddings):
            raise RuntimeError("expected embeddings of one dtype and width")
        width = _lib.kernel_width(dim, dtype)
        lengths = torch.tensor([e.size(0) for e in embeddings], dtype=torch.int64)
        blob = torch.cat([e.to(dev) for e in embeddings], dim=0)
        if width != dim:
            blob = torch.nn.functional.pad(blob, (0, width - dim))
        blob = blob.contiguous()
        offs = torch.zeros(lengths.numel() + 1, dtype=torch.int64)
        torch.cumsum(lengths, 0, out=offs[1:])
        offsets = offs.to(torch.int32).to(dev)
        labels, n_clusters = cluster_pages(blob, offsets, lengths, pool_factor)
        counts = n_clusters.cpu().to(torch.int64)                       # one synchronisation: output sizes
        out_off = torch.zeros(lengths.numel() + 1, dtype=torch.int64)
        torch.cumsum(counts, 0, out=out_off[1:])
        pooled = torch.empty((max(int(out_off[-1]), 1), dim), dtype=dtype, device=dev)
        L = _lib.lib()
        with torch.cuda.device(dev):
            rc = L.msim_pool_reduce(_lib.dtype_code(dtype), _lib.ptr(blob), _lib.ptr(offsets), int(lengths.numel()), dim, width,
                                    _lib.ptr(labels), _lib.ptr(out_off.to(torch.int32).to(dev)), _lib.ptr(pooled), dim,
                                    _lib.current_stream_handle(dev))
        _lib.check(rc, "msim_pool_reduce")
        out_off_l, offs_l = out_off.tolist(), offs.tolist()
        pooled_list = [pooled[out_off_l[i] : out_off_l[i + 1]].to(e.device) for i, e in enumerate(embeddings)]
        if not want_maps:
            return pooled_list, None
        # cluster id -> token indices (ascending), including the ids that stayed empty (:124-131): one stable sort per page
        labels_host = labels.cpu()
        maps = []
        for i, e in enumerate(embeddings):
            lab = labels_host[offs_l[i] : offs_l[i + 1]].to(torch.int64)
            max_clusters = max(e.size(0) // pool_factor, 1)
            order = torch.sort(lab, stable=True).indices
            parts = order.split(torch.bincount(lab, minlength=max_clusters).tolist())
            maps.append({c: (parts[c],) for c in range(max_clusters)})
        return pooled_list, maps

    # ---- base_token_pooling.py:106-167
    def pool_embeddings(self, embeddings, return_dict: bool = False, padding: bool = False, padding_side: str = "left",
                        num_workers: Optional[int] = None, **pool_kwargs):
        if isinstance(embeddings, list) and not embeddings:
            return TokenPoolingOutput(pooled_embeddings=[], cluster_id_to_indices=[])
        self._validate_embeddings(embeddings)
        prepared = self._prepare_embeddings(embeddings, padding, padding_side)
        pooled, mapping = self._pool_embeddings_impl(prepared, num_workers=num_workers, want_maps=return_dict, **pool_kwargs)
        if isinstance(embeddings, torch.Tensor) and embeddings.dim() == 3:
            pooled = torch.nn.utils.rnn.pad_sequence(pooled, batch_first=True, padding_value=0.0, padding_side=padding_side)
        if not return_dict:
            return pooled
        return TokenPoolingOutput(pooled_embeddings=pooled, cluster_id_to_indices=mapping)
