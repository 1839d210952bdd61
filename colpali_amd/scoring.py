"""MI355X-native drop-in for the reference's late-interaction scorer.

Mirrors `BaseVisualRetrieverProcessor.score_multi_vector`
(colpali_engine/utils/processing_utils.py:132-187): same signature, argument
meaning, errors and return contract (a new fp32 CPU tensor [n_queries, n_passages]).
All arithmetic runs in the hand-written gfx950 kernels behind the C ABI
(include/maxsim.h); there is no torch/CPU compute path here.
"""
from __future__ import annotations

import logging
from typing import List, Optional, Union

import torch

from . import _lib
from .corpus import PackedCorpus, pack_passages, pack_queries

logger = logging.getLogger(__name__)


def get_torch_device(device: str = "auto") -> str:
    """Same policy as colpali_engine/utils/torch_utils.py:12-31."""
    if device == "auto":
        if torch.cuda.is_available():
            device = "cuda:0"
        elif torch.backends.mps.is_available():
            device = "mps"
        else:
            device = "cpu"
        logger.info(f"Using device: {device}")
    return device


def _require_gpu(device: Union[str, torch.device]) -> torch.device:
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError(
            f"colpali_amd.score_multi_vector runs on an AMD Instinct MI355X only (requested device: {dev}). "
            "There is deliberately no CPU fallback; use the reference scorer for CPU scoring.")
    if not torch.cuda.is_available():
        raise RuntimeError("colpali_amd: no ROCm GPU is visible to torch")
    return dev


def maxsim_scores(queries: torch.Tensor, corpus: PackedCorpus, *, ref_rounding: bool = False,
                  out: Optional[torch.Tensor] = None, ref_bf16: Optional[bool] = None) -> torch.Tensor:
    """Device-level entry: [n_q, Lq, width] device tensor x packed corpus -> fp32 [n_q, n] on the device.

    Asynchronous on torch's current stream.  `ref_rounding=True` reproduces the rounding the reference
    applies when torch evaluates processing_utils.py:179 in the embeddings' own 16-bit dtype
    (`ref_bf16` is the older name of the same switch).
    """
    if ref_bf16 is not None:
        ref_rounding = ref_bf16
    L = _lib.lib()
    if queries.dim() != 3 or not queries.is_contiguous():
        raise ValueError("queries must be a contiguous [n_q, Lq, width] tensor")
    if queries.dtype != corpus.blob.dtype:   # torch.einsum raises on mixed dtypes too
        raise RuntimeError(f"expected queries and passages of one dtype, got {queries.dtype} and {corpus.blob.dtype}")
    dt = _lib.dtype_code(queries.dtype)
    if queries.device != corpus.device:
        raise ValueError("queries and corpus live on different devices")
    n_q, Lq, dim = queries.shape
    if dim != corpus.blob.shape[1]:
        raise RuntimeError(f"queries have embedding width {dim}, the corpus {corpus.blob.shape[1]}")
    n = len(corpus)
    if out is None:
        out = torch.empty((n_q, n), dtype=torch.float32, device=queries.device)
    elif out.shape != (n_q, n) or out.dtype != torch.float32 or out.stride(1) != 1:
        raise ValueError("out must be fp32 [n_q, n] with unit inner stride")
    with torch.cuda.device(queries.device):
        rc = L.msim_fwd(dt, _lib.ptr(queries), n_q, Lq, _lib.ptr(corpus.blob), _lib.ptr(corpus.offsets),
                        _lib.ptr(corpus.clamp0), n, dim, _lib.ptr(out), out.stride(0) if n_q > 1 else max(n, 1),
                        _lib.MSIM_FLAG_REF_ROUNDING if ref_rounding else 0, None,
                        _lib.current_stream_handle(queries.device))
    _lib.check(rc, "msim_fwd")
    return out


def score_multi_vector(
    qs: Union[torch.Tensor, List[torch.Tensor]],
    ps: Union[torch.Tensor, List[torch.Tensor]],
    batch_size: int = 128,
    device: Optional[Union[str, torch.device]] = None,
) -> torch.Tensor:
    """Late-interaction / MaxSim scores, drop-in for processing_utils.py:132-187.

    `qs` / `ps`: a list of [len_i, dim] tensors or one padded [n, max_len, dim] tensor.
    `batch_size` keeps its reference meaning for the one thing it changes in the
    reference's *results*: which passages are zero-padded together (a passage shorter
    than the longest of its block also sees similarity 0 in every per-token max).
    Returns a new fp32 tensor [n_queries, n_passages] on the CPU.
    """
    device = device or get_torch_device("auto")
    if len(qs) == 0:
        raise ValueError("No queries provided")
    if len(ps) == 0:
        raise ValueError("No passages provided")
    dev = _require_gpu(device)
    q = pack_queries(qs, dev)
    corpus = pack_passages(ps, dev, batch_size=batch_size)
    scores = maxsim_scores(q, corpus).cpu()
    assert scores.shape[0] == len(qs), f"Expected {len(qs)} scores, got {scores.shape[0]}"
    return scores.to(torch.float32)
