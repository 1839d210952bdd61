"""Embedding head on the MI355X: the last lines of every Col* model forward, fused, writing the scorer's corpus format.

Reference lines (identical in every model family):
    colpali_engine/models/paligemma/colpali/modeling_colpali.py:67-77
    colpali_engine/models/qwen2/colqwen2/modeling_colqwen2.py:65-74
        proj = self.custom_text_proj(hidden_states)
        proj = proj / proj.norm(dim=-1, keepdim=True)
        proj = proj * attention_mask.unsqueeze(-1)
        [proj = proj * image_mask]                         # mask_non_image_embeddings
and the road from there to the scorer (README.md:121-126: `torch.unbind(embeddings.to("cpu"))`, later
`pad_sequence` + H2D per block inside score_multi_vector, processing_utils.py:172-178).

`embedding_head`  -- dense drop-in for those lines: same [B, S, 128] tensor.
`CorpusWriter`    -- the MI355X-native road: rows go straight from the model's hidden states into the resident
                     packed corpus the MaxSim kernels stream; masked positions are dropped and replaced by the
                     per-document clamp0 flag (a zero row and "similarity 0 takes part in the max" are the same
                     thing), so the corpus holds real patches only and never leaves the GPU.
All arithmetic runs in colpali_amd/csrc/embed_head.hip behind msim_embed_head (include/maxsim.h); no torch fallback.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from .corpus import PackedCorpus, block_clamp0

HEAD_DIM = 128
_TILE = 256


def _check(hidden: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], mask: torch.Tensor) -> None:
    if hidden.dim() != 3:
        raise ValueError("hidden_states must be [batch, sequence, hidden]")
    if hidden.device.type != "cuda":
        raise RuntimeError("colpali_amd.embedding_head runs on an MI355X only (no CPU fallback)")
    if hidden.dtype not in (torch.bfloat16, torch.float16):
        raise NotImplementedError(f"embedding head: hidden states of dtype {hidden.dtype}; bf16 / fp16 only")
    if weight.shape != (HEAD_DIM, hidden.shape[2]) or weight.dtype != hidden.dtype or weight.device != hidden.device:
        raise ValueError(f"weight must be [{HEAD_DIM}, hidden] with the dtype/device of the hidden states")
    if bias is not None and (bias.shape != (HEAD_DIM,) or bias.dtype != hidden.dtype or bias.device != hidden.device):
        raise ValueError(f"bias must be [{HEAD_DIM}] with the dtype/device of the hidden states")
    if hidden.shape[2] % 64 != 0:
        raise NotImplementedError(f"embedding head: hidden size {hidden.shape[2]} is not a multiple of 64")
    if mask.shape != hidden.shape[:2]:
        raise ValueError("attention_mask must be [batch, sequence]")


def _launch(hidden: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], row_map: torch.Tensor,
            out: torch.Tensor) -> None:
    L = _lib.lib()
    x = hidden.contiguous()
    M, H = x.shape[0] * x.shape[1], x.shape[2]
    with torch.cuda.device(x.device):
        rc = L.msim_embed_head(_lib.dtype_code(x.dtype), _lib.ptr(x), M, H, _lib.ptr(weight.contiguous()),
                               _lib.ptr(bias.contiguous() if bias is not None else None), HEAD_DIM, _lib.ptr(row_map),
                               _lib.ptr(out), out.stride(0), _lib.current_stream_handle(x.device))
    _lib.check(rc, "msim_embed_head")


_MASK_KINDS = {torch.bool: 0, torch.uint8: 0, torch.int8: 0, torch.int16: 1, torch.int32: 2, torch.int64: 3, torch.float32: 4,
               torch.bfloat16: 5, torch.float16: 6}


def _prep_mask(t: torch.Tensor, device) -> torch.Tensor:
    """A mask as the kernels read it: flat, contiguous, on the device, of a dtype they know (anything else is compared with 0 first)."""
    t = t.reshape(-1)
    if t.dtype not in _MASK_KINDS:
        t = t != 0
    if t.device != torch.device(device) or not t.is_contiguous():
        t = t.to(device).contiguous()
    return t


def _dense_row_map(attention_mask: torch.Tensor, extra_mask: Optional[torch.Tensor], M: int, device) -> torch.Tensor:
    """int32 row map of the dense output (kept rows stay in place, masked positions become zero rows), tile padding included:
    one launch of msim_embed_head_row_map whatever the masks' dtypes (modeling_colpali.py:72, :74-77)."""
    mask = _prep_mask(attention_mask, device)
    extra = None if extra_mask is None else _prep_mask(extra_mask, device)
    if mask.numel() != M or (extra is not None and extra.numel() != M):
        raise ValueError("attention_mask / extra_mask must have one entry per hidden-state position")
    row_map = torch.empty(((M + _TILE - 1) // _TILE * _TILE,), dtype=torch.int32, device=device)
    L = _lib.lib()
    with torch.cuda.device(device):
        rc = L.msim_embed_head_row_map(_lib.ptr(mask), _MASK_KINDS[mask.dtype], _lib.ptr(extra),
                                       _MASK_KINDS[extra.dtype] if extra is not None else 0, M, _lib.ptr(row_map),
                                       _lib.current_stream_handle(torch.device(device)))
    _lib.check(rc, "msim_embed_head_row_map")
    return row_map


class _EmbeddingHeadFn(torch.autograd.Function):
    """The fused head inside a training graph (modeling_colpali.py:65-78 is part of the graph the reference trainers
    back-propagate through, trainer/contrastive_trainer.py:135-162).  Forward = the fused kernel.  Backward: the Linear
    output is recomputed by a library GEMM, the norm / mask Jacobian runs in `msim_embed_head_bwd` (fp32 inside, one
    rounding), and dX = dproj W, dW = dproj^T X, db = sum dproj are library GEMMs / a reduction again -- the same three
    products torch's own autograd runs for nn.Linear."""

    @staticmethod
    def forward(ctx, hidden, weight, bias, row_map):
        B, S, _ = hidden.shape
        out = torch.empty((B * S, HEAD_DIM), dtype=hidden.dtype, device=hidden.device)
        _launch(hidden, weight, bias, row_map, out)
        ctx.save_for_backward(hidden, weight, bias, row_map)
        return out.view(B, S, HEAD_DIM)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        hidden, weight, bias, row_map = ctx.saved_tensors
        B, S, H = hidden.shape
        M = B * S
        x2 = hidden.reshape(M, H)
        g = grad_out.reshape(M, HEAD_DIM).to(hidden.dtype).contiguous()
        with torch.autocast("cuda", enabled=False):
            proj = torch.nn.functional.linear(x2, weight, bias)                  # :67 again: a plain library GEMM
            dproj = torch.empty_like(proj)
            L = _lib.lib()
            with torch.cuda.device(hidden.device):
                rc = L.msim_embed_head_bwd(_lib.dtype_code(hidden.dtype), _lib.ptr(proj), _lib.ptr(g), _lib.ptr(row_map), M,
                                           HEAD_DIM, _lib.ptr(dproj), _lib.current_stream_handle(hidden.device))
            _lib.check(rc, "msim_embed_head_bwd")
            d_hidden = (dproj @ weight).view(B, S, H) if ctx.needs_input_grad[0] else None
            d_weight = dproj.t() @ x2 if ctx.needs_input_grad[1] else None
            d_bias = dproj.sum(dim=0) if bias is not None and ctx.needs_input_grad[2] else None
        return d_hidden, d_weight, d_bias, None


def embedding_head(hidden_states: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                   attention_mask: torch.Tensor, extra_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Dense drop-in for modeling_colpali.py:67-77: [B, S, hidden] -> [B, S, 128] (unit rows, masked rows zero).

    `extra_mask` is the optional image-token mask of `mask_non_image_embeddings` ([B, S] or [B, S, 1]).
    Differentiable: when autograd is recording and the hidden states, the weight or the bias require a gradient, the
    result carries a graph node whose backward is `_EmbeddingHeadFn.backward` (never a silent detach)."""
    _check(hidden_states, weight, bias, attention_mask)
    B, S, _ = hidden_states.shape
    row_map = _dense_row_map(attention_mask, extra_mask, B * S, hidden_states.device)
    if torch.is_grad_enabled() and (hidden_states.requires_grad or weight.requires_grad
                                    or (bias is not None and bias.requires_grad)):
        return _EmbeddingHeadFn.apply(hidden_states, weight, bias, row_map)
    out = torch.empty((B * S, HEAD_DIM), dtype=hidden_states.dtype, device=hidden_states.device)
    _launch(hidden_states, weight, bias, row_map, out)
    return out.view(B, S, HEAD_DIM)


class CorpusWriter:
    """Device-resident packed corpus under construction: `append` runs the fused head on a batch of hidden states and
    writes the unmasked rows of every page back to back; `finish` returns the PackedCorpus the scorer streams.

    Scores of the finished corpus equal what the reference computes from `list(torch.unbind(model(**batch)))`
    (README.md:121-126) with the same `batch_size` blocking: a page's masked positions are zero rows there, here they
    are dropped and the page carries the clamp0 flag; pages of different padded lengths sharing a scorer block get the
    flag through the same `block_clamp0` rule as pack_passages.
    """

    def __init__(self, capacity_rows: int, device, dtype: torch.dtype = torch.bfloat16, score_batch_size: Optional[int] = 128):
        self.device = torch.device(device)
        self.blob = torch.empty((max(capacity_rows, 1), HEAD_DIM), dtype=dtype, device=self.device)
        self.capacity = capacity_rows
        self.score_batch_size = score_batch_size
        self._rows_dev = torch.zeros((), dtype=torch.int64, device=self.device)   # exact count, stays on the device
        self._rows_upper = 0                                                       # host-side bound (no sync per append)
        self._counts, self._padded_lens = [], []

    def append(self, hidden_states: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
               attention_mask: torch.Tensor, extra_mask: Optional[torch.Tensor] = None) -> int:
        """Returns the number of pages appended.  Asynchronous: no host synchronisation.
        Indexing only: the rows are written into the resident blob, outside any autograd graph -- inputs that are being
        recorded for a gradient are refused instead of silently detached (call it under torch.no_grad(), as README.md:121
        does for inference, or use `embedding_head`, which is differentiable)."""
        _check(hidden_states, weight, bias, attention_mask)
        if torch.is_grad_enabled() and (hidden_states.requires_grad or weight.requires_grad
                                        or (bias is not None and bias.requires_grad)):
            raise RuntimeError("CorpusWriter.append writes into the resident corpus and has no backward: call it under "
                               "torch.no_grad() (indexing), or use colpali_amd.embedding_head inside a training graph")
        if hidden_states.dtype != self.blob.dtype or hidden_states.device != self.device:
            raise ValueError("hidden states must have the writer's dtype and device")
        B, S, _ = hidden_states.shape
        if B == 0 or S == 0:        # nothing to write: the row counter must not be replaced by a buffer no kernel filled
            if S == 0 and B > 0:
                self._counts.append(torch.zeros((B,), dtype=torch.int64, device=self.device))
                self._padded_lens.append(torch.zeros((B,), dtype=torch.int64))
            return B
        if self._rows_upper + B * S > self.capacity:
            # the bound counts masked positions too (no sync per append); reconcile it with the exact device-side count once
            # before refusing -- a rejected append leaves the writer unchanged
            self._rows_upper = self.rows_written()
            if self._rows_upper + B * S > self.capacity:
                raise RuntimeError(f"CorpusWriter capacity of {self.capacity} rows exceeded: {self._rows_upper} rows written, "
                                   f"the batch may add up to {B * S}")
        self._rows_upper += B * S
        # destination row of every kept position, the page counts and the new row total: two launches of msim_embed_head_writer_map
        # (the host used to assemble them from ten torch launches per batch)
        mask = _prep_mask(attention_mask, self.device)
        extra = None if extra_mask is None else _prep_mask(extra_mask, self.device)
        if mask.numel() != B * S or (extra is not None and extra.numel() != B * S):
            raise ValueError("attention_mask / extra_mask must have one entry per hidden-state position")
        counts = torch.empty((B,), dtype=torch.int64, device=self.device)
        row_map = torch.empty(((B * S + _TILE - 1) // _TILE * _TILE,), dtype=torch.int32, device=self.device)
        rows_after = torch.empty((), dtype=torch.int64, device=self.device)
        L = _lib.lib()
        with torch.cuda.device(self.device):
            rc = L.msim_embed_head_writer_map(_lib.ptr(mask), _MASK_KINDS[mask.dtype], _lib.ptr(extra),
                                              _MASK_KINDS[extra.dtype] if extra is not None else 0, B, S, _lib.ptr(self._rows_dev),
                                              _lib.ptr(counts), _lib.ptr(row_map), _lib.ptr(rows_after),
                                              _lib.current_stream_handle(self.device))
        _lib.check(rc, "msim_embed_head_writer_map")
        _launch(hidden_states, weight, bias, row_map, self.blob)
        self._rows_dev = rows_after
        self._counts.append(counts)
        self._padded_lens.append(torch.full((B,), S, dtype=torch.int64))
        return B

    def rows_written(self) -> int:
        """Exact number of rows written so far (one host synchronisation); also tightens the host-side bound."""
        self._rows_upper = int(self._rows_dev.item())
        return self._rows_upper

    def finish(self, id_base: int = 0) -> PackedCorpus:
        """One host synchronisation (the page lengths are needed on the host for the block flags)."""
        counts = torch.cat(self._counts)
        lengths = counts.cpu()
        padded = torch.cat(self._padded_lens)
        offsets = torch.zeros(lengths.numel() + 1, dtype=torch.int64)
        torch.cumsum(lengths, 0, out=offsets[1:])
        flags = (lengths < padded).to(torch.uint8)                                  # page had masked positions
        if self.score_batch_size is not None:
            flags |= block_clamp0(padded, self.score_batch_size)                   # shorter than its scorer block
        total = int(offsets[-1])
        return PackedCorpus(blob=self.blob[: max(total, 1)], offsets=offsets.to(torch.int32).to(self.device),
                            clamp0=flags.to(self.device) if bool(flags.any()) else None, lengths=lengths, id_base=id_base)
