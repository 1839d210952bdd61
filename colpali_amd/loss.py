"""MI355X-native drop-ins for the reference's late-interaction (ColBERT) losses.

Mirrors colpali_engine/loss/late_interaction_losses.py: `ColbertModule` (:6-107),
`ColbertLoss` (:110-164), `ColbertNegativeCELoss` (:167-252), `ColbertPairwiseCELoss` (:255-313),
`ColbertPairwiseNegativeCELoss` (:316-398), `ColbertSigmoidLoss` (:401-465)
-- same constructor arguments, same forward signature, no persistent state (checkpoints stay
interchangeable).  The MaxSim core every one of them starts with,

    raw = einsum("bnd,csd->bcns", Q, D); scores = raw.amax(dim=3).sum(dim=2)      (:297-298, :91)

runs in the fused gfx950 kernels (no [B,C,Lq,Ld] tensor is materialised, forward or backward).
For the two in-batch losses (`ColbertPairwiseCELoss`, `ColbertLoss`) the [B, C]-sized epilogue
(lengths, normalisation, filtering, diagonal / topk(2) / where / softplus resp. cross entropy,
:296, :300-313, :164) AND its gradient are one more launch (msim_loss_epilogue): a training step
has no host synchronisation and captures as one hipGraph.  The backward recomputes the per-token
arg-max only for the (query, doc) pairs that carry a gradient -- the two per query the epilogue
emits for the pairwise loss -- instead of keeping the similarity tensor alive.  The sigmoid and
explicit-negative losses keep their (tiny) epilogues in torch on top of the same fused cores; the sigmoid loss has its own epilogue
mode since round 6.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F  # noqa: N812

from . import _lib
from .corpus import PackedCorpus
from .scoring import maxsim_scores


_dense_offsets_cache = _lib.StreamConstCache(32)
_dense_lengths = {}


def _dense_corpus(d: torch.Tensor) -> PackedCorpus:
    """View a dense [C, Ld, width] tensor as a packed corpus (no copy; zero rows stay physical rows)."""
    C, Ld, _ = d.shape
    # the row offsets of a [C, Ld] box and its host-side lengths are constants of the shape: cached per (shape, device, stream) -- an
    # eager training step is bound by its host time (the reference trainer captures no graphs), and this was a launch + two
    # allocations per loss call
    offsets = _dense_offsets_cache.get(("off", C, Ld), d.device, lambda: (
        torch.arange(0, (C + 1) * Ld, Ld, dtype=torch.int32, device=d.device) if Ld else torch.zeros(C + 1, dtype=torch.int32, device=d.device)))
    lengths = _dense_lengths.get((C, Ld))
    if lengths is None:
        if len(_dense_lengths) > 64:
            _dense_lengths.clear()
        lengths = _dense_lengths[(C, Ld)] = torch.full((C,), Ld, dtype=torch.int64, device="cpu")
    return PackedCorpus(blob=d.view(C * Ld, d.shape[2]), offsets=offsets, clamp0=None, lengths=lengths, avg_rows=Ld)


def _check_embeddings(q: torch.Tensor, d: torch.Tensor) -> None:
    if q.dim() != 3 or d.dim() != 3 or q.shape[2] != d.shape[2]:
        raise ValueError("expected query_embeddings [B, Lq, dim] and doc_embeddings [C, Ld, dim]")
    if q.dtype != d.dtype:
        raise RuntimeError(f"expected query and doc embeddings of one dtype, got {q.dtype} and {d.dtype}")
    if q.device.type != "cuda" or d.device != q.device:
        raise RuntimeError("colpali_amd losses run on an MI355X only (no CPU fallback): move the embeddings to the GPU")
    if q.dtype not in (torch.bfloat16, torch.float16, torch.float32):
        raise NotImplementedError(
            f"colpali_amd losses take bf16/fp16/fp32 embeddings (got {q.dtype}). Converting silently would change the "
            "loss, so this is an error")
    _lib.kernel_width(q.shape[2], q.dtype)   # raises for rows above 4 KiB


def _widen(x: torch.Tensor) -> torch.Tensor:
    """Zero-pad the embedding width to what the kernels stream (differentiable; a no-op for 128 x 16-bit rows)."""
    width = _lib.kernel_width(x.shape[-1], x.dtype)
    return x if width == x.shape[-1] else F.pad(x, (0, width - x.shape[-1]))


def _box_scores(qc: torch.Tensor, dc: torch.Tensor, corpus: PackedCorpus, want_lengths: bool = False):
    """fp32 [B, C] MaxSim scores of two dense boxes (with `want_lengths`: a tuple (scores, int32 [B] token counts of the queries or
    None) -- the transposed kernel counts them while it streams the query rows).  Long queries against short documents (the trainer's symmetric direction,
    trainer/contrastive_trainer.py:202-206: pages [B, 780, 128] as queries, queries [C, 32, 128] as documents; bf16 / f16, width 128)
    take the transposed kernel (msim_fwd_transposed: the long side streams, the short side is resident); everything else msim_fwd."""
    B, Lq, width = qc.shape
    C, Ld, _ = dc.shape
    if qc.dtype in (torch.bfloat16, torch.float16) and width == 128 and Lq > 128 and 0 < Ld <= 128 and B > 0 and C > 0:
        scores = torch.empty((B, C), dtype=torch.float32, device=qc.device)
        lengths = torch.empty((B,), dtype=torch.int32, device=qc.device) if want_lengths else None
        with torch.cuda.device(qc.device):
            rc = _lib.lib().msim_fwd_transposed(_lib.dtype_code(qc.dtype), _lib.ptr(qc), B, Lq, _lib.ptr(dc), C, Ld, width, _lib.ptr(scores),
                                                C, _lib.ptr(lengths), _lib.current_stream_handle(qc.device))
        _lib.check(rc, "msim_fwd_transposed")
        return (scores, lengths) if want_lengths else scores
    scores = maxsim_scores(qc, corpus)
    return (scores, None) if want_lengths else scores


def _dense_t_ok(qc: torch.Tensor, dc: torch.Tensor) -> bool:
    """Long queries against short documents with a gradient on EVERY pair (ColbertLoss / ColbertSigmoidLoss in the trainer's symmetric
    direction, trainer/contrastive_trainer.py:202-206): the shape whose backward runs as two GEMMs on the matrix cores
    (msim_dense_t_bwd) instead of 1.6 GB of gathers.  bf16 / f16, width 128, documents of at most 64 rows."""
    B, Lq, width = qc.shape
    C, Ld, _ = dc.shape
    if not (qc.dtype in (torch.bfloat16, torch.float16) and width == 128 and Lq > 128 and 0 < Ld <= 64 and B > 0 and C > 0):
        return False
    return bool(_lib.lib().msim_dense_t_supported(_lib.dtype_code(qc.dtype), B, Lq, C, Ld, width))


def _dense_t_forward(qc: torch.Tensor, dc: torch.Tensor, want_lengths: bool = False):
    """(scores fp32 [B, C], int32 [B] token counts or None, routing uint8 [B, C, Lq_pad]): msim_fwd_transposed_route."""
    L = _lib.lib()
    B, Lq, width = qc.shape
    C, Ld, _ = dc.shape
    dev = qc.device
    scores = torch.empty((B, C), dtype=torch.float32, device=dev)
    lengths = torch.empty((B,), dtype=torch.int32, device=dev) if want_lengths else None
    route = torch.empty((L.msim_dense_t_route_bytes(B, Lq, C),), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = L.msim_fwd_transposed_route(_lib.dtype_code(qc.dtype), _lib.ptr(qc), B, Lq, _lib.ptr(dc), C, Ld, width, _lib.ptr(scores), C,
                                         _lib.ptr(lengths), _lib.ptr(route), _lib.current_stream_handle(dev))
    _lib.check(rc, "msim_fwd_transposed_route")
    return scores, lengths, route


def _dense_t_backward(qc: torch.Tensor, dc: torch.Tensor, G: torch.Tensor, route: torch.Tensor, g_scale=None):
    """(dQ, dD) in the embeddings' dtype for dense dLoss/dscores G fp32 [B, C] and the forward's routing: msim_dense_t_bwd."""
    L = _lib.lib()
    B, Lq, width = qc.shape
    C, Ld, _ = dc.shape
    dev = qc.device
    G = G.to(torch.float32)
    if G.stride(-1) != 1 or (B > 1 and G.stride(0) < C):
        G = G.contiguous()
    dq = torch.empty_like(qc)
    dd = torch.empty_like(dc)
    gs, gs_code = _scale_arg(g_scale)
    with torch.cuda.device(dev):
        ws = torch.empty((L.msim_dense_t_bwd_workspace_bytes(B, Lq, C, Ld, width),), dtype=torch.uint8, device=dev)
        rc = L.msim_dense_t_bwd(_lib.dtype_code(qc.dtype), _lib.ptr(qc), B, Lq, _lib.ptr(dc), C, Ld, width, _lib.ptr(G),
                                G.stride(0) if B > 1 else C, _lib.ptr(gs), gs_code, _lib.ptr(route), _lib.ptr(dq), _lib.ptr(dd), _lib.ptr(ws),
                                _lib.current_stream_handle(dev))
    _lib.check(rc, "msim_dense_t_bwd")
    return dq, dd


class _MaxSim(torch.autograd.Function):
    """scores[b, c] = sum_n max_s <Q[b,n], D[c,s]> with a recompute backward.

    `dense_grad` is the caller's hint that every (query, doc) pair will receive a gradient (softmax / sigmoid losses): the
    forward then runs the arg-max pair kernel over all pairs once and keeps the [B*C, Lq] int32 routing, instead of the
    score-only kernel now and the same arg-max pass again when the gradient arrives."""

    @staticmethod
    def forward(ctx, q: torch.Tensor, d: torch.Tensor, dense_grad: bool = False) -> torch.Tensor:
        qc, dc = q.contiguous(), d.contiguous()
        corpus = _dense_corpus(dc)
        B, C = qc.shape[0], dc.shape[0]
        ctx.dense_t = False
        if dense_grad and any(ctx.needs_input_grad[:2]) and B * C > 0:
            if _dense_t_ok(qc, dc):         # long queries x short documents: byte routing now, two GEMMs when the gradient arrives
                scores, _, route = _dense_t_forward(qc, dc)
                ctx.dense_t = True
                ctx.save_for_backward(qc, dc, corpus.offsets, route)
                return scores
            scores = torch.empty((B, C), dtype=torch.float32, device=qc.device)   # returned as is (not a view: callers modify it in place)
            _, argmax = maxsim_all_pairs(qc, dc, corpus.offsets, scores_out=scores)
            ctx.save_for_backward(qc, dc, corpus.offsets, argmax)
            return scores
        scores = _box_scores(qc, dc, corpus)
        ctx.save_for_backward(qc, dc, corpus.offsets, None)
        return scores

    @staticmethod
    def backward(ctx, grad_scores: torch.Tensor):
        qc, dc, offsets, argmax = ctx.saved_tensors
        if ctx.dense_t:
            dq, dd = _dense_t_backward(qc, dc, grad_scores, argmax)
            return (dq if ctx.needs_input_grad[0] else None, dd if ctx.needs_input_grad[1] else None, None)
        dq, dd = maxsim_backward(qc, dc, offsets, grad_scores, argmax_all=argmax)
        return (dq if ctx.needs_input_grad[0] else None, dd if ctx.needs_input_grad[1] else None, None)


def maxsim_pairs(qc: torch.Tensor, dc: torch.Tensor, offsets: torch.Tensor, pairs: torch.Tensor,
                 want_scores: bool = True, want_argmax: bool = True, scores_out=None):
    """MaxSim (+ arg-max routing) for an int32 [n_pairs, 2] list of (query, doc) index pairs.
    `scores_out`: a contiguous fp32 tensor of n_pairs elements to write the scores into (any shape)."""
    L = _lib.lib()
    B, Lq, dim = qc.shape
    C = offsets.numel() - 1
    n_pairs = pairs.shape[0]
    dev = qc.device
    scores = (scores_out if scores_out is not None else torch.empty((n_pairs,), dtype=torch.float32, device=dev)) if want_scores else None
    argmax = torch.empty((n_pairs, Lq), dtype=torch.int32, device=dev) if want_argmax else None
    with torch.cuda.device(dev):
        # dc is the dense [C, Ld, width] box: every document has Ld rows (the bound only selects the kernel: long queries against
        # short documents -- the trainer's symmetric direction -- take the transposed pair kernel)
        rc = L.msim_pairs_argmax(_lib.dtype_code(qc.dtype), _lib.ptr(qc), B, Lq, _lib.ptr(dc), _lib.ptr(offsets), None, C,
                                 dim, int(dc.shape[1]) if dc.dim() == 3 else 0, _lib.ptr(pairs), n_pairs, _lib.ptr(scores),
                                 _lib.ptr(argmax), _lib.current_stream_handle(dev))
    _lib.check(rc, "msim_pairs_argmax")
    return scores, argmax


def maxsim_all_pairs(qc: torch.Tensor, dc: torch.Tensor, offsets: torch.Tensor, want_scores: bool = True, scores_out=None):
    """MaxSim scores [B, C] (optional) and the arg-max routing [(b * C + c), Lq] of EVERY (query, doc) pair -- the forward of the
    losses whose upstream gradient is dense.  bf16 / f16 embeddings of width 128 with Lq <= 128 take msim_allpairs_argmax (a wave scores
    up to four queries against one document: a document is streamed once per group of queries, not once per pair); everything else the
    pair-list kernel over the row-major all-pairs list (same outputs, same layout)."""
    B, Lq, dim = qc.shape
    C = offsets.numel() - 1
    dev = qc.device
    if not (qc.dtype in (torch.bfloat16, torch.float16) and dim == 128 and Lq <= 128 and B * C > 0):
        return maxsim_pairs(qc, dc, offsets, _all_pairs(B, C, dev), want_scores=want_scores, scores_out=scores_out)
    scores = (scores_out if scores_out is not None else torch.empty((B, C), dtype=torch.float32, device=dev)) if want_scores else None
    argmax = torch.empty((B * C, Lq), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().msim_allpairs_argmax(_lib.dtype_code(qc.dtype), _lib.ptr(qc), B, Lq, _lib.ptr(dc), _lib.ptr(offsets), None, C, dim,
                                             _lib.ptr(scores), C, _lib.ptr(argmax), _lib.current_stream_handle(dev))
    _lib.check(rc, "msim_allpairs_argmax")
    return scores, argmax


def maxsim_backward(qc: torch.Tensor, dc: torch.Tensor, offsets: torch.Tensor, grad_scores: torch.Tensor, argmax_all=None):
    """(dQ [B,Lq,width], dD [C,Ld,width], in the embeddings' dtype) for upstream dLoss/dscores [B, C].  Every (query, doc) pair is a pair of
    the list (zero gradients contribute zero): no data-dependent count, hence no host synchronisation.  `argmax_all`: the
    [B*C, Lq] routing in row-major order when the forward kept it; otherwise it is recomputed here (one pass of the
    arg-max pair kernel, about the cost of the forward).  The losses whose gradient is 2-sparse per query (pairwise) do not
    come through here: their epilogue kernel hands the backward its 2B pairs directly (_FusedInBatchLoss).
    Cost, whatever the sparsity of `grad_scores`: a [B*C, Lq] int32 routing tensor (4 * B * C * Lq bytes: 1 MiB at B = 32,
    C = 256, Lq = 32) and a B*C-pair gather / scatter -- the price of never reading a count back.  A pair with gradient 0
    contributes 0 * row, i.e. NaN if that row holds inf / NaN, exactly like the dense matmuls of the reference's autograd."""
    B, C = qc.shape[0], dc.shape[0]
    dev = qc.device
    if B * C == 0:
        return (torch.zeros(qc.shape, dtype=qc.dtype, device=dev), torch.zeros(dc.shape, dtype=dc.dtype, device=dev))
    pairs = _all_pairs(B, C, dev)
    if argmax_all is None:
        _, argmax_all = maxsim_all_pairs(qc, dc, offsets, want_scores=False)
    gp = grad_scores.to(torch.float32).reshape(-1).contiguous()
    return _pairs_backward(qc, dc, offsets, pairs, _all_pairs_order(B, C, dev), gp, argmax_all)


def _scale_arg(g_scale):
    """(pointer, dtype code) of an optional device scalar the backward kernels multiply every pair's gradient by -- autograd's
    upstream gradient of the loss, taken as it arrives (0-dim, any of the three dtypes) instead of a `coef * grad` launch."""
    if g_scale is None:
        return None, 0
    if g_scale.dtype not in (torch.bfloat16, torch.float16, torch.float32) or g_scale.numel() != 1:
        g_scale = g_scale.reshape(-1)[:1].to(torch.float32)
    return g_scale, _lib.dtype_code(g_scale.dtype)


def _pairs_backward(qc, dc, offsets, pairs, order, gp, argmax, g_scale=None):
    """msim_pairs_bwd: (dQ, dD) IN THE EMBEDDINGS' DTYPE (one rounding of the fp32 sums inside the kernels: no cast launches, half
    the bytes written) for a pair list sorted by query with its by-document permutation and routing; `g_scale`: see _scale_arg."""
    L = _lib.lib()
    B, Lq, dim = qc.shape
    C, Ld, _ = dc.shape
    dev = qc.device
    dq = torch.empty((B, Lq, dim), dtype=qc.dtype, device=dev)
    dd = torch.empty((C, Ld, dim), dtype=dc.dtype, device=dev)
    gs, gs_code = _scale_arg(g_scale)
    code = _lib.dtype_code(qc.dtype)
    with torch.cuda.device(dev):
        # scratch of the dense dD form (short documents, long entry lists: the trainer's symmetric direction), per call
        ws_bytes = L.msim_pairs_bwd_workspace_bytes(B, Lq, C, dim, Ld, pairs.shape[0])
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev) if ws_bytes else None
        rc = L.msim_pairs_bwd(code, _lib.ptr(qc), B, Lq, _lib.ptr(dc), _lib.ptr(offsets), C, dim, Ld,
                              _lib.ptr(pairs), _lib.ptr(order), _lib.ptr(gp), _lib.ptr(gs), gs_code, _lib.ptr(argmax), pairs.shape[0],
                              code, _lib.ptr(dq), _lib.ptr(dd), _lib.ptr(ws), _lib.current_stream_handle(dev))
    _lib.check(rc, "msim_pairs_bwd")
    return dq, dd


class _MaxSimPairs(torch.autograd.Function):
    """scores[p] = MaxSim(Q[pairs[p,0]], D[pairs[p,1]]) for an explicit pair list sorted by query index."""

    @staticmethod
    def forward(ctx, q: torch.Tensor, d: torch.Tensor, pairs: torch.Tensor) -> torch.Tensor:
        qc, dc = q.contiguous(), d.contiguous()
        corpus = _dense_corpus(dc)
        scores, _ = maxsim_pairs(qc, dc, corpus.offsets, pairs, want_argmax=False)
        ctx.save_for_backward(qc, dc, corpus.offsets, pairs)
        return scores

    @staticmethod
    def backward(ctx, grad_scores: torch.Tensor):
        qc, dc, offsets, pairs = ctx.saved_tensors
        gp = grad_scores.to(torch.float32).contiguous()
        order = torch.sort(pairs[:, 1].to(torch.int64), stable=True).indices.to(torch.int32).contiguous()
        _, argmax = maxsim_pairs(qc, dc, offsets, pairs, want_scores=False)
        dq, dd = _pairs_backward(qc, dc, offsets, pairs, order, gp, argmax)
        return (dq if ctx.needs_input_grad[0] else None, dd if ctx.needs_input_grad[1] else None, None)


def _check_pairs(pairs: torch.Tensor, B: int, C: int) -> None:
    """Range check of a caller-supplied pair list (one host sync): the kernels index Q, D and the routing with these numbers."""
    if pairs.dim() != 2 or pairs.shape[1] != 2 or pairs.dtype != torch.int32 or not pairs.is_contiguous():
        raise ValueError("pairs must be a contiguous int32 [n_pairs, 2] tensor of (query, doc) indices")
    if pairs.shape[0] == 0:
        return
    lo = pairs.amin(dim=0).tolist()
    hi = pairs.amax(dim=0).tolist()
    if lo[0] < 0 or lo[1] < 0 or hi[0] >= B or hi[1] >= C:
        raise ValueError(f"pairs index outside [0, {B}) x [0, {C}): queries {lo[0]}..{hi[0]}, docs {lo[1]}..{hi[1]}")
    if pairs.shape[0] > 1 and bool((pairs[1:, 0] < pairs[:-1, 0]).any()):
        raise ValueError("pairs must be sorted by query index")


def maxsim_paired(query_embeddings: torch.Tensor, doc_embeddings: torch.Tensor, pairs: torch.Tensor,
                  validate: bool = True) -> torch.Tensor:
    """Differentiable MaxSim of listed (query, doc) pairs: fp32 [n_pairs].  `pairs` int32 [n,2], sorted by query.

    Fused form of the paired contractions "bnd,bsd->bns" / "bnd,blsd->blns" followed by amax/sum
    (late_interaction_losses.py:235-240, :381-386).  `validate=False` skips the range check of `pairs` (and its host
    sync) for lists that are valid by construction."""
    query_embeddings, doc_embeddings = _autocast_inputs(query_embeddings, doc_embeddings)
    _check_embeddings(query_embeddings, doc_embeddings)
    if validate:
        _check_pairs(pairs, query_embeddings.shape[0], doc_embeddings.shape[0])
    return _MaxSimPairs.apply(_widen(query_embeddings), _widen(doc_embeddings), pairs)


def smooth_pairs(qc: torch.Tensor, dc: torch.Tensor, offsets: torch.Tensor, pairs: torch.Tensor, tau: float,
                 want_scores: bool = True, want_lse: bool = True, scores_out=None):
    """Smooth-max score (+ per-token logsumexp of sim / tau) for an int32 [n_pairs, 2] list of (query, doc) pairs."""
    L = _lib.lib()
    B, Lq, dim = qc.shape
    C = offsets.numel() - 1
    n_pairs = pairs.shape[0]
    dev = qc.device
    scores = (scores_out if scores_out is not None else torch.empty((n_pairs,), dtype=torch.float32, device=dev)) if want_scores else None
    lse = torch.empty((n_pairs, Lq), dtype=torch.float32, device=dev) if want_lse else None
    with torch.cuda.device(dev):
        rc = L.msim_smooth_pairs(_lib.dtype_code(qc.dtype), _lib.ptr(qc), B, Lq, _lib.ptr(dc), _lib.ptr(offsets), C, dim,
                                 _lib.ptr(pairs), n_pairs, tau, _lib.ptr(scores), _lib.ptr(lse), _lib.current_stream_handle(dev))
    _lib.check(rc, "msim_smooth_pairs")
    return scores, lse


_all_pairs_cache = _lib.StreamConstCache(32)


def _all_pairs(B: int, C: int, device: torch.device) -> torch.Tensor:
    """int32 [B*C, 2]: every (query, doc) pair in row-major order (a read-only constant, cached per shape, device and stream)."""
    def make():
        b = torch.arange(B, dtype=torch.int32, device=device).repeat_interleave(C)
        c = torch.arange(C, dtype=torch.int32, device=device).repeat(B)
        return torch.stack([b, c], dim=1).contiguous()

    return _all_pairs_cache.get(("pairs", B, C), device, make)


def _all_pairs_order(B: int, C: int, device: torch.device) -> torch.Tensor:
    """int32 [B*C]: indices of the row-major all-pairs list sorted by document, then query (what a stable sort by doc gives)."""
    return _all_pairs_cache.get(("order", B, C), device,
                                lambda: torch.arange(B * C, dtype=torch.int32, device=device).view(B, C).t().contiguous().view(-1))


def _smooth_backward(qc, dc, offsets, pairs, gp, tau, lse=None, order=None):
    """(dQ, dD) fp32 for the listed pairs (sorted by query) with upstream gradients gp; `lse` [n_pairs, Lq] if the forward kept
    it; `order`: the stable by-document permutation of the list if the caller has it."""
    L = _lib.lib()
    B, Lq, dim = qc.shape
    C, Ld, _ = dc.shape
    dev = qc.device
    dq = torch.empty((B, Lq, dim), dtype=torch.float32, device=dev)
    dd = torch.empty((C, Ld, dim), dtype=torch.float32, device=dev)
    n_pairs = pairs.shape[0]
    if n_pairs == 0:
        return dq.zero_(), dd.zero_()
    if order is None:
        order = torch.sort(pairs[:, 1].to(torch.int64), stable=True).indices.to(torch.int32).contiguous()
    if lse is None:
        _, lse = smooth_pairs(qc, dc, offsets, pairs, tau, want_scores=False)
    with torch.cuda.device(dev):
        ws_bytes = L.msim_smooth_bwd_workspace_bytes(B, Lq, dim)
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev) if ws_bytes else None
        rc = L.msim_smooth_pairs_bwd(_lib.dtype_code(qc.dtype), _lib.ptr(qc), B, Lq, _lib.ptr(dc), _lib.ptr(offsets), C, dim, Ld,
                                     _lib.ptr(pairs), _lib.ptr(order), _lib.ptr(gp), _lib.ptr(lse), n_pairs, tau,
                                     _lib.ptr(dq), _lib.ptr(dd), _lib.ptr(ws), _lib.current_stream_handle(dev))
    _lib.check(rc, "msim_smooth_pairs_bwd")
    return dq, dd


class _MaxSimSmooth(torch.autograd.Function):
    """scores[b, c] = sum_n tau * logsumexp_s(<Q[b,n], D[c,s]> / tau)  (late_interaction_losses.py:40-44, :88-90)."""

    @staticmethod
    def forward(ctx, q: torch.Tensor, d: torch.Tensor, tau: float) -> torch.Tensor:
        L = _lib.lib()
        qc, dc = q.contiguous(), d.contiguous()
        corpus = _dense_corpus(dc)
        B, Lq, dim = qc.shape
        C = dc.shape[0]
        ctx.tau = tau
        if any(ctx.needs_input_grad[:2]) and B * C > 0:
            # training: the pair-list kernel over ALL pairs returns the scores and the per-token logsumexp the backward needs in one
            # pass (the dense kernel would have to be followed by exactly this recompute when the gradient arrives)
            scores = torch.empty((B, C), dtype=torch.float32, device=qc.device)   # returned as is (not a view: callers modify it in place)
            _, lse = smooth_pairs(qc, dc, corpus.offsets, _all_pairs(B, C, qc.device), tau, scores_out=scores)
            ctx.save_for_backward(qc, dc, corpus.offsets, lse)
            return scores
        scores = torch.empty((B, C), dtype=torch.float32, device=qc.device)
        with torch.cuda.device(qc.device):
            rc = L.msim_smooth_fwd(_lib.dtype_code(qc.dtype), _lib.ptr(qc), B, Lq, _lib.ptr(dc), _lib.ptr(corpus.offsets), C, dim,
                                   tau, _lib.ptr(scores), max(C, 1), _lib.current_stream_handle(qc.device))
        _lib.check(rc, "msim_smooth_fwd")
        ctx.save_for_backward(qc, dc, corpus.offsets, None)
        return scores

    @staticmethod
    def backward(ctx, grad_scores: torch.Tensor):
        qc, dc, offsets, lse = ctx.saved_tensors
        B, C = qc.shape[0], dc.shape[0]
        # every pair is listed (zero gradients contribute zero): no data-dependent count, no host synchronisation
        pairs, gp = _all_pairs(B, C, qc.device), grad_scores.to(torch.float32).reshape(-1).contiguous()
        dq, dd = _smooth_backward(qc, dc, offsets, pairs, gp, ctx.tau, lse=lse, order=_all_pairs_order(B, C, qc.device))
        return (dq.to(qc.dtype) if ctx.needs_input_grad[0] else None,
                dd.to(dc.dtype) if ctx.needs_input_grad[1] else None, None)


class _MaxSimPairsSmooth(torch.autograd.Function):
    """Smooth-max score of an explicit pair list sorted by query index."""

    @staticmethod
    def forward(ctx, q: torch.Tensor, d: torch.Tensor, pairs: torch.Tensor, tau: float) -> torch.Tensor:
        qc, dc = q.contiguous(), d.contiguous()
        corpus = _dense_corpus(dc)
        scores, _ = smooth_pairs(qc, dc, corpus.offsets, pairs, tau, want_lse=False)
        ctx.save_for_backward(qc, dc, corpus.offsets, pairs)
        ctx.tau = tau
        return scores

    @staticmethod
    def backward(ctx, grad_scores: torch.Tensor):
        qc, dc, offsets, pairs = ctx.saved_tensors
        dq, dd = _smooth_backward(qc, dc, offsets, pairs, grad_scores.to(torch.float32).contiguous(), ctx.tau)
        return (dq.to(qc.dtype) if ctx.needs_input_grad[0] else None,
                dd.to(dc.dtype) if ctx.needs_input_grad[1] else None, None, None)


def _widen32(x: torch.Tensor) -> torch.Tensor:
    """Zero-pad the width to a multiple of 32 bytes (what the smooth-max kernels stream; differentiable)."""
    per = 32 // x.element_size()
    width = (x.shape[-1] + per - 1) // per * per
    return x if width == x.shape[-1] else F.pad(x, (0, width - x.shape[-1]))


def maxsim_smooth(query_embeddings: torch.Tensor, doc_embeddings: torch.Tensor, tau: float) -> torch.Tensor:
    """Differentiable fused smooth-max late interaction: fp32 [B, C]."""
    query_embeddings, doc_embeddings = _autocast_inputs(query_embeddings, doc_embeddings)
    _check_embeddings(query_embeddings, doc_embeddings)
    return _MaxSimSmooth.apply(_widen32(query_embeddings), _widen32(doc_embeddings), float(tau))


def maxsim_smooth_paired(query_embeddings: torch.Tensor, doc_embeddings: torch.Tensor, pairs: torch.Tensor,
                         tau: float, validate: bool = True) -> torch.Tensor:
    """Differentiable smooth-max score of listed (query, doc) pairs: fp32 [n_pairs]; `pairs` int32 [n,2] sorted by query."""
    query_embeddings, doc_embeddings = _autocast_inputs(query_embeddings, doc_embeddings)
    _check_embeddings(query_embeddings, doc_embeddings)
    if validate:
        _check_pairs(pairs, query_embeddings.shape[0], doc_embeddings.shape[0])
    return _MaxSimPairsSmooth.apply(_widen32(query_embeddings), _widen32(doc_embeddings), pairs, float(tau))


def _epilogue_workspace(B: int, C: int, device: torch.device):
    """Zero-filled scratch of msim_loss_epilogue (its ticket counter and per-row terms: cross-workgroup state of ONE launch), or
    None for the batches one workgroup handles without any (B <= 1024, B * C <= 262 144: BASELINE config 5 among them).
    Allocated per call -- about 12 * B bytes from the caching allocator, capturable -- so two launches in flight (other
    streams, threads, replays of captured graphs) can never share a ticket (round-2 advisor finding)."""
    need = _lib.lib().msim_loss_epilogue_workspace_bytes(B, C)
    return torch.zeros((need,), dtype=torch.uint8, device=device) if need else None


MODE_PAIRWISE, MODE_INFONCE, MODE_SIGMOID = 0, 1, 2


class _FusedInBatchLoss(torch.autograd.Function):
    """loss = epilogue(MaxSim(Q, D)) for ColbertPairwiseCELoss / ColbertLoss, forward and backward without a host
    synchronisation and without torch ops between the kernels (hipGraph-capturable as a whole):

      forward   fused MaxSim (hard or smooth max; for InfoNCE the pair-list kernel over all pairs, which also leaves the routing /
                logsumexp the dense gradient needs) -> msim_loss_epilogue (normalisation, filtering, loss value, dLoss/dscores)
      backward  pairwise: the 2B (query, doc) pairs the epilogue emitted -> arg-max (or logsumexp) recompute for those pairs only
                -> msim_pairs_bwd;   InfoNCE: dense G with the routing kept by the forward -> msim_pairs_bwd.
    `stats` (returned for the caller's bound check) = [loss, min, max of the normalised scores]."""

    @staticmethod
    def forward(ctx, q, d, mode, offset, temperature, normalize, filtering, filter_threshold, filter_factor, smooth, tau, loss_in_dtype):
        L = _lib.lib()
        qc, dc = q.contiguous(), d.contiguous()
        corpus = _dense_corpus(dc)
        B, Lq, width = qc.shape
        C = dc.shape[0]
        dev = qc.device
        need_grad = any(ctx.needs_input_grad[:2])
        aux = q_lengths = None                      # InfoNCE + gradients: [B*C, Lq] routing (hard max) or logsumexp (smooth max)
        ctx.dense_t = False
        dense = mode in (MODE_INFONCE, MODE_SIGMOID)     # a gradient on every (query, document) pair
        if dense and need_grad and not smooth and B * C > 0 and _dense_t_ok(qc, dc):
            # the trainer's symmetric direction: byte routing from the transposed kernel (and the token counts, as for the pairwise loss)
            scores, q_lengths, aux = _dense_t_forward(qc, dc, want_lengths=True)
            ctx.dense_t = True
        elif dense and need_grad:
            scores = torch.empty((B, C), dtype=torch.float32, device=dev)
            if smooth:
                _, aux = smooth_pairs(qc, dc, corpus.offsets, _all_pairs(B, C, dev), tau, scores_out=scores)
            else:
                _, aux = maxsim_all_pairs(qc, dc, corpus.offsets, scores_out=scores)
        elif smooth:
            scores = torch.empty((B, C), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                rc = L.msim_smooth_fwd(_lib.dtype_code(qc.dtype), _lib.ptr(qc), B, Lq, _lib.ptr(dc), _lib.ptr(corpus.offsets), C, width,
                                       tau, _lib.ptr(scores), max(C, 1), _lib.current_stream_handle(dev))
            _lib.check(rc, "msim_smooth_fwd")
        else:
            scores, q_lengths = _box_scores(qc, dc, corpus, want_lengths=True)
        stats = torch.empty((3,), dtype=torch.float32, device=dev)
        G = pairs = coef = order = None
        if mode == MODE_PAIRWISE:
            pairs = torch.empty((2 * B, 2), dtype=torch.int32, device=dev)
            coef = torch.empty((2 * B,), dtype=torch.float32, device=dev)
            order = torch.empty((2 * B,), dtype=torch.int32, device=dev)
        elif need_grad:
            G = torch.empty((B, C), dtype=torch.float32, device=dev)
        ws = _epilogue_workspace(B, C, dev)
        # the loss in the embeddings' dtype straight from the kernel (what the reference returns); fp32 when the caller wants that
        # (autocast): then it is a copy of the statistics' first entry
        loss = torch.empty((), dtype=qc.dtype, device=dev) if loss_in_dtype else None
        with torch.cuda.device(dev):
            rc = L.msim_loss_epilogue(mode, _lib.ptr(scores), C, B, C, _lib.ptr(qc), _lib.dtype_code(qc.dtype), Lq, width, offset,
                                      float(temperature), int(normalize), int(filtering), float(filter_threshold),
                                      float(filter_factor), _lib.ptr(G), _lib.ptr(pairs), _lib.ptr(coef), _lib.ptr(order),
                                      _lib.ptr(ws), _lib.ptr(stats), _lib.ptr(loss), _lib.ptr(q_lengths), _lib.current_stream_handle(dev))
        _lib.check(rc, "msim_loss_epilogue")
        ctx.mode, ctx.smooth, ctx.tau = mode, smooth, tau
        ctx.save_for_backward(qc, dc, corpus.offsets, G, pairs, coef, order, aux)
        ctx.mark_non_differentiable(stats)
        if loss is None:
            loss = stats[0].clone()                    # the loss as a tensor of its own (not a view of the statistics buffer)
        return loss, stats

    @staticmethod
    def backward(ctx, grad_loss, _grad_stats):
        qc, dc, offsets, G, pairs, coef, order, aux = ctx.saved_tensors
        B, C = qc.shape[0], dc.shape[0]
        if ctx.mode == MODE_PAIRWISE:
            if ctx.smooth:
                dq, dd = _smooth_backward(qc, dc, offsets, pairs, coef * grad_loss.to(torch.float32), ctx.tau, order=order)
            else:
                # the upstream gradient goes into the kernels as the device scalar it is; gradients come out in the embeddings' dtype
                _, argmax = maxsim_pairs(qc, dc, offsets, pairs, want_scores=False)
                dq, dd = _pairs_backward(qc, dc, offsets, pairs, order, coef, argmax, g_scale=grad_loss)
        else:
            if ctx.dense_t:
                dq, dd = _dense_t_backward(qc, dc, G, aux, g_scale=grad_loss)
            else:
                all_pairs, all_order = _all_pairs(B, C, qc.device), _all_pairs_order(B, C, qc.device)
                if ctx.smooth:
                    gp = (G * grad_loss.to(torch.float32)).reshape(-1)
                    dq, dd = _smooth_backward(qc, dc, offsets, all_pairs, gp, ctx.tau, lse=aux, order=all_order)
                else:
                    dq, dd = _pairs_backward(qc, dc, offsets, all_pairs, all_order, G.reshape(-1), aux, g_scale=grad_loss)
        return (dq.to(qc.dtype) if ctx.needs_input_grad[0] else None,
                dd.to(dc.dtype) if ctx.needs_input_grad[1] else None) + (None,) * 10


def _autocast_inputs(q: torch.Tensor, d: torch.Tensor):
    """Under torch.autocast the reference's einsum (late_interaction_losses.py:297) is an autocast-to-lower-precision
    op: fp32 embeddings are cast to the autocast dtype before the contraction (models emit fp32 there because
    `proj / proj.norm()` promotes, modeling_colqwen2.py:68).  Do exactly that cast -- it is differentiable, so the
    gradients arrive in the embeddings' own dtype -- and only then hand over to the kernels."""
    if torch.is_autocast_enabled("cuda"):
        lowp = torch.get_autocast_dtype("cuda")
        if lowp in (torch.bfloat16, torch.float16):
            if q.dtype in (torch.float32, torch.bfloat16, torch.float16) and q.dtype != lowp:
                q = q.to(lowp)
            if d.dtype in (torch.float32, torch.bfloat16, torch.float16) and d.dtype != lowp:
                d = d.to(lowp)
    return q, d


def _loss_dtype(query_embeddings: torch.Tensor) -> torch.dtype:
    """Dtype of the scalar the reference returns: the embeddings' own dtype (bf16 in -> bf16 loss), except
    under torch.autocast, where its token sum / softplus / cross_entropy are autocast-to-fp32 ops and the loss is fp32."""
    return torch.float32 if torch.is_autocast_enabled("cuda") else query_embeddings.dtype


def _check_offset(B: int, C: int, offset: int) -> None:
    """The reference fails on shapes when the positives [offset, offset + B) do not fit the C gathered documents
    (`scores.diagonal(offset)` / `scores[idx, pos_idx]` / `doc_embeddings[offset:offset+B]`); say so instead of indexing
    past the score matrix."""
    if offset < 0 or offset + B > C:
        raise IndexError(f"offset {offset} + batch {B} exceeds the {C} documents given (positives must lie in [offset, offset + B))")


def maxsim(query_embeddings: torch.Tensor, doc_embeddings: torch.Tensor, dense_grad: bool = False) -> torch.Tensor:
    """Differentiable fused MaxSim: fp32 [B, C] (late_interaction_losses.py:297-298 without the 4-D tensor).
    `dense_grad=True`: the caller expects a gradient on every pair (see _MaxSim)."""
    query_embeddings, doc_embeddings = _autocast_inputs(query_embeddings, doc_embeddings)
    _check_embeddings(query_embeddings, doc_embeddings)
    return _MaxSim.apply(_widen(query_embeddings), _widen(doc_embeddings), dense_grad)


_bounds_modules = None


def _register_bounds_flush(module) -> None:
    """Weak registry of modules with reports in flight; flushed once at interpreter exit."""
    global _bounds_modules
    if _bounds_modules is None:
        import atexit
        import weakref

        _bounds_modules = weakref.WeakSet()

        def _flush_all():
            for m in list(_bounds_modules):
                try:
                    m._flush_bounds(wait=True)
                except Exception:   # interpreter shutdown: the device may be gone already
                    pass

        atexit.register(_flush_all)
    _bounds_modules.add(module)


class ColbertModule(torch.nn.Module):
    """Shared hyper-parameters and [B, C]-sized helpers (late_interaction_losses.py:6-107)."""

    def __init__(self, max_batch_size: int = 1024, tau: float = 0.1, norm_tol: float = 1e-3,
                 filter_threshold: float = 0.95, filter_factor: float = 0.5):
        super().__init__()
        # non-persistent, like the reference (:27): the module contributes nothing to a state_dict
        self.register_buffer("idx_buffer", torch.arange(max_batch_size), persistent=False)
        self.tau = tau
        self.norm_tol = norm_tol
        self.filter_threshold = filter_threshold
        self.filter_factor = filter_factor

    def _get_idx(self, batch_size: int, offset: int, device: torch.device):
        if self.idx_buffer.device == torch.device(device):
            rows = self.idx_buffer[:batch_size]
        else:
            # the reference copies its (CPU) buffer to the device on every call -- a blocking pageable H2D copy, i.e. a host
            # synchronisation per step unless the module itself was moved to the GPU; same values from a device-side arange
            rows = torch.arange(min(batch_size, self.idx_buffer.numel()), device=device, dtype=self.idx_buffer.dtype)
        return rows, rows + offset

    def _smooth_max(self, scores: torch.Tensor, dim: int) -> torch.Tensor:
        return torch.logsumexp(scores / self.tau, dim=dim) * self.tau

    def _apply_normalization(self, scores: torch.Tensor, lengths: torch.Tensor) -> torch.Tensor:
        out = scores / (lengths.unsqueeze(1) if scores.ndim == 2 else lengths)
        self._report_bounds(torch.stack(torch.aminmax(out.detach())).to(torch.float32))
        return out

    def _report_bounds(self, lo_hi: torch.Tensor) -> None:
        """The reference prints when normalised scores leave [-tol, 1 + tol] (:62-70) -- and synchronises the host to find
        out.  Here the two numbers travel to pinned memory asynchronously and the message of step k is printed when step
        k + 1 (or a later one) finds them arrived: same diagnostic, no synchronisation in the training step.  Nothing is
        recorded while a hipGraph is being captured."""
        if lo_hi.device.type != "cuda":                       # CPU-sized helper use (the reference's own unit tests): check now
            lo, hi = float(lo_hi[-2]), float(lo_hi[-1])
            if lo < -self.norm_tol or hi > 1 + self.norm_tol:
                print(f"Scores out of bounds after normalization: min={lo:.4f}, max={hi:.4f}, tol={self.norm_tol}")
            return
        if torch.cuda.is_current_stream_capturing():
            return
        self._flush_bounds(wait=False)
        pending = self.__dict__.setdefault("_bounds_pending", [])
        if len(pending) >= 64:                               # never grow without bound: wait for the oldest report
            self._flush_bounds(wait=True, only_first=True)
        # a pinned buffer and an event per report in flight, REUSED: allocating pinned memory costs ~15 us of host time per call,
        # which is most of what this diagnostic cost an eager training step
        free = self.__dict__.setdefault("_bounds_free", [])
        host, ev = free.pop() if free else (torch.empty((2,), dtype=torch.float32, pin_memory=True), torch.cuda.Event())
        host.copy_(lo_hi[-2:], non_blocking=True)
        ev.record(torch.cuda.current_stream(lo_hi.device))
        pending.append((host, ev))
        _register_bounds_flush(self)

    def _flush_bounds(self, wait: bool = True, only_first: bool = False) -> None:
        """Print every queued report whose copy has arrived, in step order (`wait=True`: all of them, blocking)."""
        pending = self.__dict__.get("_bounds_pending")
        while pending:
            host, ev = pending[0]
            if wait:
                ev.synchronize()
            elif not ev.query():
                return
            pending.pop(0)
            lo, hi = float(host[0]), float(host[1])
            free = self.__dict__.setdefault("_bounds_free", [])
            if len(free) < 64:
                free.append((host, ev))
            if lo < -self.norm_tol or hi > 1 + self.norm_tol:
                print(f"Scores out of bounds after normalization: min={lo:.4f}, max={hi:.4f}, tol={self.norm_tol}")
            if only_first:
                return

    def flush_bounds(self) -> None:
        """Block until every pending out-of-bounds report of this module has been printed (the last training step's report has
        no later step to ride on; this also runs at interpreter exit)."""
        self._flush_bounds(wait=True)

    def __getstate__(self):
        # the in-flight diagnostics (pinned buffers, events) are not part of the module: pickle and copy.deepcopy (which goes
        # through __reduce_ex__ and therefore through here) skip them
        state = dict(self.__dict__)
        state.pop("_bounds_pending", None)
        state.pop("_bounds_free", None)
        return state

    def _aggregate(self, scores_raw: torch.Tensor, use_smooth_max: bool, dim_max: int, dim_sum: int) -> torch.Tensor:
        reduced = self._smooth_max(scores_raw, dim=dim_max) if use_smooth_max else scores_raw.amax(dim=dim_max)
        return reduced.sum(dim=dim_sum)

    def _filter_high_negatives(self, scores: torch.Tensor, pos_idx: torch.Tensor) -> None:
        rows, _ = self._get_idx(scores.size(0), 0, scores.device)
        limit = self.filter_threshold * scores[rows, pos_idx].unsqueeze(1)
        too_high = scores > limit
        too_high[rows, pos_idx] = False
        # in place like the reference (`scores[mask] *= factor`), written without boolean-mask indexing (its nonzero() is a
        # host synchronisation): same values, same gradient
        scores.mul_(torch.where(too_high, self.filter_factor, 1.0).to(scores.dtype))

    def _fused_inbatch_loss(self, mode: int, query_embeddings, doc_embeddings, offset: int) -> torch.Tensor:
        q, d = _autocast_inputs(query_embeddings, doc_embeddings)
        _check_embeddings(q, d)
        B, C = q.shape[0], d.shape[0]
        _check_offset(B, C, offset)
        if B > self.idx_buffer.numel():
            raise RuntimeError(f"batch of {B} queries exceeds max_batch_size={self.idx_buffer.numel()} (idx_buffer, :27)")
        if mode == MODE_PAIRWISE and C < 2:
            raise RuntimeError("selected index k out of range")          # what scores.topk(2, dim=1) raises (:310)
        widen = _widen32 if self.use_smooth_max else _widen
        out_dtype = _loss_dtype(query_embeddings)
        loss, stats = _FusedInBatchLoss.apply(widen(q), widen(d), mode, int(offset), float(self.temperature),
                                              bool(self.normalize_scores), bool(self.pos_aware_negative_filtering),
                                              float(self.filter_threshold), float(self.filter_factor),
                                              bool(self.use_smooth_max), float(self.tau), out_dtype == q.dtype)
        if self.normalize_scores:
            self._report_bounds(stats)
        return loss.to(out_dtype)

    # -- shared front end of every in-batch loss: lengths, fused MaxSim, optional normalisation / filtering
    def _inbatch_scores(self, query_embeddings, doc_embeddings, offset, dense_grad=False):
        _check_offset(query_embeddings.shape[0], doc_embeddings.shape[0], offset)
        lengths = (query_embeddings[:, :, 0] != 0).sum(dim=1)          # :296 -- first component, not a norm test
        if self.use_smooth_max:                                        # :88-89: tau * logsumexp(raw / tau) instead of amax
            scores = maxsim_smooth(query_embeddings, doc_embeddings, self.tau)
        else:
            scores = maxsim(query_embeddings, doc_embeddings, dense_grad)
        if self.normalize_scores:
            scores = self._apply_normalization(scores, lengths)
        rows, pos_idx = self._get_idx(scores.size(0), offset, scores.device)
        if self.pos_aware_negative_filtering:
            self._filter_high_negatives(scores, pos_idx)
        return scores, rows, pos_idx


class ColbertPairwiseCELoss(ColbertModule):
    """Pairwise softplus loss on the hardest in-batch negative (late_interaction_losses.py:255-313)."""

    def __init__(self, temperature: float = 1.0, normalize_scores: bool = True, use_smooth_max: bool = False,
                 pos_aware_negative_filtering: bool = False, max_batch_size: int = 1024, tau: float = 0.1,
                 norm_tol: float = 1e-3, filter_threshold: float = 0.95, filter_factor: float = 0.5):
        super().__init__(max_batch_size, tau, norm_tol, filter_threshold, filter_factor)
        self.temperature = temperature
        self.normalize_scores = normalize_scores
        self.use_smooth_max = use_smooth_max
        self.pos_aware_negative_filtering = pos_aware_negative_filtering

    def forward(self, query_embeddings: torch.Tensor, doc_embeddings: torch.Tensor, offset: int = 0) -> torch.Tensor:
        # :296-313 in two launches: fused MaxSim, then msim_loss_epilogue (lengths, normalisation, filtering, diagonal, top-2 with
        # the exact-equality selection, softplus mean -- and the two gradient-carrying (query, doc) pairs per query)
        return self._fused_inbatch_loss(MODE_PAIRWISE, query_embeddings, doc_embeddings, offset)


class ColbertLoss(ColbertModule):
    """InfoNCE over in-batch documents (late_interaction_losses.py:110-164)."""

    def __init__(self, temperature: float = 0.02, normalize_scores: bool = True, use_smooth_max: bool = False,
                 pos_aware_negative_filtering: bool = False, max_batch_size: int = 1024, tau: float = 0.1,
                 norm_tol: float = 1e-3, filter_threshold: float = 0.95, filter_factor: float = 0.5):
        super().__init__(max_batch_size, tau, norm_tol, filter_threshold, filter_factor)
        self.temperature = temperature
        self.normalize_scores = normalize_scores
        self.use_smooth_max = use_smooth_max
        self.pos_aware_negative_filtering = pos_aware_negative_filtering
        self.ce_loss = torch.nn.CrossEntropyLoss()

    def forward(self, query_embeddings: torch.Tensor, doc_embeddings: torch.Tensor, offset: int = 0) -> torch.Tensor:
        # :152-164 in two launches: the pair-list MaxSim over all pairs (keeps the routing for the dense gradient), then
        # msim_loss_epilogue (lengths, normalisation, filtering, cross entropy and its gradient)
        return self._fused_inbatch_loss(MODE_INFONCE, query_embeddings, doc_embeddings, offset)


class ColbertSigmoidLoss(ColbertModule):
    """Sigmoid loss over the in-batch square (late_interaction_losses.py:401-465)."""

    def __init__(self, temperature: float = 0.02, normalize_scores: bool = True, use_smooth_max: bool = False,
                 pos_aware_negative_filtering: bool = False, max_batch_size: int = 1024, tau: float = 0.1,
                 norm_tol: float = 1e-3, filter_threshold: float = 0.95, filter_factor: float = 0.5):
        super().__init__(max_batch_size, tau, norm_tol, filter_threshold, filter_factor)
        self.temperature = temperature
        self.normalize_scores = normalize_scores
        self.use_smooth_max = use_smooth_max
        self.pos_aware_negative_filtering = pos_aware_negative_filtering
        self.ce_loss = torch.nn.CrossEntropyLoss()

    def forward(self, query_embeddings: torch.Tensor, doc_embeddings: torch.Tensor, offset: int = 0) -> torch.Tensor:
        # :444-465 in two launches (round 6): the MaxSim forward that keeps the routing, then msim_loss_epilogue(MSIM_LOSS_SIGMOID) --
        # lengths, normalisation, filtering, the +1 / -1 sign square, the softplus mean and dLoss/dscores; until round 5 the sign mask,
        # its scatter, the division, softplus and mean were six torch launches and a B x B mask per step
        B, C = query_embeddings.shape[0], doc_embeddings.shape[0]
        if C != B:      # what `-scores.view(-1) * pos_mask` raises in the reference (:459-465): the loss is defined on the in-batch square
            raise RuntimeError(f"The size of tensor a ({B * C}) must match the size of tensor b ({B * B}) at non-singleton dimension 0")
        return self._fused_inbatch_loss(MODE_SIGMOID, query_embeddings, doc_embeddings, offset)


class _ExplicitNegativesMixin:
    """Shared forward of the two explicit-negative losses (late_interaction_losses.py:215-252 and :361-398):
    both compute softplus((neg - pos) / T).mean() over the listed negatives and optionally blend an in-batch term."""

    def _explicit_negative_term(self, query_embeddings, doc_embeddings, neg_doc_embeddings, offset):
        if self.use_smooth_max:
            paired = lambda q, d, pairs: maxsim_smooth_paired(q, d, pairs, self.tau, validate=False)   # noqa: E731
        else:
            paired = lambda q, d, pairs: maxsim_paired(q, d, pairs, validate=False)                    # noqa: E731
        B = query_embeddings.size(0)
        _check_offset(B, doc_embeddings.shape[0], offset)
        if neg_doc_embeddings.dim() != 4 or neg_doc_embeddings.size(0) != B:
            raise ValueError("expected neg_doc_embeddings [B, n_neg, L_neg, dim] with the queries' batch size")
        n_neg = neg_doc_embeddings.size(1)
        dev = query_embeddings.device
        lengths = (query_embeddings[:, :, 0] != 0).sum(dim=1)
        rows = torch.arange(B, dtype=torch.int32, device=dev)
        pos_pairs = torch.stack([rows, rows + offset], dim=1).contiguous()                 # (b, offset + b)
        pos_scores = paired(query_embeddings, doc_embeddings, pos_pairs)                   # "bnd,bsd->bns" -> amax -> sum
        neg_flat = neg_doc_embeddings.reshape(B * n_neg, neg_doc_embeddings.size(2), neg_doc_embeddings.size(3))
        neg_pairs = torch.stack([rows.repeat_interleave(n_neg),
                                 torch.arange(B * n_neg, dtype=torch.int32, device=dev)], dim=1).contiguous()
        neg_scores = paired(query_embeddings, neg_flat, neg_pairs).view(B, n_neg)          # "bnd,blsd->blns" -> amax -> sum
        if self.normalize_scores:
            pos_scores = self._apply_normalization(pos_scores, lengths)
            neg_scores = self._apply_normalization(neg_scores, lengths)
        return F.softplus((neg_scores - pos_scores.unsqueeze(1)) / self.temperature).mean()


class ColbertNegativeCELoss(_ExplicitNegativesMixin, ColbertModule):
    """Explicit-negative loss with an optional in-batch InfoNCE term (late_interaction_losses.py:167-252)."""

    def __init__(self, temperature: float = 0.02, normalize_scores: bool = True, use_smooth_max: bool = False,
                 pos_aware_negative_filtering: bool = False, in_batch_term_weight: float = 0.5,
                 max_batch_size: int = 1024, tau: float = 0.1, norm_tol: float = 1e-3,
                 filter_threshold: float = 0.95, filter_factor: float = 0.5):
        super().__init__(max_batch_size, tau, norm_tol, filter_threshold, filter_factor)
        self.temperature = temperature
        self.normalize_scores = normalize_scores
        self.use_smooth_max = use_smooth_max
        self.pos_aware_negative_filtering = pos_aware_negative_filtering
        self.in_batch_term_weight = in_batch_term_weight
        self.ce_loss = torch.nn.CrossEntropyLoss()
        assert in_batch_term_weight >= 0, "in_batch_term_weight must be non-negative"
        assert in_batch_term_weight <= 1, "in_batch_term_weight must be less than 1"
        self.inner_loss = ColbertLoss(temperature=temperature, normalize_scores=normalize_scores,
                                      use_smooth_max=use_smooth_max,
                                      pos_aware_negative_filtering=pos_aware_negative_filtering,
                                      max_batch_size=max_batch_size, tau=tau, norm_tol=norm_tol,
                                      filter_threshold=filter_threshold, filter_factor=filter_factor)

    def forward(self, query_embeddings: torch.Tensor, doc_embeddings: torch.Tensor,
                neg_doc_embeddings: torch.Tensor, offset: int = 0) -> torch.Tensor:
        loss = self._explicit_negative_term(query_embeddings, doc_embeddings, neg_doc_embeddings, offset)
        if self.in_batch_term_weight > 0:                                                   # :248-250
            loss_ib = self.inner_loss(query_embeddings, doc_embeddings, offset).to(loss.dtype)
            loss = loss * (1 - self.in_batch_term_weight) + loss_ib * self.in_batch_term_weight
        return loss.to(_loss_dtype(query_embeddings))


class ColbertPairwiseNegativeCELoss(_ExplicitNegativesMixin, ColbertModule):
    """Explicit-negative loss with an optional in-batch pairwise term (late_interaction_losses.py:316-398)."""

    def __init__(self, temperature: float = 0.02, normalize_scores: bool = True, use_smooth_max: bool = False,
                 pos_aware_negative_filtering: bool = False, in_batch_term_weight: float = 0.5,
                 max_batch_size: int = 1024, tau: float = 0.1, norm_tol: float = 1e-3,
                 filter_threshold: float = 0.95, filter_factor: float = 0.5):
        super().__init__(max_batch_size, tau, norm_tol, filter_threshold, filter_factor)
        self.temperature = temperature
        self.normalize_scores = normalize_scores
        self.use_smooth_max = use_smooth_max
        self.pos_aware_negative_filtering = pos_aware_negative_filtering
        self.in_batch_term_weight = in_batch_term_weight
        assert in_batch_term_weight >= 0, "in_batch_term_weight must be non-negative"
        assert in_batch_term_weight <= 1, "in_batch_term_weight must be less than 1"
        self.inner_pairwise = ColbertPairwiseCELoss(temperature=temperature, normalize_scores=normalize_scores,
                                                    use_smooth_max=use_smooth_max,
                                                    pos_aware_negative_filtering=pos_aware_negative_filtering,
                                                    max_batch_size=max_batch_size, tau=tau, norm_tol=norm_tol,
                                                    filter_threshold=filter_threshold, filter_factor=filter_factor)

    def forward(self, query_embeddings: torch.Tensor, doc_embeddings: torch.Tensor,
                neg_doc_embeddings: torch.Tensor, offset: int = 0) -> torch.Tensor:
        loss = self._explicit_negative_term(query_embeddings, doc_embeddings, neg_doc_embeddings, offset)
        if self.in_batch_term_weight > 0:                                                   # :394-396
            loss_ib = self.inner_pairwise(query_embeddings, doc_embeddings, offset).to(loss.dtype)
            loss = loss * (1 - self.in_batch_term_weight) + loss_ib * self.in_batch_term_weight
        return loss.to(_loss_dtype(query_embeddings))
