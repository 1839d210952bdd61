"""MI355X-native drop-in for the reference's late-interaction scorer.

Mirrors `BaseVisualRetrieverProcessor.score_multi_vector`
(colpali_engine/utils/processing_utils.py:132-187): same signature, argument
meaning, errors and return contract (a new fp32 CPU tensor [n_queries, n_passages]).
All arithmetic runs in the hand-written gfx950 kernels behind the C ABI
(include/maxsim.h); there is no torch/CPU compute path here.
"""
from __future__ import annotations

import logging
import os
from typing import List, Optional, Union

import torch

from . import _lib
from .corpus import (PackedCorpus, PackedQueries, _check_embeddings, _staging, _widen, block_clamp0, check_query_list, copy_stream,
                     host_list_image, pack_passages, pack_queries)

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
            f"colpali_amd: this entry point runs on an AMD Instinct MI355X only (requested device: {dev}); "
            "score_multi_vector and score_single_vector have a host path (device='cpu'; it lives in the same library, which links "
            "the HIP runtime: the ROCm user-space libraries must be installed even on a GPU-less host), nothing else does.")
    if not torch.cuda.is_available():
        raise RuntimeError("colpali_amd: no ROCm GPU is visible to torch")
    return dev


def _host_threads() -> int:
    """Native threads of the host-core scorer: COLPALI_AMD_HOST_THREADS, else torch's intra-op count capped by what the container
    really grants (_lib.effective_cpus: affinity and cgroup CPU quota)."""
    return max(1, min(int(os.environ.get("COLPALI_AMD_HOST_THREADS", "0")) or min(torch.get_num_threads(), _lib.effective_cpus()), 256))


def _score_on_host(qs, ps, batch_size: int, ref_rounding: bool) -> torch.Tensor:
    """`device="cpu"`: the library's own host-core scorer (msim_fwd_host_lists, colpali_amd/csrc/maxsim_host.cpp) straight on the
    caller's host tensors -- the reference computes on the device it is given (processing_utils.py:161, :172-179), so a CPU request
    is served on the CPU.  Nothing is packed or copied (torch.cat of a thousand pages costs more than scoring them on a many-core
    host), ragged queries are multiplied at their real lengths, the passages' block zero-padding travels as the same clamp0 flags
    the GPU path uses.  No torch arithmetic, no oracle; a GPU request never comes here."""
    import numpy as np

    from .corpus import block_clamp0

    L = _lib.lib()

    def rows_of(x, what):
        """(keep-alive list, pointers, row counts, dim, dtype) of a list of [len, dim] tensors or a [n, len, dim] tensor."""
        if isinstance(x, torch.Tensor):
            if x.dim() != 3:
                raise ValueError(f"a {what} tensor must be 3-D (n, max_len, dim)")
            _check_embeddings(x, what)
            t = x.to("cpu").contiguous()
            n, ln, dim = t.shape
            step = ln * dim * t.element_size()
            return [t], np.asarray([t.data_ptr() + i * step for i in range(n)], dtype=np.uint64), np.full(n, ln, dtype=np.int64), dim, t.dtype
        keep = []
        for t in x:
            if t.dim() != 2:
                raise ValueError(f"each {what[:-1] if what.endswith('s') else what} must be 2-D (sequence_length, dim)")
            _check_embeddings(t, what)
            if t.dtype != x[0].dtype:
                raise RuntimeError(f"expected {what} of one dtype, got {x[0].dtype} and {t.dtype}")
            if t.shape[1] != x[0].shape[1]:
                raise RuntimeError(f"expected {what} of one embedding width, got {x[0].shape[1]} and {t.shape[1]}")
            keep.append(t.to("cpu").contiguous())
        return (keep, np.asarray([t.data_ptr() if t.numel() else 0 for t in keep], dtype=np.uint64),
                np.asarray([t.shape[0] for t in keep], dtype=np.int64), int(x[0].shape[1]), x[0].dtype)

    q_keep, q_ptr, q_rows, q_dim, q_dtype = rows_of(qs, "queries")
    p_keep, p_ptr, p_rows, p_dim, p_dtype = rows_of(ps, "passages")
    if q_dtype != p_dtype:
        raise RuntimeError(f"expected queries and passages of one dtype, got {q_dtype} and {p_dtype}")
    if q_dim != p_dim:
        raise RuntimeError(f"queries have embedding width {q_dim}, the corpus {p_dim}")
    clamp0 = None
    if not isinstance(ps, torch.Tensor):          # a 3-D passage tensor keeps its physical zero rows: no flag needed
        lengths = torch.from_numpy(p_rows)
        flags = block_clamp0(lengths, batch_size)
        if bool((lengths == 0).any()):
            for j in range(0, len(p_rows), batch_size):      # an all-empty block makes the reference's max() over an empty dim raise
                if int(lengths[j: j + batch_size].max()) == 0:
                    raise RuntimeError("max(): Expected reduction dim 3 to have non-zero size.")
        clamp0 = flags.numpy() if bool(flags.any()) else None
    n_q, n_d = len(q_rows), len(p_rows)
    out = torch.empty((n_q, n_d), dtype=torch.float32)
    rc = L.msim_fwd_host_lists(_lib.dtype_code(q_dtype), q_ptr.ctypes.data, q_rows.ctypes.data, n_q, p_ptr.ctypes.data,
                               p_rows.ctypes.data, clamp0.ctypes.data if clamp0 is not None else None, n_d, q_dim, out.data_ptr(),
                               out.stride(0) if n_q > 1 else max(n_d, 1), _lib.MSIM_FLAG_REF_ROUNDING if ref_rounding else 0,
                               _host_threads())
    del q_keep, p_keep
    if rc != 0:
        msg = L.msim_host_last_error().decode("utf-8", "replace")
        raise (NotImplementedError if rc == -2 else ValueError if rc == -1 else RuntimeError)(f"msim_fwd_host_lists: {msg}")
    return out


def _fwd_workspace(nbytes: int, device: torch.device) -> Optional[torch.Tensor]:
    """msim_fwd's scratch (include/maxsim.h: one per launch in flight; the call initialises what it uses).  Allocated per
    call from torch's caching allocator -- stream-ordered reuse, capturable, nothing shared between streams, threads or
    graph replays."""
    return torch.empty((nbytes,), dtype=torch.uint8, device=device) if nbytes else None


_MAX_FWD_SCRATCH = 1 << 30        # bytes of msim_fwd scratch per launch (long queries: see maxsim_scores)


def _ref_rounding_from_env() -> bool:
    """COLPALI_AMD_REF_ROUNDING=1: the drop-in `score_multi_vector` returns what the reference LITERALLY returns for 16-bit
    embeddings -- every similarity rounded to the input dtype before the max, the token sum rounded to it
    (processing_utils.py:179 evaluated on bf16 / fp16 tensors) -- instead of the fp32-accurate score (default; the two are
    4.7e-3 apart on bf16 inputs, SURVEY finding 3).  The reference's signature has no room for the switch, hence the
    environment; `maxsim_scores(..., ref_rounding=True)` is the explicit form."""
    return os.environ.get("COLPALI_AMD_REF_ROUNDING", "0") not in ("", "0")


def maxsim_scores(queries: Union[torch.Tensor, PackedQueries], corpus: PackedCorpus, *, ref_rounding: bool = False,
                  out: Optional[torch.Tensor] = None, ref_bf16: Optional[bool] = None) -> torch.Tensor:
    """Device-level entry: queries x packed corpus -> fp32 [n_q, n] on the device.

    `queries`: a `PackedQueries` (the flat layout `pack_queries` returns on the GPU: ragged lengths, zero rows dropped ->
    msim_fwd_ragged) or a contiguous [n_q, Lq, width] device tensor (every row is scored as it stands -> msim_fwd).
    Asynchronous on torch's current stream.  `ref_rounding=True` reproduces the rounding the reference
    applies when torch evaluates processing_utils.py:179 in the embeddings' own 16-bit dtype
    (`ref_bf16` is the older name of the same switch).
    """
    if ref_bf16 is not None:
        ref_rounding = ref_bf16
    L = _lib.lib()
    flat = isinstance(queries, PackedQueries)
    if not flat and (queries.dim() != 3 or not queries.is_contiguous()):
        raise ValueError("queries must be a PackedQueries or a contiguous [n_q, Lq, width] tensor")
    if queries.dtype != corpus.blob.dtype:   # torch.einsum raises on mixed dtypes too
        raise RuntimeError(f"expected queries and passages of one dtype, got {queries.dtype} and {corpus.blob.dtype}")
    dt = _lib.dtype_code(queries.dtype)
    if queries.device != corpus.device:
        raise ValueError("queries and corpus live on different devices")
    n_q = len(queries)
    dim = queries.tokens.shape[1] if flat else queries.shape[2]
    if dim != corpus.blob.shape[1]:
        raise RuntimeError(f"queries have embedding width {dim}, the corpus {corpus.blob.shape[1]}")
    n = len(corpus)
    if out is None:
        out = torch.empty((n_q, n), dtype=torch.float32, device=queries.device)
    elif out.shape != (n_q, n) or out.dtype != torch.float32 or out.stride(1) != 1:
        raise ValueError("out must be fp32 [n_q, n] with unit inner stride")
    flags = _lib.MSIM_FLAG_REF_ROUNDING if ref_rounding else 0
    if n:       # launch-shape hint (include/maxsim.h: MSIM_FLAG_AVG_ROWS): the average document length, which the host side knows
        flags |= min(65535, max(1, corpus.average_rows())) << 8
    ld = out.stride(0) if n_q > 1 else max(n, 1)
    if flat:
        with torch.cuda.device(queries.device):
            oh = queries.offsets_host
            ws = _fwd_workspace(L.msim_fwd_ragged_workspace_bytes(dt, oh.data_ptr(), n_q, n, dim), queries.device)
            rc = L.msim_fwd_ragged(dt, _lib.ptr(queries.tokens), _lib.ptr(queries.offsets), oh.data_ptr(), n_q, _lib.ptr(corpus.blob),
                                   _lib.ptr(corpus.offsets), _lib.ptr(corpus.clamp0), n, dim, _lib.ptr(out), ld, flags, _lib.ptr(ws),
                                   _lib.current_stream_handle(queries.device))
            _lib.check(rc, "msim_fwd_ragged")
        return out
    Lq = queries.shape[1]
    # long queries are scored in 128-token pieces whose partial sums live in the workspace (n_q x pieces x n x 4 bytes): keep
    # that scratch bounded by scoring the queries in groups (rows of `out` are independent)
    ws_bytes = L.msim_fwd_workspace_bytes(dt, n_q, Lq, n, dim)
    group = n_q if ws_bytes <= _MAX_FWD_SCRATCH else max(1, int(n_q * _MAX_FWD_SCRATCH // ws_bytes))
    with torch.cuda.device(queries.device):
        for q0 in range(0, n_q, group):
            nq = min(group, n_q - q0)
            ws = _fwd_workspace(L.msim_fwd_workspace_bytes(dt, nq, Lq, n, dim), queries.device)
            rc = L.msim_fwd(dt, _lib.ptr(queries[q0:q0 + nq]), nq, Lq, _lib.ptr(corpus.blob), _lib.ptr(corpus.offsets),
                            _lib.ptr(corpus.clamp0), n, dim, _lib.ptr(out[q0:q0 + nq]), ld,
                            flags, _lib.ptr(ws), _lib.current_stream_handle(queries.device))
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
    ref_rounding = _ref_rounding_from_env()
    if torch.device(device).type == "cpu":
        scores = _score_on_host(qs, ps, batch_size, ref_rounding)
        assert scores.shape[0] == len(qs), f"Expected {len(qs)} scores, got {scores.shape[0]}"
        return scores
    dev = _require_gpu(device)
    # The host side of the call -- the native gather threads (started by their first parallel region: they keep the mask), the pinned
    # staging buffers (first touched here) and the thread that issues the copies -- runs on the GPU's own NUMA node for the length
    # of the call (colpali_amd/_lib.py: gpu_local_cpus); the caller's affinity is restored on the way out.
    with _lib.on_gpu_local_cpus(dev):
        checked = False
        if not isinstance(ps, torch.Tensor):
            if not isinstance(qs, torch.Tensor):
                check_query_list(qs)                  # the queries' errors come first, as before; they are PACKED under the first chunk's gather
                checked = True
            elif qs.dim() != 3:
                raise ValueError("a query tensor must be 3-D (n_queries, max_len, dim)")
            scores = _score_host_list_pipelined(qs, ps, dev, batch_size, ref_rounding, checked)
            if scores is not None:
                assert scores.shape[0] == len(qs), f"Expected {len(qs)} scores, got {scores.shape[0]}"
                return scores
        q = pack_queries(qs, dev, _checked=checked)
        cols = []
        for lo, hi in passage_ranges(ps, batch_size, _corpus_budget_bytes(dev)):
            corpus = pack_passages(ps[lo:hi], dev, batch_size=batch_size)
            cols.append(maxsim_scores(q, corpus, ref_rounding=ref_rounding).cpu())
            del corpus
    scores = cols[0] if len(cols) == 1 else torch.cat(cols, dim=1)
    assert scores.shape[0] == len(qs), f"Expected {len(qs)} scores, got {scores.shape[0]}"
    return scores.to(torch.float32)


TIMELINE = None        # bench.py / tools set this to a list: (label, time.perf_counter()) stamps of the drop-in call's phases


def _stamp(label: str) -> None:
    if TIMELINE is not None:
        import time

        TIMELINE.append((label, time.perf_counter()))


_PIPE_RANGE_BYTES = 48 << 20      # passage sub-ranges of about this size are scored while the next ones upload


def _score_host_list_pipelined(qs, ps, dev: torch.device, batch_size: int, ref_rounding: bool, qs_checked: bool = False) -> Optional[torch.Tensor]:
    """The drop-in call as evaluators make it -- a Python list of per-page HOST tensors (README.md:121-126) -- as one pipeline:

        one pass of checks -> [native gather of chunk k+1 into pinned memory | H2D of chunk k on a copy stream | MaxSim of the passage
        sub-ranges that have arrived, on the caller's stream] -> one D2H of the [n_q, n_p] matrix.

    The corpus is a 264 MB PCIe upload at BASELINE config 2's geometry (1000 pages x 1030 patches): the call's floor is that upload
    (4.7 ms at the 57 GB/s this host's pinned H2D reaches); checks, gather and the kernels hide under it.  The queries are packed and
    every device buffer is allocated while the first chunk is being gathered (round 6).  Sub-ranges are cut at
    multiples of `batch_size`, and the clamp0 flags come from the same blocking, so every passage keeps the block mates the reference
    pads it with (processing_utils.py:175-178): results are bit-identical to scoring the packed corpus in one launch.  Returns None
    for inputs this road does not take (a tensor that is not on the host; a corpus above the device budget; rows that need zero
    columns appended): the caller's general path serves those."""
    import numpy as np

    _stamp("begin")
    info = host_list_image(ps)
    if info is None:
        return None
    keep, srcs, rows, dim, dtype = info
    q_is_box = isinstance(qs, torch.Tensor)
    q_dtype = qs.dtype if q_is_box else qs[0].dtype
    q_dim = int(qs.shape[2]) if q_is_box else int(qs[0].shape[1])
    if q_is_box:
        _check_embeddings(qs, "queries")
    if q_dtype != dtype:
        raise RuntimeError(f"expected queries and passages of one dtype, got {q_dtype} and {dtype}")
    if _lib.kernel_width(dim, dtype) != dim:
        return None
    n = len(keep)
    row_bytes = dim * keep[0].element_size()
    prefix = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(rows * row_bytes, out=prefix[1:])
    total = int(prefix[n])
    total_rows = total // row_bytes
    if total == 0 or total > _corpus_budget_bytes(dev) or total_rows >= 2**31:
        return None
    lengths = torch.from_numpy(rows)
    flags = block_clamp0(lengths, batch_size)
    if bool((rows == 0).any()):
        for j in range(0, n, batch_size):          # an all-empty block makes the reference's max() over an empty dim raise
            if int(rows[j: j + batch_size].max()) == 0:
                raise RuntimeError("max(): Expected reduction dim 3 to have non-zero size.")
    n_q = len(qs)
    if q_dim != dim:
        raise RuntimeError(f"queries have embedding width {q_dim}, the corpus {dim}")
    _lib.place_gather_threads(dev, srcs[:: max(1, n // 8)][:8])      # next to the caller's pages (one move_pages query of 8 addresses)
    _stamp("checked")
    main = torch.cuda.current_stream(dev)
    side = copy_stream(dev)
    # passage sub-ranges: whole blocks of `batch_size`, about _PIPE_RANGE_BYTES each, at most eight
    n_blocks = (n + batch_size - 1) // batch_size
    n_sub = max(1, min(n_blocks, 8, total // _PIPE_RANGE_BYTES))
    cuts = sorted({min(n, ((n_blocks * i + n_sub - 1) // n_sub) * batch_size) for i in range(1, n_sub + 1)} | {n})
    state = [0, 0]                               # first passage not yet scored, next cut
    dv = {}                                      # what prepare() makes: q, blob, offsets, clamp0, out

    with torch.cuda.device(dev):

        def prepare() -> torch.Tensor:
            # runs while the native thread gathers the first chunk: query packing (its own small upload) and every allocation
            dv["q"] = pack_queries(qs, dev, _checked=qs_checked)
            blob = torch.empty((total_rows, dim), dtype=dtype, device=dev)
            blob.record_stream(side)   # written on the copy stream: whatever happens below, its memory is not reused before that stream is done
            off_host = torch.zeros(n + 1, dtype=torch.int32)
            off_host[1:] = torch.from_numpy((prefix[1:] // row_bytes).astype(np.int32))
            dv["blob"] = blob
            dv["offsets"] = off_host.to(dev, non_blocking=True)
            dv["clamp0"] = flags.to(dev, non_blocking=True) if bool(flags.any()) else None
            dv["out"] = torch.empty((n_q, n), dtype=torch.float32, device=dev)
            side.wait_stream(main)                       # the blob's memory may still be in use by earlier work of the caller's stream
            _stamp("prepared")
            return blob.view(torch.uint8).view(-1)

        def score_arrived(bytes_done: int) -> None:
            q, blob, offsets, clamp0, out = dv["q"], dv["blob"], dv["offsets"], dv["clamp0"], dv["out"]
            while state[1] < len(cuts) and int(prefix[cuts[state[1]]]) <= bytes_done:
                lo, hi = state[0], cuts[state[1]]
                ev = torch.cuda.Event()
                ev.record(side)
                main.wait_event(ev)
                if hi > lo:
                    # the WHOLE blob with the sub-range's slice of the absolute row offsets: nothing is re-based, the kernels only touch
                    # rows of passages lo .. hi-1, which have arrived
                    part = PackedCorpus(blob=blob, offsets=offsets[lo:hi + 1], clamp0=None if clamp0 is None else clamp0[lo:hi],
                                        lengths=lengths[lo:hi], avg_rows=max(1, int(rows[lo:hi].sum()) // (hi - lo)))
                    maxsim_scores(q, part, ref_rounding=ref_rounding, out=out[:, lo:hi])
                state[0] = hi
                state[1] += 1

        _staging.of(dev).upload_image(srcs, prefix, n, None, side, on_chunk=score_arrived, prepare=prepare)
        _stamp("issued")
        scores = dv["out"].cpu()
        _stamp("done")
    del keep
    return scores


def _corpus_budget_bytes(dev: torch.device) -> int:
    """Device memory one packed passage range may take: COLPALI_AMD_CORPUS_BUDGET_MB, else 60 % of the free HBM."""
    mb = os.environ.get("COLPALI_AMD_CORPUS_BUDGET_MB")
    if mb:
        return int(mb) << 20
    free, _ = torch.cuda.mem_get_info(dev)
    return int(free * 0.6)


def passage_ranges(ps, batch_size: int, budget_bytes: int):
    """Contiguous passage ranges [lo, hi) whose packed rows fit `budget_bytes`, cut at multiples of `batch_size` so that
    every passage keeps the block mates the reference pads it with (processing_utils.py:175-178).  A corpus that fits --
    the normal case on 288 GB -- is one range: packed and uploaded once.  A single block above the budget is still one range
    (the reference needs that block on the device too)."""
    n = len(ps)
    if isinstance(ps, torch.Tensor):
        per = [ps.shape[1] * ps.shape[2] * ps.element_size()] * n
    else:
        per = [p.shape[0] * p.shape[1] * p.element_size() for p in ps]
    if sum(per) <= budget_bytes:
        return [(0, n)]
    ranges, lo, acc = [], 0, 0
    for b0 in range(0, n, batch_size):
        blk = sum(per[b0 : b0 + batch_size])
        if acc and acc + blk > budget_bytes:
            ranges.append((lo, b0))
            lo, acc = b0, 0
        acc += blk
    ranges.append((lo, n))
    return ranges


def similarity_matrix(a: torch.Tensor, b: torch.Tensor, *, ref_rounding: bool = False,
                      out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Device-level entry: out[i, j] = <a[i], b[j]> in fp32 for [n_a, dim] x [n_b, dim] device tensors (msim_sim_matrix)."""
    L = _lib.lib()
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[1]:
        raise ValueError("expected [n_a, dim] and [n_b, dim]")
    if a.dtype != b.dtype:
        raise RuntimeError(f"expected both operands of one dtype, got {a.dtype} and {b.dtype}")
    if a.device.type != "cuda" or b.device != a.device:
        raise RuntimeError("colpali_amd.similarity_matrix runs on an MI355X only (no CPU fallback)")
    _check_embeddings(a, "similarity operand")
    aw, bw = _widen(a).contiguous(), _widen(b).contiguous()
    n_a, n_b = a.shape[0], b.shape[0]
    if out is None:
        out = torch.empty((n_a, n_b), dtype=torch.float32, device=a.device)
    elif out.shape != (n_a, n_b) or out.dtype != torch.float32 or (n_b > 1 and out.stride(1) != 1):
        raise ValueError("out must be fp32 [n_a, n_b] with unit inner stride")
    with torch.cuda.device(a.device):
        rc = L.msim_sim_matrix(_lib.dtype_code(a.dtype), _lib.ptr(aw), n_a, _lib.ptr(bw), n_b, aw.shape[1], _lib.ptr(out),
                               out.stride(0) if n_a > 1 else max(n_b, 1),
                               _lib.MSIM_FLAG_REF_ROUNDING if ref_rounding else 0, _lib.current_stream_handle(a.device))
    _lib.check(rc, "msim_sim_matrix")
    return out


def score_single_vector(
    qs: Union[torch.Tensor, List[torch.Tensor]],
    ps: Union[torch.Tensor, List[torch.Tensor]],
    device: Optional[Union[str, torch.device]] = None,
) -> torch.Tensor:
    """Dot-product scores of single-vector (bi-encoder) embeddings, drop-in for processing_utils.py:103-130:
    a new fp32 tensor [n_queries, n_passages] on `device` (the reference does not move this one to the CPU)."""
    device = device or get_torch_device("auto")
    if isinstance(qs, list) and isinstance(ps, list):
        if len(qs) == 0:
            raise ValueError("No queries provided")
        if len(ps) == 0:
            raise ValueError("No passages provided")
        qs, ps = torch.stack(qs), torch.stack(ps)
    if torch.device(device).type == "cpu":      # the library's host-core path (msim_sim_matrix_host): a CPU request is served on the CPU
        a, b = qs.to("cpu"), ps.to("cpu")
        if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[1]:
            raise ValueError("expected [n_a, dim] and [n_b, dim]")
        if a.dtype != b.dtype:
            raise RuntimeError(f"expected both operands of one dtype, got {a.dtype} and {b.dtype}")
        _check_embeddings(a, "similarity operand")
        a, b = a.contiguous(), b.contiguous()
        L = _lib.lib()
        scores = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32)
        rc = L.msim_sim_matrix_host(_lib.dtype_code(a.dtype), a.data_ptr(), a.shape[0], b.data_ptr(), b.shape[0], a.shape[1],
                                    scores.data_ptr(), max(b.shape[0], 1), 0, _host_threads())
        if rc != 0:
            raise RuntimeError(f"msim_sim_matrix_host: {L.msim_host_last_error().decode('utf-8', 'replace')}")
        assert scores.shape[0] == len(qs), f"Expected {len(qs)} scores, got {scores.shape[0]}"
        return scores
    dev = _require_gpu(device)
    scores = similarity_matrix(qs.to(dev), ps.to(dev))
    assert scores.shape[0] == len(qs), f"Expected {len(qs)} scores, got {scores.shape[0]}"
    return scores.to(torch.float32)


def get_similarity_maps_from_embeddings(image_embeddings: torch.Tensor, query_embeddings: torch.Tensor, n_patches,
                                        image_mask: torch.Tensor) -> List[torch.Tensor]:
    """Per-image similarity maps [query_tokens, n_patches_x, n_patches_y], drop-in for
    colpali_engine/interpretability/similarity_map_utils.py:9-55 (same checks, same axis order, same dtype)."""
    if isinstance(n_patches, tuple):
        n_patches = [n_patches] * image_embeddings.size(0)
    maps: List[torch.Tensor] = []
    for idx in range(image_embeddings.size(0)):
        nx, ny = n_patches[idx]
        if image_mask[idx].sum() != nx * ny:
            raise ValueError(
                f"The number of patches ({nx} x {ny} = {nx * ny}) "
                f"does not match the number of non-padded image tokens ({image_mask[idx].sum()}).")
        patches = image_embeddings[idx][image_mask[idx]]                       # (h w) c, h = n_patches_y, w = n_patches_x
        sim = similarity_matrix(query_embeddings[idx].contiguous(), patches.contiguous())   # [n, h*w]
        maps.append(sim.view(-1, ny, nx).permute(0, 2, 1).to(image_embeddings.dtype))   # "(h w) -> w h": [n, x, y]
    return maps
