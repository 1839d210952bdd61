"""Device-resident packed corpus: the data layout the gfx950 kernels stream.

    blob    bf16|f16|f32 [total_rows, width]   all passages' patch embeddings back to back (width = dim, zero-padded
                                     to a multiple of 32 bytes when it is not the tuned 128 x 16-bit row)
    offsets int32 [n + 1]            passage c owns rows offsets[c] .. offsets[c+1]-1
    clamp0  uint8 [n] or None        1 = the reference zero-pads this passage inside its
                                     passage block, so a similarity of 0 joins every
                                     per-token max (processing_utils.py:175-178)

The reference re-pads and re-uploads every passage block for every query block
(processing_utils.py:170-178); here a corpus is packed and uploaded once.
"""
from __future__ import annotations

import ctypes
import operator
import os
import threading
from dataclasses import dataclass
from typing import List, Optional, Sequence, Union

import numpy as np
import torch

from . import _lib as _lib_mod
from ._lib import kernel_width

EMBED_DIM = 128   # the width the tuned kernels are built for (ColPali / ColQwen2 projection dim)


def block_clamp0(lengths: torch.Tensor, batch_size: int) -> torch.Tensor:
    """Per-passage "sees zero padding" flag under the reference's blocking.

    processing_utils.py:175-178 pads passages j .. j+batch_size-1 to the longest of
    the block with zero rows (pad_sequence, padding_value=0): every passage shorter
    than its block's maximum gains similarity-0 candidates.
    """
    if batch_size <= 0:
        raise ValueError("batch_size must be positive")
    # numpy on purpose: a torch CPU op on this path (repeat_interleave / max over a dim) wakes torch's whole intra-op pool -- 128
    # OpenMP threads on the GPU box, each spinning ~5 ms after the parallel region: 700 ms of CPU time per 9 ms call, which ran the
    # container's CPU quota (cgroup cpu.max: 16 CPUs per 100 ms) dry and froze the process for the rest of the period.  That was the
    # "70-90 ms stall in one call out of four" of rounds 1-4 (profiles/r05_logs/dropin_stalls.log: nr_throttled +1 in every slow call)
    import numpy as np

    ln = lengths.detach().to("cpu").numpy().astype(np.int64, copy=False)
    n = ln.size
    if n == 0:
        return torch.zeros((0,), dtype=torch.uint8)
    if batch_size >= n:                   # one block (and no padding to a multiple of a huge batch_size)
        flags = ln < ln.max()
    else:
        pad = (-n) % batch_size
        padded = np.concatenate([ln, np.full((pad,), -1, dtype=np.int64)]) if pad else ln
        block_max = np.repeat(padded.reshape(-1, batch_size).max(axis=1), batch_size)[:n]
        flags = ln < block_max
    return torch.from_numpy(flags.astype(np.uint8))


@dataclass
class PackedCorpus:
    blob: torch.Tensor                 # bf16 | f16 | f32 [rows, width], on the GPU
    offsets: torch.Tensor              # int32 [n+1], on the GPU
    clamp0: Optional[torch.Tensor]     # uint8 [n] on the GPU, or None
    lengths: torch.Tensor              # int64 [n], on the host
    id_base: int = 0                   # global id of passage 0 (sharded corpora)
    avg_rows: Optional[int] = None     # average passage length (launch-shape hint); None: derived once from `lengths` and kept

    def __len__(self) -> int:
        return int(self.lengths.numel())

    def average_rows(self) -> int:
        """The launch-shape hint of include/maxsim.h (MSIM_FLAG_AVG_ROWS), computed ONCE per corpus and with numpy: a torch CPU
        reduction here sat in every retrieval step and, above ~32k passages, woke torch's intra-op pool -- the cause of the 70-90 ms
        cgroup-throttling stalls block_clamp0 was moved off torch for (round-5 advisor finding)."""
        if self.avg_rows is None:
            n = len(self)
            if not n:
                self.avg_rows = 0
            elif self.lengths.device.type == "cpu":
                self.avg_rows = int(self.lengths.numpy().sum(dtype=np.int64)) // n
            else:       # `lengths` is a host tensor by contract; under a `with torch.device("cuda")` default the packers' torch.full lands there
                self.avg_rows = int(self.lengths.sum()) // n
        return self.avg_rows

    @property
    def device(self) -> torch.device:
        return self.blob.device

    @property
    def nbytes(self) -> int:
        return self.blob.numel() * self.blob.element_size()


def _check_embeddings(x: torch.Tensor, what: str) -> None:
    if x.dim() not in (2, 3):
        raise ValueError(f"{what}: expected 2-D or 3-D embeddings")
    if x.dtype not in (torch.bfloat16, torch.float16, torch.float32):
        raise NotImplementedError(
            f"{what}: dtype {x.dtype}; the gfx950 kernels take bfloat16 (what the ColPali/ColQwen2 forward emits), "
            "float16 or float32 embeddings. Converting silently would change the scores, so this is an error.")
    kernel_width(x.shape[-1], x.dtype)   # raises for rows above 4 KiB


def _widen(x: torch.Tensor) -> torch.Tensor:
    """[rows, dim] -> [rows, kernel_width(dim)] with zero columns appended (a no-op for the tuned shape)."""
    width = kernel_width(x.shape[-1], x.dtype)
    if width == x.shape[-1]:
        return x
    return torch.nn.functional.pad(x, (0, width - x.shape[-1]))


def _env_int(name: str, default: int, lo: int, hi: int) -> int:
    """An integer tuning knob from the environment, clamped to [lo, hi]; a malformed value falls back to the default (importing the
    package must never raise over a typo in a knob)."""
    try:
        return max(lo, min(hi, int(os.environ[name])))
    except (KeyError, ValueError):
        return default


# half of what the container grants, at most 8; COLPALI_AMD_COPY_THREADS: tuning knob (tools/dropin_profile.py)
_COPY_THREADS = _env_int("COLPALI_AMD_COPY_THREADS", max(1, min(8, _lib_mod.effective_cpus() // 2)), 1, 256)
_EDGE_CHUNK_BYTES = _env_int("COLPALI_AMD_EDGE_CHUNK_MB", 16, 0, 1024) << 20     # first and last chunk of a pipelined upload (0: off)
_EDGE_LAST_BYTES = (lambda mb: -1 if mb < 0 else mb << 20)(_env_int("COLPALI_AMD_EDGE_LAST_MB", -1, -1, 1024))   # the last chunk on its own (-1: as the first)
# pinned host memory per staging buffer = two halves that alternate: while one half is on its way to the GPU the passages of the
# next chunk are memcpy'd into the other, so a call costs max(host memcpy, PCIe upload) instead of their sum -- 32 MiB per half
# is large enough for both to run at full speed and small enough for a 264 MB corpus (1000 ColPali pages) to overlap almost fully
STAGING_BYTES = _env_int("COLPALI_AMD_STAGING_MB", 64, 2, 16384) << 20
# 0: every chunk's gather blocks the calling thread before its copy is issued, as before round 6 (A/B knob: tools/dropin_phases.py)
_ASYNC_GATHER = _env_int("COLPALI_AMD_ASYNC_GATHER", 1, 0, 1) == 1


class _Staging:
    """Bounded, reusable pinned host buffer for the drop-in's host -> device upload.

    The passages are memcpy'd into one half of the buffer while the other half is on its way to the GPU (asynchronous H2D
    into the preallocated device blob), so the pinned memory this process holds is STAGING_BYTES whatever the corpus size
    -- the reference's own footprint is O(batch) too (processing_utils.py:175-178).  Pinning costs ~0.1 s per call at this
    size, so the buffer is kept; an event per half guards its reuse."""

    def __init__(self):
        self.buf: Optional[torch.Tensor] = None
        self.events = [None, None]
        self.lock = threading.Lock()
        self.next_half = 0

    def _halves(self, want: int):
        half = max(min(want, STAGING_BYTES // 2), 1 << 20)
        if self.buf is None or self.buf.numel() < 2 * half:
            for ev in self.events:
                if ev is not None:
                    ev.synchronize()
            # pinned pages come from the allocating thread's NUMA node: the GPU's own (_lib.gpu_local_cpus), whoever asks first
            with _lib_mod.on_gpu_local_cpus(torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None):
                self.buf = torch.empty((2 * half,), dtype=torch.uint8, pin_memory=True)
        return self.buf.numel() // 2

    def upload(self, ps: Sequence[torch.Tensor], dim: int, device: torch.device,
               slot_rows: Optional[int] = None) -> torch.Tensor:
        """Rows of all tensors back to back ([sum of lengths, dim]); with `slot_rows`, tensor i starts at row i * slot_rows of
        a zero-filled [len(ps) * slot_rows, dim] image (the queries' zero padding)."""
        dtype = ps[0].dtype
        es = ps[0].element_size()
        total = sum(int(p.shape[0]) for p in ps) if slot_rows is None else len(ps) * slot_rows
        nbytes = total * dim * es
        dev = torch.empty((total, dim), dtype=dtype, device=device)
        if nbytes == 0:
            return dev
        dev_bytes = dev.view(torch.uint8).view(-1)
        with self.lock:
            half = self._halves(nbytes)
            base = self.buf.data_ptr()
            # (destination offset in the image, source pointer, bytes) per passage, split at chunk borders so that every chunk
            # fits one half of the staging buffer; the copies themselves run in msim_host_gather (native threads, one call per
            # chunk: a thousand small memmoves from python threads fight over the interpreter lock)
            import numpy as np

            L = _lib_mod.lib()
            keep, offs, srcs, sizes = [], [], [], []
            o = 0
            for p in ps:
                n = int(p.shape[0]) * dim * es
                if n:
                    src = p if p.is_contiguous() else p.contiguous()
                    keep.append(src)
                    sp, done = src.data_ptr(), 0
                    while done < n:
                        room = half - ((o + done) % half)
                        take = min(room, n - done)
                        offs.append(o + done)
                        srcs.append(sp + done)
                        sizes.append(take)
                        done += take
                o += n if slot_rows is None else slot_rows * dim * es
            offs_a = np.asarray(offs, dtype=np.int64)
            srcs_a = np.asarray(srcs, dtype=np.uint64)
            sizes_a = np.asarray(sizes, dtype=np.int64)
            stream = torch.cuda.current_stream(device)
            k = 0
            n_pieces = len(offs)
            for c0 in range(0, nbytes, half):
                c1 = min(nbytes, c0 + half)
                h = (c0 // half) & 1
                hb = base + h * half
                if self.events[h] is not None:
                    self.events[h].synchronize()          # the previous upload has left this half
                if slot_rows is not None:
                    ctypes.memset(hb, 0, c1 - c0)
                k1 = k
                while k1 < n_pieces and offs[k1] < c1:
                    k1 += 1
                if k1 > k:
                    rel = offs_a[k:k1] - c0                # offsets inside this half
                    rc = L.msim_host_gather(hb, srcs_a[k:k1].ctypes.data, rel.ctypes.data, sizes_a[k:k1].ctypes.data,
                                            k1 - k, _COPY_THREADS)
                    if rc != 0:
                        raise RuntimeError(f"msim_host_gather failed: {L.msim_last_error().decode()}")
                k = k1
                dev_bytes[c0:c1].copy_(self.buf[h * half : h * half + (c1 - c0)], non_blocking=True)
                self.events[h] = torch.cuda.Event()
                self.events[h].record(stream)
            del keep
        return dev


    def upload_image(self, srcs, prefix, n: int, dst_bytes: Optional[torch.Tensor], side_stream, on_chunk=None, prepare=None) -> None:
        """The byte image of n host buffers (buffer i = image bytes prefix[i] .. prefix[i+1]-1 at address srcs[i]; numpy uint64 /
        int64 arrays) -> `dst_bytes` (a uint8 device view of the same length), through the two pinned halves: a native thread gathers
        chunk k + 1 into one half (msim_host_gather_range_begin / _wait: a persistent pool, one request per chunk, nothing per page on
        the Python side) while the other half is on its way to the GPU on `side_stream` AND while this thread issues that copy and runs
        `on_chunk(bytes_uploaded_so_far)` -- the caller launches work on what has arrived (scoring.py).  The halves alternate ACROSS calls
        too, so back-to-back uploads keep overlapping.  `prepare()`, if given, runs on this thread while the FIRST chunk is being
        gathered (nothing else can overlap that gather: no upload is running yet) and returns the destination when `dst_bytes` is None
        -- the drop-in packs its queries and allocates its device buffers there (round 6: 0.2-0.3 ms off the front of the call)."""
        total = int(prefix[n])
        if total == 0:
            if prepare is not None:
                prepare()
            return
        L = _lib_mod.lib()
        with self.lock:
            half = self._halves(total)
            base = self.buf.data_ptr()
            chunks = _chunk_schedule(total, half)
            in_flight = False

            def begin(k: int) -> int:
                h = self.next_half
                self.next_half ^= 1
                if self.events[h] is not None:
                    self.events[h].synchronize()          # the previous upload has left this half
                c0, c1 = chunks[k]
                rc = L.msim_host_gather_range_begin(base + h * half, srcs.ctypes.data, prefix.ctypes.data, n, c0, c1, _COPY_THREADS)
                if rc != 0:
                    raise RuntimeError(f"msim_host_gather_range_begin failed: {L.msim_host_last_error().decode()}")
                return h

            try:
                if not _ASYNC_GATHER and prepare is not None:
                    got = prepare()
                    dst_bytes = got if dst_bytes is None else dst_bytes
                h = begin(0)
                in_flight = True
                if _ASYNC_GATHER and prepare is not None:
                    got = prepare()
                    dst_bytes = got if dst_bytes is None else dst_bytes
                for k, (c0, c1) in enumerate(chunks):
                    rc = L.msim_host_gather_range_wait()
                    in_flight = False
                    if rc != 0:
                        raise RuntimeError(f"msim_host_gather_range failed: {L.msim_host_last_error().decode()}")
                    with torch.cuda.stream(side_stream):
                        dst_bytes[c0:c1].copy_(self.buf[h * half : h * half + (c1 - c0)], non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(side_stream)
                    self.events[h] = ev
                    if _ASYNC_GATHER and k + 1 < len(chunks):   # the copy of chunk k is queued: gather k + 1 into the other half (waits
                        h = begin(k + 1)                        # for chunk k - 1's copy to have left it) while this thread goes on below
                        in_flight = True
                    if on_chunk is not None:
                        on_chunk(c1)
                    if not _ASYNC_GATHER and k + 1 < len(chunks):
                        h = begin(k + 1)
                        in_flight = True
            finally:
                if in_flight:                             # never leave with the native thread still writing / reading the caller's pages
                    L.msim_host_gather_range_wait()


def _chunk_schedule(total: int, half: int):
    """[c0, c1) byte ranges of a pipelined upload through staging halves of `half` bytes.  Nothing overlaps the gather of the FIRST
    chunk (no upload is running yet) nor the upload of the LAST one (nothing is left to gather): both are kept small
    (_EDGE_CHUNK_BYTES), the chunks in between are whole halves.  Measured at BASELINE config 2 (264 MB), toggled inside one process and
    interleaved (tools/ab_dropin_edge.py, profiles/r05_logs/ab_dropin_edge.log): 16 MiB edges 7.65 -> 7.19 ms and 8.18 -> 7.97 ms
    (2 / 4 / 8 MiB: -0.0 .. -0.2 ms: their per-chunk cost eats what the shorter head and tail give)."""
    edge = min(_EDGE_CHUNK_BYTES, half)
    last = min(_EDGE_LAST_BYTES, half) if _EDGE_LAST_BYTES >= 0 else edge
    if edge <= 0 or last <= 0 or total <= 2 * half:
        return [(c0, min(total, c0 + half)) for c0 in range(0, total, half)]
    cuts = [0, edge]
    while total - cuts[-1] > half + last:
        cuts.append(cuts[-1] + half)
    if total - cuts[-1] > last:
        cuts.append(total - last)
    cuts.append(total)
    return list(zip(cuts[:-1], cuts[1:]))


class _PerDevice:
    """One staging buffer PER GPU: its pinned pages are first touched on that GPU's NUMA node (_halves), so in a process that drives
    GPUs on both sockets every upload stages through memory next to its own GPU (round-5 advisor finding: one process-global buffer
    stayed on the first GPU's socket)."""

    def __init__(self):
        self._by_dev = {}
        self._lock = threading.Lock()

    def of(self, device) -> _Staging:
        idx = torch.device(device).index
        if idx is None:
            idx = torch.cuda.current_device() if torch.cuda.is_available() else -1
        with self._lock:
            st = self._by_dev.get(idx)
            if st is None:
                st = self._by_dev[idx] = _Staging()
            return st

    # the single-GPU spelling (tools/, tests): the buffer of the current device
    def upload(self, ps, dim, device, slot_rows=None):
        return self.of(device).upload(ps, dim, device, slot_rows)


_staging = _PerDevice()
_staging_q = _PerDevice()    # queries: a second buffer, so that packing the queries never waits for the corpus upload
_copy_streams = {}


def copy_stream(device: torch.device) -> torch.cuda.Stream:
    """The side stream the drop-in's uploads run on (one per device): H2D copies of the next passage range overlap the scoring of
    the previous one on the caller's stream."""
    device = torch.device(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = _copy_streams.get(key)
    if st is None:
        st = _copy_streams[key] = torch.cuda.Stream(device=device)
    return st


_dim_of = torch.Tensor.dim
_dtype_of = operator.attrgetter("dtype")
_shape_of = operator.attrgetter("shape")
_is_cpu = operator.attrgetter("is_cpu")
_is_contig = torch.Tensor.is_contiguous
_data_ptr = torch.Tensor.data_ptr
_stride_of = torch.Tensor.stride
_numel_of = torch.Tensor.numel


def host_list_image(ps: Sequence[torch.Tensor], what: str = "passages"):
    """ONE pass over a list of host tensors: the reference's checks (2-D, one dtype, one width) and what the upload needs --
    (keep-alive list, source addresses uint64 [n], rows int64 [n], dim, dtype) -- or None when a tensor does not live on the host.
    (A thousand pages x a handful of attribute reads each is ~1 ms of Python: it is done once, not once per helper.)"""
    first = ps[0]
    dtype, dim = first.dtype, (first.shape[1] if first.dim() == 2 else -1)
    # the common case in C-speed passes (round 6): a per-page Python loop with seven attribute reads was 0.6 ms of a 4-7 ms call for
    # 1000 pages; `map` over attrgetters is ~2.5 x faster.  Anything unusual (a non-2-D page, a second dtype or width, a device tensor,
    # a non-contiguous page) falls through to the loop below, which raises the reference's errors in the reference's order.
    if dim > 0 and type(ps) in (list, tuple):
        n = len(ps)
        # contiguous + stride (dim, 1) + at least two rows <=> shape (rows, dim) -- read without building a thousand torch.Size tuples
        # (that pass alone was 0.7 of 1.2 ms); a page of fewer than two rows (its stride says nothing) takes the loop below
        if (set(map(_dim_of, ps)) == {2} and set(map(_dtype_of, ps)) == {dtype} and all(map(_is_cpu, ps))
                and all(map(_is_contig, ps)) and set(map(_stride_of, ps)) == {(dim, 1)}):
            numel = np.fromiter(map(_numel_of, ps), dtype=np.int64, count=n)
            rows = numel // dim
            if bool((rows >= 2).all()) and bool((rows * dim == numel).all()):
                _check_embeddings(first, what)
                srcs = np.fromiter(map(_data_ptr, ps), dtype=np.uint64, count=n)
                return list(ps), srcs, rows, int(dim), dtype
    keep = []
    for p in ps:
        if p.dim() != 2:
            raise ValueError(f"each {what[:-1]} must be 2-D (sequence_length, dim)")
        if p.dtype != dtype:
            raise RuntimeError(f"expected {what} of one dtype, got {dtype} and {p.dtype}")
        if p.shape[1] != dim:
            raise RuntimeError(f"expected {what} of one embedding width, got {dim} and {p.shape[1]}")
        if p.device.type != "cpu":
            return None
        keep.append(p if p.is_contiguous() else p.contiguous())
    _check_embeddings(first, what)
    srcs = np.fromiter((t.data_ptr() for t in keep), dtype=np.uint64, count=len(keep))
    rows = np.fromiter((t.shape[0] for t in keep), dtype=np.int64, count=len(keep))
    return keep, srcs, rows, int(dim), dtype


def _gather_rows(ps: Sequence[torch.Tensor], dim: int, device: torch.device) -> torch.Tensor:
    """All passages' rows back to back on `device`: [sum of lengths, dim].

    Host lists are the drop-in's normal input (README.md:121-126 keeps `list(torch.unbind(embeddings.to("cpu")))`).
    `torch.cat` of a thousand bf16 tensors costs 0.5-0.6 s on a 128-thread host (its parallel loop is all contention)
    -- more than the reference's whole blocked scorer on the same GPU; one memcpy per passage into a pinned staging
    buffer followed by ONE asynchronous H2D copy is what makes the drop-in faster than the reference end to end, not
    only inside the kernel."""
    device = torch.device(device)
    if device.type == "cuda" and all(p.device.type == "cpu" for p in ps):
        return _staging.upload(ps, dim, device)
    return torch.cat([p.reshape(-1, dim) for p in ps], dim=0).to(device, non_blocking=True)


def pack_passages(ps: Union[torch.Tensor, Sequence[torch.Tensor]], device: torch.device,
                  batch_size: Optional[int] = 128, id_base: int = 0) -> PackedCorpus:
    """Pack passages for the device.  `batch_size=None` disables the reference's
    block zero-padding semantics (every passage is scored on its own rows only)."""
    if isinstance(ps, torch.Tensor):
        if ps.dim() != 3:
            raise ValueError("a passage tensor must be 3-D (n_passages, max_len, dim)")
        _check_embeddings(ps, "passages")
        n, L, dim = ps.shape
        # slicing a 3-D tensor in blocks and pad_sequence over its rows is a re-stack: zero rows that are
        # physically present take part in the max on their own, no clamp flag needed
        blob = _widen(ps.reshape(n * L, dim).to(device, non_blocking=True)).contiguous()
        lengths = torch.full((n,), L, dtype=torch.int64)
        clamp0 = None
    else:
        if len(ps) == 0:
            raise ValueError("No passages provided")
        for p in ps:
            if p.dim() != 2:
                raise ValueError("each passage must be 2-D (sequence_length, dim)")
            _check_embeddings(p, "passages")
            if p.dtype != ps[0].dtype:
                raise RuntimeError(f"expected passages of one dtype, got {ps[0].dtype} and {p.dtype}")
            if p.shape[1] != ps[0].shape[1]:
                raise RuntimeError(f"expected passages of one embedding width, got {ps[0].shape[1]} and {p.shape[1]}")
        dim = ps[0].shape[1]
        lengths = torch.tensor([p.shape[0] for p in ps], dtype=torch.int64)
        blob = _widen(_gather_rows(ps, dim, device)).contiguous()
        clamp0 = None
        if batch_size is not None:
            flags = block_clamp0(lengths, batch_size)
            if bool(flags.any()):
                clamp0 = flags.to(device)
            if bool((lengths == 0).any()):
                # an all-empty block makes the reference's max() over an empty dim raise
                for j in range(0, len(ps), batch_size):
                    if int(lengths[j : j + batch_size].max()) == 0:
                        raise RuntimeError("max(): Expected reduction dim 3 to have non-zero size.")
    if blob.numel() == 0:  # keep a valid device pointer
        blob = torch.zeros((1, blob.shape[1]), dtype=blob.dtype, device=device)
    offsets = torch.zeros(lengths.numel() + 1, dtype=torch.int64)
    torch.cumsum(lengths, 0, out=offsets[1:])
    if int(offsets[-1]) >= 2**31:
        raise NotImplementedError("more than 2^31 patch rows in one shard")
    return PackedCorpus(blob=blob, offsets=offsets.to(torch.int32).to(device), clamp0=clamp0,
                        lengths=lengths, id_base=id_base)


@dataclass
class PackedQueries:
    """Queries in the FLAT layout the tuned kernels take (include/maxsim.h: msim_fwd_ragged): every query's real tokens back
    to back, like the packed corpus.  Queries are ragged in real use (processing_utils.py:86 appends 10 augmentation tokens to
    a question of any length) and a zero row -- the model's padded positions, modeling_colpali.py:72 -- adds exactly 0 to every
    score, so neither padding to the batch maximum nor the zero rows inside a query are carried to the GPU's matrix cores."""
    tokens: torch.Tensor               # bf16 | f16 [T, 128], on the GPU
    offsets: torch.Tensor              # int32 [n_q + 1], on the GPU
    offsets_host: torch.Tensor         # int32 [n_q + 1], on the host (the launch plan is made from it)

    def __len__(self) -> int:
        return int(self.offsets_host.numel()) - 1

    @property
    def device(self) -> torch.device:
        return self.tokens.device

    @property
    def dtype(self) -> torch.dtype:
        return self.tokens.dtype

    @property
    def lengths(self) -> torch.Tensor:
        return (self.offsets_host[1:] - self.offsets_host[:-1]).to(torch.int64)

    def select(self, lo: int, hi: int) -> "PackedQueries":
        """Queries lo .. hi-1 (views of the same token matrix; offsets rebased)."""
        base = int(self.offsets_host[lo])
        oh = (self.offsets_host[lo:hi + 1] - base).contiguous()
        return PackedQueries(tokens=self.tokens[base:int(self.offsets_host[hi])] if hi > lo else self.tokens[:0],
                             offsets=oh.to(self.tokens.device, non_blocking=True), offsets_host=oh)


MAX_FLAT_QUERY_TOKENS = 1024       # a query's tokens share one workgroup (maxsim_abi.hip: flat_plan); longer ones keep the box layout
WIDE_DIM = 320                     # ColQwen3's projection (models/qwen3/colqwen3/modeling_colqwen3.py:48): the panel kernels (K1sP / K1bP / K1bPF)
MAX_FLAT_QUERY_TOKENS_WIDE = 512   # ... whose query block holds 8 waves x 4 units of 16 tokens (maxsim_abi.hip: panels_flat_plan)


def _is_flat_shape(dtype: torch.dtype, dim: int) -> bool:
    return dtype in (torch.bfloat16, torch.float16) and dim in (EMBED_DIM, WIDE_DIM)


def max_flat_query_tokens(dim: int) -> int:
    return MAX_FLAT_QUERY_TOKENS_WIDE if dim == WIDE_DIM else MAX_FLAT_QUERY_TOKENS


def _flat_from_host_list(qs: Sequence[torch.Tensor], dim: int, device: torch.device, compact: bool) -> Optional["PackedQueries"]:
    """Host list -> flat layout: native threads count and copy the rows that are not all-zero straight into the pinned staging
    buffer, one asynchronous upload (no torch CPU op on the way: see pack_queries)."""
    import numpy as np

    L = _lib_mod.lib()
    es = qs[0].element_size()
    row_bytes = dim * es
    keep = [q if q.is_contiguous() else q.contiguous() for q in qs]
    n = len(keep)
    srcs = np.asarray([q.data_ptr() for q in keep], dtype=np.uint64)
    rows = np.asarray([int(q.shape[0]) for q in keep], dtype=np.int64)
    if compact:
        counts = np.zeros(n, dtype=np.int32)
        rc = L.msim_host_count_nonzero_rows(srcs.ctypes.data, rows.ctypes.data, row_bytes, n, counts.ctypes.data, _COPY_THREADS)
        if rc != 0:
            raise RuntimeError(f"msim_host_count_nonzero_rows failed: {L.msim_last_error().decode()}")
    else:
        counts = rows.astype(np.int32)
    if int(counts.max(initial=0)) > max_flat_query_tokens(dim):
        return None
    off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, out=off[1:])
    total = int(off[-1])
    offsets_host = torch.from_numpy(off.astype(np.int32))
    tokens = torch.empty((max(total, 1), dim), dtype=qs[0].dtype, device=device)
    if total == 0:
        tokens.zero_()
    else:
        # through the bounded pinned halves of the queries' staging buffer (two halves, an event each: a query set of any size pins
        # STAGING_BYTES of host memory, never its own size -- round-4 advisor finding)
        st = _staging_q.of(device)
        stream = torch.cuda.current_stream(device)
        dst_bytes = tokens.view(torch.uint8).view(-1)
        if not compact:
            prefix = np.zeros(n + 1, dtype=np.int64)
            np.cumsum(rows * row_bytes, out=prefix[1:])
            st.upload_image(srcs, prefix, n, dst_bytes[:total * row_bytes], stream)
        else:
            with st.lock:
                half = st._halves(total * row_bytes)
                base = st.buf.data_ptr()
                g0 = 0
                while g0 < n:                         # groups of whole queries whose compacted rows fit one half (a query is <= 256 KiB)
                    g1 = g0 + 1
                    while g1 < n and (off[g1 + 1] - off[g0]) * row_bytes <= half:
                        g1 += 1
                    nb = int(off[g1] - off[g0]) * row_bytes
                    if nb:
                        h = st.next_half
                        st.next_half ^= 1
                        if st.events[h] is not None:
                            st.events[h].synchronize()          # the previous upload has left this half
                        dst_row = np.ascontiguousarray(off[g0:g1] - off[g0])
                        rc = L.msim_host_gather_nonzero_rows(base + h * half, srcs[g0:g1].ctypes.data, rows[g0:g1].ctypes.data, row_bytes,
                                                             dst_row.ctypes.data, g1 - g0, _COPY_THREADS)
                        if rc != 0:
                            raise RuntimeError(f"host query gather failed: {L.msim_last_error().decode()}")
                        lo = int(off[g0]) * row_bytes
                        dst_bytes[lo:lo + nb].copy_(st.buf[h * half: h * half + nb], non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(stream)
                        st.events[h] = ev
                    g0 = g1
    del keep
    return PackedQueries(tokens=tokens, offsets=offsets_host.to(device, non_blocking=True), offsets_host=offsets_host)


def _flat_from_device_box(box: torch.Tensor, compact: bool) -> Optional["PackedQueries"]:
    """[n_q, Lq, 128] on the GPU -> flat layout.  With `compact` the zero rows are dropped by msim_query_compact; the per-query
    counts come back to the host (one small D2H + synchronisation: the launch plan needs the lengths)."""
    n_q, Lq, dim = box.shape
    box = box.contiguous()
    if not compact or n_q == 0 or Lq == 0:
        if Lq > max_flat_query_tokens(dim):
            return None
        oh = (torch.arange(n_q + 1, dtype=torch.int64) * Lq).to(torch.int32)
        tok = box.reshape(n_q * Lq, dim)
        if tok.shape[0] == 0:
            tok = torch.zeros((1, dim), dtype=box.dtype, device=box.device)
        return PackedQueries(tokens=tok, offsets=oh.to(box.device, non_blocking=True), offsets_host=oh)
    if Lq > 4096:
        return None
    L = _lib_mod.lib()
    row_bytes = dim * box.element_size()
    counts = torch.empty((n_q,), dtype=torch.int32, device=box.device)
    with torch.cuda.device(box.device):
        stream = _lib_mod.current_stream_handle(box.device)
        _lib_mod.check(L.msim_query_compact(_lib_mod.ptr(box), n_q, Lq, row_bytes, None, _lib_mod.ptr(counts), None, stream),
                       "msim_query_compact")
        counts_h = counts.cpu()
        if int(counts_h.max()) > max_flat_query_tokens(dim):
            return None
        oh = torch.zeros(n_q + 1, dtype=torch.int32)
        torch.cumsum(counts_h, 0, out=oh[1:])
        off = oh.to(box.device, non_blocking=True)
        total = int(oh[-1])
        tokens = torch.empty((max(total, 1), dim), dtype=box.dtype, device=box.device)
        if total == 0:
            tokens.zero_()
        else:
            _lib_mod.check(L.msim_query_compact(_lib_mod.ptr(box), n_q, Lq, row_bytes, _lib_mod.ptr(off), None, _lib_mod.ptr(tokens),
                                                stream), "msim_query_compact")
    return PackedQueries(tokens=tokens, offsets=off, offsets_host=oh)


def check_query_list(qs: Sequence[torch.Tensor]) -> None:
    """The checks pack_queries makes on a list of queries, on their own (the drop-in validates first and packs later, under the
    first chunk of the corpus upload)."""
    if len(qs) == 0:
        raise ValueError("No queries provided")
    for q in qs:
        if q.dim() != 2:
            raise ValueError("each query must be 2-D (sequence_length, dim)")
        _check_embeddings(q, "queries")
        if q.dtype != qs[0].dtype:
            raise RuntimeError(f"expected queries of one dtype, got {qs[0].dtype} and {q.dtype}")
        if q.shape[1] != qs[0].shape[1]:
            raise RuntimeError(f"expected queries of one embedding width, got {qs[0].shape[1]} and {q.shape[1]}")


def pack_queries(qs: Union[torch.Tensor, Sequence[torch.Tensor]], device: torch.device, *, layout: str = "auto",
                 compact: bool = True, _checked: bool = False) -> Union[torch.Tensor, "PackedQueries"]:
    """Queries for the device.

    layout="flat" (what "auto" picks on the GPU for bf16 / f16 embeddings of width 128 or 320): a `PackedQueries` -- every query's
    tokens back to back; with `compact` (default) rows that are entirely zero are dropped, which changes no score: a zero
    query row scores exactly 0 against everything (its max is 0).
    layout="box": [n_q, Lq, width], zero padded (rows beyond a query's length and columns beyond its width) -- every other
    dtype / width, CPU tensors, and queries of more than 1024 tokens (512 at width 320).  processing_utils.py:172 pads each 128-query block to
    its own longest query; padding all queries to the global maximum returns identical values for the same reason.
    """
    if layout not in ("auto", "flat", "box"):
        raise ValueError("layout must be 'auto', 'flat' or 'box'")
    device = torch.device(device)
    if isinstance(qs, torch.Tensor):
        if qs.dim() != 3:
            raise ValueError("a query tensor must be 3-D (n_queries, max_len, dim)")
        _check_embeddings(qs, "queries")
        if layout != "box" and device.type == "cuda" and _is_flat_shape(qs.dtype, qs.shape[2]):
            flat = _flat_from_device_box(qs.to(device, non_blocking=True), compact)
            if flat is not None:
                return flat
        if layout == "flat":
            raise NotImplementedError("the flat query layout takes bf16 / f16 embeddings of width 128 (320) on the GPU, at most "
                                      f"{MAX_FLAT_QUERY_TOKENS} ({MAX_FLAT_QUERY_TOKENS_WIDE}) tokens per query")
        return _widen(qs.to(device, non_blocking=True)).contiguous()
    if not _checked:
        check_query_list(qs)
    dim = int(qs[0].shape[1])
    if layout != "box" and device.type == "cuda" and _is_flat_shape(qs[0].dtype, dim):
        if all(q.device.type == "cpu" for q in qs):
            flat = _flat_from_host_list(qs, dim, device, compact)
        else:
            l_max = max(int(q.shape[0]) for q in qs)
            box = torch.nn.utils.rnn.pad_sequence([q.to(device) for q in qs], batch_first=True, padding_value=0) if l_max else None
            flat = _flat_from_device_box(box, True) if box is not None else None
        if flat is not None:
            return flat
    if layout == "flat":
        raise NotImplementedError("the flat query layout takes bf16 / f16 embeddings of width 128 (320) on the GPU, at most "
                                  f"{MAX_FLAT_QUERY_TOKENS} ({MAX_FLAT_QUERY_TOKENS_WIDE}) tokens per query")
    if device.type == "cuda" and all(q.device.type == "cpu" for q in qs):
        # no torch CPU op on the way: pad_sequence's parallel loop costs tens of ms on a 128-thread host when it runs
        # between other multi-threaded work (measured 24 ms for 100 x 32 x 128), a memset + one memcpy per query costs 0.1
        l_max = max(int(q.shape[0]) for q in qs)
        if l_max == 0:
            padded = torch.zeros((len(qs), 0, dim), dtype=qs[0].dtype, device=device)
        else:
            padded = _staging_q.upload(qs, dim, device, slot_rows=l_max).view(len(qs), l_max, dim)
    else:
        padded = torch.nn.utils.rnn.pad_sequence(list(qs), batch_first=True, padding_value=0).to(device, non_blocking=True)
    return _widen(padded).contiguous()
