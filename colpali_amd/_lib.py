"""ctypes binding of colpali_amd/csrc/libmaxsim_gfx950.so (C ABI: include/maxsim.h).

The library is the product: there is no Python/torch/CPU fallback.  If it is not
built, or cannot be loaded, importing a scorer raises immediately.
"""
from __future__ import annotations

import ctypes
import os

import torch  # noqa: F401  -- must be imported first: it maps the HIP runtime (libamdhip64.so) we bind to

_HERE = os.path.dirname(os.path.abspath(__file__))
# COLPALI_AMD_LIB: load another build of the same ABI instead -- measurement builds under tools/_ab/ only (`make ab`, `make trace`,
# a kept previous build: A/B runs inside one gpurun); any other path is refused, the product loads the in-tree library
_AB_DIR = os.path.realpath(os.path.join(_HERE, "..", "tools", "_ab"))


def _lib_path() -> str:
    override = os.environ.get("COLPALI_AMD_LIB")
    if override:
        real = os.path.realpath(override)
        if os.path.dirname(real) != _AB_DIR:
            raise RuntimeError(f"COLPALI_AMD_LIB={override}: only measurement builds under {_AB_DIR} can replace the in-tree library")
        return real
    return os.path.join(_HERE, "csrc", "libmaxsim_gfx950.so")


LIB_PATH = _lib_path()

MSIM_FLAG_REF_ROUNDING = 0x1
ABI_VERSION = 20


def dtype_code(dtype) -> int:
    """MSIM_DTYPE_* code of a torch dtype the kernels take natively."""
    if dtype == torch.bfloat16:
        return 0
    if dtype == torch.float16:
        return 1
    if dtype == torch.float32:
        return 2
    raise NotImplementedError(f"dtype {dtype}: the gfx950 kernels take bfloat16, float16 or float32 embeddings")


def kernel_width(dim: int, dtype) -> int:
    """Physical row width (elements) the kernels need for a logical embedding width `dim`.

    bf16/f16 rows of 128 go to the tuned kernels as they are; every other shape goes to the generic kernels,
    whose rows are a multiple of 32 bytes: the width is padded with zero columns (no dot product changes)."""
    es = 4 if dtype == torch.float32 else 2
    if es == 2 and dim == 128:
        return dim
    per = 32 // es
    width = (dim + per - 1) // per * per
    if width * es > 4096:
        raise NotImplementedError(f"embedding rows of {dim} x {es} bytes exceed the 4 KiB the gfx950 kernels support")
    return width

_lib = None


class MaxSimLibraryError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MaxSimLibraryError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C colpali_amd/csrc`). colpali_amd has no CPU/torch fallback."
        )
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, u32, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_uint32, ctypes.c_size_t
    L.msim_abi_version.restype = i32
    L.msim_last_error.restype = ctypes.c_char_p
    L.msim_fwd_workspace_bytes.argtypes = [i32, i32, i32, i32, i32]
    L.msim_fwd_workspace_bytes.restype = sz
    L.msim_fwd.argtypes = [i32, vp, i32, i32, vp, vp, vp, i32, i32, vp, i64, u32, vp, vp]
    L.msim_fwd.restype = i32
    L.msim_fwd_host.argtypes = [i32, vp, i32, i32, vp, vp, vp, i32, i32, vp, i64, u32, i32]
    L.msim_fwd_host.restype = i32
    L.msim_host_last_error.restype = ctypes.c_char_p
    L.msim_fwd_host_lists.argtypes = [i32, vp, vp, i32, vp, vp, vp, i32, i32, vp, i64, u32, i32]
    L.msim_fwd_host_lists.restype = i32
    L.msim_sim_matrix_host.argtypes = [i32, vp, i32, vp, i32, i32, vp, i64, u32, i32]
    L.msim_sim_matrix_host.restype = i32
    L.msim_fwd_plan.argtypes = [vp, i32, i32, vp]
    L.msim_fwd_plan.restype = i32
    L.msim_fwd_ragged_workspace_bytes.argtypes = [i32, vp, i32, i32, i32]
    L.msim_fwd_ragged_workspace_bytes.restype = sz
    L.msim_fwd_ragged.argtypes = [i32, vp, vp, vp, i32, vp, vp, vp, i32, i32, vp, i64, u32, vp, vp]
    L.msim_fwd_ragged.restype = i32
    L.msim_query_compact.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp]
    L.msim_query_compact.restype = i32
    L.msim_host_count_nonzero_rows.argtypes = [vp, vp, i64, i64, vp, i32]
    L.msim_host_count_nonzero_rows.restype = i32
    L.msim_host_gather_nonzero_rows.argtypes = [vp, vp, vp, i64, vp, i64, i32]
    L.msim_host_gather_nonzero_rows.restype = i32
    L.msim_fwd_transposed.argtypes = [i32, vp, i32, i32, vp, i32, i32, i32, vp, i64, vp, vp]
    L.msim_fwd_transposed.restype = i32
    L.msim_dense_t_supported.argtypes = [i32, i32, i32, i32, i32, i32]
    L.msim_dense_t_supported.restype = i32
    L.msim_dense_t_route_bytes.argtypes = [i32, i32, i32]
    L.msim_dense_t_route_bytes.restype = sz
    L.msim_fwd_transposed_route.argtypes = [i32, vp, i32, i32, vp, i32, i32, i32, vp, i64, vp, vp, vp]
    L.msim_fwd_transposed_route.restype = i32
    L.msim_dense_t_bwd_workspace_bytes.argtypes = [i32, i32, i32, i32, i32]
    L.msim_dense_t_bwd_workspace_bytes.restype = sz
    L.msim_dense_t_bwd.argtypes = [i32, vp, i32, i32, vp, i32, i32, i32, vp, i64, vp, i32, vp, vp, vp, vp, vp]
    L.msim_dense_t_bwd.restype = i32
    L.msim_host_gather_range.argtypes = [vp, vp, vp, i64, i64, i64, i32]
    L.msim_host_gather_range.restype = i32
    L.msim_host_gather_range_begin.argtypes = [vp, vp, vp, i64, i64, i64, i32]
    L.msim_host_gather_range_begin.restype = i32
    L.msim_host_gather_range_wait.argtypes = []
    L.msim_host_gather_range_wait.restype = i32
    L.msim_host_threads_affinity.argtypes = [vp, i32]
    L.msim_host_threads_affinity.restype = i32
    L.msim_pairs_argmax.argtypes = [i32, vp, i32, i32, vp, vp, vp, i32, i32, i32, vp, i32, vp, vp, vp]
    L.msim_pairs_argmax.restype = i32
    L.msim_allpairs_argmax.argtypes = [i32, vp, i32, i32, vp, vp, vp, i32, i32, vp, i64, vp, vp]
    L.msim_allpairs_argmax.restype = i32
    L.msim_pairs_bwd.argtypes = [i32, vp, i32, i32, vp, vp, i32, i32, i32, vp, vp, vp, vp, i32, vp, i32, i32, vp, vp, vp, vp]
    L.msim_pairs_bwd.restype = i32
    L.msim_pairs_bwd_workspace_bytes.argtypes = [i32, i32, i32, i32, i32, i32]
    L.msim_pairs_bwd_workspace_bytes.restype = sz
    f32 = ctypes.c_float
    L.msim_smooth_fwd.argtypes = [i32, vp, i32, i32, vp, vp, i32, i32, f32, vp, i64, vp]
    L.msim_smooth_fwd.restype = i32
    L.msim_smooth_pairs.argtypes = [i32, vp, i32, i32, vp, vp, i32, i32, vp, i32, f32, vp, vp, vp]
    L.msim_smooth_pairs.restype = i32
    L.msim_smooth_bwd_workspace_bytes.argtypes = [i32, i32, i32]
    L.msim_smooth_bwd_workspace_bytes.restype = sz
    L.msim_smooth_pairs_bwd.argtypes = [i32, vp, i32, i32, vp, vp, i32, i32, i32, vp, vp, vp, vp, i32, f32, vp, vp, vp, vp]
    L.msim_smooth_pairs_bwd.restype = i32
    L.msim_embed_head.argtypes = [i32, vp, i64, i32, vp, vp, i32, vp, vp, i64, vp]
    L.msim_embed_head.restype = i32
    L.msim_embed_head_row_map.argtypes = [vp, i32, vp, i32, i64, vp, vp]
    L.msim_embed_head_row_map.restype = i32
    L.msim_embed_head_writer_map.argtypes = [vp, i32, vp, i32, i32, i32, vp, vp, vp, vp, vp]
    L.msim_embed_head_writer_map.restype = i32
    L.msim_embed_head_bwd.argtypes = [i32, vp, vp, vp, i64, i32, vp, vp]
    L.msim_embed_head_bwd.restype = i32
    L.msim_sim_matrix.argtypes = [i32, vp, i32, vp, i32, i32, vp, i64, u32, vp]
    L.msim_sim_matrix.restype = i32
    L.msim_pool_cluster.argtypes = [i32, vp, vp, i32, i32, i32, vp, i32, vp, vp, vp, vp, vp]
    L.msim_pool_cluster.restype = i32
    L.msim_pool_reduce.argtypes = [i32, vp, vp, i32, i32, i32, vp, vp, vp, i32, vp]
    L.msim_pool_reduce.restype = i32
    L.msim_loss_epilogue_workspace_bytes.argtypes = [i32, i32]
    L.msim_loss_epilogue_workspace_bytes.restype = sz
    L.msim_loss_epilogue.argtypes = [i32, vp, i64, i32, i32, vp, i32, i32, i32, i32, f32, i32, i32, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.msim_loss_epilogue.restype = i32
    L.msim_host_gather.argtypes = [vp, vp, vp, vp, i64, i32]
    L.msim_host_gather.restype = i32
    L.msim_topk_workspace_bytes.argtypes = [i32, i64, i32]
    L.msim_topk_workspace_bytes.restype = sz
    L.msim_topk_f32.argtypes = [vp, vp, i32, i64, i64, i32, i64, vp, vp, vp, vp]
    L.msim_topk_f32.restype = i32
    if L.msim_abi_version() != ABI_VERSION:
        raise MaxSimLibraryError(f"ABI version mismatch: library reports {L.msim_abi_version()}, binding expects {ABI_VERSION} "
                                 "(rebuild: make -C colpali_amd/csrc)")
    _lib = L
    return L


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().msim_last_error().decode("utf-8", "replace")
        if rc == -2:
            raise NotImplementedError(f"{what}: {msg}")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise MaxSimLibraryError(f"{what} failed (code {rc}): {msg}")


class StreamConstCache:
    """Small read-only device constants (index lists, row ids), cached per (key, device, stream).  An entry is only ever
    used on the stream it was allocated on, so evicting it -- oldest first, under a lock -- hands its memory back to the
    caching allocator in stream order; nothing is cleared wholesale under another thread's feet."""

    def __init__(self, capacity: int):
        import collections
        import threading

        self._d = collections.OrderedDict()
        self._lock = threading.Lock()
        self._cap = capacity

    def get(self, key, device, make):
        k = (key, str(device), torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0)
        with self._lock:
            t = self._d.get(k)
            if t is not None:
                self._d.move_to_end(k)
                return t
        t = make()
        with self._lock:
            self._d[k] = t
            while len(self._d) > self._cap:
                self._d.popitem(last=False)
        return t

    def __len__(self):
        return len(self._d)


def effective_cpus() -> int:
    """CPUs this process may really use: the scheduler affinity AND the container's CPU quota (cgroup v2 `cpu.max`, v1
    `cpu.cfs_quota_us / cpu.cfs_period_us`).  A GPU box reports 256 CPUs and grants 16: native thread counts taken from
    os.cpu_count() / torch.get_num_threads() (128 there) run the quota dry, and the kernel then freezes the whole process for the
    rest of the 100 ms period (round 5: profiles/r05_logs/dropin_stalls.log)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(p)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                quota = q / p
        except Exception:
            pass
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return max(1, n)


_gpu_cpus = {}


def _parse_cpulist(text: str):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_local_cpus(device):
    """CPUs of the NUMA node the GPU hangs off (`local_cpulist` of its PCI function in sysfs) that this process may run on, or None
    when the host has one node / sysfs does not say / COLPALI_AMD_NUMA=0.  The GPU boxes are two-socket hosts with four GPUs per
    socket and no cpuset on the container: a drop-in call whose gather threads and pinned staging buffer sit on the GPU's own socket
    takes 6.2 ms, one that the scheduler spread over both 7-8.5 ms (profiles/r05_logs/ab_dropin_knobs.log, host_topology.log)."""
    idx = device.index if getattr(device, "index", None) is not None else torch.cuda.current_device()
    if idx in _gpu_cpus:
        return _gpu_cpus[idx]
    cpus = None
    if os.environ.get("COLPALI_AMD_NUMA", "1") != "0":
        try:
            p = torch.cuda.get_device_properties(idx)
            bus = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
            local = _parse_cpulist(open(f"/sys/bus/pci/devices/{bus}/local_cpulist").read())
            allowed = os.sched_getaffinity(0)
            local &= allowed
            if len(local) >= 2 and local != allowed:
                cpus = frozenset(local)
        except Exception:
            cpus = None
    _gpu_cpus[idx] = cpus
    return cpus


class on_gpu_local_cpus:
    """Context manager: the calling thread runs on the GPU's own NUMA node inside the block (threads it starts there inherit the
    mask and KEEP it: the native pool's workers are started by their first parallel region; memory it first-touches there is
    node-local).  A hint only: any failure leaves the affinity as it was.
    SIDE EFFECT, by design and process-wide: a thread FIRST CREATED inside the block keeps the narrowed mask for its lifetime -- the
    library's own gather pool (wanted), but also e.g. torch's OpenMP pool if its first parallel region happens to run inside a
    score_multi_vector call.  A host that needs its CPU pools spread over both sockets sets COLPALI_AMD_NUMA=0 (no narrowing at all)
    or runs one torch CPU op before the first scoring call; staging buffers are per GPU (corpus._PerDevice)."""

    def __init__(self, device):
        self.cpus = gpu_local_cpus(device) if torch.cuda.is_available() else None
        self.saved = None

    def __enter__(self):
        if self.cpus is not None:
            try:
                saved = os.sched_getaffinity(0)
                if saved != self.cpus:
                    os.sched_setaffinity(0, self.cpus)
                    self.saved = saved
            except Exception:
                self.saved = None
        return self

    def __exit__(self, *exc):
        if self.saved is not None:
            try:
                os.sched_setaffinity(0, self.saved)
            except Exception:
                pass
        return False


_SYS_MOVE_PAGES = 279          # x86-64


def nodes_of_addresses(addrs) -> list:
    """NUMA node of the page behind every address (move_pages(2) with a NULL node list only asks); [] when the kernel does not say."""
    try:
        n = len(addrs)
        if n == 0:
            return []
        libc = ctypes.CDLL(None, use_errno=True)
        pages = (ctypes.c_void_p * n)(*[int(a) & ~4095 for a in addrs])
        status = (ctypes.c_int * n)()
        libc.syscall.restype = ctypes.c_long
        rc = libc.syscall(ctypes.c_long(_SYS_MOVE_PAGES), ctypes.c_int(0), ctypes.c_ulong(n), pages, ctypes.c_void_p(0), status, ctypes.c_int(0))
        if rc != 0:
            return []
        return [int(x) for x in status if x >= 0]
    except Exception:
        return []


_node_cpus = {}


def node_cpus(node: int):
    """CPUs of a NUMA node this process may run on (frozenset), or None."""
    if node in _node_cpus:
        return _node_cpus[node]
    cpus = None
    try:
        local = _parse_cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read()) & os.sched_getaffinity(0)
        if len(local) >= 2:
            cpus = frozenset(local)
    except Exception:
        cpus = None
    _node_cpus[node] = cpus
    return cpus


_gather_threads_on = [None]      # the CPU set the library's host threads were last moved to (None: wherever they were created)
GATHER_NODE_POLICY = os.environ.get("COLPALI_AMD_GATHER_NODE", "gpu")     # "gpu" | "pages" (tools/dropin_numa_pages.py flips it per run)


def place_gather_threads(device, sample_addrs) -> None:
    """The library's host threads (gather pool + driver thread) onto the NUMA node of the GPU THIS call targets -- explicitly, through
    msim_host_threads_affinity, so that a process that drives GPUs on both sockets moves them per call and nothing depends on which
    thread happened to start them (round-5 advisor finding).  Policy "pages" puts them next to the CALLER'S pages instead (one
    move_pages(2) query of a few addresses): the round-5 review's suggestion, measured in round 6 and NOT the default -- with the page
    tensors migrated to the other socket the call takes 6.5-7.0 ms with the threads on the GPU's node and 8.3 ms with the threads on
    the pages' node (config 2 geometry; pages on the GPU's node: 6.1 ms; profiles/r06_logs/dropin_numa_pages.log): a gather that
    WRITES the pinned staging buffer across the socket link loses more than one that reads the pages across it.  A hint only: any
    failure leaves the threads where they are."""
    if os.environ.get("COLPALI_AMD_NUMA", "1") == "0":
        return
    try:
        want = gpu_local_cpus(device)
        if GATHER_NODE_POLICY == "pages":
            nodes = nodes_of_addresses(sample_addrs)
            if nodes:
                node = max(set(nodes), key=nodes.count)
                want = node_cpus(node) or want
        if want is None or want == _gather_threads_on[0]:
            return
        arr = (ctypes.c_int32 * len(want))(*sorted(want))
        if lib().msim_host_threads_affinity(arr, len(want)) == 0:
            _gather_threads_on[0] = want
    except Exception:
        pass


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


def current_stream_handle(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream
