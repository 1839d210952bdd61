"""CPU tests of the host-side mirror: packing, blocking semantics, argument checks, ABI surface.

No compute call is made here (there is no GPU); the HIP library is only loaded and its
exported symbols compared with include/maxsim.h.
"""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import colpali_amd
from colpali_amd import corpus as C
from oracle import maxsim_oracle as mo

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def bf(n, dim=128, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, dim, generator=g).to(torch.bfloat16)


def test_abi_library_loads_and_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "maxsim.h")).read()
    declared = set(re.findall(r"\b(msim_[a-z0-9_]+)\s*\(", header))
    assert {"msim_abi_version", "msim_last_error", "msim_fwd", "msim_pairs_argmax", "msim_pairs_bwd", "msim_topk_f32"} <= declared
    lib = ctypes.CDLL(colpali_amd._lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/maxsim.h but not exported"
    lib.msim_abi_version.restype = ctypes.c_int
    m = re.search(r"#define MSIM_ABI_VERSION (\d+)", header)
    assert lib.msim_abi_version() == int(m.group(1))
    # ... and exports nothing else: the entry points ARE the header (measurement aids live in tools/probe/, include/maxsim_probe.h)
    import subprocess

    nm = subprocess.run(["nm", "-D", "--defined-only", colpali_amd._lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if ln.split()[-2:-1] == ["T"] and ln.split()[-1].startswith("msim_")}
    assert exported == declared, (sorted(exported - declared), sorted(declared - exported))
    assert not any("probe" in s for s in exported)


def test_probe_library_is_separate_from_the_product():
    """bench.py's ceilings come from tools/probe/libmaxsim_probe.so (include/maxsim_probe.h); it exports what that header declares."""
    from tools import probe

    header = open(os.path.join(ROOT, "include", "maxsim_probe.h")).read()
    declared = set(re.findall(r"\b(msim_[a-z0-9_]+)\s*\(", header))
    assert declared == {"msim_probe_last_error", "msim_probe_stream", "msim_probe_mfma"}
    if not os.path.exists(probe.LIB_PATH):
        pytest.skip("tools/probe/libmaxsim_probe.so not built")
    lib = ctypes.CDLL(probe.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name)
    src = "".join(open(os.path.join(ROOT, "colpali_amd", f)).read() for f in os.listdir(os.path.join(ROOT, "colpali_amd")) if f.endswith(".py"))
    assert "probe" not in src            # the package never loads it


def test_abi_rejects_bad_arguments_without_touching_a_gpu():
    L = colpali_amd._lib.lib()
    # a row that is not a multiple of 32 bytes, or above 4 KiB -> MSIM_EUNSUPPORTED before any device work
    rc = L.msim_fwd(0, 16, 1, 32, 16, 16, None, 1, 100, 16, 1, 0, None, None)
    assert rc == -2 and b"dim" in L.msim_last_error()
    rc = L.msim_fwd(2, 16, 1, 32, 16, 16, None, 1, 2048, 16, 1, 0, None, None)
    assert rc == -2 and b"dim" in L.msim_last_error()
    rc = L.msim_fwd(0, None, 1, 32, 16, 16, None, 1, 128, 16, 1, 0, None, None)
    assert rc == -1
    rc = L.msim_fwd(1, 16, 1, 32, 16, 16, None, 4, 128, 16, 2, 0, None, None)   # ld < n_d
    assert rc == -1
    assert L.msim_fwd(0, 16, 0, 32, 16, 16, None, 4, 128, 16, 4, 0, None, None) == 0   # empty problem is a no-op
    assert L.msim_fwd(7, 16, 1, 32, 16, 16, None, 1, 128, 16, 1, 0, None, None) == -2   # unknown dtype code
    assert L.msim_topk_f32(16, None, 1, 10, 10, 2000, 0, 16, 16, None, None) == -2      # k too large
    with pytest.raises(NotImplementedError):
        colpali_amd._lib.check(-2, "x")
    with pytest.raises(ValueError):
        colpali_amd._lib.check(-1, "x")


def test_block_clamp0_matches_oracle_definition():
    rng = np.random.default_rng(0)
    for n, bs in [(1, 128), (7, 3), (128, 128), (129, 128), (300, 7), (10, 1)]:
        lens = rng.integers(1, 50, size=n)
        got = C.block_clamp0(torch.from_numpy(lens), bs).numpy()
        np.testing.assert_array_equal(got, mo.block_clamp0(lens, bs))


def test_pack_passages_list_layout_and_flags():
    ps = [bf(5, seed=1), bf(9, seed=2), bf(9, seed=3), bf(2, seed=4)]
    pc = C.pack_passages(ps, torch.device("cpu"), batch_size=2)
    assert len(pc) == 4 and pc.blob.shape == (25, 128) and pc.blob.dtype == torch.bfloat16
    assert pc.offsets.tolist() == [0, 5, 14, 23, 25] and pc.offsets.dtype == torch.int32
    assert pc.clamp0.tolist() == [1, 0, 0, 1]
    assert torch.equal(pc.blob[5:14], ps[1])
    # equal lengths inside every block: no flag array at all
    assert C.pack_passages([bf(4), bf(4)], torch.device("cpu")).clamp0 is None
    assert C.pack_passages(ps, torch.device("cpu"), batch_size=None).clamp0 is None


def test_pack_passages_tensor_keeps_physical_zero_rows():
    p = torch.stack([bf(6, seed=1), bf(6, seed=2)])
    p[0, 4:] = 0
    pc = C.pack_passages(p, torch.device("cpu"))
    assert pc.clamp0 is None and pc.offsets.tolist() == [0, 6, 12]
    assert torch.equal(pc.blob.view(2, 6, 128), p)


def test_pack_queries_pads_with_zero_rows():
    q = C.pack_queries([bf(3, seed=1), bf(7, seed=2)], torch.device("cpu"))
    assert q.shape == (2, 7, 128) and torch.count_nonzero(q[0, 3:]) == 0


def test_unsupported_dtype_and_width_are_errors_not_silent_conversions():
    with pytest.raises(NotImplementedError, match="dtype"):
        C.pack_queries([torch.randn(3, 128, dtype=torch.float64)], torch.device("cpu"))
    with pytest.raises(NotImplementedError, match="4 KiB"):
        C.pack_passages([torch.randn(3, 1100)], torch.device("cpu"))
    with pytest.raises(RuntimeError, match="one embedding width"):
        C.pack_passages([bf(3, dim=64), bf(3, dim=128)], torch.device("cpu"))


def test_generic_widths_are_zero_padded_to_32_byte_rows():
    # fp32 keeps its dtype (reference tests use fp32, dim=32: tests/utils/test_processing_utils.py:8);
    # widths that are not a multiple of 32 bytes get zero columns, which change no dot product
    q = C.pack_queries([torch.randn(3, 32), torch.randn(5, 32)], torch.device("cpu"))
    assert q.dtype == torch.float32 and q.shape == (2, 5, 32)
    pc = C.pack_passages([bf(4, dim=100, seed=1), bf(6, dim=100, seed=2)], torch.device("cpu"))
    assert pc.blob.shape == (10, 112) and torch.count_nonzero(pc.blob[:, 100:]) == 0
    assert torch.equal(pc.blob[:4, :100], bf(4, dim=100, seed=1))
    assert colpali_amd._lib.kernel_width(128, torch.bfloat16) == 128
    assert colpali_amd._lib.kernel_width(320, torch.bfloat16) == 320
    assert colpali_amd._lib.kernel_width(30, torch.float32) == 32


def test_empty_inputs_raise_like_the_reference_and_a_gpu_request_is_never_served_elsewhere():
    with pytest.raises(ValueError, match="No queries provided"):
        colpali_amd.score_multi_vector([], [bf(2)])
    with pytest.raises(ValueError, match="No passages provided"):
        colpali_amd.score_multi_vector([bf(2)], [])
    # device="cpu" is the caller's choice and is served by the library's host path (tests/test_host_path.py); a GPU request on a host
    # without a GPU fails loudly -- it is never quietly computed somewhere else
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no ROCm GPU"):
            colpali_amd.score_multi_vector([bf(2)], [bf(2)], device="cuda:0")
    with pytest.raises(RuntimeError, match="MI355X only"):
        colpali_amd.similarity_matrix(bf(2), bf(2))


def test_all_empty_passage_block_raises_like_the_reference():
    with pytest.raises(RuntimeError, match="non-zero size"):
        C.pack_passages([bf(0), bf(0), bf(3)], torch.device("cpu"), batch_size=2)


def test_get_torch_device_policy():
    assert colpali_amd.get_torch_device("cpu") == "cpu"
    assert colpali_amd.get_torch_device("auto") == ("cuda:0" if torch.cuda.is_available() else "cpu")


def test_shard_range_partitions_exactly():
    for n, w in [(10, 3), (1_000_000, 8), (5, 8), (0, 2)]:
        spans = [colpali_amd.shard_range(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "colpali_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("no oracle", ""), f"{f} mentions the oracle"


def test_all_pairs_list_and_document_order():
    """The dense-gradient losses hand the kernels the row-major all-pairs list and its by-document order (loss.py)."""
    from colpali_amd import loss

    B, C = 3, 5
    pairs = loss._all_pairs(B, C, torch.device("cpu"))
    assert pairs.dtype == torch.int32 and pairs.shape == (B * C, 2)
    assert pairs.tolist() == [[b, c] for b in range(B) for c in range(C)]
    order = loss._all_pairs_order(B, C, torch.device("cpu"))
    want = torch.sort(pairs[:, 1].to(torch.int64), stable=True).indices.to(torch.int32)     # what the sparse path computes
    assert torch.equal(order, want)
    assert loss._all_pairs(B, C, torch.device("cpu")) is pairs                              # cached


def test_topk_workspace_plan_is_consistent():
    """msim_topk_workspace_bytes follows the level plan of msim_topk_f32: none when one workgroup finishes a row, otherwise the
    ping-pong buffers of the first two levels (12 bytes per surviving candidate), monotone in the row length."""
    import ctypes
    from colpali_amd import _lib

    L = _lib.lib()

    def later_seg(k):
        s = 512
        while s < 4 * k:
            s *= 2
        return s

    def first_seg(n_q, n, k):
        s = later_seg(k)
        while s < 4096 and n_q * -(-n // s) > 1024:
            s *= 2
        return s

    def is_last(n, seg):
        return n <= 2 * seg and n <= 4096

    def a16(x):
        return (x + 15) // 16 * 16

    for n_q, n, k in [(1, 100, 10), (4, 125000, 10), (4, 125000, 100), (1000, 125000, 10), (1, 1_000_000, 1000), (7, 4097, 1), (3, 1024, 10)]:
        got = L.msim_topk_workspace_bytes(n_q, n, k)
        s0, s1 = first_seg(n_q, n, k), later_seg(k)
        if is_last(n, s0):
            want = 0
        else:
            na = -(-n // s0) * k
            nb = 0 if is_last(na, s1) else -(-na // s1) * k
            want = a16(n_q * na * 4) + a16(n_q * na * 8) + a16(n_q * nb * 4) + a16(n_q * nb * 8)
        assert got == want, (n_q, n, k, got, want)
    assert L.msim_topk_workspace_bytes(4, 125000, 5000) == 0     # k above the kernel's limit: rejected, no plan


def test_bounded_staging_upload_reproduces_cat_and_padding(monkeypatch):
    """The drop-in's host -> device upload goes through a pinned buffer of bounded size in double-buffered chunks; chunk
    borders fall inside passages.  Exercised on the CPU with the two CUDA-only pieces (pinning, events) stubbed."""
    class _Ev:
        def synchronize(self):
            pass

        def record(self, stream):
            pass

    real_empty = torch.empty
    monkeypatch.setattr(torch.cuda, "Event", _Ev)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda d=None: None)
    monkeypatch.setattr(torch, "empty", lambda *a, **k: real_empty(*a, **{kk: v for kk, v in k.items() if kk != "pin_memory"}))
    monkeypatch.setattr(C, "STAGING_BYTES", 2 << 20)
    st = C._Staging()
    ps = [bf(n, seed=n) for n in (5000, 1, 0, 3000, 7000, 12, 9000)]          # 6 MiB through two 1 MiB halves
    out = st.upload(ps, 128, torch.device("cpu"))
    assert torch.equal(out, torch.cat(ps)) and st.buf.numel() == 2 << 20
    qs = [bf(n, seed=n) for n in (5, 32, 17)] * 300
    out = st.upload(qs, 128, torch.device("cpu"), slot_rows=32)
    assert torch.equal(out, torch.nn.utils.rnn.pad_sequence(qs, batch_first=True).reshape(-1, 128))
    assert st.buf.numel() == 2 << 20                                           # never grown past the cap


def test_passage_ranges_cut_at_block_multiples():
    from colpali_amd.scoring import passage_ranges

    ps = [bf(10)] * 10                                                        # 2560 bytes each
    assert passage_ranges(ps, 4, 10**9) == [(0, 10)]
    assert passage_ranges(ps, 4, 2560 * 8) == [(0, 8), (8, 10)]
    assert passage_ranges(ps, 4, 2560 * 5) == [(0, 4), (4, 8), (8, 10)]
    assert passage_ranges(ps, 4, 1) == [(0, 4), (4, 8), (8, 10)]             # a block above the budget is still one range
    t = torch.zeros(7, 3, 128, dtype=torch.bfloat16)
    assert passage_ranges(t, 2, 3 * 256 * 4) == [(0, 4), (4, 7)]


def test_bench_refuses_more_ranks_than_visible_gpus():
    """`python bench.py --gpus 2` must never degrade silently to one rank: without two visible GPUs (none here) and without
    the explicit BENCH_SHARE_GPU=1 plumbing override it exits non-zero before spawning anything."""
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2 and "refusing" in r.stderr and r.stdout.strip() == ""
    # a launcher that started the wrong number of ranks is an error too
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       env=env2, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_bench_self_launch_builds_one_rank_per_gpu_on_the_loopback():
    """What `python bench.py --gpus 8` executes when no launcher started it (the driver's 8-GPU run may come either way): the
    command and environment, without starting anything."""
    import importlib.util
    import sys

    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cmd, env = bench.rank_launch_command(8, ["--gpus", "8", "--steps", "5", "--warmup", "2"], 29511, {"PATH": "/usr/bin"})
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-7] == os.path.join(ROOT, "bench.py") and cmd[-6:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    assert env["BENCH_SELF_LAUNCHED"] == "1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and int(env["OMP_NUM_THREADS"]) >= 1
    # an 8-GPU request on a box with fewer GPUs is refused like the 2-GPU one (exit code 2, nothing on stdout)
    import subprocess

    clean = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True, env=clean, timeout=300)
    assert r.returncode == 2 and "refusing" in r.stderr and r.stdout.strip() == ""
    # the multi-GPU line runs a bounded regime list (eight ranks, each with its own oracle sample, inside the driver's clock)
    assert len(bench.REGIMES_MULTI.split(",")) <= 5 and "1000" in bench.REGIMES_MULTI.split(",")
    for spec_ in bench.REGIMES_SINGLE.split(",") + bench.REGIMES_MULTI.split(","):
        n, lens, label = bench.parse_regime(spec_, 32)
        assert n == len(lens) and min(lens) >= 1


def test_native_host_packing_drops_exactly_the_all_zero_rows():
    """msim_host_count_nonzero_rows / msim_host_gather_nonzero_rows (the host side of the flat query layout: include/maxsim.h): rows
    whose bytes are all zero are dropped, everything else -- a row with one non-zero element, with -0.0 -- keeps its order."""
    import numpy as np

    L = colpali_amd._lib.lib()
    g = torch.Generator().manual_seed(1)
    qs = [torch.randn(n, 128, generator=g).to(torch.bfloat16) for n in (5, 0, 9, 3)]
    qs[0][1] = 0
    qs[0][4] = 0
    qs[2][:4] = 0                       # left padding
    qs[2][6, 17] = 0                    # one zero element does not make a zero row
    qs[3][:] = 0
    qs[3][2, 0] = -0.0                  # -0.0 has a non-zero byte: kept (dropping it would be just as exact; the test pins the rule)
    keep = [q.contiguous() for q in qs]
    srcs = np.asarray([q.data_ptr() if q.numel() else 0 for q in keep], dtype=np.uint64)
    rows = np.asarray([q.shape[0] for q in keep], dtype=np.int64)
    counts = np.zeros(len(keep), dtype=np.int32)
    assert L.msim_host_count_nonzero_rows(srcs.ctypes.data, rows.ctypes.data, 256, len(keep), counts.ctypes.data, 3) == 0
    want_rows = [[0, 2, 3], [], [4, 5, 6, 7, 8], [2]]
    assert counts.tolist() == [len(w) for w in want_rows]
    off = np.zeros(len(keep) + 1, dtype=np.int64)
    np.cumsum(counts, out=off[1:])
    dst = torch.full((int(off[-1]) + 1, 128), 7.0).to(torch.bfloat16)
    assert L.msim_host_gather_nonzero_rows(dst.data_ptr(), srcs.ctypes.data, rows.ctypes.data, 256, off[:-1].copy().ctypes.data,
                                           len(keep), 2) == 0
    want = torch.cat([q[w] for q, w in zip(keep, want_rows) if w])
    assert torch.equal(dst[:-1].view(torch.int16), want.view(torch.int16)) and bool((dst[-1] == 7.0).all())
    assert L.msim_host_count_nonzero_rows(srcs.ctypes.data, rows.ctypes.data, 0, len(keep), counts.ctypes.data, 1) == -1
    assert L.msim_host_count_nonzero_rows(None, rows.ctypes.data, 256, len(keep), counts.ctypes.data, 1) == -1


def test_flat_plan_errors_and_limits_without_a_gpu():
    """msim_fwd_ragged validates on the host before any device work: unsupported shapes, broken offsets."""
    import numpy as np

    L = colpali_amd._lib.lib()
    off = np.asarray([0, 12, 52], dtype=np.int32)
    args = lambda dtype=0, dim=128, o=off, n_q=2: (dtype, 16, 16, o.ctypes.data, n_q, 16, 16, None, 5, dim, 16, 5, 0, None, None)  # noqa: E731
    assert L.msim_fwd_ragged(*args(dtype=2)) == -2 and L.msim_fwd_ragged(*args(dim=64)) == -2
    assert L.msim_fwd_ragged(*args(o=np.asarray([1, 12, 52], dtype=np.int32))) == -1
    assert L.msim_fwd_ragged(*args(o=np.asarray([0, 12, 5], dtype=np.int32))) == -1
    assert L.msim_fwd_ragged(*args(n_q=0)) == 0                                    # empty problem: a no-op
    assert L.msim_fwd_ragged_workspace_bytes(0, off.ctypes.data, 2, 5, 128) == 0   # K1s: no scratch


def test_the_launch_plan_is_the_ladder_design_md_describes():
    """msim_fwd_plan (host-only): the kernel shape per batch -- DESIGN.md 3.0's ladder in 16-token units, the balanced multi-block plans,
    the eight-vs-ten-units cost rule, the 64-queries-per-block limit, long queries as 128-token pieces."""
    import numpy as np

    L = colpali_amd._lib.lib()

    def plan(lens=None, n_q=0, lq=0):
        out = np.zeros(5, dtype=np.int32)
        if lens is not None:
            off = np.zeros(len(lens) + 1, dtype=np.int32)
            np.cumsum(np.asarray(lens, dtype=np.int32), out=off[1:])
            rc = L.msim_fwd_plan(off.ctypes.data, len(lens), 0, out.ctypes.data)
        else:
            rc = L.msim_fwd_plan(None, n_q, lq, out.ctypes.data)
        assert rc == 0, L.msim_last_error()
        return tuple(int(v) for v in out)

    # (kernel, units | waves, max units per wave, blocks, units of the heaviest wave)
    assert plan(n_q=1, lq=32) == (0, 2, 0, 1, 2) and plan(n_q=4, lq=32) == (0, 8, 0, 1, 8)         # K1s: the headline is 8 units
    assert plan(n_q=4, lq=20) == (0, 5, 0, 1, 5) and plan(lens=[25, 25, 25, 25]) == (0, 7, 0, 1, 7)  # real lengths: fewer units
    assert plan(n_q=4, lq=40) == (1, 2, 8, 1, 5)                                                    # 10 units: the pair form, 5 + 5
    assert plan(n_q=8, lq=32) == (1, 2, 8, 1, 8)                                                    # pair form
    assert plan(n_q=9, lq=32) == (1, 4, 5, 1, 5) and plan(n_q=10, lq=32) == (1, 4, 5, 1, 5)          # round 4: 17..20 units on four waves of <= 5 units,
    assert plan(n_q=8, lq=40) == (1, 4, 5, 1, 5) and plan(n_q=11, lq=32) == (1, 4, 8, 1, 6)          # three workgroups per CU (168 registers); 22 units: the 8-unit kernel
    assert plan(n_q=16, lq=32) == (1, 4, 8, 1, 8) and plan(n_q=20, lq=32) == (1, 4, 10, 1, 10)
    assert plan(n_q=32, lq=32) == (1, 8, 8, 1, 8) and plan(n_q=40, lq=32) == (1, 8, 10, 1, 10)
    assert plan(n_q=1000, lq=32) == (1, 8, 8, 32, 8)                                                # 32 blocks: one round of an XCD's CUs
    assert plan(n_q=1000, lq=40) == (1, 8, 10, 32, 10) and plan(n_q=1000, lq=20) == (1, 8, 8, 20, 8)     # 40 blocks of 64 units would take two rounds and read the corpus twice: 32 ten-unit blocks, one round
    assert plan(n_q=1200, lq=32) == (1, 8, 8, 38, 8) and plan(n_q=1280, lq=32) == (1, 8, 10, 32, 10)   # ... only when that round is full: 30 ten-unit blocks would idle two CUs per XCD
    assert plan(n_q=80, lq=32) == (1, 8, 8, 3, 7)      # three blocks of 27/27/26 queries: 7 units on the heaviest wave (round 3's 32-token
    #                                                     tiles made that 4 tiles = 8 units, and two ten-unit blocks won: 3 x 7 < 1.1 x 2 x 10 now)
    assert plan(n_q=9, lq=8) == (1, 2, 8, 1, 3)                                                     # more than 8 queries never go to K1s
    assert plan(lens=[7] * 70)[3] == 2 and plan(lens=[13] * 64) == (1, 8, 8, 1, 7)                  # 64 queries per block at most
    g = np.random.default_rng(0)
    k, nw, maxu, nb, heavy = plan(lens=g.integers(12, 49, 1000).tolist())
    assert (k, nw, maxu) == (1, 8, 8) and 29 <= nb <= 31 and heavy == 8                             # ~30 balanced blocks of whole queries
    assert plan(n_q=3, lq=780) == (1, 8, 8, 3, 7)                                                   # 21 pieces of 128 tokens: 8 + 8 + 5 per block
    out = np.zeros(5, dtype=np.int32)
    assert L.msim_fwd_plan(np.asarray([0, 1300], dtype=np.int32).ctypes.data, 1, 0, out.ctypes.data) == -2   # one query above a block


def test_bench_counts_real_tokens_only():
    """bench.py's roofline arithmetic for ragged batches: FLOP and bytes are those of the REAL query tokens -- padding an
    implementation adds is never credited (round-3 review: every measured leg used Lq = 32, where that cannot show)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_under_test2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n, lens, label = bench.parse_regime("1000xr12-48", 32)
    assert n == 1000 and label == "U{12..48}" and min(lens) >= 12 and max(lens) <= 48
    assert bench.parse_regime("1000xr12-48", 32)[1] == lens                      # seeded: the same batch every run
    r = bench.regime_numbers(n, 32, 125000, 1024, 600.0, q_tokens=sum(lens))
    assert r["flops_per_launch"] == 2.0 * sum(lens) * 125000 * 1024 * 128
    assert r["bound"] == "mfma" and abs(r["frac"] - r["flops_per_launch"] / 0.6 / 1e12 / 2500.0) < 1e-12
    u = bench.regime_numbers(4, 32, 125000, 1024, 4.8)                          # the headline: HBM-bound, bytes = corpus + queries + scores
    assert u["bound"] == "hbm" and u["algorithmic_bytes_per_launch"] == 125000 * 1024 * 256 + 4 * 32 * 256 + 4 * 125000 * 4


def test_launch_plans_of_random_batches_respect_every_capacity():
    """Invariants of flat_plan on 400 random batches (host-only): a wave never holds more units than the kernel is compiled for, the
    blocks cover the tokens and respect the 8-lanes-per-query limit, K1s only takes <= 8 queries in <= 8 units."""
    import numpy as np

    L = colpali_amd._lib.lib()
    rng = np.random.default_rng(7)
    out = np.zeros(5, dtype=np.int32)
    for _ in range(400):
        n_q = int(rng.integers(1, 300))
        hi = int(rng.choice([1, 8, 20, 48, 130, 700]))
        lens = rng.integers(0, hi + 1, n_q).astype(np.int32)
        off = np.zeros(n_q + 1, dtype=np.int32)
        np.cumsum(lens, out=off[1:])
        assert L.msim_fwd_plan(off.ctypes.data, n_q, 0, out.ctypes.data) == 0, L.msim_last_error()
        kernel, a, maxu, blocks, heavy = (int(v) for v in out)
        units = -(-int(off[-1]) // 16)
        if kernel == 0:
            assert n_q <= 8 and units <= 8 and a == max(units, 1) and blocks == 1
        else:
            nw = a
            assert nw in (2, 4, 8) and maxu in (5, 8, 10) and 1 <= heavy <= maxu
            assert blocks * nw * maxu >= units and blocks * nw * 8 >= n_q          # capacity in units and in queries
            assert blocks == 1 or nw == 8                                            # several blocks only on the 8-wave form


def test_loss_offset_and_pair_checks_run_before_any_device_work():
    from colpali_amd import loss as Lm

    with pytest.raises(IndexError):
        Lm._check_offset(4, 6, 3)
    Lm._check_offset(4, 7, 3)
    with pytest.raises(ValueError):
        Lm._check_pairs(torch.tensor([[0, 0], [1, 6]], dtype=torch.int32), 4, 6)
    with pytest.raises(ValueError):
        Lm._check_pairs(torch.tensor([[1, 0], [0, 1]], dtype=torch.int32), 4, 6)          # not sorted by query
    Lm._check_pairs(torch.tensor([[0, 5], [3, 0]], dtype=torch.int32), 4, 6)


def test_native_host_gather_copies_every_buffer_to_its_offset():
    """msim_host_gather (include/maxsim.h): the drop-in's staging memcpy, native and multi-threaded; no device involved."""
    L = colpali_amd._lib.lib()
    rng = np.random.default_rng(3)
    bufs = [rng.integers(0, 255, size=n, dtype=np.uint8) for n in (5 << 20, 0, 1, 3 << 20, 7, 9 << 20)]
    offs = np.cumsum([0] + [b.size + 5 for b in bufs[:-1]]).astype(np.int64)          # gaps of 5 bytes stay untouched
    dst = np.full(int(offs[-1]) + bufs[-1].size + 3, 0xEE, dtype=np.uint8)
    srcs = np.array([b.ctypes.data if b.size else 0 for b in bufs], dtype=np.uint64)
    sizes = np.array([b.size for b in bufs], dtype=np.int64)
    for threads in (1, 8, 64):
        dst[:] = 0xEE
        assert L.msim_host_gather(dst.ctypes.data, srcs.ctypes.data, offs.ctypes.data, sizes.ctypes.data, len(bufs), threads) == 0
        for b, o in zip(bufs, offs):
            np.testing.assert_array_equal(dst[o : o + b.size], b)
            if b.size:
                assert dst[o + b.size] == 0xEE
    assert L.msim_host_gather(dst.ctypes.data, None, offs.ctypes.data, sizes.ctypes.data, len(bufs), 4) == -1
    assert L.msim_host_gather(dst.ctypes.data, srcs.ctypes.data, offs.ctypes.data, sizes.ctypes.data, 0, 4) == 0


def test_measurement_tools_parse():
    """tools/*.py and tools/*.sh are the scripts behind every log under profiles/: they must at least parse."""
    import ast
    import glob
    import subprocess

    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools")
    scripts = sorted(glob.glob(os.path.join(root, "*.py")))
    assert scripts
    for path in scripts:
        with open(path) as f:
            ast.parse(f.read(), filename=path)
    for path in sorted(glob.glob(os.path.join(root, "*.sh"))):
        assert subprocess.run(["bash", "-n", path]).returncode == 0, path


def test_fwd_workspace_bytes_reports_scratch_exactly_when_several_query_blocks_share_a_range():
    """Round-2 advisor finding: msim_fwd_workspace_bytes must agree with the launch path's use of the workspace (the convoy
    counters of K1b: needed whenever the plan has more than one query block).  Host-only function, no GPU needed."""
    L = colpali_amd._lib.lib()

    def ws(n_q, lq, dtype=0, dim=128):
        return L.msim_fwd_workspace_bytes(dtype, n_q, lq, 1000, dim)

    # the plan (maxsim_abi.hip: flat_plan) works in 16-token units of the flat token matrix: one query block holds up to
    # 256 / 320 / 512 / 640 / 1024 / 1280 tokens (pair, 4 waves x 5 units, 4 waves, 4 x 10, 8 waves, 8 x 10) and 16 / 32 / 64 queries
    assert ws(1, 32) == 0 and ws(4, 32) == 0                       # K1s
    assert ws(8, 32) == 0 and ws(16, 32) == 0 and ws(32, 32) == 0  # one query block (pair / 4-wave / 8-wave form)
    assert ws(17, 32) == 0 and ws(20, 32) == 0 and ws(33, 32) == 0 and ws(40, 32) == 0   # ONE block: ten units per wave (round 3's five tiles)
    assert ws(41, 32) == 4096 and ws(64, 32) == 4096 and ws(80, 32) == 4096 and ws(1000, 32) == 4096   # several blocks
    # 96-token queries are 6 units each, wherever they sit: 5 / 9 / 10 / 13 of them still fit one block (round 3: a wave held whole
    # queries of 32-token tiles, and 5 of them were already two blocks)
    assert ws(5, 96) == 0 and ws(9, 96) == 0 and ws(10, 96) == 0 and ws(13, 96) == 0 and ws(14, 96) == 4096
    assert ws(17, 64) == 0 and ws(20, 64) == 0 and ws(21, 64) == 4096
    assert ws(20, 40) == 0 and ws(32, 40) == 0 and ws(33, 40) == 4096      # real query lengths: 32 x 40 tokens = 1280 = one 8 x 10 block
    assert ws(65, 8) == 4096 and ws(64, 8) == 0                            # the 64-queries-per-block limit (8 lanes per query in the reduction)
    # ragged queries: the same plan from the host copy of the token offsets
    import numpy as np

    def wsr(lens, dtype=0, dim=128):
        off = np.zeros(len(lens) + 1, dtype=np.int32)
        np.cumsum(np.asarray(lens, dtype=np.int32), out=off[1:])
        return L.msim_fwd_ragged_workspace_bytes(dtype, off.ctypes.data, len(lens), 1000, dim)

    assert wsr([12, 40, 33]) == 0 and wsr([40] * 32) == 0 and wsr([40] * 32 + [1]) == 4096 and wsr([0] * 64 + [5]) == 4096
    assert wsr([32] * 100, dtype=2) == 0 and wsr([32] * 100, dim=320) == 0      # not the flat path's shapes
    assert ws(100, 32, dtype=2) == 0 and ws(100, 32, dim=320) == 0 and ws(100, 200, dtype=2) == 0   # generic / panel kernels: none
    # queries longer than 128 tokens (16-bit, width 128): 128-token segments on K1b -- counters + n_q x segments x n_d partial sums
    assert ws(100, 200) == 4096 + 100 * 2 * 1000 * 4 and ws(1, 780) == 4096 + 7 * 1000 * 4 and ws(3, 129, dtype=1) == 4096 + 3 * 2 * 1000 * 4


def test_the_shipped_library_never_reads_the_environment():
    """A/B knobs and trace hooks are compiled into measurement builds only (make ab / make trace)."""
    import subprocess

    out = subprocess.run(["nm", "-D", "--undefined-only", colpali_amd._lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in out
    blob = open(colpali_amd._lib.LIB_PATH, "rb").read()
    assert b"MSIM_HEAD_VARIANT" not in blob and b"MSIM_BATCH_NW" not in blob and b"TRACE_PTR" not in blob


def test_library_override_is_confined_to_measurement_builds(monkeypatch, tmp_path):
    monkeypatch.setenv("COLPALI_AMD_LIB", str(tmp_path / "evil.so"))
    with pytest.raises(RuntimeError, match="measurement builds"):
        colpali_amd._lib._lib_path()
    monkeypatch.setenv("COLPALI_AMD_LIB", os.path.join(ROOT, "tools", "_ab", "libmaxsim_ab.so"))
    assert colpali_amd._lib._lib_path().endswith(os.path.join("tools", "_ab", "libmaxsim_ab.so"))
    monkeypatch.delenv("COLPALI_AMD_LIB")
    assert colpali_amd._lib._lib_path().endswith(os.path.join("csrc", "libmaxsim_gfx950.so"))


def test_loss_modules_survive_deepcopy_and_pickle_with_reports_in_flight():
    import copy
    import pickle
    import threading

    m = colpali_amd.ColbertPairwiseCELoss(temperature=0.5)
    m.__dict__["_bounds_pending"] = [(threading.Lock(), threading.Lock())]     # unpicklable stand-ins for (pinned tensor, event)
    for clone in (copy.deepcopy(m), pickle.loads(pickle.dumps(m))):
        assert clone.temperature == 0.5 and "_bounds_pending" not in clone.__dict__
        assert clone.state_dict() == {}                                        # zero state, like the reference (:27)
    assert len(m.__dict__["_bounds_pending"]) == 1


def test_stream_const_cache_evicts_oldest_first_and_is_bounded():
    cache = colpali_amd._lib.StreamConstCache(3)
    made = []
    for i in range(5):
        cache.get(i, "cpu", lambda i=i: made.append(i) or i)
    assert len(cache) == 3 and made == [0, 1, 2, 3, 4]
    assert cache.get(4, "cpu", lambda: "new") == 4 and cache.get(0, "cpu", lambda: "again") == "again"


def test_chunk_schedule_covers_the_image_with_small_edges():
    """The pipelined upload's chunks (colpali_amd/corpus.py): contiguous, every byte once, none above a staging half; images of more
    than two halves start and end with the small edge chunk."""
    from colpali_amd import corpus as C

    half = 32 << 20
    for total in (1, 5 << 20, half, 2 * half, 2 * half + 1, 263_680_000, 7 * half + 12345, 100 * half):
        sch = C._chunk_schedule(total, half)
        assert sch[0][0] == 0 and sch[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(sch, sch[1:]))
        assert all(0 < c1 - c0 <= half for c0, c1 in sch)
        if total > 2 * half and C._EDGE_CHUNK_BYTES:
            edge = min(C._EDGE_CHUNK_BYTES, half)
            assert sch[0][1] - sch[0][0] == edge and sch[-1][1] - sch[-1][0] == edge


def test_gpu_local_cpus_is_a_hint_that_never_fails(monkeypatch):
    """The NUMA-local host side of the drop-in (colpali_amd/_lib.py): cpulist parsing; without a GPU the context manager leaves the
    caller's affinity alone; COLPALI_AMD_NUMA=0 switches the hint off."""
    import os

    from colpali_amd import _lib

    assert _lib._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert _lib._parse_cpulist("") == set()
    before = os.sched_getaffinity(0)
    with _lib.on_gpu_local_cpus(torch.device("cuda:0")) as ctx:
        if not torch.cuda.is_available():
            assert ctx.cpus is None
        assert os.sched_getaffinity(0) == (ctx.cpus or before)
    assert os.sched_getaffinity(0) == before
