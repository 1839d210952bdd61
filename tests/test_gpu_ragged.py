"""GPU parity of the FLAT query layout (round 4): ragged query lengths in every shape the launch plan can pick, zero rows
dropped, scores independent of the batch a query is scored in.  Checker: the C oracle (oracle/maxsim_oracle.c restating
processing_utils.py:170-186, truth tier) on the same seeded inputs -- never the HIP path against itself.

Tolerance: |got - truth| <= 1e-5 * max(|truth|, 1) (north star: 1e-3 relative).
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import maxsim_oracle as mo

pytestmark = pytest.mark.gpu

RTOL = 1e-5
DEV = torch.device("cuda:0")


def close(got, want, rtol=RTOL):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    return np.max(np.abs(got - want) / np.maximum(np.abs(want), 1.0)) <= rtol


@pytest.fixture(scope="module")
def amd():
    import colpali_amd

    assert torch.cuda.is_available()
    colpali_amd._lib.lib()
    return colpali_amd


def unit_rows(n, g, dtype=torch.bfloat16, dim=128):
    return torch.nn.functional.normalize(torch.randn(n, dim, generator=g), dim=-1).to(dtype)


def unit_row_list(lens, g, dtype=torch.bfloat16, dim=128):
    """One tensor of unit rows cut into pieces of the given lengths (one torch call instead of one per piece: a many-core host spends
    tens of milliseconds in every small CPU op)."""
    lens = [int(n) for n in lens]
    return list(unit_rows(sum(lens), g, dtype, dim).split(lens)) if lens else []


def oracle(qs, ps, batch_size=10**9):
    return mo.score_multi_vector([q.float().numpy() for q in qs], [p.float().numpy() for p in ps], batch_size=batch_size, mode="f32")


def docs(g, n, lo, hi, dtype=torch.bfloat16, dim=128):
    return unit_row_list(torch.randint(lo, hi + 1, (n,), generator=g).tolist(), g, dtype, dim)


# (query lengths, what the plan makes of them) -- flat_plan in colpali_amd/csrc/maxsim_abi.hip
PLANS = [
    ([5], "K1s, 1 unit"),
    ([16], "K1s, 1 full unit"),
    ([17], "K1s, 2 units"),
    ([20, 40, 33], "K1s, 6 units, three queries straddling units"),
    ([12] * 8, "K1s, 6 units, 8 queries (the most one wave reduces)"),
    ([40, 40, 40], "K1s, 8 units (120 tokens)"),
    ([1] * 8, "K1s, 8 one-token queries in one unit"),
    ([12] * 9, "pair form: 9 queries > K1s's 8"),
    ([10] * 30, "4 waves x 5 units: 30 short queries (more than the pair's 16), 19 units"),
    ([40] * 4, "pair form: 160 tokens = 10 units, 5 + 5"),
    ([33, 47, 12, 40, 21, 38], "pair form: 191 tokens = 12 units"),
    ([40] * 7, "4 waves x 5 units (three workgroups per CU): 280 tokens = 18 units"),
    ([40] * 8, "4 waves x 5 units: 320 tokens = 20 units exactly"),
    ([20] * 17, "4-wave form: 17 queries > the pair's 16 (22 units: the 8-unit kernel)"),
    ([48] * 10, "4-wave form: 480 tokens = 30 units"),
    ([40] * 15, "4-wave form x 10 units: 600 tokens"),
    ([31] * 30, "8-wave form: 930 tokens = 59 units"),
    ([40] * 31, "8-wave form x 10 units: 1240 tokens"),
    ([13] * 64, "8-wave form: 64 queries (the limit of one block's reduction)"),
    ([7] * 70, "two blocks by the query limit: 70 short queries"),
    ([40] * 40, "two blocks: 1600 tokens"),
    ([200, 33, 780, 12], "long ragged queries in one block: 1025 tokens"),
    ([0, 19, 0, 40], "empty queries score 0"),
]


@pytest.mark.parametrize("lens,what", PLANS, ids=[w for _, w in PLANS])
def test_flat_queries_in_every_plan_shape_match_the_oracle(amd, lens, what):
    g = torch.Generator().manual_seed(1000 + sum(lens) + len(lens))
    qs = unit_row_list(lens, g)
    ps = docs(g, 150, 1, 300) + unit_row_list([1024, 32, 33, 128, 129], g)
    want = oracle(qs, ps, batch_size=16)
    got = amd.score_multi_vector(qs, ps, batch_size=16, device="cuda:0").numpy()
    assert got.shape == (len(qs), len(ps))
    assert close(got, want), what
    q = amd.pack_queries(qs, DEV)
    assert isinstance(q, amd.PackedQueries) and q.lengths.tolist() == lens
    # the same through the resident-corpus entry, the whole corpus as ONE of the reference's passage blocks
    corpus = amd.pack_passages(ps, DEV, batch_size=len(ps))
    assert close(amd.maxsim_scores(q, corpus).cpu().numpy(), oracle(qs, ps)), what


def test_ragged_thousand_queries_many_blocks(amd):
    """BASELINE config 4's query batch with real lengths: 1000 queries of U{12..48} tokens -> ~30 balanced blocks on the 8-wave form,
    convoy counters in the workspace; and a corpus large enough for every XCD range to hold documents."""
    g = torch.Generator().manual_seed(77)
    lens = torch.randint(12, 49, (1000,), generator=g).tolist()
    qs = unit_row_list(lens, g)
    ps = docs(g, 700, 20, 140)
    got = amd.score_multi_vector(qs, ps, device="cuda:0").numpy()
    assert close(got, oracle(qs, ps, batch_size=128))


def test_more_query_blocks_than_one_launch_carries(amd):
    """The block table travels in the kernel arguments, 256 blocks per launch: 17 000 one-token queries are 266 blocks (64 queries each,
    the reduction's limit), i.e. two launches writing disjoint score rows."""
    g = torch.Generator().manual_seed(41)
    tok = unit_rows(17000, g)
    qs = list(tok.split(1))
    ps = docs(g, 40, 1, 70)
    got = amd.maxsim_scores(amd.pack_queries(qs, DEV), amd.pack_passages(ps, DEV, batch_size=None)).cpu().numpy()
    want = oracle(qs, ps, batch_size=1)           # blocks of one passage: no block padding, like batch_size=None here
    assert got.shape == (17000, 40) and close(got, want)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_zero_rows_are_dropped_and_change_no_score(amd, dtype):
    """The model's padded positions are zero rows (modeling_colpali.py:72 right-padded, modeling_colqwen2.py:69 left-padded).  The flat
    layout drops them; the box layout multiplies them.  Same scores -- bit for bit when the padding follows the tokens (the token sum
    visits the real tokens in the same order and adds 0 for the others), to fp32 summation order when it precedes them."""
    g = torch.Generator().manual_seed(5)
    lens = [12, 33, 40, 7, 25]
    l_max = 48
    ps = docs(g, 120, 5, 400, dtype)
    corpus = amd.pack_passages(ps, DEV, batch_size=len(ps))
    real = [unit_rows(n, g, dtype) for n in lens]
    want = oracle(real, ps)
    for side in ("right", "left"):
        box = torch.zeros((len(lens), l_max, 128), dtype=dtype)
        for i, q in enumerate(real):
            if side == "right":
                box[i, :q.shape[0]] = q
            else:
                box[i, l_max - q.shape[0]:] = q
        s_box = amd.maxsim_scores(box.to(DEV), corpus).cpu().numpy()                 # every row multiplied: msim_fwd
        for src in (box, box.to(DEV), list(torch.unbind(box))):
            flat = amd.pack_queries(src, DEV)
            assert isinstance(flat, amd.PackedQueries) and flat.lengths.tolist() == lens, side
            s_flat = amd.maxsim_scores(flat, corpus).cpu().numpy()
            assert close(s_flat, want) and close(s_box, want)
            if side == "right":
                assert np.array_equal(s_flat, s_box)
        kept = amd.pack_queries(box, DEV, compact=False)                              # explicit: keep the zero rows
        assert kept.lengths.tolist() == [l_max] * len(lens)
        assert np.array_equal(amd.maxsim_scores(kept, corpus).cpu().numpy(), s_box)


def test_a_query_scores_the_same_bits_in_any_batch(amd):
    """A query's token sum runs in an order fixed by its own length: the same bits alone (K1s), in a pair-form batch, in a 4-wave
    batch, in one 8-wave block and in a multi-block launch, at any position."""
    g = torch.Generator().manual_seed(9)
    lens = torch.randint(10, 49, (90,), generator=g).tolist()
    qs = unit_row_list(lens, g)
    ps = docs(g, 300, 30, 500)
    corpus = amd.pack_passages(ps, DEV)
    full = amd.maxsim_scores(amd.pack_queries(qs, DEV), corpus).cpu().numpy()           # several blocks
    for lo, hi in [(0, 1), (3, 4), (0, 3), (10, 16), (20, 33), (40, 70), (89, 90)]:
        part = amd.maxsim_scores(amd.pack_queries(qs[lo:hi], DEV), corpus).cpu().numpy()
        assert np.array_equal(part, full[lo:hi]), (lo, hi)
    packed = amd.pack_queries(qs, DEV)
    sel = packed.select(17, 29)
    assert np.array_equal(amd.maxsim_scores(sel, corpus).cpu().numpy(), full[17:29])
    # the box entry (msim_fwd) on queries of one length agrees bit for bit with the flat entry on the same queries
    same = [unit_rows(40, g) for _ in range(12)]
    a = amd.maxsim_scores(torch.stack(same).to(DEV), corpus).cpu().numpy()
    b = amd.maxsim_scores(amd.pack_queries(same, DEV), corpus).cpu().numpy()
    assert np.array_equal(a, b)


def test_literal_tier_on_ragged_queries(amd):
    """MSIM_FLAG_REF_ROUNDING on the flat path: within one bf16 ulp of the oracle's literal tier (bf16(fp32 dot) -> max -> fp32 sum ->
    bf16, oracle/maxsim_oracle.c following processing_utils.py:179 on bf16 tensors)."""
    g = torch.Generator().manual_seed(21)
    lens = [12, 40, 33, 48, 20, 17, 29]
    qs = [unit_rows(n, g) for n in lens]
    ps = docs(g, 100, 10, 300)
    want = mo.score_multi_vector([q.float().numpy() for q in qs], [p.float().numpy() for p in ps], batch_size=128, mode="bf16ref")
    lit = amd.maxsim_scores(amd.pack_queries(qs, DEV), amd.pack_passages(ps, DEV), ref_rounding=True).cpu().numpy()
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(want), 1e-30))) - 7)
    assert np.all(np.abs(lit - want) <= ulp) and np.mean(lit == want) > 0.9


def test_msim_fwd_ragged_through_the_c_abi(amd):
    """The entry point as a C caller uses it: raw pointers, the host copy of the offsets, error codes."""
    L = amd._lib.lib()
    g = torch.Generator().manual_seed(3)
    lens = [33, 12, 40, 25, 48, 19]
    qs = [unit_rows(n, g) for n in lens]
    ps = [unit_rows(150, g) for _ in range(80)]        # one length: no block padding, no clamp flags
    corpus = amd.pack_passages(ps, DEV, batch_size=None)
    tokens = torch.cat(qs).to(DEV)
    off_h = np.zeros(len(lens) + 1, dtype=np.int32)
    np.cumsum(lens, out=off_h[1:])
    off_d = torch.from_numpy(off_h).to(DEV)
    out = torch.full((len(lens), len(ps) + 3), -7.0, dtype=torch.float32, device=DEV)
    st = torch.cuda.current_stream().cuda_stream

    def call(dtype=0, dim=128, off_host=off_h, ld=out.stride(0), flags=0, tok=tokens):
        return L.msim_fwd_ragged(dtype, tok.data_ptr(), off_d.data_ptr(), off_host.ctypes.data, len(lens), corpus.blob.data_ptr(),
                                 corpus.offsets.data_ptr(), None, len(ps), dim, out.data_ptr(), ld, flags, None, st)

    assert call() == 0
    torch.cuda.synchronize()
    assert close(out[:, :len(ps)].cpu().numpy(), oracle(qs, ps))
    assert bool((out[:, len(ps):] == -7.0).all())                        # the padding columns of the caller's matrix are untouched
    assert call(dtype=2) == -2 and b"msim_fwd" in L.msim_last_error()     # fp32: not the flat path's shape
    assert call(dim=64) == -2                                             # 128 and 320 are the flat path's widths
    assert call(ld=len(ps) - 1) == -1
    assert call(flags=0x80) == -1
    bad = off_h.copy()
    bad[2] = bad[1] - 1
    assert call(off_host=bad) == -1
    bad = off_h.copy()
    bad[0] = 1
    assert call(off_host=bad) == -1
    assert L.msim_fwd_ragged(0, None, off_d.data_ptr(), off_h.ctypes.data, len(lens), corpus.blob.data_ptr(), corpus.offsets.data_ptr(),
                             None, len(ps), 128, out.data_ptr(), out.stride(0), 0, None, st) == -1
    huge = np.array([0, 1281], dtype=np.int32)                            # one query above a block's 1280 tokens
    big = torch.zeros((1281, 128), dtype=torch.bfloat16, device=DEV)
    assert L.msim_fwd_ragged(0, big.data_ptr(), off_d.data_ptr(), huge.ctypes.data, 1, corpus.blob.data_ptr(), corpus.offsets.data_ptr(),
                             None, len(ps), 128, out.data_ptr(), out.stride(0), 0, None, st) == -2


def test_flat_forward_captures_in_a_hipgraph(amd):
    """Nothing in msim_fwd_ragged allocates, synchronises or reads device memory on the host: it captures, and the replay on new
    query tokens of the same lengths equals the eager call bit for bit (1000 queries: convoy counters reset inside the graph)."""
    g = torch.Generator().manual_seed(31)
    lens = torch.randint(12, 49, (300,), generator=g).tolist()
    qs = [unit_rows(n, g) for n in lens]
    ps = docs(g, 200, 40, 300)
    corpus = amd.pack_passages(ps, DEV)
    q = amd.pack_queries(qs, DEV)
    L = amd._lib.lib()
    n, n_q = len(ps), len(lens)
    ws = torch.empty((max(int(L.msim_fwd_ragged_workspace_bytes(0, q.offsets_host.data_ptr(), n_q, n, 128)), 16),), dtype=torch.uint8, device=DEV)
    out = torch.zeros((n_q, n), dtype=torch.float32, device=DEV)
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            rc = L.msim_fwd_ragged(0, q.tokens.data_ptr(), q.offsets.data_ptr(), q.offsets_host.data_ptr(), n_q, corpus.blob.data_ptr(),
                                   corpus.offsets.data_ptr(), ctypes.c_void_p(corpus.clamp0.data_ptr()) if corpus.clamp0 is not None else None,
                                   n, 128, out.data_ptr(), out.stride(0), 0, ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
            assert rc == 0
    new_tokens = torch.cat([unit_rows(n_, g) for n_ in lens]).to(DEV)
    q.tokens.copy_(new_tokens)
    graph.replay()
    torch.cuda.synchronize()
    eager = amd.maxsim_scores(q, corpus)
    assert torch.equal(out, eager)


# ---------------------------------------------------------------------------------------------------------------------------------
# Width 320 (ColQwen3: colpali_engine/models/qwen3/colqwen3/modeling_colqwen3.py:48) on the flat layout: K1bPF (maxsim_panels.hip),
# 8 waves x <= 4 units per query block (512 tokens, 64 queries) and, since round 6, ONE block of 2 waves (<= 8 units, <= 16 queries) or
# 4 waves (<= 16 units, <= 32 queries) for the small batches -- plus K1sP for a uniform call of <= 4 token tiles.
WIDE = 320
PLANS_WIDE = [
    ([5], "one query, one unit"),
    ([16], "one full unit"),
    ([17], "two units"),
    ([20, 40, 33], "three ragged queries straddling units"),
    ([32] * 4, "uniform, four tiles: the box K1sP streams"),
    ([12] * 8, "uniform, eight tiles of a box = 6 units here"),
    ([40] * 3, "uniform Lq 40: 120 tokens = 8 units, one per wave"),
    ([33, 47, 12, 40, 21, 38, 9, 27, 44], "271 tokens = 17 units: three waves hold 3, five hold 2"),
    ([31] * 16, "496 tokens = 31 units: every wave but one holds 4"),
    ([32] * 16, "512 tokens: a full block"),
    ([31] * 30, "930 tokens: two blocks by the token limit"),
    ([7] * 70, "two blocks by the query limit (64 queries)"),
    ([200, 33, 270, 12], "long ragged queries: 515 tokens, two blocks"),
    ([512], "one query filling a block"),
    ([0, 19, 0, 40], "empty queries score 0"),
    ([40] * 40, "1600 tokens: four blocks"),
    ([40] * 4, "160 tokens = 10 units: the four-wave shape, 3 + 3 + 2 + 2"),
    ([25] * 8, "200 tokens = 13 units on four waves, queries straddling units"),
    ([16] * 16, "16 units and 16 queries: four waves, every wave holds 4"),
    ([8] * 32, "32 queries of 8 tokens: the four-wave shape at its query limit"),
    ([8] * 33, "33 queries: one more than four waves reduce -- the eight-wave shape"),
    ([7] * 17, "119 tokens = 8 units but 17 queries: four waves, not two"),
    ([64, 64], "128 tokens = 8 units in two queries: the two-wave shape, every wave holds 4"),
    ([1] * 16, "16 one-token queries in one unit on two waves (the second wave idles)"),
]


@pytest.mark.parametrize("lens,what", PLANS_WIDE, ids=[w for _, w in PLANS_WIDE])
def test_wide_flat_queries_in_every_plan_shape_match_the_oracle(amd, lens, what):
    g = torch.Generator().manual_seed(3200 + sum(lens) + len(lens))
    qs = unit_row_list(lens, g, dim=WIDE)
    ps = docs(g, 150, 1, 300, dim=WIDE) + unit_row_list([1024, 32, 33, 128, 129], g, dim=WIDE)
    want = oracle(qs, ps, batch_size=16)
    got = amd.score_multi_vector(qs, ps, batch_size=16, device="cuda:0").numpy()
    assert got.shape == (len(qs), len(ps))
    assert close(got, want), what
    q = amd.pack_queries(qs, DEV)
    assert isinstance(q, amd.PackedQueries) and q.lengths.tolist() == lens
    corpus = amd.pack_passages(ps, DEV, batch_size=len(ps))
    assert close(amd.maxsim_scores(q, corpus).cpu().numpy(), oracle(qs, ps)), what


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_wide_ragged_thousand_queries_many_blocks(amd, dtype):
    """1000 queries of U{12..48} tokens at width 320: ~60 balanced blocks over several document ranges per XCD; with empty documents
    and one-slab documents (the barrier of their own) in the corpus."""
    g = torch.Generator().manual_seed(78)
    lens = torch.randint(12, 49, (1000,), generator=g).tolist()
    qs = unit_row_list(lens, g, dtype, WIDE)
    ps = docs(g, 700, 0, 140, dtype, WIDE)
    corpus = amd.pack_passages(ps, DEV, batch_size=None)
    got = amd.maxsim_scores(amd.pack_queries(qs, DEV), corpus).cpu().numpy()
    empty = np.asarray([p.shape[0] == 0 for p in ps])
    assert empty.any() and np.all(np.isneginf(got[:, empty]))             # a max over nothing (no passage block pads these: batch_size=None)
    want = oracle(qs, [p for p in ps if p.shape[0]], batch_size=1)        # blocks of one passage: no block padding either
    assert close(got[:, ~empty], want)


def test_wide_zero_rows_are_dropped_and_the_box_entry_agrees(amd):
    """Zero padding rows are dropped by pack_queries at width 320 too; the box entry (msim_fwd) of a uniform Lq = 40 batch runs on the
    same flat kernel (48 rows per query instead of K1bP's 64) and returns the flat entry's bits."""
    g = torch.Generator().manual_seed(6)
    lens = [12, 33, 40, 7, 25]
    l_max = 48
    ps = docs(g, 120, 5, 400, dim=WIDE)
    corpus = amd.pack_passages(ps, DEV, batch_size=len(ps))
    real = [unit_rows(n, g, dim=WIDE) for n in lens]
    want = oracle(real, ps)
    box = torch.zeros((len(lens), l_max, WIDE), dtype=torch.bfloat16)
    for i, q in enumerate(real):
        box[i, :q.shape[0]] = q
    s_box = amd.maxsim_scores(box.to(DEV), corpus).cpu().numpy()          # every row multiplied: msim_fwd, K1bP (Lq 48 -> two tiles)
    for src in (box, box.to(DEV), list(torch.unbind(box))):
        flat = amd.pack_queries(src, DEV)
        assert isinstance(flat, amd.PackedQueries) and flat.lengths.tolist() == lens
        assert close(amd.maxsim_scores(flat, corpus).cpu().numpy(), want)
    assert close(s_box, want)
    same = [unit_rows(40, g, dim=WIDE) for _ in range(12)]
    a = amd.maxsim_scores(torch.stack(same).to(DEV), corpus).cpu().numpy()
    b = amd.maxsim_scores(amd.pack_queries(same, DEV), corpus).cpu().numpy()
    assert np.array_equal(a, b) and close(a, oracle(same, ps))
    three_tiles = [unit_rows(90, g, dim=WIDE) for _ in range(6)]                 # more tiles per query than K1bP takes: units on K1bPF
    c = amd.maxsim_scores(torch.stack(three_tiles).to(DEV), corpus).cpu().numpy()
    assert close(c, oracle(three_tiles, ps))


def test_wide_query_scores_the_same_bits_in_any_flat_batch(amd):
    """On K1bPF a query's token sum runs in an order fixed by its own length: the same bits in a one-block launch, in a multi-block
    launch and at any position (batches of mixed lengths, so none of them is the uniform <= 4-tile call K1sP takes)."""
    g = torch.Generator().manual_seed(10)
    lens = torch.randint(10, 49, (90,), generator=g).tolist()
    lens[0], lens[1] = 11, 37
    qs = unit_row_list(lens, g, dim=WIDE)
    ps = docs(g, 300, 30, 500, dim=WIDE)
    corpus = amd.pack_passages(ps, DEV)
    full = amd.maxsim_scores(amd.pack_queries(qs, DEV), corpus).cpu().numpy()
    assert close(full, oracle(qs, ps, batch_size=128))
    for lo, hi in [(0, 2), (0, 7), (10, 16), (20, 33), (40, 70), (85, 90)]:
        part = amd.maxsim_scores(amd.pack_queries(qs[lo:hi], DEV), corpus).cpu().numpy()
        assert np.array_equal(part, full[lo:hi]), (lo, hi)


def test_wide_literal_tier_and_too_long_queries(amd):
    """MSIM_FLAG_REF_ROUNDING on K1bPF: within one bf16 ulp of the oracle's literal tier.  A query above a block's 512 tokens keeps the
    box layout (generic kernels) and still matches."""
    g = torch.Generator().manual_seed(22)
    lens = [12, 40, 33, 48, 20, 17, 29]
    qs = [unit_rows(n, g, dim=WIDE) for n in lens]
    ps = docs(g, 100, 10, 300, dim=WIDE)
    want = mo.score_multi_vector([q.float().numpy() for q in qs], [p.float().numpy() for p in ps], batch_size=128, mode="bf16ref")
    lit = amd.maxsim_scores(amd.pack_queries(qs, DEV), amd.pack_passages(ps, DEV), ref_rounding=True).cpu().numpy()
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(want), 1e-30))) - 7)
    assert np.all(np.abs(lit - want) <= ulp) and np.mean(lit == want) > 0.9
    long_qs = [unit_rows(513, g, dim=WIDE), unit_rows(20, g, dim=WIDE)]
    packed = amd.pack_queries(long_qs, DEV)
    assert not isinstance(packed, amd.PackedQueries)
    assert close(amd.score_multi_vector(long_qs, ps, device="cuda:0").numpy(), oracle(long_qs, ps, batch_size=128))


def test_wide_msim_fwd_ragged_through_the_c_abi(amd):
    """msim_fwd_ragged at width 320 as a C caller uses it; a query above 512 tokens is refused (-2) with a message naming the limit."""
    L = amd._lib.lib()
    g = torch.Generator().manual_seed(4)
    lens = [33, 12, 40, 25, 48, 19]
    qs = [unit_rows(n, g, dim=WIDE) for n in lens]
    ps = [unit_rows(150, g, dim=WIDE) for _ in range(80)]
    corpus = amd.pack_passages(ps, DEV, batch_size=None)
    tokens = torch.cat(qs).to(DEV)
    off_h = np.zeros(len(lens) + 1, dtype=np.int32)
    np.cumsum(lens, out=off_h[1:])
    off_d = torch.from_numpy(off_h).to(DEV)
    out = torch.full((len(lens), len(ps) + 3), -7.0, dtype=torch.float32, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    assert L.msim_fwd_ragged_workspace_bytes(0, off_h.ctypes.data, len(lens), len(ps), WIDE) == 0
    rc = L.msim_fwd_ragged(0, tokens.data_ptr(), off_d.data_ptr(), off_h.ctypes.data, len(lens), corpus.blob.data_ptr(),
                           corpus.offsets.data_ptr(), None, len(ps), WIDE, out.data_ptr(), out.stride(0), 0, None, st)
    assert rc == 0
    torch.cuda.synchronize()
    assert close(out[:, :len(ps)].cpu().numpy(), oracle(qs, ps))
    assert bool((out[:, len(ps):] == -7.0).all())
    huge = np.array([0, 513], dtype=np.int32)
    big = torch.zeros((513, WIDE), dtype=torch.bfloat16, device=DEV)
    rc = L.msim_fwd_ragged(0, big.data_ptr(), off_d.data_ptr(), huge.ctypes.data, 1, corpus.blob.data_ptr(), corpus.offsets.data_ptr(),
                           None, len(ps), WIDE, out.data_ptr(), out.stride(0), 0, None, st)
    assert rc == -2 and b"512" in L.msim_last_error()
