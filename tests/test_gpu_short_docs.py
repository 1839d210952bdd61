"""GPU parity of K1b on SHORT documents (pooled pages, 64-row documents): document ends at every position of a chunk's four 32-row
slabs, tails of every size, documents without rows, clamp flags, a slice of a larger corpus.

Round 6 changed what K1b's producer requests -- a wave's share of a chunk that lies wholly past the document's end is no longer
fetched (the ring keeps whatever it held there; those rows are masked or never computed) -- and how the token sums are folded (DPP
row shifts); both must leave every bit where it was.  MSIM_FLAG_AVG_ROWS is a launch-shape hint and never a result: the same corpus
scored under a short and under a long hint must give the SAME BITS (run against a measurement build with
MSIM_BATCH_PACKED=1 the same file checks K1bK, the several-documents-per-chunk form that was measured and not kept), and both must sit within 1e-5 of
the float64 truth (processing_utils.py:179).
"""
import dataclasses

import numpy as np
import pytest
import torch

from oracle import maxsim_oracle as mo

pytestmark = pytest.mark.gpu

RTOL = 1e-5


@pytest.fixture(scope="module")
def amd():
    import colpali_amd

    assert torch.cuda.is_available()
    colpali_amd._lib.lib()  # must load: no fallback
    return colpali_amd


def _rows(g, lens, dtype=torch.bfloat16):
    rows = torch.nn.functional.normalize(torch.randn(max(sum(lens), 1), 128, generator=g), dim=-1).to(dtype)
    return [t.clone() for t in rows[: sum(lens)].split(lens)]


def _close(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return np.max(np.abs(got - want) / np.maximum(np.abs(want), 1.0)) <= RTOL


def _oracle(qs, ps, batch_size=128):
    return mo.score_multi_vector([q.float().numpy() for q in qs], [p.float().numpy() for p in ps], batch_size=batch_size, mode="f32")


def _both(amd, q, corpus, **kw):
    """the same call under a short and under a long average-length hint"""
    short_hint = amd.maxsim_scores(q, dataclasses.replace(corpus, avg_rows=32), **kw).cpu()
    plain = amd.maxsim_scores(q, dataclasses.replace(corpus, avg_rows=4096), **kw).cpu()
    return short_hint, plain


# every position of a document end inside a chunk of four 32-row slabs, tails of every size
EDGE_LENS = [1, 2, 31, 32, 33, 63, 64, 65, 95, 96, 97, 127, 128, 129, 159, 160, 161, 200, 255, 256, 257, 343, 400]


@pytest.mark.parametrize("n_q,lq,ragged", [(40, 32, False), (100, 32, True), (64, 40, False), (33, 48, True), (300, 32, False)])
def test_slab_edges_against_oracle_and_bitwise_against_k1b(amd, n_q, lq, ragged):
    g = torch.Generator().manual_seed(4000 + n_q + lq)
    q_lens = torch.randint(1, lq + 1, (n_q,), generator=g).tolist() if ragged else [lq] * n_q
    d_lens = (EDGE_LENS * 9)[: 200]
    perm = torch.randperm(len(d_lens), generator=g).tolist()
    d_lens = [d_lens[i] for i in perm]
    qs, ps = _rows(g, q_lens), _rows(g, d_lens)
    dev = torch.device("cuda:0")
    corpus = amd.pack_passages(ps, dev, batch_size=None)
    q = amd.pack_queries(qs, dev)
    short_hint, plain = _both(amd, q, corpus)
    assert torch.equal(short_hint, plain)
    assert _close(short_hint.numpy(), _truth(qs, ps))


def _truth(qs, ps):
    """sum over tokens of the max over the passage's own rows, float64 (no block padding: batch_size=None packing); one product per
    passage against all query tokens, token sums per query by index_add"""
    Q = torch.cat([q.double() for q in qs])
    owner = torch.repeat_interleave(torch.arange(len(qs)), torch.tensor([q.shape[0] for q in qs]))
    out = torch.zeros((len(qs), len(ps)), dtype=torch.float64)
    for j, p in enumerate(ps):
        if p.shape[0] == 0:
            out[:, j] = float("-inf")
            continue
        out[:, j].index_add_(0, owner, (Q @ p.double().T).max(dim=1).values)
    return out.numpy()


@pytest.mark.parametrize("doc_len", [64, 32, 96, 17])
def test_uniform_short_documents_bitwise_and_against_oracle(amd, doc_len):
    g = torch.Generator().manual_seed(doc_len)
    qs, ps = _rows(g, [32] * 70), _rows(g, [doc_len] * 1500)
    dev = torch.device("cuda:0")
    corpus, q = amd.pack_passages(ps, dev), amd.pack_queries(qs, dev)
    short_hint, plain = _both(amd, q, corpus)
    assert torch.equal(short_hint, plain)
    assert _close(short_hint.numpy(), _oracle(qs, ps))


def test_block_padding_clamp_flags_and_the_literal_tier(amd):
    # reference blocks of 7 passages: the shorter ones of a block see its zero padding rows (clamp0) -- the flags are per document
    g = torch.Generator().manual_seed(77)
    d_lens = torch.randint(1, 150, (700,), generator=g).tolist()
    qs, ps = _rows(g, torch.randint(1, 33, (90,), generator=g).tolist()), _rows(g, d_lens)
    dev = torch.device("cuda:0")
    corpus, q = amd.pack_passages(ps, dev, batch_size=7), amd.pack_queries(qs, dev)
    assert corpus.clamp0 is not None
    short_hint, plain = _both(amd, q, corpus)
    assert torch.equal(short_hint, plain)
    assert _close(short_hint.numpy(), _oracle(qs, ps, batch_size=7))
    lit_p, lit_b = _both(amd, q, corpus, ref_bf16=True)
    assert torch.equal(lit_p, lit_b)
    want = mo.score_multi_vector([x.float().numpy() for x in qs], [x.float().numpy() for x in ps], batch_size=7, mode="bf16ref")
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(want), 1e-3))) - 7)
    assert np.all(np.abs(lit_p.numpy() - want) <= ulp)


def test_documents_without_rows_and_runs_longer_than_the_offset_window(amd):
    # empty documents inside a block (the reference scores them 0 under its zero padding; -inf cannot occur through the drop-in) and a
    # run of 150 of them: producer and consumer skip them without a chunk
    g = torch.Generator().manual_seed(5)
    d_lens = [40, 0, 0, 64, 1, 0] * 20 + [0] * 150 + [33, 0, 96] * 30 + [0, 0, 0]
    qs, ps = _rows(g, [32] * 48), _rows(g, d_lens)
    dev = torch.device("cuda:0")
    corpus, q = amd.pack_passages(ps, dev, batch_size=len(ps)), amd.pack_queries(qs, dev)
    short_hint, plain = _both(amd, q, corpus)
    assert torch.equal(short_hint, plain)
    assert _close(short_hint.numpy(), _oracle(qs, ps, batch_size=len(ps)))
    # without block semantics an empty document is a max over nothing: -inf for a query with tokens
    raw = dataclasses.replace(corpus, clamp0=None)
    short_hint, plain = _both(amd, q, raw)
    assert torch.equal(short_hint, plain)
    empty = torch.tensor(d_lens) == 0
    assert torch.isinf(short_hint[:, empty]).all() and (short_hint[:, empty] < 0).all() and torch.isfinite(short_hint[:, ~empty]).all()


def test_float16_and_many_query_blocks_with_the_convoy(amd):
    g = torch.Generator().manual_seed(11)
    q_lens = torch.randint(12, 49, (1000,), generator=g).tolist()
    qs, ps = _rows(g, q_lens, torch.float16), _rows(g, torch.randint(20, 140, (4000,), generator=g).tolist(), torch.float16)
    dev = torch.device("cuda:0")
    corpus, q = amd.pack_passages(ps, dev, batch_size=None), amd.pack_queries(qs, dev)
    short_hint, plain = _both(amd, q, corpus)
    assert torch.equal(short_hint, plain)
    sub = list(range(0, 1000, 97))
    assert _close(short_hint[sub][:, :300].numpy(), _truth([qs[i] for i in sub], ps[:300]))


def test_a_corpus_slice_with_absolute_offsets(amd):
    # the drop-in's pipelined sub-ranges hand the kernels a SLICE of a larger corpus' offsets (d_off[0] != 0)
    g = torch.Generator().manual_seed(12)
    qs, ps = _rows(g, [32] * 64), _rows(g, torch.randint(1, 130, (900,), generator=g).tolist())
    dev = torch.device("cuda:0")
    corpus, q = amd.pack_passages(ps, dev, batch_size=None), amd.pack_queries(qs, dev)
    whole = amd.maxsim_scores(q, dataclasses.replace(corpus, avg_rows=32)).cpu()
    lo, hi = 301, 855
    part = dataclasses.replace(corpus, offsets=corpus.offsets[lo: hi + 1], lengths=corpus.lengths[lo:hi], avg_rows=32,
                               clamp0=None)
    got = amd.maxsim_scores(q, part).cpu()
    assert torch.equal(got, whole[:, lo:hi])


@pytest.mark.parametrize("n_q", [8, 10, 16, 20, 32])
def test_pair_and_four_wave_forms_on_tiny_documents(amd, n_q):
    # the pair form's chunk is one slab (a tail of <= 16 rows: its second wave requests nothing, ring of four chunks = a two-chunk
    # request history), the four-wave forms' two slabs (<= 32 rows: waves 2 and 3 request nothing); runs of tiny documents make
    # every pattern of requested / skipped shares in the history
    g = torch.Generator().manual_seed(900 + n_q)
    d_lens = ([1, 5, 16, 17, 20, 32, 33, 40, 48, 49, 64, 65, 9, 3] * 30)[: 400]
    perm = torch.randperm(len(d_lens), generator=g).tolist()
    d_lens = [d_lens[i] for i in perm] + [7] * 40 + [16] * 40 + [17] * 40 + [33] * 40
    qs, ps = _rows(g, [32] * n_q), _rows(g, d_lens)
    dev = torch.device("cuda:0")
    corpus, q = amd.pack_passages(ps, dev, batch_size=None), amd.pack_queries(qs, dev)
    got = amd.maxsim_scores(q, corpus).cpu()
    assert _close(got.numpy(), _truth(qs, ps))
    one = amd.maxsim_scores(amd.pack_queries(qs[:1], dev), corpus).cpu()        # K1s, which requests every slab
    assert torch.equal(one[0], got[0])
