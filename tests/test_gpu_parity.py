"""GPU parity: the HIP path (through the C ABI) against the oracle / golden vectors.

Tolerances (written here as the north star demands):
  * fp32-truth tier: |got - truth| <= 1e-5 * max(|truth|, 1)   (north star allows 1e-3 relative;
    bf16 products are exact in fp32, only the fp32 summation order differs)
  * literal REF_BF16 tier: at most one bf16 ulp away from the reference's bf16 CPU output
"""
import numpy as np
import pytest
import torch

from oracle import maxsim_oracle as mo
from tests.conftest import load_golden
from tests.helpers import config1_inputs, ragged_from_golden

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def close(got, want):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    return np.max(np.abs(got - want) / np.maximum(np.abs(want), 1.0)) <= RTOL


def bits_to_bf16(bits: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(bits.view(np.int16).copy()).view(torch.bfloat16)


@pytest.fixture(scope="module")
def amd():
    import colpali_amd

    assert torch.cuda.is_available()
    colpali_amd._lib.lib()  # must load: no fallback
    return colpali_amd


@pytest.mark.parametrize("name", ["score_ragged_d128.npz", "score_ragged_d320.npz"])      # d320: ColQwen3's width on K1bPF (round 5)
def test_golden_ragged_all_block_sizes(amd, name):
    z = load_golden(name)
    qs, ps = ragged_from_golden(z)
    qs = [bits_to_bf16(q) for q in qs]
    ps = [bits_to_bf16(p) for p in ps]
    for bs in z["batch_sizes"]:
        got = amd.score_multi_vector(qs, ps, batch_size=int(bs), device="cuda:0")
        assert got.device.type == "cpu" and got.dtype == torch.float32 and got.shape == (len(qs), len(ps))
        assert close(got.numpy(), z[f"truth_bs{bs}"]), f"batch_size={bs}"


def test_golden_config1_truth_and_literal(amd):
    z = load_golden("score_config1.npz")
    qs, ps = config1_inputs(z)
    got = amd.score_multi_vector(qs, ps, device="cuda:0").numpy()
    assert close(got, z["truth"])
    dev = torch.device("cuda:0")
    lit = amd.maxsim_scores(amd.pack_queries(qs, dev), amd.pack_passages(ps, dev), ref_bf16=True).cpu().numpy()
    ulp = 2.0 ** (np.floor(np.log2(np.abs(z["literal"]))) - 7)
    assert np.all(np.abs(lit - z["literal"]) <= ulp)
    assert np.mean(lit == z["literal"]) > 0.9


def test_literal_tier_is_reachable_through_the_drop_in_signature(amd, monkeypatch):
    """COLPALI_AMD_REF_ROUNDING=1: score_multi_vector (whose reference signature has no room for a switch) returns what the
    reference literally returns for bf16 inputs -- the live reference's own bf16 CPU output, score_config1.npz['literal']."""
    z = load_golden("score_config1.npz")
    qs, ps = config1_inputs(z)
    monkeypatch.setenv("COLPALI_AMD_REF_ROUNDING", "1")
    lit = amd.score_multi_vector(qs, ps, device="cuda:0").numpy()
    ulp = 2.0 ** (np.floor(np.log2(np.abs(z["literal"]))) - 7)
    assert np.all(np.abs(lit - z["literal"]) <= ulp) and np.mean(lit == z["literal"]) > 0.9
    assert np.all(lit == lit.astype(np.float32)) and not close(lit, z["truth"])     # bf16-valued: not the fp32-accurate tier
    monkeypatch.setenv("COLPALI_AMD_REF_ROUNDING", "0")
    assert close(amd.score_multi_vector(qs, ps, device="cuda:0").numpy(), z["truth"])


def test_golden_negative_similarities_and_zero_padding(amd):
    z = load_golden("score_negative_clamp.npz")
    q, short, long_ = (bits_to_bf16(z[k]) for k in ("q_bits", "short_bits", "long_bits"))
    assert close(amd.score_multi_vector([q], [short], device="cuda:0").numpy(), z["truth_alone"])
    assert close(amd.score_multi_vector([q], [short, long_], device="cuda:0").numpy(), z["truth_block"])
    assert close(amd.score_multi_vector([q], [short, long_], batch_size=1, device="cuda:0").numpy(), z["truth_split"])


def test_golden_tensor3d_inputs(amd):
    z = load_golden("score_tensor3d.npz")
    q = bits_to_bf16(z["q_bits"]).reshape(*z["q_shape"])
    p = bits_to_bf16(z["p_bits"]).reshape(*z["p_shape"])
    assert close(amd.score_multi_vector(q, p, device="cuda:0").numpy(), z["truth"])


def _random_case(seed, n_q, lq_max, n_d, ld_max, fixed_ld=None):
    g = torch.Generator().manual_seed(seed)

    def units(lens):        # one torch call per list, cut into pieces: a many-core host spends tens of ms in every small CPU op
        rows = torch.nn.functional.normalize(torch.randn(sum(lens), 128, generator=g), dim=-1).to(torch.bfloat16)
        return [t.clone() for t in rows.split(lens)]

    q_lens = torch.randint(1, lq_max + 1, (n_q,), generator=g).tolist()
    d_lens = [fixed_ld] * n_d if fixed_ld else torch.randint(1, ld_max + 1, (n_d,), generator=g).tolist()
    return units(q_lens), units(d_lens)


def _oracle(qs, ps, batch_size):
    return mo.score_multi_vector([q.float().numpy() for q in qs], [p.float().numpy() for p in ps],
                                 batch_size=batch_size, mode="f32")


@pytest.mark.parametrize("n_q,lq_max,n_d,ld_max,bs", [
    (1, 32, 300, 1100, 128),     # single query, ragged long documents, tails of every size
    (3, 32, 257, 200, 128),
    (4, 32, 64, 1024, 16),
    (5, 20, 100, 300, 7),        # more queries than one pass holds
    (9, 33, 40, 130, 128),       # two token tiles per query
    (2, 100, 30, 90, 128),       # four token tiles per query
    (3, 128, 20, 64, 128),
    (8, 32, 300, 500, 128),      # K1b pair form (NW=2): 8 token tiles, 4 per wave
    (5, 32, 257, 700, 128),      # pair form: 3 + 2 tiles
    (7, 31, 64, 1030, 16),       # pair form: 4 + 3 tiles, clamp0
    (3, 64, 90, 260, 128),       # pair form, two-tile queries: 2 + 1 queries per wave
    (2, 96, 60, 200, 128),       # pair form, three-tile queries: one per wave
    (2, 128, 40, 150, 128),      # pair form, four-tile queries
    (10, 32, 257, 700, 128),     # pair form with FIVE tiles per wave (round 3): 5 + 5
    (9, 30, 120, 400, 16),       # 5 + 4, clamp0
    (18, 32, 200, 300, 128),     # 4-wave form, five tiles per wave: 5/5/4/4
    (38, 32, 150, 260, 128),     # 8-wave form, one block of 38: 5/5/5/5/5/5/4/4
    (33, 32, 120, 200, 128),     # one block of 33: a single five-tile wave
    (77, 32, 90, 150, 128),      # two blocks of 39 + 38 on five-tile waves (instead of three of 26)
    (13, 32, 300, 500, 128),     # K1b<1>: waves with 2 and with 1 tile in one block
    (20, 32, 100, 400, 128),     # K1b<1>: 3 / 2 tiles per wave
    (41, 32, 500, 300, 128),     # K1b<1>: two balanced query blocks (21 + 20)
    (70, 32, 150, 300, 128),     # K1b<1>: three blocks (24 + 23 + 23)
    (100, 32, 200, 1100, 5),     # many queries, long ragged documents, small reference blocks (clamp0 everywhere)
    (17, 64, 90, 260, 128),      # K1b<2>: two blocks of 9 + 8 two-tile queries
    (40, 64, 60, 200, 128),      # K1b<2>: three blocks (14 + 13 + 13)
    (6, 50, 120, 200, 128),      # K1b<2>: idle waves in the only block
    (10, 96, 60, 200, 128),      # K1b<3>: one three-tile query per wave, two blocks
    (12, 128, 40, 150, 128),     # K1b<4>
])
def test_random_ragged_against_oracle(amd, n_q, lq_max, n_d, ld_max, bs):
    qs, ps = _random_case(1000 + n_q * 7 + n_d, n_q, lq_max, n_d, ld_max)
    qs[0] = qs[0][: lq_max] if qs[0].shape[0] >= lq_max else torch.cat(
        [qs[0], torch.nn.functional.normalize(torch.randn(lq_max - qs[0].shape[0], 128), dim=-1).to(torch.bfloat16)])
    got = amd.score_multi_vector(qs, ps, batch_size=bs, device="cuda:0").numpy()
    assert close(got, _oracle(qs, ps, bs))


def test_many_documents_more_than_waves(amd):
    # more documents than resident waves (256 CUs x 4): every wave walks several documents
    qs, ps = _random_case(5, 2, 32, 5000, 0, fixed_ld=40)
    got = amd.score_multi_vector(qs, ps, device="cuda:0").numpy()
    assert close(got, _oracle(qs, ps, 128))


def test_batch_regime_fixed_length_corpus_and_literal_mode(amd):
    # 64 queries x 400 pages of 1030 patches (ColPali-v1.2 geometry), truth tier + REF_BF16 tier
    qs, ps = _random_case(77, 64, 32, 400, 0, fixed_ld=1030)
    got = amd.score_multi_vector(qs, ps, device="cuda:0").numpy()
    assert close(got, _oracle(qs, ps, 128))
    dev = torch.device("cuda:0")
    lit = amd.maxsim_scores(amd.pack_queries(qs, dev), amd.pack_passages(ps, dev), ref_bf16=True).cpu().numpy()
    want = mo.score_multi_vector([q.float().numpy() for q in qs], [p.float().numpy() for p in ps], mode="bf16ref")
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(want), 1e-3))) - 7)
    assert np.all(np.abs(lit - want) <= ulp) and np.mean(lit == want) > 0.9


def test_stream_and_batch_kernels_agree_bitwise_on_shared_queries(amd):
    # the same query scored alone (K1s) and inside batches of 6 (K1b pair form), 12 (4 waves) and 40 (8 waves) must give the same
    # fp32 value: all kernels run the identical MFMA chain per (token tile, slab) and the same reduction tree
    qs, ps = _random_case(9, 40, 32, 300, 700)
    dev = torch.device("cuda:0")
    corpus = amd.pack_passages(ps, dev)
    lq = max(q.shape[0] for q in qs)
    padded = [torch.cat([q, q.new_zeros(lq - q.shape[0], 128)]) for q in qs]
    big = amd.maxsim_scores(amd.pack_queries(padded, dev), corpus).cpu()
    for i in (0, 7, 39):
        one = amd.maxsim_scores(amd.pack_queries([padded[i]], dev), corpus).cpu()
        assert torch.equal(one[0], big[i])
    for n in (6, 10, 12, 20, 38):                                   # pair form 3+3, FIVE tiles per wave (5+5; 5/5/5/5; 5/.../4), 4-wave form
        part = amd.maxsim_scores(amd.pack_queries(padded[:n], dev), corpus).cpu()
        assert torch.equal(part, big[:n])


@pytest.mark.parametrize("n_q,lq_max,n_d,ld_max", [(2, 32, 200, 700), (7, 32, 150, 300), (40, 40, 120, 500)])
def test_float16_embeddings_against_oracle(amd, n_q, lq_max, n_d, ld_max):
    # fp16 inputs go to the f16 MFMA (products of fp16 values are exact in fp32, like bf16)
    qs, ps = _random_case(31 + n_q, n_q, lq_max, n_d, ld_max)
    qs = [q.float().to(torch.float16) for q in qs]
    ps = [p.float().to(torch.float16) for p in ps]
    got = amd.score_multi_vector(qs, ps, device="cuda:0").numpy()
    assert close(got, _oracle(qs, ps, 128))


def test_mixed_dtypes_are_an_error_like_in_the_reference(amd):
    q = [torch.zeros(4, 128, dtype=torch.bfloat16)]
    p = [torch.zeros(4, 128, dtype=torch.float16)]
    with pytest.raises(RuntimeError, match="one dtype"):
        amd.score_multi_vector(q, p, device="cuda:0")
    with pytest.raises(NotImplementedError, match="dtype"):
        amd.score_multi_vector([torch.zeros(4, 128, dtype=torch.float64)], [torch.zeros(4, 128, dtype=torch.float64)], device="cuda:0")


def test_empty_inputs_raise_before_any_device_work(amd):
    with pytest.raises(ValueError, match="No queries provided"):
        amd.score_multi_vector([], [torch.zeros(2, 128, dtype=torch.bfloat16)], device="cuda:0")
    with pytest.raises(ValueError, match="No passages provided"):
        amd.score_multi_vector([torch.zeros(2, 128, dtype=torch.bfloat16)], [], device="cuda:0")


def test_transpose_detecting_asymmetric_inputs(amd):
    # one-hot style rows: a swapped operand / wrong fragment mapping cannot pass this
    q = torch.zeros(32, 128)
    d = torch.zeros(70, 128)
    for i in range(32):
        q[i, (3 * i) % 128] = 1.0 + i / 64
        q[i, (5 * i + 1) % 128] = -0.5
    for j in range(70):
        d[j, (7 * j) % 128] = 0.25 + j / 128
        d[j, (3 * j + 2) % 128] = 1.0
    q, d = q.to(torch.bfloat16), d.to(torch.bfloat16)
    got = amd.score_multi_vector([q], [d], device="cuda:0").numpy()
    want = (q.float() @ d.float().T).max(dim=1).values.sum().item()
    assert abs(got[0, 0] - want) <= 1e-5 * max(abs(want), 1)


# ---------------------------------------------------------------------------------------------------------
# Generic kernels (K1g): fp32 embeddings, widths other than 128, queries longer than 128 tokens.

def test_reference_unit_test_shape_fp32_dim32_list_and_tensor(amd):
    """Mirror of the reference's tests/utils/test_processing_utils.py:15-35 (fp32, dim=32, list vs padded tensor),
    on the golden inputs whose outputs were produced by the live reference."""
    z = load_golden("score_fp32_d32.npz")
    q, p = torch.from_numpy(z["q"]), torch.from_numpy(z["p"])
    qs = list(torch.split(q, z["q_lens"].tolist()))
    ps = list(torch.split(p, z["p_lens"].tolist()))
    from_list = amd.score_multi_vector(qs, ps, device="cuda:0")
    assert from_list.shape == (len(qs), len(ps)) and from_list.dtype == torch.float32 and from_list.device.type == "cpu"
    qs_padded = torch.nn.utils.rnn.pad_sequence(qs, batch_first=True)
    ps_padded = torch.nn.utils.rnn.pad_sequence(ps, batch_first=True)
    from_tensor = amd.score_multi_vector(qs_padded, ps_padded, device="cuda:0")
    assert from_tensor.shape == (len(qs), len(ps))
    assert torch.allclose(from_list, from_tensor), "Scores from list and tensor inputs should match"
    assert close(from_list.numpy(), z["scores_list"]) and close(from_tensor.numpy(), z["scores_tensor"])


def _random_generic(seed, n_q, lq_max, n_d, ld_max, dim, dtype):
    g = torch.Generator().manual_seed(seed)
    def unit(n):
        return torch.nn.functional.normalize(torch.randn(n, dim, generator=g), dim=-1).to(dtype)
    q_lens = torch.randint(1, lq_max + 1, (n_q,), generator=g).tolist()
    q_lens[0] = lq_max
    d_lens = torch.randint(1, ld_max + 1, (n_d,), generator=g).tolist()
    return [unit(n) for n in q_lens], [unit(n) for n in d_lens]


@pytest.mark.parametrize("dtype,dim,n_q,lq_max,n_d,ld_max,bs", [
    (torch.float32, 128, 3, 32, 200, 300, 128),    # fp32 embeddings of the usual width, T=4 whole queries
    (torch.float32, 128, 9, 100, 60, 130, 16),     # four token tiles per query
    (torch.float32, 128, 2, 200, 40, 90, 128),     # seven token tiles: sub-passes with same-thread accumulation
    (torch.float32, 32, 5, 7, 64, 20, 128),        # the reference unit test's width
    (torch.float32, 320, 4, 40, 50, 100, 128),     # ColQwen3 width in fp32 (1280-byte rows: T=2)
    (torch.float32, 1024, 2, 33, 10, 70, 128),     # 4 KiB rows: T=1
    (torch.bfloat16, 320, 6, 32, 150, 400, 128),   # ColQwen3 (colqwen3 dim=320)
    (torch.bfloat16, 320, 40, 32, 100, 300, 7),
    (torch.bfloat16, 128, 3, 150, 80, 200, 128),   # dim 128 but queries longer than the tuned kernels hold
    (torch.bfloat16, 100, 4, 20, 70, 90, 128),     # width padded 100 -> 112 with zero columns
    (torch.bfloat16, 64, 17, 32, 90, 260, 5),
    (torch.float16, 64, 5, 32, 90, 260, 128),
    (torch.float16, 512, 3, 64, 30, 100, 128),
])
def test_generic_kernels_against_oracle(amd, dtype, dim, n_q, lq_max, n_d, ld_max, bs):
    qs, ps = _random_generic(dim * 3 + n_q, n_q, lq_max, n_d, ld_max, dim, dtype)
    got = amd.score_multi_vector(qs, ps, batch_size=bs, device="cuda:0").numpy()
    assert close(got, _oracle(qs, ps, bs))


def test_generic_many_documents_and_literal_rounding(amd):
    # more documents than resident waves; REF rounding mode on the generic bf16 path equals the literal oracle tier
    qs, ps = _random_generic(3, 3, 32, 5000, 40, 320, torch.bfloat16)
    got = amd.score_multi_vector(qs, ps, device="cuda:0").numpy()
    assert close(got, _oracle(qs, ps, 128))
    dev = torch.device("cuda:0")
    lit = amd.maxsim_scores(amd.pack_queries(qs, dev), amd.pack_passages(ps[:300], dev), ref_rounding=True).cpu().numpy()
    want = mo.score_multi_vector([q.float().numpy() for q in qs], [p.float().numpy() for p in ps[:300]], mode="bf16ref")
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(want), 1e-3))) - 7)
    assert np.all(np.abs(lit - want) <= ulp) and np.mean(lit == want) > 0.9


def test_generic_and_tuned_kernels_agree(amd):
    # a 128-wide bf16 query scored by the tuned kernels (16x16x32 MFMA tiles, 32 k per step) and, as a zero-padded box of 160
    # tokens handed to msim_fwd WITHOUT scratch, by the generic kernel K1g (32x32x16 tiles, 16 k per step): the same exact products
    # summed in a different order inside the matrix unit -- equal to fp32 summation-order noise, not bit for bit
    from colpali_amd import _lib

    qs, ps = _random_case(21, 3, 32, 200, 500)
    dev = torch.device("cuda:0")
    corpus = amd.pack_passages(ps, dev)
    tuned = amd.maxsim_scores(amd.pack_queries(qs, dev), corpus).cpu()
    box = amd.pack_queries([torch.cat([q, q.new_zeros(160 - q.shape[0], 128)]) for q in qs], dev, layout="box")
    assert box.shape == (3, 160, 128)
    generic = torch.empty((3, len(ps)), dtype=torch.float32, device=dev)
    rc = _lib.lib().msim_fwd(0, _lib.ptr(box), 3, 160, _lib.ptr(corpus.blob), _lib.ptr(corpus.offsets), _lib.ptr(corpus.clamp0),
                             len(ps), 128, _lib.ptr(generic), len(ps), 0, None, _lib.current_stream_handle(dev))   # NULL scratch: K1g
    _lib.check(rc, "msim_fwd")
    generic = generic.cpu()
    assert float(((tuned - generic).abs() / generic.abs().clamp_min(1.0)).max()) < 2e-6
    assert not torch.equal(tuned, generic)          # a different kernel really ran


def test_transpose_detecting_asymmetric_inputs_fp32(amd):
    q = torch.zeros(32, 48)
    d = torch.zeros(70, 48)
    for i in range(32):
        q[i, (3 * i) % 48] = 1.0 + i / 64
        q[i, (5 * i + 1) % 48] = -0.5
    for j in range(70):
        d[j, (7 * j) % 48] = 0.25 + j / 128
        d[j, (3 * j + 2) % 48] += 1.0
    got = amd.score_multi_vector([q], [d], device="cuda:0").numpy()
    want = (q.double() @ d.double().T).max(dim=1).values.sum().item()
    assert abs(got[0, 0] - want) <= 1e-5 * max(abs(want), 1)


# ---------------------------------------------------------------------------------------------------------
# Panel kernels (K1sP / K1bP / K1bPF): the tuned path for dim = 320 (ColQwen3), 16-bit.  The drop-in packs width-320 queries into
# the flat layout since round 5 (ragged lists -> K1bPF; a uniform list of <= 4 tiles -> K1sP); the query BOX entry (maxsim_scores on
# a [n_q, Lq, 320] tensor -> msim_fwd) keeps K1sP (<= 4 tiles in all) and K1bP (whole queries of 32 / 64 rows) and sends the rest to
# K1bPF: both entries are checked against the oracle for every shape.

def _box320(qs, dtype):
    """the queries zero-padded to one length that is a multiple of 32 (the shape K1sP / K1bP take as it stands)"""
    lq = (max(q.shape[0] for q in qs) + 31) // 32 * 32
    box = torch.zeros((len(qs), lq, 320), dtype=dtype)
    for i, q in enumerate(qs):
        box[i, :q.shape[0]] = q
    return box


@pytest.mark.parametrize("dtype,n_q,lq_max,n_d,ld_max,bs", [
    (torch.bfloat16, 1, 32, 300, 1100, 128),    # box: K1sP<1,1>: ragged long documents, tails of every size
    (torch.bfloat16, 2, 20, 257, 200, 128),     # box: K1sP<2,1>
    (torch.bfloat16, 3, 32, 64, 1024, 16),      # box: K1sP<3,1>
    (torch.bfloat16, 4, 32, 100, 300, 7),       # box: K1sP<4,1>, clamp0 everywhere
    (torch.bfloat16, 1, 64, 90, 260, 128),      # box: K1sP<2,2>
    (torch.bfloat16, 2, 50, 120, 200, 128),     # box: K1sP<4,2>
    (torch.bfloat16, 1, 96, 60, 200, 128),      # box: K1sP<3,3>
    (torch.bfloat16, 1, 128, 40, 150, 128),     # box: K1sP<4,4>
    (torch.bfloat16, 8, 32, 300, 500, 128),     # box: K1bP<1,1>: exactly one workgroup's queries
    (torch.bfloat16, 5, 32, 300, 500, 5),       # box: K1bP<1,1>: partial
    (torch.bfloat16, 13, 32, 300, 500, 128),    # box: K1bP<2,1>
    (torch.bfloat16, 33, 32, 500, 300, 128),    # box: K1bP<2,1>, three query blocks, partial last block
    (torch.bfloat16, 17, 64, 90, 260, 128),     # box: K1bP<2,2>
    (torch.bfloat16, 9, 100, 40, 130, 128),     # box: four tiles per query, more than K1bP takes -> K1bPF (the generic kernel before round 5)
    (torch.float16, 3, 32, 200, 700, 128),
    (torch.float16, 40, 40, 120, 500, 128),
])
def test_dim320_panel_kernels_against_oracle(amd, dtype, n_q, lq_max, n_d, ld_max, bs):
    qs, ps = _random_generic(n_q * 31 + n_d, n_q, lq_max, n_d, ld_max, 320, dtype)
    want = _oracle(qs, ps, bs)
    got = amd.score_multi_vector(qs, ps, batch_size=bs, device="cuda:0").numpy()        # flat entry: K1bPF / K1sP
    assert close(got, want)
    dev = torch.device("cuda:0")
    corpus = amd.pack_passages(ps, dev, batch_size=bs)
    box = amd.maxsim_scores(_box320(qs, dtype).to(dev), corpus).cpu().numpy()           # box entry: K1sP / K1bP (zero rows add exactly 0)
    assert close(box, want)


def test_dim320_panel_kernels_agree_bitwise_with_each_other_and_closely_with_the_generic_kernel(amd):
    # the same 16x16x32 MFMA chain in the same k order and the same reduction tree in K1sP (query alone) and K1bP (inside a
    # batch): bit-identical.  K1bPF (the same queries in the flat layout) runs the same chain and adds the tokens in K1b's order, K1g
    # (same query zero-padded to 520 tokens: more than a K1bPF block holds, which only the generic kernel takes) runs 32x32x16 tiles:
    # both equal up to fp32 summation order
    qs, ps = _random_generic(11, 20, 32, 200, 500, 320, torch.bfloat16)
    dev = torch.device("cuda:0")
    corpus = amd.pack_passages(ps, dev)
    box = _box320(qs, torch.bfloat16)
    big = amd.maxsim_scores(box.to(dev), corpus).cpu()                                    # K1bP<2,1>
    flat = amd.maxsim_scores(amd.pack_queries(qs, dev), corpus).cpu()                     # K1bPF
    assert float(((flat - big).abs() / big.abs().clamp_min(1.0)).max()) < 2e-6
    for i in (0, 7, 19):
        one = amd.maxsim_scores(box[i:i + 1].to(dev), corpus).cpu()                       # K1sP<1,1>
        assert torch.equal(one[0], big[i])
        long_q = torch.cat([qs[i], qs[i].new_zeros(520 - qs[i].shape[0], 320)])
        gen = amd.maxsim_scores(long_q[None].to(dev), corpus).cpu()                       # K1g
        assert float(((gen[0] - big[i]).abs() / big[i].abs().clamp_min(1.0)).max()) < 2e-6


def test_dim320_many_documents_and_literal_rounding(amd):
    qs, ps = _random_generic(5, 2, 32, 6000, 40, 320, torch.bfloat16)        # more documents than resident waves
    got = amd.score_multi_vector(qs, ps, device="cuda:0").numpy()
    assert close(got, _oracle(qs, ps, 128))
    qs, ps = _random_generic(6, 12, 32, 300, 400, 320, torch.bfloat16)
    dev = torch.device("cuda:0")
    want = mo.score_multi_vector([q.float().numpy() for q in qs], [p.float().numpy() for p in ps], mode="bf16ref")
    ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(want), 1e-3))) - 7)
    corpus = amd.pack_passages(ps, dev)
    for queries in (amd.pack_queries(qs, dev), _box320(qs, torch.bfloat16).to(dev), _box320(qs[:3], torch.bfloat16).to(dev)):   # K1bPF, K1bP, K1sP
        lit = amd.maxsim_scores(queries, corpus, ref_rounding=True).cpu().numpy()
        w, u = want[:lit.shape[0]], ulp[:lit.shape[0]]
        assert np.all(np.abs(lit - w) <= u) and np.mean(lit == w) > 0.9


@pytest.mark.parametrize("n_q,lq,n_d,ld_max,dtype", [
    (3, 780, 200, 300, torch.bfloat16),       # pages as queries: 7 segments, the last one of 12 tokens
    (2, 129, 150, 700, torch.bfloat16),       # one token beyond a segment
    (5, 200, 64, 100, torch.float16),         # 128 + 72
    (11, 256, 300, 64, torch.bfloat16),       # exactly two full segments; 22 pseudo-queries = 88 tiles: three query blocks
])
def test_long_queries_score_as_segments_on_the_tuned_kernels(amd, n_q, lq, n_d, ld_max, dtype):
    """Queries longer than 128 tokens (bf16 / f16, width 128) are scored as 128-token segments on K1b and the partial token sums
    added in segment order (msim_fwd with its workspace); without the workspace the generic kernels take them.  Both against the
    oracle, and against each other to fp32 summation-order noise; ragged queries (zero padding rows) and clamp0 included."""
    from colpali_amd import _lib

    qs, ps = _random_case(500 + n_q + lq, n_q, lq, n_d, ld_max)
    qs[0] = torch.nn.functional.normalize(torch.randn(lq, 128), dim=-1).to(torch.bfloat16)      # one query of the full length
    qs, ps = [q.float().to(dtype) for q in qs], [p.float().to(dtype) for p in ps]
    want = _oracle(qs, ps, 7)
    got = amd.score_multi_vector(qs, ps, batch_size=7, device="cuda:0").numpy()                    # blocks of 7: clamp0 flags everywhere
    assert close(got, want)
    dev = torch.device("cuda:0")
    q, corpus = amd.pack_queries(qs, dev, layout="box"), amd.pack_passages(ps, dev, batch_size=7)   # the box entry: pieces on K1b
    L = _lib.lib()
    assert L.msim_fwd_workspace_bytes(_lib.dtype_code(dtype), n_q, q.shape[1], n_d, 128) == 4096 + n_q * ((q.shape[1] + 127) // 128) * n_d * 4
    generic = torch.empty((n_q, n_d), dtype=torch.float32, device=dev)
    rc = L.msim_fwd(_lib.dtype_code(dtype), _lib.ptr(q), n_q, q.shape[1], _lib.ptr(corpus.blob), _lib.ptr(corpus.offsets),
                    _lib.ptr(corpus.clamp0), n_d, 128, _lib.ptr(generic), n_d, 0, None, _lib.current_stream_handle(dev))
    _lib.check(rc, "msim_fwd")
    assert close(generic.cpu().numpy(), want)
    assert np.max(np.abs(generic.cpu().numpy() - got) / np.maximum(np.abs(want), 1.0)) <= 4e-6
    if dtype == torch.bfloat16:                                                                     # the literal tier, rounded once, on the total
        lit = amd.maxsim_scores(q, corpus, ref_rounding=True).cpu().numpy()
        ref = mo.score_multi_vector([x.float().numpy() for x in qs], [x.float().numpy() for x in ps], batch_size=7, mode="bf16ref")
        ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(ref), 1e-3))) - 7)
        assert np.all(np.abs(lit - ref) <= ulp) and np.mean(lit == ref) > 0.8


def test_long_query_scratch_is_bounded_by_scoring_the_queries_in_groups(amd, monkeypatch):
    from colpali_amd import scoring

    qs, ps = _random_case(77, 9, 300, 120, 200)
    dev = torch.device("cuda:0")
    q, corpus = amd.pack_queries(qs, dev, layout="box"), amd.pack_passages(ps, dev)
    whole = amd.maxsim_scores(q, corpus).cpu()
    monkeypatch.setattr(scoring, "_MAX_FWD_SCRATCH", 4096 + 2 * 3 * 120 * 4)          # room for two queries' partial sums at a time
    grouped = amd.maxsim_scores(q, corpus).cpu()
    assert torch.equal(whole, grouped)
    assert close(whole.numpy(), _oracle(qs, ps, 128))


@pytest.mark.parametrize("batch_size", [128, 64])
def test_pipelined_dropin_from_host_lists_is_bit_identical_to_one_launch_over_the_packed_corpus(amd, batch_size):
    """score_multi_vector from a Python list of host page tensors is a pipeline (scoring.py:_score_host_list_pipelined): the corpus
    goes up in chunks on a copy stream and passage SUB-RANGES (whole blocks of `batch_size`) are scored while the next ones upload.
    The result must be the bits of ONE launch over the packed corpus -- clamp0 from the same blocking, ragged pages, enough bytes for
    several sub-ranges (> 96 MB) -- and within 1e-5 of the oracle on a sample."""
    import numpy as np

    from colpali_amd import scoring
    from oracle import maxsim_oracle as mo

    g = torch.Generator().manual_seed(77 + batch_size)
    lens = torch.randint(200, 1031, (700,), generator=g).tolist()
    ps = [torch.nn.functional.normalize(torch.randn(n, 128, generator=g), dim=-1).to(torch.bfloat16) for n in lens]
    qs = [torch.nn.functional.normalize(torch.randn(int(n), 128, generator=g), dim=-1).to(torch.bfloat16)
          for n in torch.randint(8, 41, (37,), generator=g)]
    assert sum(p.numel() * 2 for p in ps) > 2 * scoring._PIPE_RANGE_BYTES
    got = amd.score_multi_vector(qs, ps, batch_size=batch_size, device="cuda:0")
    dev = torch.device("cuda:0")
    want = amd.maxsim_scores(amd.pack_queries(qs, dev), amd.pack_passages(ps, dev, batch_size=batch_size)).cpu()
    assert got.dtype == torch.float32 and got.device.type == "cpu" and torch.equal(got, want)
    again = amd.score_multi_vector(qs, ps, batch_size=batch_size, device="cuda:0")        # the staging halves alternate across calls
    assert torch.equal(again, got)
    pick = [0, 3, 127, 128, 350, 699]
    ref = mo.score_multi_vector([q.float().numpy() for q in qs[:5]], [ps[i].float().numpy() for i in pick], batch_size=1)
    # (the sample is scored without block mates, i.e. un-clamped: compare the pages that are the longest of their real block)
    long_enough = [k for k, i in enumerate(pick) if lens[i] == max(lens[(i // batch_size) * batch_size:(i // batch_size + 1) * batch_size])]
    for k in long_enough:
        assert np.max(np.abs(got[:5, pick[k]].numpy() - ref[:, k]) / np.maximum(np.abs(ref[:, k]), 1.0)) <= 1e-5
