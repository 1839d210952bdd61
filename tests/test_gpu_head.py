"""GPU parity of the fused embedding head (K3, msim_embed_head) against oracle/head_oracle.py and the live-reference golden.

Tolerances: the kernel reproduces the reference's rounding chain in the model dtype (Linear output, norm, quotient each
rounded once), so outputs are compared element-wise with the literal tier (the reference's own lines evaluated on CPU in
that dtype): >= 99.5 % of the elements bit-equal, at most 1e-4 of them more than one ulp of the 16-bit dtype apart, none
more than two (2e-6 absolute floor for outputs that cancelled to almost nothing) -- the fp32 accumulation order of the
K = hidden-size dot product differs from the CPU GEMM's, which occasionally flips a rounding;
against the float64 truth tier: 2^-6 relative for bf16, 2^-9 for fp16, on |value| >= 1e-3.
"""
import numpy as np
import pytest
import torch

from oracle import head_oracle as ho
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import colpali_amd

    colpali_amd._lib.lib()
    return colpali_amd


def _bf16(bits: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(bits.view(np.int16).copy()).view(torch.bfloat16)


def grid_distance(got: torch.Tensor, want: torch.Tensor, atol: float = 2e-6):
    """Per-element distance in ulps of the 16-bit dtype (0 where |diff| <= atol), and the bit-equal fraction.
    The absolute floor covers outputs that are tiny because the K-long dot product cancelled: there the fp32
    accumulation-order error (~1e-7 of the sum of |products|, ~4e-7 after the normalisation) spans several grid steps of
    a number that small -- for the CPU GEMM too."""
    g, w = got.cpu().float(), want.cpu().float()
    mant = 7 if got.dtype == torch.bfloat16 else 10
    ulp = torch.exp2(torch.floor(torch.log2(w.abs().clamp_min(1e-30))) - mant)
    diff = (g - w).abs()
    d = torch.where(diff <= atol, torch.zeros_like(diff), diff / ulp)
    return d, float((g == w).float().mean())


def assert_same_rounding_chain(got, want):
    """One rounding where torch has one: >= 99.5 % of the elements bit-equal, at most 1e-4 of them more than one ulp
    apart and none more than two (a flipped rounding of the Linear output moves the quotient by up to one ulp before ITS
    rounding).  The reference's own CPU output sits exactly this far from the exactly-accumulated chain
    (measured: 99.99 % equal, 1 element of 527 360 at two ulps)."""
    d, same = grid_distance(got, want)
    assert same >= 0.995, same
    assert float((d > 1).float().mean()) <= 1e-4 and float(d.max()) <= 2.0, (float((d > 1).float().mean()), float(d.max()))


def check(got, hidden, weight, bias, mask, extra=None):
    assert_same_rounding_chain(got, ho.head_literal(hidden, weight, bias, mask, extra))
    truth = ho.head_truth(hidden, weight, bias, mask, extra)
    big = truth.abs() >= 1e-3
    rel = 2.0**-6 if got.dtype == torch.bfloat16 else 2.0**-9
    assert torch.all(((got.cpu().double() - truth).abs() <= rel * truth.abs())[big])
    keep = (mask != 0) if extra is None else ((mask != 0) & (extra.reshape(mask.shape) != 0))
    assert torch.count_nonzero(got.cpu()[~keep]) == 0                      # masked positions are exactly zero


def test_golden_live_reference_forward(amd):
    z = load_golden("head_colpali_tiny.npz")
    h, w, b = (_bf16(z[k]) for k in ("hidden_bf16", "weight_bf16", "bias_bf16"))
    mask = torch.from_numpy(z["attention_mask"])
    got = amd.embedding_head(h.cuda(), w.cuda(), b.cuda(), mask.cuda())
    assert got.shape == (5, 37, 128) and got.dtype == torch.bfloat16
    assert_same_rounding_chain(got, _bf16(z["out_bf16"]))
    check(got, h, w, b, mask)


def _case(seed, B, S, H, dtype, pad="right"):
    g = torch.Generator().manual_seed(seed)
    hidden = (torch.randn(B, S, H, generator=g) * 2.0).to(dtype)
    weight = (torch.randn(128, H, generator=g) / H**0.5).to(dtype)
    bias = (torch.randn(128, generator=g) * 0.1).to(dtype)
    mask = torch.ones(B, S, dtype=torch.long)
    for b in range(1, B):
        n = int(torch.randint(1, S, (1,), generator=g))
        if pad == "left":
            mask[b, : S - n] = 0          # ColQwen2 pads on the left (modeling_colqwen2.py:36)
        else:
            mask[b, n:] = 0
    return hidden, weight, bias, mask


@pytest.mark.parametrize("B,S,H,dtype,pad", [
    (3, 50, 64, torch.bfloat16, "right"),        # a single K chunk
    (4, 1030, 2048, torch.bfloat16, "right"),    # ColPali-v1.2 geometry (PaliGemma-3B text width)
    (5, 779, 1536, torch.bfloat16, "left"),      # ColQwen2-v1.0 geometry (Qwen2-VL-2B), left padded
    (2, 300, 3584, torch.bfloat16, "left"),      # Qwen2-VL-7B width
    (3, 200, 1536, torch.float16, "right"),
    (1, 17, 128, torch.bfloat16, "right"),       # fewer rows than one wave's share
])
def test_random_hidden_states_against_oracle(amd, B, S, H, dtype, pad):
    hidden, weight, bias, mask = _case(B * 100 + S, B, S, H, dtype, pad)
    got = amd.embedding_head(hidden.cuda(), weight.cuda(), bias.cuda(), mask.cuda())
    check(got, hidden, weight, bias, mask)


def test_more_tiles_than_compute_units_and_ragged_tail(amd):
    hidden, weight, bias, mask = _case(9, 301, 257, 256, torch.bfloat16)   # 77357 rows: 303 tiles, the last one partial
    got = amd.embedding_head(hidden.cuda(), weight.cuda(), bias.cuda(), mask.cuda())
    check(got, hidden, weight, bias, mask)


def test_image_mask_and_missing_bias(amd):
    hidden, weight, bias, mask = _case(4, 3, 90, 512, torch.bfloat16)
    extra = (torch.arange(90)[None, :] % 3 != 0).expand(3, 90).unsqueeze(-1)   # mask_non_image_embeddings style [B, S, 1]
    got = amd.embedding_head(hidden.cuda(), weight.cuda(), None, mask.cuda(), extra.cuda())
    check(got, hidden, weight, None, mask, extra)


def test_unsupported_inputs_fail_loudly(amd):
    hidden, weight, bias, mask = _case(1, 2, 10, 64, torch.bfloat16)
    with pytest.raises(NotImplementedError, match="bf16"):
        amd.embedding_head(hidden.float().cuda(), weight.float().cuda(), bias.float().cuda(), mask.cuda())
    with pytest.raises(NotImplementedError, match="multiple of 64"):
        amd.embedding_head(hidden[..., :40].contiguous().cuda(), weight[:, :40].contiguous().cuda(), bias.cuda(), mask.cuda())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        amd.embedding_head(hidden, weight, bias, mask)


def test_corpus_writer_equals_the_reference_road_to_the_scorer(amd):
    """README.md:121-126: ds.extend(list(torch.unbind(model(**batch)))) then processor.score(qs, ds).  The writer drops the
    masked rows and flags the page instead; scores must be identical (bit for bit: same kernel, a max does not care
    about row order or about how many zero rows there are)."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(77)
    H = 1536
    weight = (torch.randn(128, H, generator=g) / H**0.5).to(torch.bfloat16).to(dev)
    bias = (torch.randn(128, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    writer = amd.CorpusWriter(capacity_rows=5 * 300 + 4 * 210 + 3 * 64, device=dev)
    pages, dense_all, masks = [], [], []
    for B, S, pad in ((5, 300, "left"), (4, 210, "right"), (3, 64, "right")):
        hidden, _, _, mask = _case(B + S, B, S, H, torch.bfloat16, pad)
        if S == 64:
            mask[:] = 1                                    # a batch without any padding
        hidden, mask = hidden.to(dev), mask.to(dev)
        assert writer.append(hidden, weight, bias, mask) == B
        dense = amd.embedding_head(hidden, weight, bias, mask)
        pages.extend(list(torch.unbind(dense)))           # what the reference user keeps
        dense_all.append(dense)
        masks.append(mask)
    corpus = writer.finish()
    assert len(corpus) == 12
    # layout: unmasked rows of every page, in order, back to back
    want_rows = torch.cat([d[m != 0] for d, m in zip(dense_all, masks)])
    assert torch.equal(corpus.blob.view(torch.int16), want_rows.view(torch.int16))
    assert corpus.lengths.tolist() == [int((m[b] != 0).sum()) for m in masks for b in range(m.shape[0])]
    qs = [torch.nn.functional.normalize(torch.randn(n, 128, generator=g), dim=-1).to(torch.bfloat16) for n in (20, 32, 11)]
    via_reference_road = amd.score_multi_vector(qs, [p.cpu() for p in pages], device=dev)
    direct = amd.maxsim_scores(amd.pack_queries(qs, dev), corpus).cpu()
    assert torch.equal(via_reference_road, direct)
    # and against the CPU oracle scorer on the literal head outputs (truth tier of the scorer)
    from oracle import maxsim_oracle as mo
    want = mo.score_multi_vector([q.float().numpy() for q in qs], [p.float().cpu().numpy() for p in pages], batch_size=128, mode="f32")
    assert np.max(np.abs(direct.numpy() - want) / np.maximum(np.abs(want), 1.0)) <= 1e-5


def test_corpus_writer_rejected_append_leaves_the_writer_usable(amd):
    # the host-side row bound counts masked positions; a rejected append must not inflate it, and the exact device-side count
    # is what decides in the end
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(2)
    H = 256
    w = (torch.randn(128, H, generator=g) / H**0.5).to(torch.bfloat16).to(dev)
    writer = amd.CorpusWriter(capacity_rows=100, device=dev)
    h = torch.randn(2, 40, H, generator=g).to(torch.bfloat16).to(dev)
    mask = torch.ones(2, 40, dtype=torch.long, device=dev)
    mask[:, 10:] = 0                                       # 20 real rows out of 80 positions
    assert writer.append(h, w, None, mask) == 2
    big = torch.randn(3, 40, H, generator=g).to(torch.bfloat16).to(dev)
    with pytest.raises(RuntimeError, match="capacity"):
        writer.append(big, w, None, torch.ones(3, 40, dtype=torch.long, device=dev))      # 20 + 120 > 100
    assert writer.rows_written() == 20
    assert writer.append(h, w, None, torch.ones(2, 40, dtype=torch.long, device=dev)) == 2   # 20 + 80 fits
    assert len(writer.finish()) == 4


# ---------------------------------------------------------------------------------------------------------------------------
# The head inside a training graph (modeling_colpali.py:65-78 is part of what the reference trainers back-propagate through).

def _truth_grads(hidden, weight, bias, mask, G, extra=None):
    """float64 autograd through oracle/head_oracle.py's restatement of the reference lines: the three gradients, and for each
    the sum of the ABSOLUTE terms of the product that forms it (|dproj| |W|, |dproj|^T |X|, sum |dproj|): dproj is rounded
    to the 16-bit model dtype before the products, so the error of a gradient scales with that sum, not with the (possibly
    cancelled) value."""
    h = hidden.double().requires_grad_(True)
    w = weight.double().requires_grad_(True)
    b = None if bias is None else bias.double().requires_grad_(True)
    cap = {}
    orig = torch.nn.functional.linear

    def linear_keep(x, ww, bb=None):
        y = orig(x, ww, bb)
        y.retain_grad()
        cap["proj"] = y
        return y

    ho.F.linear = linear_keep
    try:
        y = ho.head_literal(h, w, b, mask, extra)
    finally:
        ho.F.linear = orig
    (y * G.double()).sum().backward()
    dproj = cap["proj"].grad.abs()
    flat = dproj.reshape(-1, dproj.shape[-1])
    bounds = ((dproj @ w.detach().abs()), flat.t() @ h.detach().abs().reshape(-1, h.shape[-1]), flat.sum(0))
    return (h.grad, w.grad, None if b is None else b.grad), bounds


def _grad_close(got, want, bound, dtype):
    """|error| <= ulp-of-the-result + one 16-bit rounding per term of the product (worst case, see _truth_grads)."""
    rel = 2.0**-8 if dtype == torch.bfloat16 else 2.0**-11
    err = (got.cpu().double() - want).abs()
    return bool(torch.all(err <= rel * want.abs() + rel * bound + 1e-30))


def test_backward_against_live_reference_autograd_golden(amd):
    z = load_golden("head_colpali_tiny.npz")
    h, w, b, G = (_bf16(z[k]) for k in ("hidden_bf16", "weight_bf16", "bias_bf16", "gout_bf16"))
    mask = torch.from_numpy(z["attention_mask"])
    hx, wx, bx = (t.cuda().requires_grad_(True) for t in (h, w, b))
    out = amd.embedding_head(hx, wx, bx, mask.cuda())
    assert out.requires_grad and out.grad_fn is not None
    (out * G.cuda()).sum().backward()
    wants, bounds = _truth_grads(h, w, b, mask, G)
    for got, want, bound, ref_key in zip((hx.grad, wx.grad, bx.grad), wants, bounds, ("dhidden_bf16", "dweight_bf16", "dbias_bf16")):
        assert got.dtype == torch.bfloat16
        assert _grad_close(got, want, bound, torch.bfloat16)
        # the live reference's own bf16 autograd (five 16-bit roundings between the upstream gradient and dproj where this
        # backward has one) is what is being replaced: it must sit around the same truth, a few times looser
        ref = _bf16(z[ref_key]).double()
        assert bool(torch.all((ref - want).abs() <= 4 * (2.0**-8) * (want.abs() + bound) + 1e-30))
    assert torch.count_nonzero(hx.grad.cpu()[mask == 0]) == 0            # masked positions receive exactly no gradient


@pytest.mark.parametrize("B,S,H,dtype,with_bias,with_extra", [
    (3, 50, 64, torch.bfloat16, True, False),
    (2, 333, 1536, torch.bfloat16, True, True),      # rows not a multiple of the 256-row tile, image mask
    (2, 130, 2048, torch.float16, False, False),
])
def test_backward_random_cases_against_float64_autograd(amd, B, S, H, dtype, with_bias, with_extra):
    hidden, weight, bias, mask = _case(B * 10 + S, B, S, H, dtype)
    bias = bias if with_bias else None
    extra = (torch.arange(S)[None, :] % 4 != 1).expand(B, S).unsqueeze(-1) if with_extra else None
    G = torch.randn(B, S, 128, generator=torch.Generator().manual_seed(S)).to(dtype)
    hx, wx = hidden.cuda().requires_grad_(True), weight.cuda().requires_grad_(True)
    bx = None if bias is None else bias.cuda().requires_grad_(True)
    out = amd.embedding_head(hx, wx, bx, mask.cuda(), None if extra is None else extra.cuda())
    (out.float() * G.cuda().float()).sum().backward()
    (want_h, want_w, want_b), (bd_h, bd_w, bd_b) = _truth_grads(hidden, weight, bias, mask, G, extra)
    assert _grad_close(hx.grad, want_h, bd_h, dtype) and _grad_close(wx.grad, want_w, bd_w, dtype)
    if bx is not None:
        assert _grad_close(bx.grad, want_b, bd_b, dtype)


def test_a_silent_detach_is_impossible(amd):
    hidden, weight, bias, mask = _case(3, 2, 40, 128, torch.bfloat16)
    h, w, b, m = hidden.cuda(), weight.cuda(), bias.cuda(), mask.cuda()
    assert not amd.embedding_head(h, w, b, m).requires_grad               # nothing to differentiate: the plain launch
    for which in range(3):
        args = [t.clone().requires_grad_(i == which) for i, t in enumerate((h, w, b))]
        out = amd.embedding_head(*args, m)
        assert out.requires_grad and out.grad_fn is not None
        out.float().square().sum().backward()
        assert args[which].grad is not None and torch.isfinite(args[which].grad.float()).all()
        with torch.no_grad():
            assert not amd.embedding_head(*args, m).requires_grad
    writer = amd.CorpusWriter(capacity_rows=100, device=h.device)
    with pytest.raises(RuntimeError, match="no_grad"):
        writer.append(h, w.clone().requires_grad_(True), b, m)
    with torch.no_grad():
        assert writer.append(h, w.clone().requires_grad_(True), b, m) == 2
