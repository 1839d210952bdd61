"""GPU: the loss drop-ins (fused MaxSim forward + sparse recompute backward) against the float64 oracle.

Inputs are bf16 (what the models emit under HF bf16 training); the oracle runs on the exact fp32
upcast in float64.  Tolerances: loss 1e-5 relative; gradients are returned in bf16 like autograd
would, so they are compared after the same rounding with one bf16 ulp of slack (2^-8 relative) plus
1e-6 absolute.  Gradients at all-zero padding rows are excluded: the reference splits a gradient
evenly among exact ties, which only occur there, and the model masks those positions
(modeling_colpali.py:72) -- see DESIGN.md "tie semantics".
"""
import numpy as np
import pytest
import torch

from oracle import li_loss_oracle as lo
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import colpali_amd

    colpali_amd._lib.lib()
    return colpali_amd


def grads_close(got, want, mask=None, paths=1):
    """`paths` = number of autograd branches whose bf16-rounded contributions torch sums in bf16 (each adds a rounding
    of the size of the partial terms, which can exceed an ulp of the sum when they cancel)."""
    got = got.float().cpu()
    want = want.float()
    want_bf = want.to(torch.bfloat16).float()
    tol = want.abs() * 2.0**-7 + 1e-6
    if paths > 1:
        tol = tol * paths + float(want.abs().max()) * 2.0**-9 * paths
    bad = (got - want_bf).abs() > tol
    if mask is not None:
        bad &= mask
    return int(bad.sum()) == 0


VARIANTS = {
    "default": dict(),
    "nonorm": dict(normalize_scores=False),
    "nonorm_T05": dict(normalize_scores=False, temperature=0.5),
    "filter": dict(normalize_scores=False, pos_aware_negative_filtering=True),
}


@pytest.mark.parametrize("cls,kind", [("ColbertPairwiseCELoss", "pairwise"), ("ColbertLoss", "infonce")])
@pytest.mark.parametrize("offset", [0, 6])
def test_small_golden_shapes_loss_and_grads(amd, cls, kind, offset):
    z = load_golden("loss_small.npz")
    Qb = torch.from_numpy(z["Q"]).to(torch.bfloat16)
    Db = torch.from_numpy(z["D"]).to(torch.bfloat16)
    q_real = (Qb.float().abs().sum(-1, keepdim=True) > 0)
    d_real = (Db.float().abs().sum(-1, keepdim=True) > 0)
    for vname, kw in VARIANTS.items():
        want_loss, want_dq, want_dd = lo.loss_and_grads(kind, Qb.float(), Db.float(), offset=offset, **kw)
        q = Qb.cuda().requires_grad_(True)
        d = Db.cuda().requires_grad_(True)
        loss = getattr(amd, cls)(**kw)(q, d, offset=offset)
        assert loss.dtype == torch.bfloat16 and loss.dim() == 0
        assert abs(float(loss.detach()) - float(want_loss)) <= 2.0**-8 * abs(float(want_loss)) + 1e-6, (cls, vname)
        loss.backward()
        assert q.grad.dtype == torch.bfloat16 and q.grad.shape == q.shape and d.grad.shape == d.shape
        assert grads_close(q.grad, want_dq, q_real.expand_as(want_dq)), (cls, vname, "dQ")
        assert grads_close(d.grad, want_dd, d_real.expand_as(want_dd)), (cls, vname, "dD")


def test_fp32_scores_of_the_fused_forward_match_oracle_tightly(amd):
    z = load_golden("loss_small.npz")
    Qb = torch.from_numpy(z["Q"]).to(torch.bfloat16)
    Db = torch.from_numpy(z["D"]).to(torch.bfloat16)
    s = amd.loss.maxsim(Qb.cuda(), Db.cuda()).cpu().double()
    want = torch.einsum("bnd,csd->bcns", Qb.double(), Db.double()).amax(3).sum(2)
    assert torch.max((s - want).abs() / want.abs().clamp_min(1.0)) < 1e-5


@pytest.mark.parametrize("B,C,Lq,Ld,offset", [(16, 64, 32, 780, 0), (16, 64, 32, 780, 32), (8, 40, 45, 300, 16)])
def test_training_shapes_pairwise(amd, B, C, Lq, Ld, offset):
    g = torch.Generator().manual_seed(B * 1000 + C + offset)
    Q = torch.nn.functional.normalize(torch.randn(B, Lq, 128, generator=g), dim=-1)
    D = torch.nn.functional.normalize(torch.randn(C, Ld, 128, generator=g), dim=-1)
    for b in range(B):   # plant the positives
        D[offset + b, 5 : 5 + Lq] = torch.nn.functional.normalize(Q[b] + 0.5 * torch.randn(Lq, 128, generator=g), dim=-1)
    Qb, Db = Q.to(torch.bfloat16), D.to(torch.bfloat16)
    want_loss, want_dq, want_dd = lo.loss_and_grads("pairwise", Qb.float(), Db.float(), offset=offset, normalize_scores=False)
    q, d = Qb.cuda().requires_grad_(True), Db.cuda().requires_grad_(True)
    loss = amd.ColbertPairwiseCELoss(normalize_scores=False)(query_embeddings=q, doc_embeddings=d, offset=offset)
    assert abs(float(loss.detach()) - float(want_loss)) <= 2.0**-8 * abs(float(want_loss)) + 1e-6
    loss.backward()
    assert grads_close(q.grad, want_dq) and grads_close(d.grad, want_dd)
    # 2-sparse upstream gradient: at most 2 documents per query receive any gradient
    touched = (d.grad.float().abs().sum(dim=(1, 2)) > 0).sum().item()
    assert touched <= 2 * B


def test_pairs_argmax_matches_oracle_routing(amd):
    from oracle import maxsim_oracle as mo

    g = torch.Generator().manual_seed(4)
    Q = torch.nn.functional.normalize(torch.randn(5, 40, 128, generator=g), dim=-1).to(torch.bfloat16)
    D = torch.nn.functional.normalize(torch.randn(7, 100, 128, generator=g), dim=-1).to(torch.bfloat16)
    pairs = torch.tensor([[0, 0], [0, 6], [1, 3], [4, 2], [4, 3], [4, 6]], dtype=torch.int32)
    offs = (torch.arange(8, dtype=torch.int32) * 100)
    s, am = amd.loss.maxsim_pairs(Q.cuda(), D.cuda(), offs.cuda(), pairs.cuda())
    ws, wam = mo.maxsim_argmax_f32(Q.float().numpy(), D.float().numpy().reshape(-1, 128), offs.numpy(), None)
    for k, (b, c) in enumerate(pairs.tolist()):
        assert abs(float(s[k]) - ws[b, c]) < 1e-5 * max(1.0, abs(ws[b, c]))
        np.testing.assert_array_equal(am[k].cpu().numpy(), wam[b, c])


def test_reference_known_answers_on_gpu(amd):
    # tests/loss/test_li_losses.py:137-147 and :76-88 restated with dim=128 bf16 tensors on the GPU
    q = torch.zeros(2, 1, 128, dtype=torch.bfloat16, device="cuda")
    loss = amd.ColbertPairwiseCELoss(temperature=1.0, normalize_scores=False)(q, q.clone())
    assert abs(float(loss) - np.log(2.0)) < 4e-3
    q3 = torch.zeros(3, 1, 128, dtype=torch.bfloat16, device="cuda")
    loss = amd.ColbertLoss(temperature=1.0, normalize_scores=False)(q3, q3.clone())
    assert abs(float(loss) - np.log(3.0)) < 8e-3


def test_loss_modules_have_no_state_and_reject_cpu(amd):
    m = amd.ColbertPairwiseCELoss()
    assert len(m.state_dict()) == 0 and len(list(m.parameters())) == 0
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(2, 1, 128, dtype=torch.bfloat16), torch.zeros(2, 1, 128, dtype=torch.bfloat16))


@pytest.mark.parametrize("cls,kind", [("ColbertNegativeCELoss", "negative_ce"), ("ColbertPairwiseNegativeCELoss", "pairwise_negative_ce")])
@pytest.mark.parametrize("offset", [0, 6])
def test_explicit_negative_variants(amd, cls, kind, offset):
    z = load_golden("loss_negatives.npz")
    Qb, Db, Nb = (torch.from_numpy(z[k]).to(torch.bfloat16) for k in ("Q", "D", "N"))
    q_real = (Qb.float().abs().sum(-1, keepdim=True) > 0)
    d_real = (Db.float().abs().sum(-1, keepdim=True) > 0)
    n_real = (Nb.float().abs().sum(-1, keepdim=True) > 0)
    variants = {"default": dict(), "nonorm_w0": dict(normalize_scores=False, in_batch_term_weight=0.0),
                "T1_w03": dict(temperature=1.0, in_batch_term_weight=0.3)}
    for vname, kw in variants.items():
        want_loss, want_dq, want_dd, want_dn = lo.negatives_loss_and_grads(kind, Qb.float(), Db.float(), Nb.float(), offset=offset, **kw)
        q, d, n = (t.cuda().requires_grad_(True) for t in (Qb, Db, Nb))
        loss = getattr(amd, cls)(**kw)(q, d, n, offset=offset)
        assert abs(float(loss.detach()) - float(want_loss)) <= 2.0**-7 * abs(float(want_loss)) + 1e-6, (cls, vname)
        loss.backward()
        assert grads_close(q.grad, want_dq, q_real.expand_as(want_dq), paths=3), (cls, vname, "dQ")
        assert grads_close(d.grad, want_dd, d_real.expand_as(want_dd), paths=2), (cls, vname, "dD")
        assert grads_close(n.grad, want_dn, n_real.expand_as(want_dn)), (cls, vname, "dN")


@pytest.mark.parametrize("lowp", [torch.bfloat16, torch.float16])
def test_fp32_embeddings_under_autocast_follow_the_reference_autocast_semantics(amd, lowp):
    """Under torch.autocast the reference's einsum runs in the autocast dtype on casts of the fp32 embeddings
    (the models emit fp32 there); the drop-in does the same cast, gradients come back in fp32."""
    g = torch.Generator().manual_seed(17)
    B, C, Lq, Ld = 8, 24, 20, 100
    Q = torch.nn.functional.normalize(torch.randn(B, Lq, 128, generator=g), dim=-1)
    D = torch.nn.functional.normalize(torch.randn(C, Ld, 128, generator=g), dim=-1)
    for b in range(B):
        D[b, :Lq] = torch.nn.functional.normalize(Q[b] + 0.5 * torch.randn(Lq, 128, generator=g), dim=-1)
    want_loss, want_dq, want_dd = lo.loss_and_grads("pairwise", Q.to(lowp).float(), D.to(lowp).float(), normalize_scores=False)
    q, d = Q.cuda().requires_grad_(True), D.cuda().requires_grad_(True)
    with torch.autocast("cuda", dtype=lowp):
        loss = amd.ColbertPairwiseCELoss(normalize_scores=False)(q, d)
    loss.backward()
    assert q.grad.dtype == torch.float32 and d.grad.dtype == torch.float32
    assert abs(float(loss.detach()) - float(want_loss)) <= 1e-3 * abs(float(want_loss)) + 1e-6
    # gradients pass through one 16-bit rounding (the kernel returns them in the autocast dtype) before the fp32 cast
    rel = 2.0**-7 if lowp == torch.bfloat16 else 2.0**-10
    assert torch.all((q.grad.cpu() - want_dq.float()).abs() <= want_dq.float().abs() * rel + 1e-5)
    assert torch.all((d.grad.cpu() - want_dd.float()).abs() <= want_dd.float().abs() * rel + 1e-5)
    # fp32 without autocast: computed in fp32 (exact-fp32 MFMA), like the reference does -- no silent down-conversion
    loss32 = amd.ColbertPairwiseCELoss(normalize_scores=False)(Q.cuda(), D.cuda())
    want32, _, _ = lo.loss_and_grads("pairwise", Q, D, normalize_scores=False)
    assert loss32.dtype == torch.float32 and abs(float(loss32) - float(want32)) <= 1e-5 * abs(float(want32)) + 1e-6


def test_float16_loss_and_grads(amd):
    g = torch.Generator().manual_seed(23)
    B, C, Lq, Ld = 6, 18, 33, 70
    Q = torch.nn.functional.normalize(torch.randn(B, Lq, 128, generator=g), dim=-1).to(torch.float16)
    D = torch.nn.functional.normalize(torch.randn(C, Ld, 128, generator=g), dim=-1).to(torch.float16)
    want_loss, want_dq, want_dd = lo.loss_and_grads("infonce", Q.float(), D.float(), offset=6, temperature=0.5)
    q, d = Q.cuda().requires_grad_(True), D.cuda().requires_grad_(True)
    loss = amd.ColbertLoss(temperature=0.5)(q, d, offset=6)
    assert loss.dtype == torch.float16
    assert abs(float(loss.detach()) - float(want_loss)) <= 2.0**-10 * abs(float(want_loss)) + 1e-4
    loss.backward()
    assert torch.all((q.grad.float().cpu() - want_dq.float()).abs() <= want_dq.float().abs() * 2.0**-9 + 2e-5)
    assert torch.all((d.grad.float().cpu() - want_dd.float()).abs() <= want_dd.float().abs() * 2.0**-9 + 2e-5)


# ---------------------------------------------------------------------------------------------------------
# fp32 embeddings (a model trained / evaluated in fp32, no autocast): generic kernels, exact-fp32 MFMA.
# Here the golden vectors are the live reference's own fp32 outputs (loss, dQ, dD from its autograd), so the
# comparison is reference-vs-HIP directly, not through the oracle.

@pytest.mark.parametrize("cls", ["ColbertPairwiseCELoss", "ColbertLoss"])
@pytest.mark.parametrize("offset", [0, 6])
def test_fp32_loss_and_grads_against_reference_goldens(amd, cls, offset):
    z = load_golden("loss_small.npz")
    Q, D = torch.from_numpy(z["Q"]), torch.from_numpy(z["D"])
    q_real = (Q.abs().sum(-1, keepdim=True) > 0)
    d_real = (D.abs().sum(-1, keepdim=True) > 0)
    for vname, kw in VARIANTS.items():
        key = f"{cls}_{vname}_off{offset}"
        if key + "_loss" not in z.files:
            continue
        q, d = Q.cuda().requires_grad_(True), D.cuda().requires_grad_(True)
        loss = getattr(amd, cls)(**kw)(q, d, offset=offset)
        assert loss.dtype == torch.float32 and loss.dim() == 0
        want = float(z[key + "_loss"])
        assert abs(float(loss.detach()) - want) <= 1e-5 * abs(want) + 1e-6, key
        loss.backward()
        for got, name, mask in ((q.grad, "dQ", q_real), (d.grad, "dD", d_real)):
            w = torch.from_numpy(z[f"{key}_{name}"])
            assert got.dtype == torch.float32 and got.shape == w.shape
            bad = ((got.cpu() - w).abs() > 1e-4 * w.abs() + 1e-6) & mask.expand_as(w)
            assert int(bad.sum()) == 0, (key, name)


@pytest.mark.parametrize("cls", ["ColbertNegativeCELoss", "ColbertPairwiseNegativeCELoss"])
def test_fp32_explicit_negatives_against_reference_goldens(amd, cls):
    z = load_golden("loss_negatives.npz")
    Q, D, N = (torch.from_numpy(z[k]) for k in ("Q", "D", "N"))
    q_real = (Q.abs().sum(-1, keepdim=True) > 0)
    n_real = (N.abs().sum(-1, keepdim=True) > 0)
    variants = {"default": dict(), "nonorm_w0": dict(normalize_scores=False, in_batch_term_weight=0.0),
                "T1_w03": dict(temperature=1.0, in_batch_term_weight=0.3)}
    for offset in (0, 6):
        for vname, kw in variants.items():
            key = f"{cls}_{vname}_off{offset}"
            q, d, n = (t.cuda().requires_grad_(True) for t in (Q, D, N))
            loss = getattr(amd, cls)(**kw)(q, d, n, offset=offset)
            want = float(z[key + "_loss"])
            assert abs(float(loss.detach()) - want) <= 1e-5 * abs(want) + 1e-6, key
            loss.backward()
            for got, name, mask in ((q.grad, "dQ", q_real), (n.grad, "dN", n_real)):
                w = torch.from_numpy(z[f"{key}_{name}"])
                bad = ((got.cpu() - w).abs() > 1e-4 * w.abs() + 1e-6) & mask.expand_as(w)
                assert int(bad.sum()) == 0, (key, name)


def test_width_320_bf16_loss_and_grads(amd):
    # ColQwen3 geometry (dim=320): generic forward, generic pair-list backward
    g = torch.Generator().manual_seed(29)
    B, C, Lq, Ld = 6, 18, 40, 90
    Q = torch.nn.functional.normalize(torch.randn(B, Lq, 320, generator=g), dim=-1).to(torch.bfloat16)
    D = torch.nn.functional.normalize(torch.randn(C, Ld, 320, generator=g), dim=-1).to(torch.bfloat16)
    want_loss, want_dq, want_dd = lo.loss_and_grads("pairwise", Q.float(), D.float(), offset=6, normalize_scores=False)
    q, d = Q.cuda().requires_grad_(True), D.cuda().requires_grad_(True)
    loss = amd.ColbertPairwiseCELoss(normalize_scores=False)(q, d, offset=6)
    assert abs(float(loss.detach()) - float(want_loss)) <= 2.0**-8 * abs(float(want_loss)) + 1e-6
    loss.backward()
    assert q.grad.shape == q.shape and d.grad.shape == d.shape
    assert grads_close(q.grad, want_dq) and grads_close(d.grad, want_dd)


def test_generic_pairs_argmax_matches_oracle_routing_fp32(amd):
    from oracle import maxsim_oracle as mo

    g = torch.Generator().manual_seed(5)
    Q = torch.nn.functional.normalize(torch.randn(5, 40, 64, generator=g), dim=-1)
    D = torch.nn.functional.normalize(torch.randn(7, 100, 64, generator=g), dim=-1)
    pairs = torch.tensor([[0, 0], [0, 6], [1, 3], [4, 2], [4, 3], [4, 6]], dtype=torch.int32)
    offs = (torch.arange(8, dtype=torch.int32) * 100)
    s, am = amd.loss.maxsim_pairs(Q.cuda(), D.cuda(), offs.cuda(), pairs.cuda())
    ws, wam = mo.maxsim_argmax_f32(Q.numpy(), D.numpy().reshape(-1, 64), offs.numpy(), None)
    for k, (b, c) in enumerate(pairs.tolist()):
        assert abs(float(s[k]) - ws[b, c]) < 1e-5 * max(1.0, abs(ws[b, c]))
        np.testing.assert_array_equal(am[k].cpu().numpy(), wam[b, c])


# ---------------------------------------------------------------------------------------------------------
# use_smooth_max=True: tau * logsumexp over document rows (late_interaction_losses.py:40-44, :88-90).
# fp32 inputs are compared with the live reference's own outputs (tests/golden/loss_smooth.npz); bf16 inputs with the
# float64 oracle on the same bf16-valued numbers.  Every row takes part here (zero padding rows contribute exp(0)), so
# there are no ties and no rows to exclude.

SMOOTH = {"tau01": dict(use_smooth_max=True),
          "tau002_nonorm_T1": dict(use_smooth_max=True, tau=0.02, normalize_scores=False, temperature=1.0)}
SMOOTH_NEG = {"tau01": dict(use_smooth_max=True),
              "tau05_T1_w03": dict(use_smooth_max=True, tau=0.5, temperature=1.0, in_batch_term_weight=0.3)}


def _close32(got, want, rtol=2e-3, atol=1e-5):
    want = torch.as_tensor(want)
    return bool(torch.all((got.float().cpu() - want).abs() <= rtol * want.abs() + atol))


@pytest.mark.parametrize("cls", ["ColbertPairwiseCELoss", "ColbertLoss"])
def test_smooth_max_fp32_against_reference_goldens(amd, cls):
    z = load_golden("loss_smooth.npz")
    zs = load_golden("loss_small.npz")
    Q, D = torch.from_numpy(zs["Q"]), torch.from_numpy(zs["D"])
    for vname, kw in SMOOTH.items():
        for offset in (0, 6):
            key = f"{cls}_{vname}_off{offset}"
            q, d = Q.cuda().requires_grad_(True), D.cuda().requires_grad_(True)
            loss = getattr(amd, cls)(**kw)(q, d, offset=offset)
            want = float(z[key + "_loss"])
            assert loss.dtype == torch.float32 and abs(float(loss.detach()) - want) <= 2e-4 * abs(want) + 1e-6, key
            loss.backward()
            assert _close32(q.grad, z[key + "_dQ"]), (key, "dQ")
            if key + "_dD" in z.files:
                assert _close32(d.grad, z[key + "_dD"]), (key, "dD")


@pytest.mark.parametrize("cls", ["ColbertNegativeCELoss", "ColbertPairwiseNegativeCELoss"])
def test_smooth_max_explicit_negatives_fp32_against_reference_goldens(amd, cls):
    z = load_golden("loss_smooth.npz")
    zn = load_golden("loss_negatives.npz")
    Q, D, N = (torch.from_numpy(zn[k]) for k in ("Q", "D", "N"))
    for vname, kw in SMOOTH_NEG.items():
        for offset in (0, 6):
            key = f"{cls}_{vname}_off{offset}"
            q, d, n = (t.cuda().requires_grad_(True) for t in (Q, D, N))
            loss = getattr(amd, cls)(**kw)(q, d, n, offset=offset)
            want = float(z[key + "_loss"])
            assert abs(float(loss.detach()) - want) <= 2e-4 * abs(want) + 1e-6, key
            loss.backward()
            assert _close32(q.grad, z[key + "_dQ"]), (key, "dQ")
            if key + "_dN" in z.files:
                assert _close32(n.grad, z[key + "_dN"]), (key, "dN")


@pytest.mark.parametrize("B,C,Lq,Ld,dim,dtype", [(8, 24, 20, 100, 128, torch.bfloat16), (5, 11, 40, 70, 128, torch.bfloat16),
                                                 (4, 9, 33, 45, 320, torch.bfloat16), (6, 12, 70, 50, 64, torch.float16)])
def test_smooth_max_16bit_inputs_against_oracle(amd, B, C, Lq, Ld, dim, dtype):
    g = torch.Generator().manual_seed(B * 100 + C)
    Q = torch.nn.functional.normalize(torch.randn(B, Lq, dim, generator=g), dim=-1).to(dtype)
    D = torch.nn.functional.normalize(torch.randn(C, Ld, dim, generator=g), dim=-1).to(dtype)
    Q[1, Lq - 3:] = 0       # padded query rows still contribute tau * log(Ld) each, as in the reference
    D[2, Ld - 7:] = 0
    want_loss, want_dq, want_dd = lo.loss_and_grads("infonce", Q.float(), D.float(), offset=2, temperature=0.5,
                                                    use_smooth_max=True, tau=0.1)
    q, d = Q.cuda().requires_grad_(True), D.cuda().requires_grad_(True)
    loss = amd.ColbertLoss(temperature=0.5, use_smooth_max=True, tau=0.1)(q, d, offset=2)
    eps = 2.0**-8 if dtype == torch.bfloat16 else 2.0**-10
    assert loss.dtype == dtype and abs(float(loss.detach()) - float(want_loss)) <= eps * abs(float(want_loss)) + 1e-4
    loss.backward()
    assert torch.all((q.grad.float().cpu() - want_dq.float()).abs() <= want_dq.float().abs() * 2 * eps + 2e-5)
    assert torch.all((d.grad.float().cpu() - want_dd.float()).abs() <= want_dd.float().abs() * 2 * eps + 2e-5)


def test_smooth_max_scores_match_float64_logsumexp(amd):
    # forward alone, dense and pair-list entry points, including a query longer than 64 tokens (sub-passes)
    g = torch.Generator().manual_seed(3)
    Q = torch.nn.functional.normalize(torch.randn(7, 75, 128, generator=g), dim=-1).to(torch.bfloat16)
    D = torch.nn.functional.normalize(torch.randn(19, 130, 128, generator=g), dim=-1).to(torch.bfloat16)
    tau = 0.05
    want = (tau * torch.logsumexp(torch.einsum("bnd,csd->bcns", Q.double(), D.double()) / tau, dim=3)).sum(2)
    got = amd.loss.maxsim_smooth(Q.cuda(), D.cuda(), tau).cpu().double()
    assert torch.max((got - want).abs() / want.abs().clamp_min(1.0)) < 1e-5
    pairs = torch.tensor([[0, 3], [0, 18], [2, 0], [6, 5], [6, 6]], dtype=torch.int32)
    got_p = amd.loss.maxsim_smooth_paired(Q.cuda(), D.cuda(), pairs.cuda(), tau).cpu().double()
    want_p = torch.stack([want[b, c] for b, c in pairs.tolist()])
    assert torch.max((got_p - want_p).abs() / want_p.abs().clamp_min(1.0)) < 1e-5


SIGMOID_VARIANTS = {"default": dict(), "nonorm_T1": dict(normalize_scores=False, temperature=1.0),
                    "filter_T05": dict(pos_aware_negative_filtering=True, temperature=0.5),
                    "smooth_T1": dict(use_smooth_max=True, temperature=1.0)}


def test_sigmoid_loss_fp32_against_reference_goldens_and_bf16_against_oracle(amd):
    """ColbertSigmoidLoss (late_interaction_losses.py:401-465), square in-batch case."""
    z = load_golden("loss_sigmoid.npz")
    zs = load_golden("loss_small.npz")
    Q, D = torch.from_numpy(zs["Q"]), torch.from_numpy(zs["D"])[:6].contiguous()
    q_real = (Q.abs().sum(-1, keepdim=True) > 0)
    d_real = (D.abs().sum(-1, keepdim=True) > 0)
    for vname, kw in SIGMOID_VARIANTS.items():
        q, d = Q.cuda().requires_grad_(True), D.cuda().requires_grad_(True)
        loss = amd.ColbertSigmoidLoss(**kw)(q, d)
        want = float(z[vname + "_loss"])
        assert loss.dtype == torch.float32 and abs(float(loss.detach()) - want) <= 2e-4 * abs(want) + 1e-6, vname
        loss.backward()
        smooth = kw.get("use_smooth_max", False)     # hard max: exclude the exact ties at all-zero padding rows
        for got, name, mask in ((q.grad, "dQ", q_real), (d.grad, "dD", d_real)):
            w = torch.from_numpy(z[f"{vname}_{name}"])
            bad = (got.cpu() - w).abs() > 2e-3 * w.abs() + 1e-5
            if not smooth:
                bad &= mask.expand_as(w)
            assert int(bad.sum()) == 0, (vname, name)
        # bf16 inputs against the float64 oracle
        Qb, Db = Q.to(torch.bfloat16), D.to(torch.bfloat16)
        want_loss, want_dq, want_dd = lo.loss_and_grads("sigmoid", Qb.float(), Db.float(), **kw)
        q, d = Qb.cuda().requires_grad_(True), Db.cuda().requires_grad_(True)
        loss = amd.ColbertSigmoidLoss(**kw)(q, d)
        assert loss.dtype == torch.bfloat16
        assert abs(float(loss.detach()) - float(want_loss)) <= 2.0**-7 * abs(float(want_loss)) + 1e-4, vname
        loss.backward()
        assert grads_close(q.grad, want_dq, None if smooth else q_real.expand_as(want_dq), paths=2), (vname, "dQ bf16")
        assert grads_close(d.grad, want_dd, None if smooth else d_real.expand_as(want_dd), paths=2), (vname, "dD bf16")


def test_sigmoid_loss_runs_on_the_fused_epilogue_and_captures_as_one_graph(amd, monkeypatch):
    """Round 6: ColbertSigmoidLoss (late_interaction_losses.py:440-465) = the MaxSim forward + ONE msim_loss_epilogue(MSIM_LOSS_SIGMOID)
    launch -- no torch softplus / scatter / mask behind the kernels, nothing that synchronises the host (the step captures as a
    hipGraph and replays to the same bits) -- and a non-square batch fails like the reference's `-scores.view(-1) * pos_mask`."""
    import torch.nn.functional as F

    def boom(*a, **k):
        raise AssertionError("torch softplus in the sigmoid loss: the epilogue is fused")

    monkeypatch.setattr(F, "softplus", boom)
    g = torch.Generator().manual_seed(11)
    Q = torch.nn.functional.normalize(torch.randn(16, 24, 128, generator=g), dim=-1).to(torch.bfloat16).cuda()
    D = torch.nn.functional.normalize(torch.randn(16, 300, 128, generator=g), dim=-1).to(torch.bfloat16).cuda()
    mod = amd.ColbertSigmoidLoss()
    q, d = Q.clone().requires_grad_(True), D.clone().requires_grad_(True)
    mod(q, d).backward()
    eager = (q.grad.clone(), d.grad.clone())
    qs, ds = Q.clone().requires_grad_(True), D.clone().requires_grad_(True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            qs.grad = ds.grad = None
            mod(qs, ds).backward()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    qs.grad = ds.grad = None
    with torch.cuda.graph(graph):
        loss = mod(qs, ds)
        loss.backward()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(qs.grad, eager[0]) and torch.equal(ds.grad, eager[1])
    with pytest.raises(RuntimeError, match="must match the size of tensor b"):
        mod(Q[:8], D)


# ---------------------------------------------------------------------------------------------------------
# BASELINE config 5 at its stated per-rank shape: bs 256 over 8 ranks -> B = 32 local queries, C = world * B = 256 gathered
# documents, offset = rank * B (trainer/contrastive_trainer.py:143-160), Lq = 32, Ld = 780 (ColQwen2, left padded).

def _config5_inputs(offset, seed=0):
    g = torch.Generator().manual_seed(5000 + offset + seed)
    B, C, Lq, Ld = 32, 256, 32, 780
    Q = torch.nn.functional.normalize(torch.randn(B, Lq, 128, generator=g), dim=-1)
    D = torch.nn.functional.normalize(torch.randn(C, Ld, 128, generator=g), dim=-1)
    q_pad = torch.randint(0, 12, (B,), generator=g)
    d_pad = torch.randint(0, 500, (C,), generator=g)
    for b in range(B):                                   # left padding, rows exactly zero (modeling_colqwen2.py:36, :69)
        Q[b, : int(q_pad[b])] = 0
    for c in range(C):
        D[c, : int(d_pad[c])] = 0
    for b in range(B):                                   # the positives: noisy copies of the query's tokens inside the page
        c = offset + b
        lo = int(d_pad[c])
        rows = lo + torch.randperm(Ld - lo, generator=g)[:Lq]
        D[c, rows] = torch.nn.functional.normalize(Q[b] + 0.6 * torch.randn(Lq, 128, generator=g), dim=-1) * (Q[b].abs().sum(-1, keepdim=True) > 0)
    # a hard negative for a few queries: another rank's page that out-scores the positive
    for b in (3, 17):
        c = (offset + b + 40) % C
        lo = int(d_pad[c])
        rows = lo + torch.randperm(Ld - lo, generator=g)[:Lq]
        D[c, rows] = torch.nn.functional.normalize(Q[b] + 0.3 * torch.randn(Lq, 128, generator=g), dim=-1) * (Q[b].abs().sum(-1, keepdim=True) > 0)
    return Q, D


@pytest.mark.parametrize("offset", [0, 96, 224])
@pytest.mark.parametrize("cls,kind", [("ColbertPairwiseCELoss", "pairwise"), ("ColbertLoss", "infonce")])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_config5_full_shape_loss_and_grads(amd, offset, cls, kind, dtype):
    Q, D = _config5_inputs(offset)
    Qx, Dx = Q.to(dtype), D.to(dtype)
    kw = dict(normalize_scores=False) if kind == "pairwise" else dict()         # shipped configs / class defaults
    want_loss, want_dq, want_dd = lo.loss_and_grads(kind, Qx.float(), Dx.float(), offset=offset, **kw)
    q_real = (Qx.float().abs().sum(-1, keepdim=True) > 0)
    d_real = (Dx.float().abs().sum(-1, keepdim=True) > 0)
    q, d = Qx.cuda().requires_grad_(True), Dx.cuda().requires_grad_(True)
    loss = getattr(amd, cls)(**kw)(query_embeddings=q, doc_embeddings=d, offset=offset)   # keyword call, contrastive_trainer.py:160
    assert loss.dtype == dtype and loss.dim() == 0
    loss.backward()
    if dtype == torch.float32:
        assert abs(float(loss.detach()) - float(want_loss)) <= 1e-5 * abs(float(want_loss)) + 1e-6
        for got, want, mask in ((q.grad, want_dq, q_real), (d.grad, want_dd, d_real)):
            bad = ((got.cpu().double() - want).abs() > 1e-4 * want.abs() + 1e-6) & mask.expand_as(want)
            assert int(bad.sum()) == 0
    else:
        assert abs(float(loss.detach()) - float(want_loss)) <= 2.0**-8 * abs(float(want_loss)) + 1e-6
        assert grads_close(q.grad, want_dq, q_real.expand_as(want_dq))
        assert grads_close(d.grad, want_dd, d_real.expand_as(want_dd))
    if kind == "pairwise":
        assert (d.grad.float().abs().sum(dim=(1, 2)) > 0).sum().item() <= 2 * Q.shape[0]


def test_positional_three_argument_call_of_the_evaluation_path(amd):
    """trainer/contrastive_trainer.py:221-224: prediction_step calls loss_func(query, doc, neg_doc) / loss_func(query, doc)
    positionally, without `offset`."""
    z = load_golden("loss_negatives.npz")
    Q, D, N = (torch.from_numpy(z[k]) for k in ("Q", "D", "N"))
    for cls in ("ColbertNegativeCELoss", "ColbertPairwiseNegativeCELoss"):
        loss = getattr(amd, cls)()(Q.cuda(), D.cuda(), N.cuda())                   # positional, offset defaults to 0
        want = float(z[f"{cls}_default_off0_loss"])
        assert abs(float(loss) - want) <= 1e-5 * abs(want) + 1e-6
    zs = load_golden("loss_small.npz")
    for cls in ("ColbertPairwiseCELoss", "ColbertLoss"):
        loss = getattr(amd, cls)()(Q.cuda(), D.cuda())                             # two positional arguments
        want = float(zs[f"{cls}_default_off0_loss"])
        assert abs(float(loss) - want) <= 1e-5 * abs(want) + 1e-6


def test_offset_beyond_the_gathered_documents_raises_like_the_reference(amd):
    # the reference fails on shapes when offset + B > C (diagonal / index out of range); never a silent garbage loss
    Q = torch.nn.functional.normalize(torch.randn(4, 8, 128), dim=-1).to(torch.bfloat16).cuda()
    D = torch.nn.functional.normalize(torch.randn(6, 16, 128), dim=-1).to(torch.bfloat16).cuda()
    N = torch.nn.functional.normalize(torch.randn(4, 2, 16, 128), dim=-1).to(torch.bfloat16).cuda()
    for cls in ("ColbertPairwiseCELoss", "ColbertLoss"):
        with pytest.raises((RuntimeError, IndexError, ValueError)):
            getattr(amd, cls)()(Q, D, offset=3)
    for cls in ("ColbertNegativeCELoss", "ColbertPairwiseNegativeCELoss"):
        with pytest.raises((RuntimeError, IndexError, ValueError)):
            getattr(amd, cls)()(Q, D, N, offset=3)
    bad = torch.tensor([[0, 0], [1, 6]], dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError):
        amd.maxsim_paired(Q, D, bad)


def test_loss_is_fp32_under_autocast_like_the_reference(amd):
    # with bf16 embeddings under torch.autocast the reference's sum / softplus / cross_entropy run in fp32 (autocast fp32 ops)
    Q = torch.nn.functional.normalize(torch.randn(4, 8, 128), dim=-1).to(torch.bfloat16).cuda()
    D = torch.nn.functional.normalize(torch.randn(4, 16, 128), dim=-1).to(torch.bfloat16).cuda()
    for cls in ("ColbertPairwiseCELoss", "ColbertLoss", "ColbertSigmoidLoss"):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            assert getattr(amd, cls)()(Q, D).dtype == torch.float32
        assert getattr(amd, cls)()(Q, D).dtype == torch.bfloat16


@pytest.mark.parametrize("cls,kind", [("ColbertPairwiseCELoss", "pairwise"), ("ColbertLoss", "infonce")])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_symmetric_loss_documents_in_the_query_slot(amd, cls, kind, dtype):
    """trainer/contrastive_trainer.py:202-206 (`compute_symetric_loss`): the second term is
    `_compute_loss_from_outputs(doc_outputs, query_outputs)` -- the loss called with the PAGES as `query_embeddings`
    ([B, 780, 128]: 25 token tiles per "query", beyond the tuned kernels' 4) and the queries as `doc_embeddings`
    ([B, 32, 128]).  Same oracle, same tolerances as the forward direction."""
    Q, D = _config5_inputs(0, seed=7)
    B = 12
    pages, queries = D[:B].to(dtype), Q[:B].to(dtype)
    for kw in (dict(), dict(normalize_scores=False)):
        want_loss, want_dp, want_dq = lo.loss_and_grads(kind, pages.float(), queries.float(), offset=0, **kw)
        p_real = (pages.float().abs().sum(-1, keepdim=True) > 0)
        q_real = (queries.float().abs().sum(-1, keepdim=True) > 0)
        p, q = pages.cuda().requires_grad_(True), queries.cuda().requires_grad_(True)
        loss = getattr(amd, cls)(**kw)(query_embeddings=p, doc_embeddings=q, offset=0)
        assert loss.dtype == dtype and loss.dim() == 0
        loss.backward()
        if dtype == torch.float32:
            assert abs(float(loss.detach()) - float(want_loss)) <= 1e-5 * abs(float(want_loss)) + 1e-6, kw
            for got, want, mask in ((p.grad, want_dp, p_real), (q.grad, want_dq, q_real)):
                if kind == "infonce" and kw:
                    # un-normalised scores of 780-token "queries" (~300) over T = 0.02: logits of ~15 000, so one fp32 ulp of a score
                    # moves a logit by 1.5e-3 and the softmax gradient with it -- ill-conditioned in fp32 for the reference too;
                    # compare in norm instead of element by element
                    diff = ((got.cpu().double() - want) * mask).norm() / (want * mask).norm()
                    assert float(diff) < 5e-3, (kw, float(diff))
                    if got is not p.grad:
                        continue
                    # ... and element by element where it IS well conditioned: the same float64 chain evaluated at the scores this
                    # library produced (softmax of OUR fp32 scores, then the oracle's d(score)/d(embedding) in float64).  What is left
                    # is the backward kernels' own error, not the softmax's sensitivity to the last bit of a score of ~300.
                    ours_scores = amd.maxsim(p.detach(), q.detach()).cpu().double()
                    p64 = pages.double().requires_grad_(True)
                    q64 = queries.double().requires_grad_(True)
                    s64, pos_idx = lo._scores(p64, q64, 0, False, False, 0.95, 0.5)
                    Gs = torch.softmax(ours_scores / 0.02, dim=1)
                    Gs[torch.arange(B), pos_idx] -= 1.0
                    (s64 * (Gs / (0.02 * B))).sum().backward()
                    for g_, w_, m_ in ((p.grad, p64.grad, p_real), (q.grad, q64.grad, q_real)):
                        bad = ((g_.cpu().double() - w_).abs() > 1e-4 * w_.abs() + 1e-6 * float(w_.abs().max())) & m_.expand_as(w_)
                        assert int(bad.sum()) == 0, kw
                    continue
                bad = ((got.cpu().double() - want).abs() > 1e-4 * want.abs() + 1e-6) & mask.expand_as(want)
                assert int(bad.sum()) == 0, kw
        else:
            assert abs(float(loss.detach()) - float(want_loss)) <= 2.0**-8 * abs(float(want_loss)) + 1e-6, kw
            # ColbertLoss at this shape runs its backward as two GEMMs on the matrix cores (msim_dense_t_bwd, round 6): dLoss/dscores enters
            # them rounded to bf16 -- exactly what the reference's own autograd multiplies by (its [B, C] score gradient IS a bf16 tensor)
            # -- so every one of a row's ~B resp. ~780 B terms carries a relative 2^-9: two "paths" of slack in grads_close's terms
            paths = 2 if kind == "infonce" else 1
            assert grads_close(p.grad, want_dp, p_real.expand_as(want_dp), paths=paths), kw
            assert grads_close(q.grad, want_dq, q_real.expand_as(want_dq), paths=paths), kw


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_dense_dd_form_equals_the_row_range_form_and_the_float64_scatter(amd, dtype):
    """msim_pairs_bwd with and without its workspace: short documents (32 rows) that collect long entry lists (all pairs of
    780-token 'queries': the trainer's symmetric direction) take the dense dD form when scratch is passed -- every document's
    pair list split over several workgroups, partial sums added in split order -- and the one-workgroup-per-row-range form
    without.  Both must equal the float64 scatter-add of g[p] * Q[b_p, i, :] into row argmax[p, i] of document c_p."""
    from colpali_amd import _lib, loss as L_

    g = torch.Generator().manual_seed(3)
    B, C, Lq, Ld = 9, 7, 780, 32
    q = torch.nn.functional.normalize(torch.randn(B, Lq, 128, generator=g), dim=-1).to(dtype).cuda()
    d = torch.nn.functional.normalize(torch.randn(C, Ld, 128, generator=g), dim=-1).to(dtype).cuda()
    d[2, 20:] = 0                                                          # a document with zero padding rows
    gp = torch.randn(B * C, generator=g).cuda()
    pairs, order = L_._all_pairs(B, C, q.device), L_._all_pairs_order(B, C, q.device)
    offsets = L_._dense_corpus(d).offsets
    _, argmax = L_.maxsim_pairs(q, d, offsets, pairs, want_scores=False)
    lib = _lib.lib()
    assert lib.msim_pairs_bwd_workspace_bytes(B, Lq, C, 128, Ld, B * C) > 0
    assert lib.msim_pairs_bwd_workspace_bytes(32, 32, 256, 128, 780, 64) == 0          # config 5, forward direction: no scratch
    outs = []
    for use_ws in (True, False):
        dq = torch.full((B, Lq, 128), float("nan"), device=q.device)
        dd = torch.full((C, Ld, 128), float("nan"), device=q.device)
        nbytes = lib.msim_pairs_bwd_workspace_bytes(B, Lq, C, 128, Ld, B * C)
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=q.device) if use_ws else None
        rc = lib.msim_pairs_bwd(_lib.dtype_code(dtype), _lib.ptr(q), B, Lq, _lib.ptr(d), _lib.ptr(offsets), C, 128, Ld, _lib.ptr(pairs),
                                _lib.ptr(order), _lib.ptr(gp), None, 0, _lib.ptr(argmax), B * C, 2, _lib.ptr(dq), _lib.ptr(dd), _lib.ptr(ws),
                                _lib.current_stream_handle(q.device))
        _lib.check(rc, "msim_pairs_bwd")
        outs.append((dq.cpu(), dd.cpu()))
    assert torch.equal(outs[0][0], outs[1][0])                            # dQ does not depend on the dD form
    want = torch.zeros(C, Ld, 128, dtype=torch.float64)
    am, pr, gg, qq = argmax.cpu().long(), pairs.cpu().long(), gp.cpu().double(), q.cpu().double()
    for p in range(B * C):
        b, c = int(pr[p, 0]), int(pr[p, 1])
        rows = am[p]
        ok = rows >= 0
        want[c].index_add_(0, rows[ok], gg[p] * qq[b][ok])
    scale = float(want.abs().max())
    for dq, dd in outs:
        assert torch.isfinite(dd).all()
        assert float((dd.double() - want).abs().max()) <= 2e-6 * scale


@pytest.mark.parametrize("case", ["dense_bf16", "dense_f32", "width320", "duplicates_overflow", "two_row_splits"])
def test_row_list_dd_form_against_the_float64_scatter(amd, case):
    """The row-list dD kernel (round 6: long documents, short entry lists -- the entries bucketed by winning row with a counting sort
    in LDS, every row adding its own entries in list order): dense all-pairs lists at widths 128 / 320, a list whose duplicates
    overflow the LDS lists at run time (the in-kernel direct walk), documents without pairs (rows of zeros), several row splits.
    fp32 output against the float64 scatter-add of g[p] * Q[b_p, i, :] into row argmax[p, i] of document c_p; run twice: bit-stable."""
    from colpali_amd import _lib, loss as L_

    g = torch.Generator().manual_seed(11)
    dtype, dim = torch.bfloat16, 128
    if case == "dense_f32":
        dtype = torch.float32
    if case == "width320":
        dim = 320
    B, C, Lq, Ld = (4, 3, 128, 200) if case == "duplicates_overflow" else (6, 3, 32, 1500) if case == "two_row_splits" else (12, 10, 32, 300)
    q = torch.nn.functional.normalize(torch.randn(B, Lq, dim, generator=g), dim=-1).to(dtype).cuda()
    d = torch.nn.functional.normalize(torch.randn(C, Ld, dim, generator=g), dim=-1).to(dtype).cuda()
    if case == "duplicates_overflow":
        lst = [(b, 0) for b in range(B) for _ in range(12)] + [(b, 1) for b in range(B)]      # document 2: no pairs at all
        lst.sort(key=lambda bc: bc[0])
        pairs = torch.tensor(lst, dtype=torch.int32).cuda()
        order = torch.sort(pairs[:, 1].to(torch.int64), stable=True).indices.to(torch.int32).contiguous()
    else:
        pairs, order = L_._all_pairs(B, C, q.device), L_._all_pairs_order(B, C, q.device)
    n_pairs = pairs.shape[0]
    gp = torch.randn(n_pairs, generator=g).cuda()
    offsets = L_._dense_corpus(d).offsets
    _, argmax = L_.maxsim_pairs(q, d, offsets, pairs, want_scores=False)
    lib = _lib.lib()
    assert lib.msim_pairs_bwd_workspace_bytes(B, Lq, C, dim, Ld, n_pairs) == 0          # long documents: no scratch, the row forms
    outs = []
    for _ in range(2):
        dq = torch.full((B, Lq, dim), float("nan"), device=q.device)
        dd = torch.full((C, Ld, dim), float("nan"), device=q.device)
        rc = lib.msim_pairs_bwd(_lib.dtype_code(dtype), _lib.ptr(q), B, Lq, _lib.ptr(d), _lib.ptr(offsets), C, dim, Ld, _lib.ptr(pairs),
                                _lib.ptr(order), _lib.ptr(gp), None, 0, _lib.ptr(argmax), n_pairs, 2, _lib.ptr(dq), _lib.ptr(dd), None,
                                _lib.current_stream_handle(q.device))
        _lib.check(rc, "msim_pairs_bwd")
        outs.append(dd.cpu())
    assert torch.equal(outs[0], outs[1])
    want = torch.zeros(C, Ld, dim, dtype=torch.float64)
    am, pr, gg, qq = argmax.cpu().long(), pairs.cpu().long(), gp.cpu().double(), q.cpu().double()
    for p in range(n_pairs):
        b, c = int(pr[p, 0]), int(pr[p, 1])
        want[c].index_add_(0, am[p], gg[p] * qq[b])
    assert torch.isfinite(outs[0]).all()
    assert float((outs[0].double() - want).abs().max()) <= 2e-6 * float(want.abs().max())
    if case == "duplicates_overflow":
        assert float(outs[0][2].abs().max()) == 0.0


@pytest.mark.parametrize("shape", [(6, 11, 32, 300), (5, 9, 780, 32), (3, 4, 45, 70)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gradients_in_the_embedding_dtype_with_the_upstream_scalar_folded_in(amd, shape, dtype):
    """msim_pairs_bwd(out_dtype = dtype, g_scale = a device scalar): dQ / dD leave the kernels in the embeddings' own 16-bit dtype
    with every pair's gradient multiplied by the scalar -- what `(coef * grad).to(fp32)` -> kernels -> `.to(dtype)` did in five
    launches.  Against the fp32 outputs of the same call without the scalar: out16 == round(fp32 * scale) up to the one product
    that is now formed before the sum (1 ulp of the 16-bit result), for the row-range dD form, the dense form (long queries, short
    documents) and documents that receive nothing (rows of zeros)."""
    from colpali_amd import _lib, loss as L_

    B, C, Lq, Ld = shape
    g = torch.Generator().manual_seed(11 + Lq)
    q = torch.nn.functional.normalize(torch.randn(B, Lq, 128, generator=g), dim=-1).to(dtype).cuda()
    d = torch.nn.functional.normalize(torch.randn(C, Ld, 128, generator=g), dim=-1).to(dtype).cuda()
    # a sparse pair list sorted by query: two documents per query, some documents never named
    docs = torch.stack([torch.arange(B) % (C - 2), (torch.arange(B) * 3 + 1) % (C - 2)], 1).sort(dim=1).values
    pairs = torch.stack([torch.arange(B).repeat_interleave(2), docs.reshape(-1)], 1).to(torch.int32).cuda().contiguous()
    order = torch.sort(pairs[:, 1].long(), stable=True).indices.to(torch.int32).contiguous()
    gp = torch.randn(2 * B, generator=g).cuda()
    offsets = L_._dense_corpus(d).offsets
    _, argmax = L_.maxsim_pairs(q, d, offsets, pairs, want_scores=False)
    lib = _lib.lib()
    code = _lib.dtype_code(dtype)
    nbytes = lib.msim_pairs_bwd_workspace_bytes(B, Lq, C, 128, Ld, 2 * B)
    assert (nbytes > 0) == (Ld <= 64 and Lq >= 256)
    ws = torch.empty((max(nbytes, 16),), dtype=torch.uint8, device=q.device)

    def run(out_dtype, scale):
        dq = torch.full((B, Lq, 128), float("nan"), dtype=out_dtype, device=q.device)
        dd = torch.full((C, Ld, 128), float("nan"), dtype=out_dtype, device=q.device)
        rc = lib.msim_pairs_bwd(code, _lib.ptr(q), B, Lq, _lib.ptr(d), _lib.ptr(offsets), C, 128, Ld, _lib.ptr(pairs), _lib.ptr(order),
                                _lib.ptr(gp), _lib.ptr(scale), _lib.dtype_code(scale.dtype) if scale is not None else 0, _lib.ptr(argmax),
                                2 * B, 2 if out_dtype == torch.float32 else code, _lib.ptr(dq), _lib.ptr(dd), _lib.ptr(ws if nbytes else None),
                                _lib.current_stream_handle(q.device))
        _lib.check(rc, "msim_pairs_bwd")
        return dq, dd

    dq32, dd32 = run(torch.float32, None)
    for scale in (torch.tensor(0.375, dtype=dtype, device=q.device), torch.tensor(-2.5, dtype=torch.float32, device=q.device)):
        dq16, dd16 = run(dtype, scale)
        for got, ref in ((dq16, dq32), (dd16, dd32)):
            want = ref * float(scale)
            assert got.dtype == dtype and torch.isfinite(got.float()).all()
            ulp = torch.exp2(torch.floor(torch.log2(want.abs().clamp_min(1e-30))) - (7 if dtype == torch.bfloat16 else 10))
            # + fp32 noise of the terms themselves (hits of opposite sign that nearly cancel leave a residue far below their size)
            assert bool(((got.float() - want).abs() <= ulp + 1e-6 * float(want.abs().max())).all())
        untouched = torch.ones(C, dtype=torch.bool)
        untouched[pairs[:, 1].long().cpu()] = False
        assert bool((dd16[untouched.cuda()] == 0).all())
    with pytest.raises(ValueError):                      # fp32 embeddings have no 16-bit output; a foreign out dtype is refused
        rc = lib.msim_pairs_bwd(code, _lib.ptr(q), B, Lq, _lib.ptr(d), _lib.ptr(offsets), C, 128, Ld, _lib.ptr(pairs), _lib.ptr(order),
                                _lib.ptr(gp), None, 0, _lib.ptr(argmax), 2 * B, 1 - code, _lib.ptr(dq32), _lib.ptr(dd32), None,
                                _lib.current_stream_handle(q.device))
        _lib.check(rc, "msim_pairs_bwd")


@pytest.mark.parametrize("n_pairs", [7, 64, 1500])
@pytest.mark.parametrize("Lq,Ld", [(32, 780), (100, 33), (780, 32), (300, 70), (200, 128)])
def test_pair_kernels_in_every_form_agree_with_the_float64_similarities(amd, n_pairs, Lq, Ld):
    """msim_pairs_argmax: one workgroup per pair (<= 1024 pairs), one wave per pair (more), and the transposed kernel (queries of more
    than 128 tokens against documents of at most 128 rows: the trainer's symmetric direction) -- scores within 1e-5 of float64, the
    routing a row that attains the maximum (to fp32 noise), -1 never (no clamp0 here); ragged documents through the packed layout."""
    from colpali_amd import _lib

    g = torch.Generator().manual_seed(Lq * 7 + Ld + n_pairs)
    n_q, n_d = 5, 9
    q = torch.nn.functional.normalize(torch.randn(n_q, Lq, 128, generator=g), dim=-1).to(torch.bfloat16)
    q[1, : Lq // 3] = 0                                                    # left padding rows
    lens = torch.randint(max(1, Ld // 2), Ld + 1, (n_d,), generator=g)
    lens[0] = Ld
    rows = [torch.nn.functional.normalize(torch.randn(int(n), 128, generator=g), dim=-1).to(torch.bfloat16) for n in lens]
    blob = torch.cat(rows).cuda()
    off = torch.zeros(n_d + 1, dtype=torch.int32)
    off[1:] = torch.cumsum(lens, 0)
    pairs = torch.stack([torch.sort(torch.randint(0, n_q, (n_pairs,), generator=g)).values, torch.randint(0, n_d, (n_pairs,), generator=g)], 1).to(torch.int32)
    lib = _lib.lib()
    qd, offd, pd = q.cuda(), off.cuda(), pairs.cuda().contiguous()
    scores = torch.full((n_pairs,), float("nan"), device="cuda")
    argmax = torch.full((n_pairs, Lq), -7, dtype=torch.int32, device="cuda")
    rc = lib.msim_pairs_argmax(0, _lib.ptr(qd), n_q, Lq, _lib.ptr(blob), _lib.ptr(offd), None, n_d, 128, Ld, _lib.ptr(pd), n_pairs,
                               _lib.ptr(scores), _lib.ptr(argmax), _lib.current_stream_handle(qd.device))
    _lib.check(rc, "msim_pairs_argmax")
    scores, argmax = scores.cpu(), argmax.cpu().long()
    q64 = q.double()
    checked = set()
    for p in range(n_pairs):
        b, c = int(pairs[p, 0]), int(pairs[p, 1])
        sim = q64[b] @ rows[c].double().t()                                 # [Lq, len]
        want = sim.max(dim=1).values
        assert abs(float(scores[p]) - float(want.sum())) <= 1e-5 * max(1.0, abs(float(want.sum())))
        if (b, c) in checked:
            continue
        checked.add((b, c))
        am = argmax[p]
        assert int(am.min()) >= 0 and int(am.max()) < int(lens[c])
        assert float((sim.gather(1, am[:, None])[:, 0] - want).abs().max()) <= 2e-6
        zero_rows = q[b].abs().sum(-1) == 0
        assert bool((am[zero_rows] == 0).all())                              # all-equal similarities: the first row wins


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n_q,Lq,n_d,Ld", [(32, 780, 256, 32), (3, 129, 5, 1), (7, 300, 70, 40), (2, 1000, 9, 128), (5, 200, 33, 70),
                                           (9, 131, 300, 16), (1, 4097, 3, 48)])
def test_transposed_forward_kernel_against_float64(amd, dtype, n_q, Lq, n_d, Ld):
    """msim_fwd_transposed (K1t: the long side streams, the short documents are resident -- the trainer's symmetric direction,
    contrastive_trainer.py:202-206) against a float64 einsum -> amax -> sum on the same 16-bit values: within 1e-5, for every unit
    count (1, 2, 3, 4, 8 units of 16 rows; lengths that are not multiples of 16: the register rows beyond the document are masked to
    -inf, so a document whose similarities are ALL negative must not see a 0), zero padding rows inside the boxes (they do take part,
    as in the reference), page tails that are not multiples of 32 / 128 rows, more documents than one block, and against msim_fwd on
    the same boxes (fp32 summation order apart)."""
    from colpali_amd import _lib

    g = torch.Generator().manual_seed(n_q * 1000 + Lq + Ld)
    q = torch.nn.functional.normalize(torch.randn(n_q, Lq, 128, generator=g), dim=-1)
    d = torch.nn.functional.normalize(torch.randn(n_d, Ld, 128, generator=g), dim=-1)
    q[0, : Lq // 4] = 0                                       # left padding rows of a page
    if n_d > 2 and Ld > 2:
        d[1, : Ld // 2] = 0                                   # left padding rows of a query-as-document
        d[2] = -q[min(1, n_q - 1), :Ld] if Lq >= Ld else d[2]  # every similarity with that page's own rows strongly negative
    q, d = q.to(dtype).cuda(), d.to(dtype).cuda()
    lib = _lib.lib()
    got = torch.full((n_q, n_d + 3), float("nan"), device="cuda")
    lens = torch.full((n_q,), -1, dtype=torch.int32, device="cuda")
    rc = lib.msim_fwd_transposed(_lib.dtype_code(dtype), _lib.ptr(q), n_q, Lq, _lib.ptr(d), n_d, Ld, 128, _lib.ptr(got), n_d + 3,
                                 _lib.ptr(lens), _lib.current_stream_handle(q.device))
    _lib.check(rc, "msim_fwd_transposed")
    assert torch.equal(lens.cpu(), (q[:, :, 0] != 0).sum(dim=1).to(torch.int32).cpu())      # late_interaction_losses.py:296, as a by-product
    assert torch.isnan(got[:, n_d:]).all()                    # nothing beyond the n_d columns is written
    want = torch.einsum("bnd,csd->bcns", q.double().cpu(), d.double().cpu()).amax(dim=3).sum(dim=2)
    err = (got[:, :n_d].double().cpu() - want).abs() / want.abs().clamp_min(1.0)
    assert float(err.max()) <= 1e-5, float(err.max())
    from colpali_amd import loss as L_

    other = amd.maxsim_scores(q, L_._dense_corpus(d))          # K1b on 128-token pieces (or the generic kernel): the same scores
    assert float(((other - got[:, :n_d]).abs() / want.abs().clamp_min(1.0).cuda()).max()) <= 1e-5
    # refused shapes
    assert lib.msim_fwd_transposed(_lib.dtype_code(dtype), _lib.ptr(q), n_q, Lq, _lib.ptr(d), n_d, 129, 128, _lib.ptr(got), n_d + 3, None, None) == -2
    assert lib.msim_fwd_transposed(2, _lib.ptr(q), n_q, Lq, _lib.ptr(d), n_d, Ld, 128, _lib.ptr(got), n_d + 3, None, None) == -2


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,C,Lq,Ld", [(32, 256, 32, 780), (5, 9, 32, 100), (7, 11, 40, 300), (3, 5, 96, 64), (2, 3, 128, 33), (9, 4, 7, 50)])
def test_all_pairs_kernel_equals_the_pair_list_kernel_on_the_all_pairs_list(amd, dtype, B, C, Lq, Ld):
    """msim_allpairs_argmax (a wave scores up to four queries against one document) against msim_pairs_argmax on the row-major
    all-pairs list: the same routing, the same scores to fp32 summation order (the per-token maxima are identical, the token sums run
    in the same order) -- every tile count (1..4 tiles per query: 4 / 2 / 1 / 1 queries per wave), query counts that do not fill a
    group, left-padded queries, ragged documents through the packed layout."""
    from colpali_amd import _lib, loss as L_

    g = torch.Generator().manual_seed(B * 100 + Lq)
    q = torch.nn.functional.normalize(torch.randn(B, Lq, 128, generator=g), dim=-1)
    q[0, : Lq // 3] = 0
    q = q.to(dtype).cuda()
    lens = torch.randint(max(1, Ld // 2), Ld + 1, (C,), generator=g)
    blob = torch.cat([torch.nn.functional.normalize(torch.randn(int(n), 128, generator=g), dim=-1) for n in lens]).to(dtype).cuda()
    off = torch.zeros(C + 1, dtype=torch.int32)
    off[1:] = torch.cumsum(lens, 0)
    off = off.cuda()
    lib = _lib.lib()
    code = _lib.dtype_code(dtype)
    pairs = L_._all_pairs(B, C, q.device)
    s_ref = torch.empty((B * C,), dtype=torch.float32, device="cuda")
    a_ref = torch.empty((B * C, Lq), dtype=torch.int32, device="cuda")
    _lib.check(lib.msim_pairs_argmax(code, _lib.ptr(q), B, Lq, _lib.ptr(blob), _lib.ptr(off), None, C, 128, 0, _lib.ptr(pairs), B * C,
                                     _lib.ptr(s_ref), _lib.ptr(a_ref), _lib.current_stream_handle(q.device)), "msim_pairs_argmax")
    s_got = torch.full((B, C + 2), float("nan"), device="cuda")
    a_got = torch.full((B * C, Lq), -7, dtype=torch.int32, device="cuda")
    _lib.check(lib.msim_allpairs_argmax(code, _lib.ptr(q), B, Lq, _lib.ptr(blob), _lib.ptr(off), None, C, 128, _lib.ptr(s_got), C + 2,
                                        _lib.ptr(a_got), _lib.current_stream_handle(q.device)), "msim_allpairs_argmax")
    assert torch.equal(a_got, a_ref)
    assert torch.isnan(s_got[:, C:]).all()
    assert float((s_got[:, :C].reshape(-1) - s_ref).abs().max()) <= 1e-5 * float(s_ref.abs().max().clamp_min(1.0))
    assert lib.msim_allpairs_argmax(code, _lib.ptr(q), B, 129, _lib.ptr(blob), _lib.ptr(off), None, C, 128, _lib.ptr(s_got), C + 2,
                                    _lib.ptr(a_got), None) == -2
