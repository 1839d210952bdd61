"""GPU: the dense hard-max backward of the transposed shape on the matrix cores (round 6; maxsim_dense_t.hip, include/maxsim.h:
msim_fwd_transposed_route / msim_dense_t_bwd) -- ColbertLoss / ColbertSigmoidLoss in the trainer's symmetric direction
(late_interaction_losses.py:140-164, :440-465; trainer/contrastive_trainer.py:202-206).

The forward's scores must be the bits msim_fwd_transposed returns; its routing bytes must name a maximal row (the first one, checked
wherever float64 can tell the rows apart); the backward must be the float64 scatter of that routing with G rounded once to the
embeddings' dtype (what the kernels multiply by), to fp32-accumulation accuracy -- and within one more rounding of the unrounded G."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(32, 780, 256, 32), (3, 129, 5, 1), (7, 300, 70, 40), (5, 200, 33, 64), (9, 131, 300, 16), (2, 1000, 9, 17), (4, 260, 8, 33)]


@pytest.fixture(scope="module")
def amd():
    import colpali_amd

    colpali_amd._lib.lib()
    return colpali_amd


def _boxes(n_q, Lq, n_d, Ld, dtype, zero_rows=True):
    g = torch.Generator().manual_seed(n_q * 1000 + Lq + 7 * Ld)
    q = torch.nn.functional.normalize(torch.randn(n_q, Lq, 128, generator=g), dim=-1)
    d = torch.nn.functional.normalize(torch.randn(n_d, Ld, 128, generator=g), dim=-1)
    if zero_rows:
        q[0, : Lq // 4] = 0                                   # left padding rows of a page
        if n_d > 2 and Ld > 2:
            d[1, : Ld // 2] = 0                               # left padding rows of a query-as-document
    return q.to(dtype).cuda(), d.to(dtype).cuda()


def _route_forward(q, d):
    from colpali_amd import _lib

    lib = _lib.lib()
    n_q, Lq, _ = q.shape
    n_d, Ld, _ = d.shape
    code = _lib.dtype_code(q.dtype)
    assert lib.msim_dense_t_supported(code, n_q, Lq, n_d, Ld, 128) == 1
    lq_pad = (Lq + 63) // 64 * 64
    assert lib.msim_dense_t_route_bytes(n_q, Lq, n_d) == n_q * n_d * lq_pad
    scores = torch.full((n_q, n_d), float("nan"), device="cuda")
    lens = torch.full((n_q,), -1, dtype=torch.int32, device="cuda")
    route = torch.full((n_q, n_d, lq_pad), 77, dtype=torch.uint8, device="cuda")
    rc = lib.msim_fwd_transposed_route(code, _lib.ptr(q), n_q, Lq, _lib.ptr(d), n_d, Ld, 128, _lib.ptr(scores), n_d, _lib.ptr(lens),
                                       _lib.ptr(route), _lib.current_stream_handle(q.device))
    _lib.check(rc, "msim_fwd_transposed_route")
    return scores, lens, route


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n_q,Lq,n_d,Ld", SHAPES)
def test_route_forward_scores_are_k1ts_bits_and_the_bytes_name_the_first_maximal_row(amd, dtype, n_q, Lq, n_d, Ld):
    from colpali_amd import _lib

    lib = _lib.lib()
    q, d = _boxes(n_q, Lq, n_d, Ld, dtype)
    scores, lens, route = _route_forward(q, d)
    plain = torch.empty((n_q, n_d), device="cuda")
    rc = lib.msim_fwd_transposed(_lib.dtype_code(dtype), _lib.ptr(q), n_q, Lq, _lib.ptr(d), n_d, Ld, 128, _lib.ptr(plain), n_d, None,
                                 _lib.current_stream_handle(q.device))
    _lib.check(rc, "msim_fwd_transposed")
    assert torch.equal(scores, plain)                          # the routing changes no bit of a score
    assert torch.equal(lens.cpu(), (q[:, :, 0] != 0).sum(dim=1).to(torch.int32).cpu())
    lq_pad = route.shape[2]                                    # bytes of rows >= Lq are unspecified: they never reach a result
    sims = torch.einsum("bnd,csd->bcns", q.double(), d.double())              # [n_q, n_d, Lq, Ld] float64 on the same 16-bit values
    r = route[:, :, :Lq].long()
    assert int(r.max()) < Ld
    top = sims.amax(dim=3)
    chosen = sims.gather(3, r.unsqueeze(-1)).squeeze(-1)
    assert float((top - chosen).abs().max()) <= 2e-6          # a maximal row (fp32 products of 16-bit values: exact up to the sum's rounding)
    # the FIRST maximal row, wherever float64 separates the best two rows by more than the kernels' rounding -- and at exact ties
    # (all-zero rows on either side: every similarity is +0.0) the lowest row index
    first = sims.argmax(dim=3)
    top2 = sims.topk(min(2, Ld), dim=3).values
    clear = (top2[..., 0] - top2[..., -1] > 1e-5) if Ld > 1 else torch.ones_like(first, dtype=torch.bool)
    assert torch.equal(r[clear], first[clear])
    zero_page_rows = (q.double().abs().sum(-1) == 0)                          # [n_q, Lq]: every similarity of such a row ties at 0
    if bool(zero_page_rows.any()):
        assert int(r[zero_page_rows.unsqueeze(1).expand_as(r)].max()) == 0
    # refused shapes: documents above 64 rows keep the list kernels; fp32 is not served here
    assert lib.msim_dense_t_supported(_lib.dtype_code(dtype), n_q, Lq, n_d, 65, 128) == 0
    assert lib.msim_dense_t_supported(2, n_q, Lq, n_d, Ld, 128) == 0
    assert lib.msim_dense_t_supported(_lib.dtype_code(dtype), n_q, Lq, n_d, Ld, 96) == 0
    assert lib.msim_fwd_transposed_route(_lib.dtype_code(dtype), _lib.ptr(q), n_q, Lq, _lib.ptr(d), n_d, 65, 128, _lib.ptr(plain), n_d, None,
                                         _lib.ptr(route), None) == -2
    assert lq_pad % 64 == 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n_q,Lq,n_d,Ld", SHAPES)
@pytest.mark.parametrize("scaled", [False, True])
def test_dense_backward_is_the_float64_scatter_of_the_routing(amd, dtype, n_q, Lq, n_d, Ld, scaled):
    from colpali_amd import loss as L_

    q, d = _boxes(n_q, Lq, n_d, Ld, dtype)
    _, _, route = _route_forward(q, d)
    g = torch.Generator().manual_seed(5 + n_q + Ld)
    G = (torch.randn(n_q, n_d + 3, generator=g) * 0.05).cuda()[:, :n_d]      # a strided view: ldg = n_d + 3
    scale = torch.tensor(0.37, dtype=dtype, device="cuda") if scaled else None
    dq, dd = L_._dense_t_backward(q, d, G, route.view(-1), g_scale=scale)
    assert dq.dtype == dtype and dd.dtype == dtype and dq.shape == q.shape and dd.shape == d.shape
    up = float(scale) if scaled else 1.0
    r = route[:, :, :Lq].long()                                                # [n_q, n_d, Lq]
    # fp16: weights below 2^-14 are subnormal in the embeddings' dtype and the matrix cores flush subnormal fp16 operands (G ~ N(0, 0.05)
    # puts ~0.3 % of them there, each worth at most 6e-5 x |row|): a floor of 2^-11 of the largest gradient instead of 1e-6
    f16 = dtype == torch.float16
    for W, rel, floor in (((G * up).to(dtype).double(), 2.0**-10 if f16 else 2.0**-7, 2.0**-11 if f16 else 1e-6),   # what the kernels multiply by: only the output's rounding (one ulp) remains
                          ((G.double() * up), 2.0**-6, 2.0**-8)):                                                        # the unrounded G: one more rounding per term
        rows = d.double()[torch.arange(n_d, device="cuda")[None, :, None], r]   # [n_q, n_d, Lq, 128]: the winning rows
        want_dq = (W[:, :, None, None] * rows).sum(dim=1)                       # [n_q, Lq, 128]
        want_dd = torch.zeros((n_d, Ld, 128), dtype=torch.float64, device="cuda")
        contrib = W[:, :, None, None] * q.double()[:, None, :, :]               # [n_q, n_d, Lq, 128]
        idx = (torch.arange(n_d, device="cuda")[None, :, None] * Ld + r).reshape(-1)
        want_dd.view(-1, 128).index_add_(0, idx, contrib.reshape(-1, 128))
        for got, want in ((dq, want_dq), (dd, want_dd)):
            tol = want.abs() * rel + floor * float(want.abs().max()) + 1e-9
            assert int(((got.double() - want).abs() > tol).sum()) == 0, (float((got.double() - want).abs().max()), float(want.abs().max()))
        del rows, contrib


@pytest.mark.parametrize("cls", ["ColbertLoss", "ColbertSigmoidLoss"])
def test_dense_losses_take_the_matrix_core_backward_and_agree_with_the_list_kernels(amd, cls, monkeypatch):
    """The same loss step through msim_dense_t_bwd (default for this shape) and through the pair-list kernels (the path every other
    shape keeps, forced here): loss bit-identical (the scores are), gradients within the bf16 rounding of G."""
    from colpali_amd import loss as L_

    B = 16
    g = torch.Generator().manual_seed(3)
    pages = torch.nn.functional.normalize(torch.randn(B, 300, 128, generator=g), dim=-1).to(torch.bfloat16)
    queries = torch.nn.functional.normalize(torch.randn(B, 24, 128, generator=g), dim=-1).to(torch.bfloat16)
    queries[:, :5] = 0
    pages[torch.arange(B), :19] = queries[:, 5:]                 # positives

    def run():
        p, q = pages.cuda().requires_grad_(True), queries.cuda().requires_grad_(True)
        loss = getattr(amd, cls)()(query_embeddings=p, doc_embeddings=q, offset=0)
        loss.backward()
        return loss.detach().float().cpu(), p.grad.float().cpu(), q.grad.float().cpu()

    calls = []
    real = L_._dense_t_backward
    monkeypatch.setattr(L_, "_dense_t_backward", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    new = run()
    assert calls, "the dense matrix-core backward was not taken"
    monkeypatch.setattr(L_, "_dense_t_ok", lambda qc, dc: False)
    old = run()
    assert float((new[0] - old[0]).abs()) <= 2.0**-8 * float(old[0].abs()) + 1e-7
    for a, b in zip(new[1:], old[1:]):
        assert float((a - b).abs().max()) <= 2.0**-6 * float(b.abs().max())


def test_dense_backward_abi_rejects_what_it_does_not_take(amd):
    """include/maxsim.h: msim_dense_t_bwd / msim_fwd_transposed_route return error codes (never touch memory) for null pointers,
    unsupported dtypes / widths / document lengths, a short leading dimension and a misaligned workspace; empty batches are a no-op."""
    from colpali_amd import _lib

    lib = _lib.lib()
    q, d = _boxes(3, 200, 5, 24, torch.bfloat16, zero_rows=False)
    _, _, route = _route_forward(q, d)
    G = torch.zeros((3, 5), device="cuda")
    dq, dd = torch.empty_like(q), torch.empty_like(d)
    nbytes = lib.msim_dense_t_bwd_workspace_bytes(3, 200, 5, 24, 128)
    assert nbytes > 0 and lib.msim_dense_t_bwd_workspace_bytes(0, 200, 5, 24, 128) == 0
    ws = torch.empty((nbytes + 16,), dtype=torch.uint8, device="cuda")
    st = _lib.current_stream_handle(q.device)

    def call(dtype=0, Q=q, D=d, Ld=24, dim=128, g=G, ldg=5, r=route, out_q=dq, out_d=dd, w=ws, n_q=3, n_d=5):
        return lib.msim_dense_t_bwd(dtype, _lib.ptr(Q), n_q, 200, _lib.ptr(D), n_d, Ld, dim, _lib.ptr(g), ldg, None, 0, _lib.ptr(r),
                                    _lib.ptr(out_q), _lib.ptr(out_d), _lib.ptr(w), st)

    assert call() == 0
    assert call(n_q=0) == 0 and call(n_d=0) == 0                       # nothing to do
    assert call(dtype=2) == -2 and call(dim=96) == -2 and call(Ld=65) == -2          # MSIM_EUNSUPPORTED
    assert call(g=None) == -1 and call(r=None) == -1 and call(w=None) == -1          # MSIM_EINVAL
    assert call(ldg=4) == -1
    assert call(w=ws[1:]) == -1                                                       # workspace not 16-byte aligned
    assert b"" != lib.msim_last_error()
    torch.cuda.synchronize()
