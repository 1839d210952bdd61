"""GPU: ABI-level behaviour -- asynchronous on the caller's stream, hipGraph-capturable, re-entrant."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import colpali_amd

    colpali_amd._lib.lib()
    return colpali_amd


def _case(n_q, n_d, ld=300, seed=0):
    g = torch.Generator().manual_seed(seed)
    q = torch.nn.functional.normalize(torch.randn(n_q, 32, 128, generator=g), dim=-1).to(torch.bfloat16)
    docs = [torch.nn.functional.normalize(torch.randn(ld, 128, generator=g), dim=-1).to(torch.bfloat16) for _ in range(n_d)]
    return q, docs


@pytest.mark.parametrize("n_q", [2, 40])
def test_forward_and_topk_are_graph_capturable(amd, n_q):
    dev = torch.device("cuda:0")
    q, docs = _case(n_q, 500)
    q = q.to(dev)
    corpus = amd.pack_passages(docs, dev, batch_size=None)
    eager = amd.maxsim_scores(q, corpus)
    es, ei = amd.topk(eager, 10)
    out = torch.zeros_like(eager)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        amd.maxsim_scores(q, corpus, out=out)          # warm-up on the side stream (one-time attribute calls)
    torch.cuda.current_stream().wait_stream(s)
    out.zero_()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):                       # no allocation / sync inside the library call
        amd.maxsim_scores(q, corpus, out=out)
    assert torch.count_nonzero(out) == 0                # capture does not execute
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
    gs, gi = amd.topk(out, 10)
    assert torch.equal(gi, ei) and torch.equal(gs, es)


def test_calls_are_ordered_on_a_non_default_stream(amd):
    dev = torch.device("cuda:0")
    q, docs = _case(3, 2000, ld=128, seed=1)
    corpus = amd.pack_passages(docs, dev, batch_size=None)
    want = amd.maxsim_scores(q.to(dev), corpus)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        qd = q.to(dev, non_blocking=True)
        got = amd.maxsim_scores(qd, corpus)             # must run after the copy on the same stream
        top = amd.topk(got, 5)
    s.synchronize()
    assert torch.equal(got, want)
    assert torch.equal(top[1], amd.topk(want, 5)[1])


def test_deterministic_across_repeated_launches(amd):
    dev = torch.device("cuda:0")
    q, docs = _case(40, 800, ld=257, seed=2)
    corpus = amd.pack_passages(docs, dev)
    a = amd.maxsim_scores(q.to(dev), corpus).clone()
    for _ in range(3):
        assert torch.equal(amd.maxsim_scores(q.to(dev), corpus), a)


def test_every_other_entry_point_is_graph_capturable(amd):
    """Smooth-max forward, similarity matrix, embedding head and the generic (fp32) scorer under hipGraph capture: nothing
    in the library allocates or synchronises."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    Q = torch.nn.functional.normalize(torch.randn(6, 20, 128, generator=g), dim=-1).to(torch.bfloat16).to(dev)
    D = torch.nn.functional.normalize(torch.randn(9, 50, 128, generator=g), dim=-1).to(torch.bfloat16).to(dev)
    hidden = torch.randn(2, 40, 256, generator=g).to(torch.bfloat16).to(dev)
    W = (torch.randn(128, 256, generator=g) / 16).to(torch.bfloat16).to(dev)
    b = torch.zeros(128, dtype=torch.bfloat16, device=dev)
    mask = torch.ones(2, 40, dtype=torch.long, device=dev)
    Qf, Df = Q.float(), D.float()
    corpus32 = amd.pack_passages(list(Df), dev, batch_size=None)

    def run():
        return (amd.loss.maxsim_smooth(Q, D, 0.1), amd.similarity_matrix(Q[0], D[0]),
                amd.embedding_head(hidden, W, b, mask), amd.maxsim_scores(Qf, corpus32))

    eager = [t.clone() for t in run()]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = run()
    for t in captured:
        t.zero_()
    graph.replay()
    torch.cuda.synchronize()
    for got, want in zip(captured, eager):
        assert torch.equal(got, want)


def test_probe_stream_runs_and_validates_arguments(amd):
    """msim_probe_stream (measurement aid, tools/probe/libmaxsim_probe.so -- not the product library): every variant launches on a
    conforming matrix; bad shapes are refused."""
    from tools import probe

    L = probe.lib()
    assert L is not None, "tools/probe/libmaxsim_probe.so is not built (__graft_entry__.build())"
    dev = torch.device("cuda:0")
    sink = torch.zeros(4, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    x128 = torch.randn((4096, 128), device=dev).to(torch.bfloat16)
    x2k = torch.randn((1024, 2048), device=dev).to(torch.bfloat16)
    assert L.msim_probe_stream(0, x128.data_ptr(), 4096, 128, sink.data_ptr(), st) == 0
    assert L.msim_probe_stream(1, x2k.data_ptr(), 1024, 2048, sink.data_ptr(), st) == 0
    assert L.msim_probe_stream(2, x2k.data_ptr(), 1024, 2048, sink.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert float(sink.abs().sum()) == 0.0
    assert L.msim_probe_stream(0, x128.data_ptr(), 4000, 128, sink.data_ptr(), st) != 0      # rows % 256
    assert L.msim_probe_stream(2, x128.data_ptr(), 4096, 128, sink.data_ptr(), st) != 0      # row shorter than a piece
    assert L.msim_probe_stream(7, x128.data_ptr(), 4096, 128, sink.data_ptr(), st) != 0
    assert L.msim_probe_stream(0, None, 4096, 128, sink.data_ptr(), st) != 0


@pytest.mark.parametrize("cls,kw", [("ColbertPairwiseCELoss", dict(normalize_scores=False)),
                                    ("ColbertPairwiseCELoss", dict(pos_aware_negative_filtering=True)),
                                    ("ColbertLoss", dict()),
                                    ("ColbertLoss", dict(use_smooth_max=True)),
                                    ("ColbertPairwiseCELoss", dict(use_smooth_max=True, normalize_scores=False))])
def test_whole_loss_step_forward_and_backward_captures_in_one_hipgraph(amd, cls, kw):
    """A training step of the loss -- MaxSim forward, the [B, C] epilogue, gradient routing and both backward kernels -- has no
    host synchronisation and no data-dependent launch shape: it is captured ONCE and replayed on new embeddings, and the
    replayed loss and gradients equal an eager run on the same numbers bit for bit."""
    dev = torch.device("cuda:0")
    B, C, Lq, Ld, offset = 8, 24, 32, 96, 8

    def batch(seed):
        g = torch.Generator().manual_seed(seed)
        Q = torch.nn.functional.normalize(torch.randn(B, Lq, 128, generator=g), dim=-1)
        D = torch.nn.functional.normalize(torch.randn(C, Ld, 128, generator=g), dim=-1)
        Q[1, :5] = 0
        for b in range(B):
            D[offset + b, :Lq] = torch.nn.functional.normalize(Q[b] + 0.5 * torch.randn(Lq, 128, generator=g), dim=-1)
        return Q.to(torch.bfloat16).to(dev), D.to(torch.bfloat16).to(dev)

    loss_fn = getattr(amd, cls)(**kw)
    sq, sd = (t.clone().requires_grad_(True) for t in batch(0))

    def step():
        sq.grad = sd.grad = None
        loss = loss_fn(sq, sd, offset=offset)
        loss.backward()
        return loss

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):                                   # warm-up: caches, one-time attribute calls
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    sq.grad = sd.grad = None
    with torch.cuda.graph(graph):
        loss_static = loss_fn(sq, sd, offset=offset)
        loss_static.backward()
    gq_static, gd_static = sq.grad, sd.grad
    for seed in (1, 2):
        nq, nd = batch(seed)
        with torch.no_grad():
            sq.copy_(nq)
            sd.copy_(nd)
        graph.replay()
        torch.cuda.synchronize()
        eq, ed = nq.clone().requires_grad_(True), nd.clone().requires_grad_(True)
        eager = loss_fn(eq, ed, offset=offset)
        eager.backward()
        assert torch.equal(loss_static, eager.detach())
        assert torch.equal(gq_static, eq.grad) and torch.equal(gd_static, ed.grad)


def test_loss_step_does_not_synchronise_the_host(amd):
    """No call on the path of a training step may block the host: with a long-running kernel queued in front, building the whole
    step (forward + backward) must return while that kernel is still running."""
    import time

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    Q = torch.nn.functional.normalize(torch.randn(16, 32, 128, generator=g), dim=-1).to(torch.bfloat16).to(dev)
    D = torch.nn.functional.normalize(torch.randn(64, 200, 128, generator=g), dim=-1).to(torch.bfloat16).to(dev)
    big = torch.randn(8192, 8192, device=dev)
    for cls, kw in (("ColbertPairwiseCELoss", {}), ("ColbertLoss", {}), ("ColbertLoss", dict(pos_aware_negative_filtering=True)),
                    ("ColbertSigmoidLoss", {}), ("ColbertPairwiseCELoss", dict(use_smooth_max=True))):
        fn = getattr(amd, cls)(**kw)
        if cls == "ColbertSigmoidLoss":
            d_in = D[:16].clone().requires_grad_(True)
        else:
            d_in = D.clone().requires_grad_(True)
        q_in = Q.clone().requires_grad_(True)
        fn(q_in, d_in).backward()                               # warm-up
        torch.cuda.synchronize()
        done = torch.cuda.Event()
        for _ in range(30):                                      # ~100+ ms of queued GPU work
            big = big @ big * 1e-4
        done.record()
        t0 = time.perf_counter()
        loss = fn(q_in, d_in)
        loss.backward()
        host_s = time.perf_counter() - t0
        still_running = not done.query()
        torch.cuda.synchronize()
        assert still_running, (cls, kw, f"the step took {host_s * 1e3:.1f} ms on the host and outlived the queued work")


def test_loss_epilogue_matches_the_torch_expression_on_adversarial_scores(amd):
    """msim_loss_epilogue against the reference's own torch lines (late_interaction_losses.py:300-313, :164) evaluated with
    autograd in fp32 on the same score matrix: ties between equal scores, a positive that is not the best document, filtering
    that hits the selected negative, padded (zero-first-component) query tokens."""
    import torch.nn.functional as F

    L = amd._lib.lib()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    B, C, Lq, offset = 7, 19, 12, 5
    Q = torch.randn(B, Lq, 128, generator=g)
    Q[2, :4, 0] = 0
    Q[5, 3, 0] = -0.0
    Q = Q.to(torch.bfloat16).to(dev)
    base = (torch.rand(B, C, generator=g) * 8 + 1)
    base[0, offset + 0] = 20.0                       # positive is the best
    base[1, 3] = base[1, offset + 1] = 15.0          # exact tie between the positive and another document
    base[2, 0] = 30.0                                # positive is not the best
    base[3, offset + 3] = 9.0; base[3, 1] = 8.9      # the hardest negative is above the filter threshold
    base[4, 2] = base[4, 7] = 25.0                   # two tied best negatives
    for mode, T, norm, filt in ((0, 1.0, False, False), (0, 0.5, True, True), (0, 1.0, True, False), (1, 0.02, True, False),
                                (1, 0.5, False, True), (0, 0.02, False, True)):
        scores = base.clone().to(dev).requires_grad_(True)
        lengths = (Q[:, :, 0] != 0).sum(dim=1)
        s = scores / lengths.unsqueeze(1) if norm else scores * 1.0
        idx = torch.arange(B, device=dev)
        pos_idx = idx + offset
        if filt:
            lim = 0.95 * s[idx, pos_idx].unsqueeze(1)
            m = s > lim
            m[idx, pos_idx] = False
            s = s * torch.where(m, 0.5, 1.0)
        if mode == 0:
            pos = s.diagonal(offset=offset)
            top2 = s.topk(2, dim=1).values
            neg = torch.where(top2[:, 0] == pos, top2[:, 1], top2[:, 0])
            want = F.softplus((neg - pos) / T).mean()
        else:
            want = F.cross_entropy(s / T, pos_idx)
        want.backward()
        raw = base.clone().to(dev).contiguous()
        G = torch.zeros((B, C), dtype=torch.float32, device=dev)
        pairs = torch.empty((2 * B, 2), dtype=torch.int32, device=dev)
        coef = torch.empty((2 * B,), dtype=torch.float32, device=dev)
        order = torch.empty((2 * B,), dtype=torch.int32, device=dev)
        assert L.msim_loss_epilogue_workspace_bytes(B, C) == 0      # a small batch: one workgroup, no scratch (NULL is legal)
        out = torch.empty((3,), dtype=torch.float32, device=dev)
        loss16 = torch.full((), float("nan"), dtype=torch.bfloat16, device=dev)
        lens32 = lengths.to(torch.int32).contiguous()
        for rep in range(2):                          # the token counts taken from Q, then handed over ready-made: the same results
            rc = L.msim_loss_epilogue(mode, raw.data_ptr(), C, B, C, Q.data_ptr(), 0, Lq, 128, offset, T, int(norm), int(filt), 0.95, 0.5,
                                      G.data_ptr(), pairs.data_ptr(), coef.data_ptr(), order.data_ptr(), None, out.data_ptr(),
                                      loss16.data_ptr(), lens32.data_ptr() if rep else None, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, L.msim_last_error()
            if rep == 0:
                torch.cuda.synchronize()
                first = (float(out[0]), coef.clone(), pairs.clone(), G.clone())
        torch.cuda.synchronize()
        assert float(out[0]) == first[0] and torch.equal(coef, first[1]) and torch.equal(pairs, first[2]) and torch.equal(G, first[3])
        torch.cuda.synchronize()
        assert abs(float(out[0]) - float(want.detach())) <= 1e-5 * abs(float(want.detach())) + 1e-6, (mode, T, norm, filt)
        assert float(loss16) == float(out[0].to(torch.bfloat16))          # the loss in the embeddings' dtype: one rounding
        if mode == 0:
            got = torch.zeros((B, C), dtype=torch.float32, device=dev)
            got.index_put_((pairs[:, 0].long(), pairs[:, 1].long()), coef, accumulate=True)
            assert pairs[:, 0].tolist() == [b for b in range(B) for _ in range(2)]
            assert all(pairs[2 * b, 1] <= pairs[2 * b + 1, 1] for b in range(B))
            docs = pairs[:, 1].long()
            perm = order.long()
            assert sorted(perm.tolist()) == list(range(2 * B))
            assert torch.equal(docs[perm], docs[torch.sort(docs, stable=True).indices])
            assert torch.equal(perm, torch.sort(docs, stable=True).indices)
        else:
            got = G
        # tied scores: torch splits / picks differently, the sum over the tied entries is what is defined
        ref = scores.grad
        scale = float(ref.abs().max())
        untied = torch.ones(B, dtype=torch.bool)
        if mode == 0:
            untied[1] = untied[4] = False
            assert torch.allclose(got.sum(dim=1), ref.sum(dim=1), rtol=1e-4, atol=1e-6 * scale), (mode, T, norm, filt)
        # InfoNCE at T = 0.02: logits reach ~80, so `logit - lse` carries an absolute error of a few ulp(80) = 1e-5 in any fp32
        # evaluation (torch's log_softmax as much as the kernel's): probabilities agree to that, not to 1e-7
        atol = (3e-5 if mode == 1 else 1e-6) * scale
        assert torch.allclose(got[untied], ref[untied], rtol=1e-4, atol=atol), (mode, T, norm, filt)
        want_lo = (raw / lengths.unsqueeze(1)).min() if norm else raw.min()
        assert abs(float(out[1]) - float(want_lo)) < 1e-5
    # argument checks
    assert L.msim_loss_epilogue(0, raw.data_ptr(), C, B, C, Q.data_ptr(), 0, Lq, 128, C - B + 1, 1.0, 0, 0, 0.95, 0.5, None, pairs.data_ptr(),
                                coef.data_ptr(), order.data_ptr(), None, out.data_ptr(), None, None, None) == -1
    assert L.msim_loss_epilogue(0, raw.data_ptr(), 1, 1, 1, Q.data_ptr(), 0, Lq, 128, 0, 1.0, 0, 0, 0.95, 0.5, None, pairs.data_ptr(),
                                coef.data_ptr(), order.data_ptr(), None, out.data_ptr(), None, None, None) == -1


@pytest.mark.parametrize("mode", [0, 1])
def test_loss_epilogue_large_batch_form_equals_the_small_batch_form_row_by_row(amd, mode):
    """B * C above 262 144 scores: one workgroup per row + the ticket (zero-filled scratch, reusable) instead of the one-workgroup
    form.  Both run the same per-row code: the pair list / G of the first rows of a large batch must equal what the small form
    emits for those rows scored alone with the same 1 / B weighting, and the loss must equal the torch expression."""
    import torch.nn.functional as F

    L = amd._lib.lib()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(4)
    B, C, Lq, offset, T = 520, 610, 8, 37, 0.5
    Q = torch.randn(B, Lq, 128, generator=g).to(torch.bfloat16).to(dev)
    raw = (torch.rand(B, C, generator=g) * 8 + 1).to(dev).contiguous()
    need = L.msim_loss_epilogue_workspace_bytes(B, C)
    assert need > 0
    G = torch.zeros((B, C), dtype=torch.float32, device=dev)
    pairs = torch.empty((2 * B, 2), dtype=torch.int32, device=dev)
    coef = torch.empty((2 * B,), dtype=torch.float32, device=dev)
    order = torch.empty((2 * B,), dtype=torch.int32, device=dev)
    out = torch.empty((3,), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    args = (mode, raw.data_ptr(), C, B, C, Q.data_ptr(), 0, Lq, 128, offset, T, 1, 1, 0.95, 0.5, G.data_ptr(), pairs.data_ptr(),
            coef.data_ptr(), order.data_ptr())
    assert L.msim_loss_epilogue(*args, None, out.data_ptr(), None, None, st) == -1           # this size needs its scratch
    ws = torch.zeros((need,), dtype=torch.uint8, device=dev)
    for _ in range(2):                                                               # twice on the same scratch: the ticket resets itself
        assert L.msim_loss_epilogue(*args, ws.data_ptr(), out.data_ptr(), None, None, st) == 0, L.msim_last_error()
    torch.cuda.synchronize()
    s = raw / (Q[:, :, 0] != 0).sum(dim=1).unsqueeze(1)
    idx = torch.arange(B, device=dev)
    lim = 0.95 * s[idx, idx + offset].unsqueeze(1)
    m = s > lim
    m[idx, idx + offset] = False
    s = s * torch.where(m, 0.5, 1.0)
    if mode == 0:
        pos = s.diagonal(offset=offset)
        top2 = s.topk(2, dim=1).values
        want = F.softplus((torch.where(top2[:, 0] == pos, top2[:, 1], top2[:, 0]) - pos) / T).mean()
        docs = pairs[:, 1].long()
        assert torch.equal(order.long(), torch.sort(docs, stable=True).indices)
    else:
        want = F.cross_entropy(s / T, idx + offset)
    assert abs(float(out[0]) - float(want)) <= 1e-5 * abs(float(want)) + 1e-6


def test_probe_mfma_runs_and_validates_arguments(amd):
    """msim_probe_mfma (measurement aid, tools/probe/libmaxsim_probe.so): every variant launches and leaves the sink untouched; bad
    arguments are refused."""
    from tools import probe

    L = probe.lib()
    assert L is not None, "tools/probe/libmaxsim_probe.so is not built (__graft_entry__.build())"
    dev = torch.device("cuda:0")
    rows, rows_small = 256 * 16 * 3 * 32, 256 * 8 * 5 * 32
    x = torch.nn.functional.normalize(torch.randn((rows, 128), device=dev), dim=-1).to(torch.bfloat16)
    sink = torch.zeros(4, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for variant in range(29):               # 12..24: K1b's exact slab body under the round-3 register plans; 25..28: the ridge's ten-tile body (round 6)
        assert L.msim_probe_mfma(variant, x.data_ptr(), rows, 50, sink.data_ptr(), st) == 0, L.msim_probe_last_error()
    torch.cuda.synchronize()
    assert float(sink.abs().sum()) == 0.0
    assert L.msim_probe_mfma(29, x.data_ptr(), rows, 50, sink.data_ptr(), st) != 0
    assert L.msim_probe_mfma(25, x.data_ptr(), 256 * 4 * 11 * 32 - 1, 50, sink.data_ptr(), st) != 0   # ten tiles + the operand rows per wave
    assert L.msim_probe_mfma(7, x.data_ptr(), rows_small, 50, sink.data_ptr(), st) == 0      # variants 0..7 take the smaller matrix
    assert L.msim_probe_mfma(8, x.data_ptr(), rows_small, 50, sink.data_ptr(), st) != 0      # the 12 / 16-wave shapes do not
    assert L.msim_probe_mfma(0, x.data_ptr(), rows_small - 1, 50, sink.data_ptr(), st) != 0
    assert L.msim_probe_mfma(0, x.data_ptr(), rows, 0, sink.data_ptr(), st) != 0
    assert L.msim_probe_mfma(0, None, rows, 50, sink.data_ptr(), st) != 0


def test_forward_workspace_is_optional_and_changes_no_score(amd):
    """msim_fwd's workspace only carries the convoy's progress counters (several query blocks streaming one document range):
    with it, without it (NULL), and with a dirty one the scores are bit-identical; the workspace size query is consistent."""
    L = amd._lib.lib()
    dev = torch.device("cuda:0")
    q, docs = _case(70, 600, ld=257, seed=4)            # 70 queries: three query blocks of 8-wave workgroups
    q = q.to(dev)
    corpus = amd.pack_passages(docs, dev, batch_size=None)
    n_q, Lq, dim = q.shape
    n = len(corpus)
    need = L.msim_fwd_workspace_bytes(0, n_q, Lq, n, dim)
    assert need > 0 and L.msim_fwd_workspace_bytes(0, 4, Lq, n, dim) == 0 and L.msim_fwd_workspace_bytes(2, n_q, Lq, n, dim) == 0
    st = torch.cuda.current_stream().cuda_stream

    def run(ws):
        out = torch.empty((n_q, n), dtype=torch.float32, device=dev)
        rc = L.msim_fwd(0, q.data_ptr(), n_q, Lq, corpus.blob.data_ptr(), corpus.offsets.data_ptr(), None, n, dim, out.data_ptr(), n, 0,
                        None if ws is None else ws.data_ptr(), st)
        assert rc == 0, L.msim_last_error()
        torch.cuda.synchronize()
        return out

    base = run(None)
    clean = torch.zeros(need, dtype=torch.uint8, device=dev)
    dirty = torch.full((need,), 0x7f, dtype=torch.uint8, device=dev)     # the call initialises what it uses
    assert torch.equal(run(clean), base) and torch.equal(run(dirty), base) and torch.equal(run(dirty), base)
    assert torch.equal(amd.maxsim_scores(q, corpus), base)
