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
    """msim_probe_stream (measurement aid): every variant launches on a conforming matrix; bad shapes are refused."""
    L = amd._lib.lib()
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
