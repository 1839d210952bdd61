"""The multi-GPU code paths on the ONE GPU a test box has: a 1-rank `nccl` (= RCCL) process group.

* `shard_topk(..., force_collective=True)` sends a single rank through the message packing, the RCCL
  `all_gather_into_tensor` on the uint8 message and the strided-view merge (colpali_amd/retrieval.py) -- the code an
  8-GPU run executes -- and must return what the non-collective path returns.
* BASELINE config 5 through the reference trainer's REAL gather: `torch.distributed.nn.functional.all_gather` -> `cat`
  (trainer/contrastive_trainer.py:14-17, trainer/colmodel_torch_training.py:172-181) feeds the fused loss a non-leaf
  `doc_embeddings` whose gradient flows back through the collective; loss, dQ and dD against the float64 oracle.
* the zero-state loss modules and the fused backward under `DistributedDataParallel` hooks.
"""
import os
import socket

import pytest
import torch

from oracle import li_loss_oracle as lo
from tests.test_gpu_loss import _config5_inputs, grads_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import colpali_amd

    colpali_amd._lib.lib()
    return colpali_amd


@pytest.fixture(scope="module")
def dist():
    import torch.distributed as d

    created = False
    if not d.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        d.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
        created = True
    assert d.get_backend() == "nccl" and d.get_world_size() == 1
    yield d
    if created:
        d.destroy_process_group()


def _unit(g, *shape):
    return torch.nn.functional.normalize(torch.randn(*shape, generator=g), dim=-1)


@pytest.mark.parametrize("n_q,n,k", [(7, 5000, 10), (3, 700, 100), (1, 37, 10), (1000, 2048, 10)])
def test_forced_collective_topk_equals_the_plain_path(amd, dist, n_q, n, k):
    g = torch.Generator().manual_seed(n_q * 131 + n)
    scores = torch.randn(n_q, n, generator=g).cuda()
    scores[:, ::7] = scores[:, 3:4]                                     # exact ties: the (score desc, id asc) order decides
    plain = amd.shard_topk(scores, k, 1000, 1, dist)
    forced = amd.shard_topk(scores, k, 1000, 1, dist, force_collective=True)
    assert torch.equal(plain[0], forced[0]) and torch.equal(plain[1], forced[1])
    assert forced[1].dtype == torch.int64 and forced[0].dtype == torch.float32
    assert int(forced[1][:, : min(k, n)].min()) >= 1000


def test_sharded_retriever_through_rccl_equals_unsharded_search(amd, dist):
    g = torch.Generator().manual_seed(11)
    docs = [_unit(g, int(n), 128).to(torch.bfloat16) for n in torch.randint(20, 300, (400,), generator=g)]
    q = torch.stack([_unit(g, 32, 128).to(torch.bfloat16) for _ in range(9)]).cuda()
    corpus = amd.pack_passages(docs, torch.device("cuda:0"))
    corpus.id_base = 5000
    plain = amd.ShardedRetriever(corpus, world=1, rank=0).search(q, k=10)
    coll = amd.ShardedRetriever(corpus, world=1, rank=0, dist=dist, force_collective=True).search(q, k=10)
    assert torch.equal(plain[0], coll[0]) and torch.equal(plain[1], coll[1])


@pytest.mark.parametrize("cls,kind", [("ColbertPairwiseCELoss", "pairwise"), ("ColbertLoss", "infonce")])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_config5_through_the_autograd_all_gather(amd, dist, cls, kind, dtype):
    """contrastive_trainer.py:14-17 + :143-160: local pages -> all_gather (autograd) -> cat -> loss(offset = rank * B).
    One rank is all a test box has, so the pages of the 'other ranks' (C = 8 * B, this rank's block at offset 96) are
    constants concatenated around the gathered block -- exactly what they are to this rank's autograd graph."""
    from torch.distributed.nn.functional import all_gather

    offset = 96
    Q, D = _config5_inputs(offset)
    Qx, Dx = Q.to(dtype), D.to(dtype)
    B = Q.shape[0]
    kw = dict(normalize_scores=False) if kind == "pairwise" else dict()
    want_loss, want_dq, want_dd = lo.loss_and_grads(kind, Qx.float(), Dx.float(), offset=offset, **kw)
    q = Qx.cuda().requires_grad_(True)
    d_local = Dx[offset:offset + B].cuda().requires_grad_(True)
    before, after = Dx[:offset].cuda(), Dx[offset + B:].cuda()
    gathered = torch.cat(all_gather(d_local), dim=0)                    # keeps the grad graph (contrastive_trainer.py:16)
    assert gathered.grad_fn is not None and not gathered.is_leaf
    docs = torch.cat([before, gathered, after], dim=0)
    loss = getattr(amd, cls)(**kw)(query_embeddings=q, doc_embeddings=docs, offset=dist.get_rank() * B + offset)
    loss.backward()
    q_real = (Qx.float().abs().sum(-1, keepdim=True) > 0)
    d_real = (Dx[offset:offset + B].float().abs().sum(-1, keepdim=True) > 0)
    want_dd_local = want_dd[offset:offset + B]
    if dtype == torch.float32:
        assert abs(float(loss.detach()) - float(want_loss)) <= 1e-5 * abs(float(want_loss)) + 1e-6
        for got, want, mask in ((q.grad, want_dq, q_real), (d_local.grad, want_dd_local, d_real)):
            bad = ((got.cpu().double() - want).abs() > 1e-4 * want.abs() + 1e-6) & mask.expand_as(want)
            assert int(bad.sum()) == 0
    else:
        assert abs(float(loss.detach()) - float(want_loss)) <= 2.0**-8 * abs(float(want_loss)) + 1e-6
        assert grads_close(q.grad, want_dq, q_real.expand_as(want_dq))
        assert grads_close(d_local.grad, want_dd_local, d_real.expand_as(want_dd_local))


@pytest.mark.parametrize("cls,kind", [("ColbertPairwiseCELoss", "pairwise"), ("ColbertLoss", "infonce")])
def test_fused_loss_under_distributed_data_parallel(amd, dist, cls, kind):
    """colmodel_torch_training.py:145-184: a DDP-wrapped producer (here the Col* head alone: Linear(hidden -> 128) + L2
    norm), pages gathered with autograd, the zero-state loss module, backward through DDP's hooks.  fp32 end to end, the
    weight gradient against the same graph evaluated in float64 on the CPU with the oracle's loss."""
    from torch.distributed.nn.functional import all_gather
    from torch.nn.parallel import DistributedDataParallel

    g = torch.Generator().manual_seed(77)
    B, Lq, Ld, H = 8, 12, 40, 64
    head = torch.nn.Linear(H, 128)
    with torch.no_grad():
        head.weight.copy_(torch.randn(128, H, generator=g) / H**0.5)
        head.bias.copy_(0.01 * torch.randn(128, generator=g))
    xq, xd = torch.randn(B, Lq, H, generator=g), torch.randn(B, Ld, H, generator=g)
    for b in range(B):
        xd[b, :Lq] = xq[b] + 0.3 * torch.randn(Lq, H, generator=g)      # the positive page contains the query

    def embed(m, x):
        y = m(x)
        return y / y.norm(dim=-1, keepdim=True)

    ref = torch.nn.Linear(H, 128).double()
    ref.load_state_dict({k: v.double() for k, v in head.state_dict().items()})
    q64, d64 = embed(ref, xq.double()), embed(ref, xd.double())
    kw = dict(normalize_scores=False) if kind == "pairwise" else dict()
    want_loss, gq, gd = lo.loss_and_grads(kind, q64.detach(), d64.detach(), offset=0, **kw)
    torch.autograd.backward([q64, d64], [gq, gd])

    ddp = DistributedDataParallel(head.cuda(), device_ids=[0])
    q, d = embed(ddp, xq.cuda()), embed(ddp, xd.cuda())
    loss = getattr(amd, cls)(**kw)(q, torch.cat(all_gather(d), dim=0), offset=dist.get_rank() * B)
    loss.backward()
    assert abs(float(loss.detach()) - float(want_loss)) <= 1e-5 * abs(float(want_loss)) + 1e-6
    for got, want in ((ddp.module.weight.grad, ref.weight.grad), (ddp.module.bias.grad, ref.bias.grad)):
        err = (got.cpu().double() - want).abs().max() / want.abs().max()
        assert float(err) <= 2e-4, float(err)
