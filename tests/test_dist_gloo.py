"""world_size=2 `gloo` tests of the sharded-retrieval host logic (CPU, no GPU).

The GPU kernels cannot run here, so the per-shard scorer and the selection kernel are replaced by
the oracle through the injection points `score_fn` / `select`; what is under test is the product's
sharding, id bookkeeping, collective plumbing and merge order: the result on every rank must be
identical to the unsharded ranking.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_docs, k, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import colpali_amd
    from colpali_amd.corpus import PackedCorpus
    from oracle import maxsim_oracle as mo
    from oracle import topk_oracle

    g = torch.Generator().manual_seed(42)
    lens = torch.randint(1, 40, (n_docs,), generator=g).tolist()
    docs = [torch.nn.functional.normalize(torch.randn(n, 128, generator=g), dim=-1).to(torch.bfloat16) for n in lens]
    docs[5] = docs[3].clone()                 # exact duplicates -> exact score ties across shards
    docs[n_docs - 2] = docs[3].clone()
    q = torch.nn.functional.normalize(torch.randn(3, 8, 128, generator=g), dim=-1).to(torch.bfloat16)

    lo, hi = colpali_amd.shard_range(n_docs, world, rank)
    shard = colpali_amd.pack_passages(docs[lo:hi], torch.device("cpu"), batch_size=None, id_base=lo)
    assert isinstance(shard, PackedCorpus) and shard.id_base == lo

    def score_fn(queries, corpus):            # oracle in place of the HIP kernel
        blob = corpus.blob.float().numpy()
        return torch.from_numpy(mo.maxsim_f32(queries.float().numpy(), blob, corpus.offsets.numpy(), None))

    r = colpali_amd.ShardedRetriever(shard, world=world, rank=rank, dist=dist, score_fn=score_fn,
                                     select=topk_oracle.torch_select)
    s, i = r.search(q, k=k)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), s=s.numpy(), i=i.numpy())

    if rank == 0:                             # unsharded truth
        full = colpali_amd.pack_passages(docs, torch.device("cpu"), batch_size=None)
        fs = score_fn(q, full).numpy()
        ws, wi = topk_oracle.topk(fs, k)
        np.savez(os.path.join(out_dir, "truth.npz"), s=ws, i=wi)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_docs,k", [(2, 37, 5), (2, 12, 10), (3, 50, 7)])
def test_sharded_topk_equals_unsharded(tmp_path, world, n_docs, k):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_docs, k, str(tmp_path)), nprocs=world, join=True)
    truth = np.load(tmp_path / "truth.npz")
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        np.testing.assert_array_equal(got["i"], truth["i"])
        np.testing.assert_array_equal(got["s"], truth["s"])
    # the planted duplicates must appear in id order
    row = truth["i"][0].tolist()
    if 3 in row and 5 in row:
        assert row.index(3) < row.index(5)
