"""GPU: selection kernel and virtual-shard merge, bit-exact against the oracle ranking."""
import numpy as np
import pytest
import torch

from oracle import topk_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import colpali_amd

    colpali_amd._lib.lib()
    return colpali_amd


@pytest.mark.parametrize("n_q,n,k", [(1, 10, 3), (3, 4096, 10), (2, 4097, 100), (5, 20000, 10), (2, 200000, 100),
                                     (1, 7, 10), (4, 1, 1), (2, 50000, 1024)])
def test_topk_matches_oracle_with_ties(amd, n_q, n, k):
    g = torch.Generator().manual_seed(n + k)
    s = torch.randn(n_q, n, generator=g)
    s = (s * 8).round() / 8 if n > 100 else s          # heavy exact ties: the id order must decide
    s[0, n // 2] = float("-inf")
    if n > 3:
        s[0, 1] = 0.0
        s[0, 2] = -0.0
    gs, gi = amd.topk(s.cuda(), k, id_base=1000)
    ws, wi = topk_oracle.topk(s.numpy(), k, id_base=1000)
    np.testing.assert_array_equal(gi.cpu().numpy(), wi)
    np.testing.assert_array_equal(gs.cpu().numpy(), ws)


@pytest.mark.parametrize("n_q,n,k,kind", [
    (100, 125000, 10, "ties"),          # BASELINE config 4's shard width: 8 filter segments per row
    (200, 40001, 100, "ties"),          # odd row length: rows start off the 16-byte grid (scalar loads), ragged last segment
    (600, 33000, 256, "normal"),        # the largest k the filter takes
    (520, 32768, 10, "ascending"),      # adversarial: every score beats the threshold -> one cut per round
    (520, 50000, 10, "descending"),     # the first round already holds the winners
    (520, 32770, 7, "constant"),        # all equal: ids decide everything
    (70, 125000, 10, "special"),        # -inf, NaN-free extremes, signed zeros
])
def test_streaming_filter_level_matches_oracle(amd, n_q, n, k, kind):
    """Many rows x long rows take the streaming threshold filter at level 0 (topk_select.hip: topk_filter_kernel): ids and scores
    bit-exact against the oracle ranking (score desc, id asc), whatever the data does to the threshold."""
    g = torch.Generator().manual_seed(n + k)
    if kind == "ties":
        s = (torch.randn(n_q, n, generator=g) * 8).round() / 8
    elif kind == "normal":
        s = torch.randn(n_q, n, generator=g)
    elif kind == "ascending":
        s = torch.arange(n, dtype=torch.float32).repeat(n_q, 1) + torch.arange(n_q, dtype=torch.float32)[:, None]
    elif kind == "descending":
        s = -torch.arange(n, dtype=torch.float32).repeat(n_q, 1)
    elif kind == "constant":
        s = torch.full((n_q, n), 3.25)
    else:
        s = torch.randn(n_q, n, generator=g)
        s[:, ::97] = float("-inf")
        s[0, 5] = 3.0e38
        s[1, :] = float("-inf")
        s[2, 10] = 0.0
        s[2, 11] = -0.0
        s[2, 12:] = -1.0
    L = amd._lib.lib()
    assert L.msim_topk_workspace_bytes(n_q, n, k) > 0
    gs, gi = amd.topk(s.cuda(), k, id_base=5000)
    ws, wi = topk_oracle.topk(s.numpy(), k, id_base=5000)
    np.testing.assert_array_equal(gi.cpu().numpy(), wi)
    np.testing.assert_array_equal(gs.cpu().numpy(), ws)


def test_topk_with_explicit_ids_and_padding(amd):
    s = torch.tensor([[1.0, 5.0, 5.0, -1.0, 5.0, 9.0]])
    ids = torch.tensor([[70, 30, 10, -1, 20, -1]])
    gs, gi = amd.topk(s.cuda(), 5, 0, ids.cuda())
    assert gi.cpu().tolist() == [[10, 20, 30, 70, -1]]
    assert gs.cpu().tolist()[0][:4] == [5.0, 5.0, 5.0, 1.0] and gs.cpu()[0, 4] == float("-inf")


def test_virtual_shards_on_one_gpu_equal_unsharded(amd):
    g = torch.Generator().manual_seed(3)
    n_docs, k = 1000, 10
    docs = [torch.nn.functional.normalize(torch.randn(64, 128, generator=g), dim=-1).to(torch.bfloat16) for _ in range(n_docs)]
    docs[777] = docs[5].clone()
    q = torch.nn.functional.normalize(torch.randn(6, 32, 128, generator=g), dim=-1).to(torch.bfloat16).cuda()
    dev = torch.device("cuda:0")
    full = amd.pack_passages(docs, dev, batch_size=None)
    fs, fi = amd.topk(amd.maxsim_scores(q, full), k)
    for world in (2, 8):
        parts = []
        for r in range(world):
            lo, hi = amd.shard_range(n_docs, world, r)
            shard = amd.pack_passages(docs[lo:hi], dev, batch_size=None, id_base=lo)
            parts.append(amd.topk(amd.maxsim_scores(q, shard), k, id_base=lo))
        ms, mi = amd.merge_gathered(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]), k)
        assert torch.equal(mi, fi) and torch.equal(ms, fs)


class _FakeDist:
    """Stands in for torch.distributed inside one process: rank r's message is whatever virtual shard r produced."""

    def __init__(self, messages, me):
        self.messages, self.me = messages, me

    def all_gather_into_tensor(self, out, mine, group=None):
        self.messages[self.me] = mine.clone()
        out.copy_(torch.cat([m.reshape(-1) for m in self.messages]))


def test_shard_topk_single_packed_message_equals_unsharded(amd):
    # the N > 1 path of shard_topk on one GPU: the selection kernel writes (scores | ids) straight into the rank's message,
    # one all-gather of that byte buffer, strided views of the gathered bytes go into the merge
    g = torch.Generator().manual_seed(11)
    n_docs, k, world = 900, 7, 3                         # n_q * k * 4 = 140 bytes: not a multiple of 8 -> exercises the padding
    docs = [torch.nn.functional.normalize(torch.randn(48, 128, generator=g), dim=-1).to(torch.bfloat16) for _ in range(n_docs)]
    q = torch.nn.functional.normalize(torch.randn(5, 32, 128, generator=g), dim=-1).to(torch.bfloat16).cuda()
    dev = torch.device("cuda:0")
    fs, fi = amd.topk(amd.maxsim_scores(q, amd.pack_passages(docs, dev, batch_size=None)), k)
    shards = []
    for r in range(world):
        lo, hi = amd.shard_range(n_docs, world, r)
        shards.append((lo, amd.maxsim_scores(q, amd.pack_passages(docs[lo:hi], dev, batch_size=None, id_base=lo))))
    n_q = q.shape[0]
    nbytes = (n_q * k * 4 + 7) // 8 * 8 + n_q * k * 8
    messages = [torch.zeros(nbytes, dtype=torch.uint8, device=dev) for _ in range(world)]
    for r in range(world):                               # first pass fills every rank's message, second pass merges with all of them
        amd.shard_topk(shards[r][1], k, shards[r][0], world, _FakeDist(messages, r))
    for r in range(world):
        ms, mi = amd.shard_topk(shards[r][1], k, shards[r][0], world, _FakeDist(messages, r))
        assert torch.equal(mi, fi) and torch.equal(ms, fs)
