"""GPU fuzz: random shapes across every forward dispatch path (K1s, K1b, K1sP, K1bP, K1g; bf16 / fp16 / fp32; widths 128, 320
and odd ones; ragged documents with empty-free random lengths; random reference block sizes) against the C oracle's truth
tier, 1e-5 relative to max(|truth|, 1).  Seeds are fixed: a failure is reproducible by its case index."""
import numpy as np
import pytest
import torch

from oracle import maxsim_oracle as mo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import colpali_amd

    colpali_amd._lib.lib()
    return colpali_amd


def _case(i):
    rng = np.random.default_rng(1000 + i)
    dtype = [torch.bfloat16, torch.bfloat16, torch.float16, torch.float32][int(rng.integers(0, 4))]
    dim = int(rng.choice([128, 128, 128, 320, 320, 64, 48, 100, 256]))
    n_q = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 13, 16, 17, 31, 33, 40]))
    lq = int(rng.choice([1, 5, 31, 32, 33, 64, 65, 97, 128, 129, 140]))
    n_d = int(rng.integers(1, 260))
    ld = int(rng.choice([1, 2, 31, 32, 33, 100, 257, 700]))
    bs = int(rng.choice([1, 3, 16, 128]))
    g = torch.Generator().manual_seed(5000 + i)

    def unit(n):
        return torch.nn.functional.normalize(torch.randn(n, dim, generator=g), dim=-1).to(dtype)

    q_lens = rng.integers(1, lq + 1, size=n_q).tolist()
    q_lens[0] = lq
    d_lens = rng.integers(1, ld + 1, size=n_d).tolist()
    return [unit(n) for n in q_lens], [unit(n) for n in d_lens], bs, (str(dtype), dim, n_q, lq, n_d, ld, bs)


@pytest.mark.parametrize("block", range(6))
def test_random_shapes_against_oracle(amd, block):
    for i in range(block * 10, block * 10 + 10):
        qs, ps, bs, desc = _case(i)
        got = amd.score_multi_vector(qs, ps, batch_size=bs, device="cuda:0").numpy()
        want = mo.score_multi_vector([q.float().numpy() for q in qs], [p.float().numpy() for p in ps], batch_size=bs, mode="f32")
        err = np.max(np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want), 1.0))
        assert err <= 1e-5, (i, desc, err)
