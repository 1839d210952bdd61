"""Quick on-GPU diagnostic (not a test): prints HIP-vs-oracle differences for a few tiny shapes."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
import colpali_amd
from oracle import maxsim_oracle as mo

torch.manual_seed(0)
def unit(n): return torch.nn.functional.normalize(torch.randn(n, 128), dim=-1).to(torch.bfloat16)
for (nq, lq, nd, ld) in [(1, 32, 1, 32), (1, 32, 1, 64), (1, 32, 3, 40), (1, 7, 5, 100), (2, 32, 9, 1024), (4, 32, 16, 1024)]:
    qs = [unit(lq) for _ in range(nq)]; ps = [unit(ld) for _ in range(nd)]
    got = colpali_amd.score_multi_vector(qs, ps, device="cuda:0").numpy()
    want = mo.score_multi_vector([q.float().numpy() for q in qs], [p.float().numpy() for p in ps])
    print((nq, lq, nd, ld), "max abs diff", np.abs(got - want).max(), "got", got.ravel()[:4], "want", want.ravel()[:4])
