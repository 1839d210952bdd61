"""torch (CPU) restatement of the reference scorer -- TEST / BASELINE INFRASTRUCTURE ONLY.

Restates colpali_engine/utils/processing_utils.py:163-186 with the same library
calls the reference makes (pad_sequence, einsum, max, sum) so that bench.py can time
"the reference's CPU scorer" on the GPU box's host cores, where /root/reference does
not exist (cpu_baseline.kind = "port").  tests/test_reference_live.py checks it against
the live reference in the build container.
"""
from __future__ import annotations

from typing import List, Union

import torch
from torch.nn.utils.rnn import pad_sequence


def score_multi_vector_cpu(qs: Union[torch.Tensor, List[torch.Tensor]],
                           ps: Union[torch.Tensor, List[torch.Tensor]],
                           batch_size: int = 128, device: str = "cpu") -> torch.Tensor:
    """`device="cuda:0"` gives what the unmodified reference does on the same MI355X (torch/hipBLASLt einsum)."""
    if len(qs) == 0:
        raise ValueError("No queries provided")          # :163-164
    if len(ps) == 0:
        raise ValueError("No passages provided")         # :165-166
    rows = []
    for i in range(0, len(qs), batch_size):              # :170
        qb = pad_sequence(list(qs[i : i + batch_size]), batch_first=True, padding_value=0).to(device)      # :172-174
        cols = []
        for j in range(0, len(ps), batch_size):          # :175
            pb = pad_sequence(list(ps[j : j + batch_size]), batch_first=True, padding_value=0).to(device)  # :176-178
            sim = torch.einsum("bnd,csd->bcns", qb, pb)  # :179
            cols.append(sim.max(dim=3)[0].sum(dim=2))
        rows.append(torch.cat(cols, dim=1).cpu())        # :180
    return torch.cat(rows, dim=0).to(torch.float32)      # :182-186
