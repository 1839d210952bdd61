"""CPU restatement of the Col* embedding head -- TEST INFRASTRUCTURE ONLY.

Restates, with the reference's own torch calls,
    colpali_engine/models/paligemma/colpali/modeling_colpali.py:67   proj = self.custom_text_proj(last_hidden_states)
                                                                :70   proj = proj / proj.norm(dim=-1, keepdim=True)
                                                                :72   proj = proj * kwargs["attention_mask"].unsqueeze(-1)
                                                                :74-77 proj = proj * image_mask        (optional)
    (colpali_engine/models/qwen2/colqwen2/modeling_colqwen2.py:65-74 are the same lines)
`literal`: the lines evaluated on CPU in the tensors' own dtype (what the reference does, rounding chain included);
`truth`:   the same in float64.
Pinned by tests/test_head_oracle_golden.py against tests/golden/head_colpali_tiny.npz: the output of the live
reference ColPali.forward (random-init tiny PaliGemma config) for the hidden states its head received.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F  # noqa: N812


def head_literal(hidden: torch.Tensor, weight: torch.Tensor, bias, attention_mask: torch.Tensor, extra_mask=None):
    proj = F.linear(hidden, weight, bias)                       # :67 nn.Linear
    proj = proj / proj.norm(dim=-1, keepdim=True)               # :70
    proj = proj * attention_mask.unsqueeze(-1)                  # :72
    if extra_mask is not None:
        proj = proj * extra_mask.reshape(*attention_mask.shape, 1)   # :74-77
    return proj


def head_truth(hidden, weight, bias, attention_mask, extra_mask=None):
    return head_literal(hidden.double(), weight.double(), None if bias is None else bias.double(), attention_mask, extra_mask)
