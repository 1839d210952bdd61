"""The fused embedding head INSIDE the reference's model classes.

Every Col* model of the reference ends its forward with the same lines (e.g.
colpali_engine/models/paligemma/colpali/modeling_colpali.py:65-78,
colpali_engine/models/qwen2/colqwen2/modeling_colqwen2.py:59-75, and the same in colqwen2_5 / colqwen3 / colqwen3_5 /
colgemma3 / colidefics3 / colqwen_omni):

    proj = self.custom_text_proj(last_hidden_states)          # `self.linear` in ColIdefics3
    proj = proj / proj.norm(dim=-1, keepdim=True)
    proj = proj * kwargs["attention_mask"].unsqueeze(-1)
    if "pixel_values" in kwargs and self.mask_non_image_embeddings:
        proj = proj * (kwargs["input_ids"] == self.config.image_token_id).unsqueeze(-1)
    return proj

`install(cls)` replaces `cls.forward` by a wrapper that runs the class's OWN forward -- its argument handling and the
whole VLM backbone stay exactly what the reference wrote, on PyTorch-ROCm -- up to the moment the projection layer is
entered: a forward pre-hook on that layer takes the hidden states and ends the forward there, and the lines above run as
ONE HIP kernel (colpali_amd.embedding_head -> msim_embed_head), differentiable like the lines it replaces.  Nothing of
a family's forward is restated here, so the wrapper serves every family, including ones added upstream later.

The wrapper steps aside (the reference's own lines run, untouched) whenever the fused kernel is not an exact stand-in:
the projection is not a plain nn.Linear (a peft LoRA wrapper adds its own term), its output width is not 128, the weights
are not bf16 / fp16 on the GPU, the hidden size is not a multiple of 64, autocast is active, the call carries no
attention_mask keyword -- or the forward is being traced by torch.compile (trainer/colmodel_torch_training.py:57-63 wraps the
model in DDP and then in torch.compile(backend="inductor", dynamic=True)): a forward that ends by raising out of a module hook is
not something dynamo can turn into a graph, and a C-ABI launch through ctypes is opaque to it, so under compilation the reference's
own four lines are what inductor compiles (tests/test_gpu_models.py runs both wrappers on the real classes).

Consequence, stated plainly: in the training configurations the reference SHIPS the fused head is
bypassed -- scripts/configs/qwen2/train_colqwen2_model.yaml:62 puts a LoRA adapter on `custom_text_proj` (not a plain nn.Linear any
more), the HF Trainer path runs under bf16 autocast, and the torch loop compiles the model.  The fused head and its backward serve
inference / corpus building and full-fine-tuning loops in eager mode; the loss kernels are unaffected by any of this.

ColModernVBert is NOT wrapped: its tail divides by `norm.clamp_min(1e-12)` (modeling_colmodernvbert.py:59), so a projection row that
underflows to zero comes out 0 there and NaN from a kernel that divides by the unclamped norm (round-4 advisor finding).
"""
from __future__ import annotations

import threading
from typing import Callable, Dict, Optional, Tuple

import torch

from .embed import HEAD_DIM, embedding_head

MODEL_CLASS_NAMES = ("ColPali", "ColQwen2", "ColQwen2_5", "ColQwen3", "ColQwen3_5", "ColGemma3", "ColIdefics3", "ColQwen2_5Omni")
_PROJECTION_ATTRS = ("custom_text_proj", "linear")

_originals: Dict[type, Callable] = {}


class _HeadReached(Exception):
    """Raised by the pre-hook on the projection layer: the backbone is done, the hidden states are in hand."""


def _projection(model) -> Optional[torch.nn.Linear]:
    for name in _PROJECTION_ATTRS:
        lin = getattr(model, name, None)
        if lin is not None:
            return lin if type(lin) is torch.nn.Linear else None      # a subclass / peft wrapper computes something else
    return None


def _fusable(lin: torch.nn.Linear, kwargs) -> bool:
    w = lin.weight
    return (w.device.type == "cuda" and w.dtype in (torch.bfloat16, torch.float16) and lin.out_features == HEAD_DIM
            and lin.in_features % 64 == 0 and (lin.bias is None or lin.bias.dtype == w.dtype)
            and isinstance(kwargs.get("attention_mask"), torch.Tensor) and not torch.is_autocast_enabled())


def _image_token_id(config):
    tok = getattr(config, "image_token_id", None)
    return tok if tok is not None else getattr(config, "image_token_index", None)


def _wrap(orig_forward: Callable) -> Callable:
    def forward(self, *args, **kwargs):
        if torch.compiler.is_compiling():        # dynamo is tracing this forward: the reference's own lines are what gets compiled
            return orig_forward(self, *args, **kwargs)
        lin = _projection(self)
        if lin is None or not _fusable(lin, kwargs):
            return orig_forward(self, *args, **kwargs)
        taken = []
        mask = kwargs["attention_mask"]
        me = threading.get_ident()

        def take(_module, inputs):
            if threading.get_ident() != me:      # another thread's forward through the same module: not this call's hidden states
                return
            hidden = inputs[0]
            if hidden.dim() == 3 and hidden.dtype == lin.weight.dtype and hidden.shape[:2] == mask.shape:
                taken.append(hidden)
                raise _HeadReached()
            # anything else is not what the fused kernel stands in for: let the reference's own lines run

        handle = lin.register_forward_pre_hook(take)
        try:
            return orig_forward(self, *args, **kwargs)       # returns only if the hook let the projection layer run
        except _HeadReached:
            pass
        finally:
            handle.remove()
        extra = None
        if "pixel_values" in kwargs and getattr(self, "mask_non_image_embeddings", False):
            extra = kwargs["input_ids"] == _image_token_id(self.config)
        return embedding_head(taken[0], lin.weight, lin.bias, mask, extra)

    forward.__wrapped__ = orig_forward
    forward.__doc__ = orig_forward.__doc__
    return forward


def install(cls: type) -> bool:
    """Put the fused head into `cls.forward` (idempotent).  Returns False for a class without a forward of its own."""
    if cls in _originals:
        return True
    orig = cls.__dict__.get("forward")
    if orig is None:
        return False
    _originals[cls] = orig
    cls.forward = _wrap(orig)
    return True


def uninstall(cls: Optional[type] = None) -> None:
    """Restore the reference's forward (of one class, or of every class `install` touched)."""
    for c in ([cls] if cls is not None else list(_originals)):
        orig = _originals.pop(c, None)
        if orig is not None:
            c.forward = orig


def installed() -> Tuple[type, ...]:
    return tuple(_originals)
