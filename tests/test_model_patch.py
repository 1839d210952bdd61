"""patch_colpali_engine(models=True) on the REAL reference classes (CPU side: the wrapper's control flow; the fused kernel itself is
checked on the GPU in tests/test_gpu_models.py).  Skipped where neither the reference checkout nor its fetched copy exists."""
import pytest
import torch

import colpali_amd
from colpali_amd import models as M
from oracle import refimport

try:
    refimport.model_root()
except RuntimeError:
    pytest.skip("no reference model files here", allow_module_level=True)

from tests.model_fixtures import text_batch, tiny_colpali, tiny_colqwen2  # noqa: E402


def test_patch_installs_and_restores_the_forward():
    model, ColPali = tiny_colpali()
    orig = ColPali.__dict__["forward"]
    colpali_amd.patch_colpali_engine(scorer=False, losses=False, models=True)
    try:
        assert ColPali in M.installed() and ColPali.__dict__["forward"] is not orig
        assert ColPali.forward.__wrapped__ is orig
        batch = text_batch()
        with torch.no_grad():
            y = model(**batch)                 # CPU weights: the wrapper steps aside, the reference's own lines run
        colpali_amd.unpatch_colpali_engine()
        assert ColPali.__dict__["forward"] is orig and ColPali not in M.installed()
        with torch.no_grad():
            assert torch.equal(model(**batch), y)
    finally:
        colpali_amd.unpatch_colpali_engine()


@pytest.mark.parametrize("family", ["colpali", "colqwen2"])
def test_wrapper_hands_the_fused_head_exactly_what_the_projection_would_have_received(monkeypatch, family):
    """The backbone runs as the reference wrote it; the pre-hook on the projection layer takes its input and the head is called with
    (hidden, weight, bias, attention_mask, image mask).  Here the head is a torch stand-in, so the patched forward must equal the
    unpatched one bit for bit -- on both padding sides."""
    model, cls = tiny_colpali() if family == "colpali" else tiny_colqwen2()
    batch = text_batch(left_pad=(family == "colqwen2"))
    with torch.no_grad():
        want = model(**batch)
    calls = []

    def stand_in(hidden, weight, bias, mask, extra=None):
        calls.append((hidden.shape, extra))
        proj = torch.nn.functional.linear(hidden, weight, bias)
        proj = proj / proj.norm(dim=-1, keepdim=True)
        proj = proj * mask.unsqueeze(-1)
        return proj if extra is None else proj * extra.unsqueeze(-1)

    monkeypatch.setattr(M, "_fusable", lambda lin, kwargs: isinstance(kwargs.get("attention_mask"), torch.Tensor))
    monkeypatch.setattr(M, "embedding_head", stand_in)
    M.install(cls)
    try:
        with torch.no_grad():
            got = model(**batch)
        assert len(calls) == 1 and calls[0][0] == (5, 37, 128) and calls[0][1] is None
        assert torch.equal(got, want)
        assert not model.custom_text_proj._forward_pre_hooks          # the hook is gone after every call
        with torch.no_grad():                                          # positional call without the keyword: the reference's lines
            with pytest.raises(KeyError):
                model(batch["input_ids"])                              # (the reference itself needs kwargs["attention_mask"])
    finally:
        M.uninstall(cls)


def test_a_wrapped_projection_layer_keeps_the_reference_path(monkeypatch):
    """peft wraps `custom_text_proj` (scripts/configs/*: target_modules name it) -- its forward adds the LoRA term, which a kernel reading
    `.weight` would drop.  Anything but a plain nn.Linear keeps the reference's lines."""
    model, cls = tiny_colpali()

    class Wrapped(torch.nn.Linear):
        def forward(self, x):
            return super().forward(x) * 2.0

    w = Wrapped(128, 128)
    w.load_state_dict(model.custom_text_proj.state_dict())
    model.custom_text_proj = w
    monkeypatch.setattr(M, "_fusable", lambda lin, kwargs: True)
    monkeypatch.setattr(M, "embedding_head", lambda *a, **k: (_ for _ in ()).throw(AssertionError("must not be called")))
    batch = text_batch()
    with torch.no_grad():
        want = model(**batch)
    M.install(cls)
    try:
        with torch.no_grad():
            assert torch.equal(model(**batch), want)
    finally:
        M.uninstall(cls)
