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


def test_patched_forward_under_torch_compile_traces_the_reference_lines_in_one_graph(monkeypatch):
    """trainer/colmodel_torch_training.py:57-63 compiles the model (torch.compile(..., dynamic=True)).  The wrapper ends an eager
    forward by raising out of a module hook -- not something dynamo can graph -- so while dynamo traces it steps aside and the
    reference's own lines are compiled: one graph, no graph break, no recompilation on a second batch shape, the eager result.
    Here with dynamo's `eager` backend (the tracing is what is under test; inductor itself runs in tests/test_gpu_models.py).
    `_fusable` is forced to True so that the ONLY reason the head does not run is the compile check."""
    import torch._dynamo as dynamo
    from torch._dynamo.utils import counters

    model, cls = tiny_colpali()
    b1, b2 = text_batch(), text_batch(B=3, S=21, seed=9)
    with torch.no_grad():
        want1, want2 = model(**b1), model(**b2)
    monkeypatch.setattr(M, "_fusable", lambda lin, kwargs: True)
    monkeypatch.setattr(M, "embedding_head", lambda *a, **k: (_ for _ in ()).throw(AssertionError("fused head reached under compile")))
    M.install(cls)
    try:
        dynamo.reset()
        counters.clear()
        compiled = torch.compile(model, backend="eager", dynamic=True)
        with torch.no_grad():
            got1, got2 = compiled(**b1), compiled(**b2)
        assert torch.equal(got1, want1) and torch.equal(got2, want2)
        assert sum(counters["graph_break"].values()) == 0, dict(counters["graph_break"])
        assert counters["stats"]["unique_graphs"] == 1, dict(counters["stats"])
    finally:
        M.uninstall()
        dynamo.reset()


def test_a_lora_style_projection_is_left_to_the_reference_lines_compiled_or_not(monkeypatch):
    """scripts/configs/qwen2/train_colqwen2_model.yaml:62 targets `custom_text_proj` with LoRA: peft replaces the nn.Linear by a
    wrapper module that adds its own term.  The fused head reads `.weight` only, so it must not run -- eager or compiled."""
    import torch._dynamo as dynamo

    class LoraLinear(torch.nn.Module):          # what peft's lora.Linear computes, in miniature
        def __init__(self, base):
            super().__init__()
            self.base_layer = base
            self.lora_A = torch.nn.Linear(base.in_features, 4, bias=False)
            self.lora_B = torch.nn.Linear(4, base.out_features, bias=False)
            self.in_features, self.out_features = base.in_features, base.out_features

        @property
        def weight(self):
            return self.base_layer.weight

        @property
        def bias(self):
            return self.base_layer.bias

        def forward(self, x):
            return self.base_layer(x) + self.lora_B(self.lora_A(x)) * 2.0

    model, cls = tiny_colpali()
    torch.manual_seed(3)
    model.custom_text_proj = LoraLinear(model.custom_text_proj)
    batch = text_batch()
    with torch.no_grad():
        want = model(**batch)
    monkeypatch.setattr(M, "_fusable", lambda lin, kwargs: True)
    monkeypatch.setattr(M, "embedding_head", lambda *a, **k: (_ for _ in ()).throw(AssertionError("fused head ran on a LoRA projection")))
    M.install(cls)
    try:
        with torch.no_grad():
            assert torch.equal(model(**batch), want)
            dynamo.reset()
            assert torch.equal(torch.compile(model, backend="eager", dynamic=True)(**batch), want)
    finally:
        M.uninstall()
        dynamo.reset()


def test_colmodernvbert_is_not_wrapped():
    """Its tail clamps the norm (modeling_colmodernvbert.py:59: `.clamp_min(1e-12)`): the fused head, which divides by the unclamped
    norm, is not an exact stand-in for it (round-4 advisor finding)."""
    assert "ColModernVBert" not in M.MODEL_CLASS_NAMES
