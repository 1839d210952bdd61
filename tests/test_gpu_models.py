"""patch_colpali_engine(models=True) on the GPU: the REAL reference model classes (random-init tiny configs) with the fused
embedding head installed, against the same instances unpatched -- i.e. against the reference's own lines
(modeling_colpali.py:65-78, modeling_colqwen2.py:59-75) executed by torch on the same GPU.

Tolerances: bf16 / fp16 outputs within one ulp of the reference's output element-wise and > 90 % bit-equal (the reference rounds the
Linear output, the norm and the quotient to the model dtype: one rounding each, reproduced by the kernel; the library GEMM's
accumulation order is the only difference); gradients against the reference's autograd within 2e-2 relative to the tensor's scale
(bf16 autograd rounds five intermediates, the fused backward one)."""
import pytest
import torch

import colpali_amd
from colpali_amd import models as M
from oracle import refimport

pytestmark = pytest.mark.gpu

try:
    refimport.model_root()
except RuntimeError:
    pytest.skip("no reference model files (tests/_reference_pkg/ is fetched by __graft_entry__.build())", allow_module_level=True)

from tests.model_fixtures import colpali_page_batch, colqwen2_page_batch, text_batch, tiny_colpali, tiny_colqwen2  # noqa: E402

DEV = "cuda:0"


def one_ulp_close(got, want, dtype):
    got, want = got.float(), want.float()
    mant = 7 if dtype == torch.bfloat16 else 10
    ulp = torch.exp2(torch.floor(torch.log2(want.abs().clamp_min(1e-30))) - mant)
    ok = (got - want).abs() <= ulp + 1e-12
    return bool(ok.all()), float((got == want).float().mean())


@pytest.fixture()
def patched():
    yield
    colpali_amd.unpatch_colpali_engine()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("family", ["colpali", "colqwen2"])
def test_patched_forward_equals_the_reference_forward(patched, family, dtype):
    model, cls = tiny_colpali() if family == "colpali" else tiny_colqwen2()
    model = model.to(DEV, dtype)
    batch = text_batch(left_pad=(family == "colqwen2"), device=DEV)
    with torch.no_grad():
        want = model(**batch)
    colpali_amd.patch_colpali_engine(scorer=False, losses=False, models=True)
    assert cls in M.installed()
    calls = []
    real = M.embedding_head
    M.embedding_head = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        with torch.no_grad():
            got = model(**batch)
    finally:
        M.embedding_head = real
    assert calls == [1], "the fused head did not run"
    assert got.dtype == want.dtype and got.shape == want.shape
    pad = batch["attention_mask"] == 0
    assert bool((got[pad] == 0).all()) and bool((want[pad] == 0).all())
    ok, same = one_ulp_close(got, want, dtype)
    assert ok and same > 0.9, same
    colpali_amd.unpatch_colpali_engine()
    with torch.no_grad():
        assert torch.equal(model(**batch), want)


@pytest.mark.parametrize("mask_non_image", [False, True])
def test_patched_colpali_with_an_image_and_the_image_token_mask(patched, mask_non_image):
    model, cls = tiny_colpali(mask_non_image_embeddings=mask_non_image)
    model = model.to(DEV, torch.bfloat16)
    batch = colpali_page_batch(device=DEV)
    with torch.no_grad():
        want = model(**batch)
    colpali_amd.patch_colpali_engine(scorer=False, losses=False, models=True)
    with torch.no_grad():
        got = model(**batch)
    ok, same = one_ulp_close(got, want, torch.bfloat16)
    assert ok and same > 0.9, same
    if mask_non_image:
        assert bool((got[:, 4:] == 0).all()) and bool((got[:, :4].float().norm(dim=-1) > 0.9).all())


@pytest.mark.parametrize("mask_non_image", [False, True])
def test_patched_colqwen2_with_an_image(patched, mask_non_image):
    """ColQwen2.forward's own preamble (un-padding `pixel_values` with image_grid_thw, modeling_colqwen2.py:50-56) and the Qwen2-VL
    backbone run unchanged in front of the fused head; `mask_non_image_embeddings` uses this family's `config.image_token_id`."""
    model, cls = tiny_colqwen2()
    model.mask_non_image_embeddings = mask_non_image
    model = model.to(DEV, torch.bfloat16)
    batch = colqwen2_page_batch(model, device=DEV)
    batch["pixel_values"] = batch["pixel_values"].to(torch.bfloat16)
    with torch.no_grad():
        want = model(**{k: (v.clone() if k == "pixel_values" else v) for k, v in batch.items()})
    colpali_amd.patch_colpali_engine(scorer=False, losses=False, models=True)
    assert cls in M.installed()
    with torch.no_grad():
        got = model(**batch)
    ok, same = one_ulp_close(got, want, torch.bfloat16)
    assert ok and same > 0.9, same
    if mask_non_image:
        img = batch["input_ids"] == model.config.image_token_id
        assert bool((got[~img] == 0).all()) and bool((got[img].float().norm(dim=-1) > 0.9).all())


def test_patched_forward_is_differentiable_like_the_reference(patched):
    """The model forward is part of the training graph (trainer/contrastive_trainer.py:135-162): gradients reach the projection's weight
    and bias and the backbone through the fused head."""
    model, cls = tiny_colpali()
    model = model.to(DEV, torch.bfloat16).train()
    batch = text_batch(device=DEV)
    G = torch.randn(5, 37, 128, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3)).to(torch.bfloat16)

    def grads():
        model.zero_grad(set_to_none=True)
        y = model(**batch)
        (y.float() * G.float()).sum().backward()
        emb = model.get_input_embeddings().weight.grad
        return y.detach(), model.custom_text_proj.weight.grad.float(), model.custom_text_proj.bias.grad.float(), emb.float()

    y0, dw0, db0, de0 = grads()
    colpali_amd.patch_colpali_engine(scorer=False, losses=False, models=True)
    y1, dw1, db1, de1 = grads()
    assert one_ulp_close(y1, y0, torch.bfloat16)[0]
    for a, b in ((dw1, dw0), (db1, db0), (de1, de0)):
        assert float((a - b).abs().max()) <= 2e-2 * float(b.abs().max()) + 1e-6
    assert float(de1.abs().max()) > 0            # the gradient really went through the backbone


def test_corpus_from_a_patched_model_scores_like_the_reference_road(patched):
    """BASELINE config 2's road end to end on a tiny model: pages -> patched ColPali.forward -> list(torch.unbind(.cpu())) ->
    score_multi_vector, against the unpatched model and the reference's blocked scorer semantics (the C oracle)."""
    import numpy as np

    from oracle import maxsim_oracle as mo

    model, cls = tiny_colpali()
    model = model.to(DEV, torch.bfloat16)
    pages, queries = colpali_page_batch(B=6, n_text=5, device=DEV), text_batch(B=3, S=12, device=DEV)
    colpali_amd.patch_colpali_engine(scorer=False, losses=False, models=True)
    with torch.no_grad():
        ps = list(torch.unbind(model(**pages).cpu()))
        qs = list(torch.unbind(model(**queries).cpu()))
    got = colpali_amd.score_multi_vector(qs, ps, device=DEV).numpy()
    want = mo.score_multi_vector([q.float().numpy() for q in qs], [p.float().numpy() for p in ps])
    assert np.max(np.abs(got - want) / np.maximum(np.abs(want), 1.0)) <= 1e-5


def _require_inductor():
    """The compile tests need a working inductor (Triton code generation for gfx950) on this box: probe it on a trivial function and
    skip -- not fail -- where the stock toolchain itself is unusable (that is not this repository's code)."""
    import torch._dynamo as dynamo

    try:
        dynamo.reset()
        x = torch.arange(64, device=DEV, dtype=torch.float32)
        y = torch.compile(lambda t: torch.sin(t) * 2 + 1, backend="inductor")(x)
        assert torch.allclose(y, torch.sin(x) * 2 + 1, atol=1e-5)
    except Exception as e:      # noqa: BLE001
        pytest.skip(f"torch.compile(backend='inductor') does not work on this box: {type(e).__name__}: {str(e)[:200]}")
    finally:
        dynamo.reset()


def test_patched_model_under_torch_compile_inductor_dynamic(patched):
    """trainer/colmodel_torch_training.py:57-63: `torch.compile(model, backend="inductor", dynamic=True)`.  While dynamo traces, the
    wrapper steps aside (colpali_amd/models.py): the reference's own lines are what inductor compiles -- forward equal to the eager
    unpatched model to the model dtype's rounding on two batch shapes, one graph, no graph break, no recompilation storm, a working
    backward; outside compilation the same patched class still runs the fused head."""
    import torch._dynamo as dynamo
    from torch._dynamo.utils import counters

    _require_inductor()
    model, cls = tiny_colpali()
    model = model.to(DEV, torch.bfloat16).train()
    b1, b2 = text_batch(device=DEV), text_batch(B=3, S=21, seed=9, device=DEV)
    with torch.no_grad():
        want1, want2 = model(**b1), model(**b2)
    colpali_amd.patch_colpali_engine(scorer=False, losses=False, models=True)
    calls = []
    real = M.embedding_head
    M.embedding_head = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        dynamo.reset()
        counters.clear()
        compiled = torch.compile(model, backend="inductor", dynamic=True)
        with torch.no_grad():
            got1, got2 = compiled(**b1), compiled(**b2)
        assert calls == [], "the fused head ran inside a compiled forward"
        for got, want in ((got1, want1), (got2, want2)):
            # inductor fuses the norm / divide / mask lines in fp32 and rounds once where eager rounds three times: two bf16 ulps
            assert float((got.float() - want.float()).abs().max()) <= 2 * 2.0**-8
        assert sum(counters["graph_break"].values()) == 0, dict(counters["graph_break"])
        assert counters["stats"]["unique_graphs"] <= 2, dict(counters["stats"])        # dynamic=True: at most one re-trace for the second shape
        y = compiled(**b1)
        (y.float() ** 2).sum().backward()
        assert model.custom_text_proj.weight.grad is not None and float(model.custom_text_proj.weight.grad.float().abs().max()) > 0
        with torch.no_grad():
            eager = model(**b1)                       # the same patched class outside compilation: the fused head
        assert calls == [1]
        assert one_ulp_close(eager, want1, torch.bfloat16)[0]
    finally:
        M.embedding_head = real
        dynamo.reset()


def test_patched_model_under_distributed_data_parallel(patched):
    """trainer/colmodel_torch_training.py:57-59 wraps the model in DistributedDataParallel before compiling it: DDP calls
    `module.forward`, i.e. the wrapper -- the fused head runs, its backward feeds DDP's gradient hooks (a 1-rank nccl group)."""
    import os
    import socket

    import torch.distributed as dist

    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device(DEV))
        created = True
    try:
        model, cls = tiny_colpali()
        model = model.to(DEV, torch.bfloat16).train()
        batch = text_batch(device=DEV)
        G = torch.randn(5, 37, 128, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3)).to(torch.bfloat16)
        model.zero_grad(set_to_none=True)
        (model(**batch).float() * G.float()).sum().backward()
        want = model.custom_text_proj.weight.grad.float().clone()
        colpali_amd.patch_colpali_engine(scorer=False, losses=False, models=True)
        calls = []
        real = M.embedding_head
        M.embedding_head = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        try:
            ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0])
            model.zero_grad(set_to_none=True)
            (ddp(**batch).float() * G.float()).sum().backward()
        finally:
            M.embedding_head = real
        assert calls == [1], "the fused head did not run under DDP"
        got = model.custom_text_proj.weight.grad.float()
        assert float((got - want).abs().max()) <= 2e-2 * float(want.abs().max()) + 1e-6
    finally:
        if created:
            dist.destroy_process_group()


def test_lora_wrapped_projection_keeps_the_reference_lines_on_the_gpu(patched):
    """scripts/configs/qwen2/train_colqwen2_model.yaml:62 puts a LoRA adapter on `custom_text_proj`: no longer a plain nn.Linear, so the
    fused head (which reads `.weight` only) must not run -- eager and compiled -- and the adapter's term must be in the output."""
    import torch._dynamo as dynamo

    class LoraLinear(torch.nn.Module):
        def __init__(self, base):
            super().__init__()
            self.base_layer = base
            self.lora_A = torch.nn.Linear(base.in_features, 4, bias=False)
            self.lora_B = torch.nn.Linear(4, base.out_features, bias=False)

        def forward(self, x):
            return self.base_layer(x) + self.lora_B(self.lora_A(x)) * 2.0

    _require_inductor()
    model, cls = tiny_colpali()
    torch.manual_seed(3)
    model.custom_text_proj = LoraLinear(model.custom_text_proj)
    model = model.to(DEV, torch.bfloat16)
    batch = text_batch(device=DEV)
    with torch.no_grad():
        want = model(**batch)
    colpali_amd.patch_colpali_engine(scorer=False, losses=False, models=True)
    calls = []
    real = M.embedding_head
    M.embedding_head = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        with torch.no_grad():
            assert torch.equal(model(**batch), want)
            dynamo.reset()
            got = torch.compile(model, backend="inductor", dynamic=True)(**batch)
        assert float((got.float() - want.float()).abs().max()) <= 2 * 2.0**-8
        assert calls == []
    finally:
        M.embedding_head = real
        dynamo.reset()
