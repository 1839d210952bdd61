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
