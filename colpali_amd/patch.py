"""Install the MI355X-native implementations into an importable `colpali_engine`.

    import colpali_amd; colpali_amd.patch_colpali_engine()

After the call
  * `BaseVisualRetrieverProcessor.score_multi_vector` (inherited by every Col*Processor and reached
    through `processor.score(...)`, e.g. models/paligemma/colpali/processing_colpali.py:96-106)
    is colpali_amd.score_multi_vector; `create_plaid_index` / `get_topk_plaid` (:189-244, the reference's only
    top-k API) build and query the exact resident index instead of fast_plaid's approximate one;
  * `colpali_engine.loss.ColbertPairwiseCELoss` / `ColbertLoss` / `ColbertSigmoidLoss`
    (and the same names in `colpali_engine.loss.late_interaction_losses`) are the fused versions, so
    YAML configs that name the class by dotted path (scripts/configs/qwen2/train_colqwen2_model.yaml:24-25)
    pick them up.
  * with `models=True`: the forward of every Col* model class that is importable (`colpali_engine.models`, or the model modules
    already imported) ends in the fused embedding head (colpali_amd/models.py): the class's own argument handling and VLM
    backbone run unchanged on PyTorch-ROCm, `custom_text_proj` + L2 norm + masks (modeling_colpali.py:67-77 and the same lines
    in the other families) become one HIP kernel.
`unpatch_colpali_engine()` restores the originals.
"""
from __future__ import annotations

import importlib
import sys

from . import loss as _loss
from . import models as _models
from .pooling import HierarchicalTokenPooler
from .retrieval import create_plaid_index, get_topk_plaid
from .scoring import get_similarity_maps_from_embeddings, score_multi_vector, score_single_vector

_LOSS_NAMES = ("ColbertPairwiseCELoss", "ColbertLoss", "ColbertSigmoidLoss", "ColbertNegativeCELoss",
               "ColbertPairwiseNegativeCELoss")
_saved = {}


def _model_classes():
    """Col* model classes of an importable colpali_engine: the whole `colpali_engine.models` package when its dependencies are
    installed, else whatever model modules the caller has imported (the package __init__ pulls every family)."""
    try:
        importlib.import_module("colpali_engine.models")
    except Exception:      # a family's dependency is missing here: patch the families that ARE imported
        pass
    found = []
    for name, mod in list(sys.modules.items()):
        if mod is None or not name.startswith("colpali_engine.models"):
            continue
        for cname in _models.MODEL_CLASS_NAMES:
            cls = getattr(mod, cname, None)
            if isinstance(cls, type) and cls not in found and getattr(cls, "__module__", "").startswith("colpali_engine.models"):
                found.append(cls)
    return found


def patch_colpali_engine(scorer: bool = True, losses: bool = True, models: bool = False) -> None:
    if models:
        for cls in _model_classes():
            _models.install(cls)
    if scorer:
        pu = importlib.import_module("colpali_engine.utils.processing_utils")
        cls = pu.BaseVisualRetrieverProcessor
        _saved.setdefault("score_multi_vector", cls.__dict__["score_multi_vector"])
        cls.score_multi_vector = staticmethod(score_multi_vector)
        _saved.setdefault("score_single_vector", cls.__dict__["score_single_vector"])
        cls.score_single_vector = staticmethod(score_single_vector)
        for name, fn in (("get_topk_plaid", get_topk_plaid), ("create_plaid_index", create_plaid_index)):
            if name in cls.__dict__:                       # the reference's experimental top-k API (processing_utils.py:189-244)
                _saved.setdefault(name, cls.__dict__[name])
                setattr(cls, name, staticmethod(fn))
        sm = sys.modules.get("colpali_engine.interpretability.similarity_map_utils")
        if sm is not None:   # only when the user has imported the interpretability helpers
            _saved.setdefault((sm.__name__, "get_similarity_maps_from_embeddings"), sm.get_similarity_maps_from_embeddings)
            sm.get_similarity_maps_from_embeddings = get_similarity_maps_from_embeddings
        for name in ("colpali_engine.compression.token_pooling.hierarchical_token_pooling",
                     "colpali_engine.compression.token_pooling", "colpali_engine.compression"):
            mod = sys.modules.get(name)   # only when the user has imported the pooling helpers
            if mod is not None and hasattr(mod, "HierarchicalTokenPooler"):
                _saved.setdefault((name, "HierarchicalTokenPooler"), mod.HierarchicalTokenPooler)
                mod.HierarchicalTokenPooler = HierarchicalTokenPooler
    if losses:
        mods = [importlib.import_module("colpali_engine.loss.late_interaction_losses")]
        pkg = sys.modules.get("colpali_engine.loss")
        if pkg is not None:
            mods.append(pkg)
        for m in mods:
            for name in _LOSS_NAMES:
                if hasattr(m, name):
                    _saved.setdefault((m.__name__, name), getattr(m, name))
                    setattr(m, name, getattr(_loss, name))


def unpatch_colpali_engine() -> None:
    _models.uninstall()
    for key, obj in list(_saved.items()):
        if key in ("score_multi_vector", "score_single_vector", "get_topk_plaid", "create_plaid_index"):
            pu = importlib.import_module("colpali_engine.utils.processing_utils")
            setattr(pu.BaseVisualRetrieverProcessor, key, obj)
        else:
            setattr(sys.modules[key[0]], key[1], obj)
        del _saved[key]
