"""Install the MI355X-native implementations into an importable `colpali_engine`.

    import colpali_amd; colpali_amd.patch_colpali_engine()

After the call
  * `BaseVisualRetrieverProcessor.score_multi_vector` (inherited by every Col*Processor and reached
    through `processor.score(...)`, e.g. models/paligemma/colpali/processing_colpali.py:96-106)
    is colpali_amd.score_multi_vector;
  * `colpali_engine.loss.ColbertPairwiseCELoss` / `ColbertLoss` / `ColbertSigmoidLoss`
    (and the same names in `colpali_engine.loss.late_interaction_losses`) are the fused versions, so
    YAML configs that name the class by dotted path (scripts/configs/qwen2/train_colqwen2_model.yaml:24-25)
    pick them up.
`unpatch_colpali_engine()` restores the originals.
"""
from __future__ import annotations

import importlib
import sys

from . import loss as _loss
from .scoring import score_multi_vector

_LOSS_NAMES = ("ColbertPairwiseCELoss", "ColbertLoss", "ColbertSigmoidLoss", "ColbertNegativeCELoss",
               "ColbertPairwiseNegativeCELoss")
_saved = {}


def patch_colpali_engine(scorer: bool = True, losses: bool = True) -> None:
    if scorer:
        pu = importlib.import_module("colpali_engine.utils.processing_utils")
        cls = pu.BaseVisualRetrieverProcessor
        _saved.setdefault("score_multi_vector", cls.__dict__["score_multi_vector"])
        cls.score_multi_vector = staticmethod(score_multi_vector)
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
    for key, obj in list(_saved.items()):
        if key == "score_multi_vector":
            pu = importlib.import_module("colpali_engine.utils.processing_utils")
            pu.BaseVisualRetrieverProcessor.score_multi_vector = obj
        else:
            setattr(sys.modules[key[0]], key[1], obj)
        del _saved[key]
