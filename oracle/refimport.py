"""Import the live reference's hot-path modules in THIS build container.

TEST INFRASTRUCTURE ONLY, and only usable where /root/reference exists (it does
not exist on the GPU box).  `colpali_engine/__init__.py` pulls every model
family (and through them torchvision, peft ...), which are not installed, so a
stub top-level package is registered and only the two modules on the hot path
are imported: utils/processing_utils.py and loss/late_interaction_losses.py.
Used by tests/golden/make_golden.py to generate the committed fixtures and by
tests/test_reference_live.py (skipped when the checkout is absent).
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_ROOT = os.environ.get("COLPALI_REFERENCE_ROOT", "/root/reference")
# where oracle/fetch_reference_tests.py leaves verbatim, git-ignored copies of the reference's model files (they travel to the
# GPU box with the working tree, like the reference's own tests under tests/_reference_tests/)
FETCHED_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests", "_reference_pkg"))


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "colpali_engine"))


def load():
    """Returns (BaseVisualRetrieverProcessor, late_interaction_losses module)."""
    if not available():
        raise RuntimeError(f"reference checkout not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True
    if "colpali_engine" not in sys.modules:
        pkg = types.ModuleType("colpali_engine")
        pkg.__path__ = [os.path.join(REFERENCE_ROOT, "colpali_engine")]
        sys.modules["colpali_engine"] = pkg
    from colpali_engine.loss import late_interaction_losses  # noqa: E402
    from colpali_engine.utils.processing_utils import BaseVisualRetrieverProcessor  # noqa: E402

    return BaseVisualRetrieverProcessor, late_interaction_losses


def load_hot_path(allow_fetched: bool = True):
    """(BaseVisualRetrieverProcessor, late_interaction_losses module, "live" | "fetched") from the reference checkout where it exists,
    else from the verbatim git-ignored copies oracle/fetch_reference_tests.py leaves under tests/_reference_pkg/ (the GPU box).
    bench.py times THESE as `cpu_baseline.kind = "reference"` and as the reference loss module on the same GPU."""
    if available():
        proc, losses = load()
        return proc, losses, "live"
    have = all(os.path.exists(os.path.join(FETCHED_ROOT, "colpali_engine", rel)) for rel in
               ("loss/late_interaction_losses.py", "utils/processing_utils.py", "utils/torch_utils.py"))
    if not (allow_fetched and have):
        raise RuntimeError("no reference hot-path files: neither the checkout nor tests/_reference_pkg/ (oracle/fetch_reference_tests.py)")
    sys.dont_write_bytecode = True
    for name, sub in (("colpali_engine", ""), ("colpali_engine.loss", "loss"), ("colpali_engine.utils", "utils")):
        if name not in sys.modules:
            pkg = types.ModuleType(name)
            pkg.__path__ = [os.path.join(FETCHED_ROOT, "colpali_engine", sub)]
            sys.modules[name] = pkg
    from colpali_engine.loss import late_interaction_losses  # noqa: E402
    from colpali_engine.utils.processing_utils import BaseVisualRetrieverProcessor  # noqa: E402

    return BaseVisualRetrieverProcessor, late_interaction_losses, "fetched"


def load_colpali_class():
    """The live reference's ColPali model class (random-init use only: no weights exist here).  The model
    sub-packages' __init__ files import every family, so they are stubbed the same way as the top-level package."""
    if not available():
        raise RuntimeError(f"reference checkout not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True
    for name, sub in (("colpali_engine", ""), ("colpali_engine.models", "models"),
                      ("colpali_engine.models.paligemma", "models/paligemma"),
                      ("colpali_engine.models.paligemma.colpali", "models/paligemma/colpali")):
        if name not in sys.modules:
            pkg = types.ModuleType(name)
            pkg.__path__ = [os.path.join(REFERENCE_ROOT, "colpali_engine", sub)]
            sys.modules[name] = pkg
    from colpali_engine.models.paligemma.colpali.modeling_colpali import ColPali  # noqa: E402

    return ColPali


def model_root() -> str:
    """The directory holding `colpali_engine/models/...`: the reference checkout here, the fetched copy on the GPU box."""
    if available():
        return REFERENCE_ROOT
    if os.path.isdir(os.path.join(FETCHED_ROOT, "colpali_engine", "models")):
        return FETCHED_ROOT
    raise RuntimeError("no reference model files: neither the checkout nor tests/_reference_pkg/ (oracle/fetch_reference_tests.py)")


def load_model_class(rel_module: str, cls_name: str):
    """A Col* model class of the (live or fetched) reference, e.g. ("models/qwen2/colqwen2/modeling_colqwen2", "ColQwen2").
    Random-init use only.  Every package on the way is a stub (the real __init__ files import every family and, through
    them, dependencies that are not installed)."""
    root = model_root()
    sys.dont_write_bytecode = True
    parts = rel_module.split("/")
    for i in range(len(parts)):
        name = ".".join(["colpali_engine"] + parts[:i])
        if name not in sys.modules:
            pkg = types.ModuleType(name)
            pkg.__path__ = [os.path.join(root, "colpali_engine", *parts[:i])]
            sys.modules[name] = pkg
    import importlib

    mod = importlib.import_module(".".join(["colpali_engine"] + parts))
    return getattr(mod, cls_name)


def load_similarity_map_utils():
    """colpali_engine/interpretability/similarity_map_utils.py of the live reference (the package __init__ pulls the
    plotting helpers and seaborn, which is not installed: stub the package, import the one module)."""
    if not available():
        raise RuntimeError(f"reference checkout not found at {REFERENCE_ROOT}")
    load()
    name = "colpali_engine.interpretability"
    if name not in sys.modules:
        pkg = types.ModuleType(name)
        pkg.__path__ = [os.path.join(REFERENCE_ROOT, "colpali_engine", "interpretability")]
        sys.modules[name] = pkg
    from colpali_engine.interpretability import similarity_map_utils  # noqa: E402

    return similarity_map_utils


def load_token_pooler():
    """HierarchicalTokenPooler of the live reference (compression/token_pooling/hierarchical_token_pooling.py)."""
    if not available():
        raise RuntimeError(f"reference checkout not found at {REFERENCE_ROOT}")
    load()
    for name, sub in (("colpali_engine.compression", "compression"),
                      ("colpali_engine.compression.token_pooling", "compression/token_pooling")):
        if name not in sys.modules:
            pkg = types.ModuleType(name)
            pkg.__path__ = [os.path.join(REFERENCE_ROOT, "colpali_engine", sub)]
            sys.modules[name] = pkg
    from colpali_engine.compression.token_pooling.hierarchical_token_pooling import HierarchicalTokenPooler  # noqa: E402

    return HierarchicalTokenPooler
