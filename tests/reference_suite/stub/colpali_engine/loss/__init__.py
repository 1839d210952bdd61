"""colpali_engine.loss as patch_colpali_engine() leaves it (colpali_engine/loss/__init__.py re-exports late_interaction_losses)."""
from colpali_amd import (ColbertLoss, ColbertModule, ColbertNegativeCELoss, ColbertPairwiseCELoss,  # noqa: F401
                         ColbertPairwiseNegativeCELoss, ColbertSigmoidLoss)
