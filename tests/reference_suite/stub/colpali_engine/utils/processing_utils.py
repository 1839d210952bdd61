"""`BaseVisualRetrieverProcessor` with exactly what colpali_amd.patch_colpali_engine() installs on the real class
(colpali_engine/utils/processing_utils.py:103-187): the two static scorers."""
import colpali_amd


class BaseVisualRetrieverProcessor:
    score_single_vector = staticmethod(colpali_amd.score_single_vector)
    score_multi_vector = staticmethod(colpali_amd.score_multi_vector)
