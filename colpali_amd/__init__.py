"""colpali_amd -- MI355X (gfx950) native late-interaction scorer.

Drop-in for the MaxSim hot path of illuin-tech/colpali:
  * score_multi_vector            <- BaseVisualRetrieverProcessor.score_multi_vector
                                     (colpali_engine/utils/processing_utils.py:132-187)
  * ColbertPairwiseCELoss (+ ColbertLoss, ColbertSigmoidLoss, ColbertModule)
                                  <- colpali_engine/loss/late_interaction_losses.py:255-313 (:110-164, :401-465, :6-107)
  * ShardedRetriever / topk       -- sharded-corpus top-k with an RCCL all-gather merge (no reference equivalent)
  * embedding_head / CorpusWriter <- the projection / L2-norm / mask tail of every Col* forward
                                     (models/paligemma/colpali/modeling_colpali.py:67-77), writing the packed corpus
The compute lives in hand-written HIP kernels behind a C ABI (include/maxsim.h,
colpali_amd/csrc/); this package is the thin host-side mirror of the reference interface.
"""
from .corpus import PackedCorpus, PackedQueries, block_clamp0, pack_passages, pack_queries
from . import loss
from .embed import CorpusWriter, embedding_head
from .loss import (ColbertLoss, ColbertModule, ColbertNegativeCELoss, ColbertPairwiseCELoss,
                   ColbertPairwiseNegativeCELoss, ColbertSigmoidLoss, maxsim, maxsim_paired)
from .pooling import HierarchicalTokenPooler, TokenPoolingOutput
from .patch import patch_colpali_engine, unpatch_colpali_engine
from .retrieval import (ExactMaxSimIndex, ShardedRetriever, create_plaid_index, get_topk_plaid, merge_gathered, shard_range,
                        shard_topk, topk)
from .scoring import (get_similarity_maps_from_embeddings, get_torch_device, maxsim_scores, score_multi_vector,
                      score_single_vector, similarity_matrix)

__all__ = [
    "CorpusWriter",
    "HierarchicalTokenPooler",
    "TokenPoolingOutput",
    "embedding_head",
    "ColbertLoss",
    "ColbertModule",
    "ColbertNegativeCELoss",
    "ColbertPairwiseNegativeCELoss",
    "maxsim_paired",
    "ColbertPairwiseCELoss",
    "ColbertSigmoidLoss",
    "PackedCorpus",
    "PackedQueries",
    "maxsim",
    "ShardedRetriever",
    "ExactMaxSimIndex",
    "create_plaid_index",
    "get_topk_plaid",
    "merge_gathered",
    "shard_range",
    "shard_topk",
    "topk",
    "block_clamp0",
    "get_torch_device",
    "maxsim_scores",
    "pack_passages",
    "patch_colpali_engine",
    "unpatch_colpali_engine",
    "pack_queries",
    "score_multi_vector",
    "score_single_vector",
    "similarity_matrix",
    "get_similarity_maps_from_embeddings",
]
