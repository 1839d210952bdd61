/*
 * include/maxsim.h -- C ABI of libmaxsim_gfx950.so, the MI355X-native MaxSim
 * (ColBERT late-interaction) scorer.
 *
 * The reference (illuin-tech/colpali) has no FFI layer: its hot path is two
 * Python call sites that delegate to torch.  These entry points are what a
 * binding for that path binds instead; each cites the reference lines it
 * replaces (paths relative to the reference checkout).  INTEGRATION.md shows
 * the ctypes stub a maintainer adds on the reference side.
 *
 * Conventions (all entry points)
 *   - plain C types only; device buffers are raw device pointers; `stream` is a
 *     hipStream_t passed as void* (NULL = the null stream);
 *   - the caller owns every buffer: nothing is allocated, freed or synchronised
 *     inside a call, so every call is asynchronous on `stream` and can be
 *     captured in a hipGraph;
 *   - return 0 on success, a negative MSIM_E* code otherwise (never throws,
 *     never aborts); msim_last_error() gives a thread-local message;
 *   - matrices are row-major and contiguous, embeddings are 16-byte aligned.
 *
 * Data layout ("packed corpus")
 *   D        bf16|f16|f32 [total_rows, dim]  every document's patch embeddings, back to back
 *   d_off    int32 [n_d + 1]         document c owns rows d_off[c] .. d_off[c+1]-1
 *   d_clamp0 uint8 [n_d] or NULL     1 = the reference would have zero-padded this
 *                                    document inside its passage block, so a
 *                                    similarity of exactly 0 also takes part in
 *                                    every per-token max
 *                                    (colpali_engine/utils/processing_utils.py:175-178,
 *                                     pad_sequence(..., padding_value=0))
 *   Q        bf16|f16|f32 [n_q, Lq, dim]     queries (same dtype as D), zero rows = padding (they add 0)
 *
 * Flat query layout (msim_fwd_ragged; bf16 / f16, dim 128 or 320) -- queries are ragged in real use
 * (colpali_engine/utils/processing_utils.py:86 appends 10 augmentation tokens to a question of any length; a batch is
 * padded to its longest member, colpali_engine/collators/visual_retriever_collator.py:82-85) and a zero row adds exactly
 * 0 to every score, so the kernels take the real tokens only:
 *   Qt       bf16|f16 [T, dim]       every query's tokens back to back
 *   q_off    int32 [n_q + 1]         query q owns tokens q_off[q] .. q_off[q+1]-1 (on the device AND on the host: the
 *                                    launch plan -- which whole queries share a workgroup -- is made on the host)
 */
#ifndef COLPALI_AMD_MAXSIM_H
#define COLPALI_AMD_MAXSIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSIM_ABI_VERSION 20

/* error codes */
#define MSIM_OK 0
#define MSIM_EINVAL (-1)      /* bad argument (null pointer, negative size, misalignment) */
#define MSIM_EUNSUPPORTED (-2) /* shape/dtype outside what the gfx950 kernels implement */
#define MSIM_ELAUNCH (-3)      /* HIP reported an error at launch/configuration time */

/* embedding element types (`dtype` argument), fed to the MFMA as is, fp32 accumulate.
 * bf16 / f16 with dim == 128 and Lq <= 128 take the tuned kernels; every other combination (fp32 embeddings,
 * other widths such as ColQwen3's 320, longer queries) takes the generic kernels, which need
 * dim * sizeof(element) to be a multiple of 32 bytes (pad the width with zero columns) and <= 4096 bytes. */
#define MSIM_DTYPE_BF16 0
#define MSIM_DTYPE_F16 1
#define MSIM_DTYPE_F32 2

/* flags for msim_fwd */
#define MSIM_FLAG_REF_ROUNDING 0x1u /* reproduce the rounding the reference applies when torch computes the
                                       contraction in the embeddings' own dtype: every similarity rounded to
                                       that dtype before the max, the token sum rounded to it
                                       (processing_utils.py:179 evaluated on bf16 / fp16 tensors) */

#define MSIM_FLAG_AVG_ROWS(n) ((uint32_t)((n) < 0 ? 0 : ((n) > 65535 ? 65535 : (n))) << 8)
                                    /* optional launch-shape hint for msim_fwd / msim_fwd_ragged: the average number of rows per document
                                       (the row offsets live on the device; the launch plan is made on the host).  Only the number of
                                       document ranges the corpus is cut into depends on it -- speed, never a score.  0 = unknown. */

int msim_abi_version(void);
const char *msim_last_error(void);

/* Number of bytes of scratch msim_fwd can use for this problem (16-byte aligned device memory, contents irrelevant: the
 * call initialises what it uses on the stream).  Non-zero exactly when the launch plan holds several query blocks (bf16 / f16,
 * width 128: more than 1280 query tokens or 64 queries in total -- a block takes whole queries, up to 1024 or 1280 tokens): the
 * workgroups of an XCD that stream the same document range for different blocks then keep in step through progress counters, so
 * that the range is fetched from HBM once and served to the others from that XCD's L2.
 * Queries longer than 128 tokens (bf16 / f16, width 128: pages as queries, the trainer's symmetric direction) are scored as
 * 128-token pieces on the tuned kernels -- MaxSim is a sum over query tokens -- and need room for the partial sums:
 * 4096 + n_q * ceil(Lq / 128) * n_d * 4 bytes.
 * Passing NULL is legal and only switches those off (no convoy; long queries take the generic kernels: the same scores up to fp32
 * summation order, 3-4 x slower on a large corpus); a non-NULL workspace must hold at least the number of bytes this function
 * reports for the same problem.  One workspace per launch in flight. */
size_t msim_fwd_workspace_bytes(int dtype, int n_q, int Lq, int n_d, int dim);

/*
 * scores[q, c] = sum_{i < Lq} max_{j in doc c} <Q[q,i,:], D[j,:]>      (fp32 accumulate)
 *
 * Replaces the arithmetic of
 *   colpali_engine/utils/processing_utils.py:179
 *       torch.einsum("bnd,csd->bcns", qs_batch, ps_batch).max(dim=3)[0].sum(dim=2)
 *   colpali_engine/loss/late_interaction_losses.py:297-298 (+ :91)
 *       torch.einsum("bnd,csd->bcns", q, d) -> amax(dim=3) -> sum(dim=2)
 * without materialising the [b, c, n, s] similarity tensor.
 *
 * scores is fp32 [n_q, ld_scores] (ld_scores >= n_d).
 */
int msim_fwd(int dtype, const void *Q, int n_q, int Lq,
                  const void *D, const int32_t *d_off, const uint8_t *d_clamp0,
                  int n_d, int dim,
                  float *scores, int64_t ld_scores,
                  uint32_t flags, void *workspace, void *stream);

/*
 * The same scores on the HOST cores: every pointer is host memory, the call is synchronous and runs on `n_threads` native
 * threads (AVX-512 / AVX2 / baseline clones picked at load time).  This is what colpali_amd.score_multi_vector runs when the
 * caller names device="cpu" -- or passes no device on a host without a GPU -- through the reference's signature
 *   colpali_engine/utils/processing_utils.py:132-187 (the reference scores on whatever device it is given: :161, :172-179;
 *   colpali_engine/utils/torch_utils.py:12-31 answers "cpu" when no accelerator is visible);
 * it is never used on behalf of a GPU request.  fp32 products and sums of the exactly widened inputs (the kernels' truth tier:
 * within 1e-5 of a float64 evaluation); MSIM_FLAG_REF_ROUNDING as in msim_fwd.  Any dim >= 1, any Lq >= 0; bf16 / f16 / f32.
 * msim_host_last_error() gives its thread-local message.
 */
const char *msim_host_last_error(void);
int msim_fwd_host(int dtype, const void *Q, int n_q, int Lq,
                  const void *D, const int32_t *d_off, const uint8_t *d_clamp0,
                  int n_d, int dim,
                  float *scores, int64_t ld_scores,
                  uint32_t flags, int n_threads);
/* The same without any packing: query q = q_rows[q] rows of `dim` elements at q_ptr[q], document c = d_rows[c] rows at d_ptr[c] -- the
 * caller's own host tensors (a list of ragged queries and passages is what the reference's callers hold: README.md:121-126); nothing
 * is copied, and only the real tokens of a ragged query are multiplied. */
int msim_fwd_host_lists(int dtype, const void *const *q_ptr, const int64_t *q_rows, int n_q,
                        const void *const *d_ptr, const int64_t *d_rows, const uint8_t *d_clamp0,
                        int n_d, int dim,
                        float *scores, int64_t ld_scores,
                        uint32_t flags, int n_threads);
/* out[i, j] = <A_i, B_j> on the host cores (score_single_vector with device="cpu": processing_utils.py:103-130) */
int msim_sim_matrix_host(int dtype, const void *A, int n_a, const void *B, int n_b, int dim,
                         float *out, int64_t ld_out, uint32_t flags, int n_threads);

/*
 * The same scores for RAGGED queries in the flat layout: scores[q, c] = sum over the tokens i of query q of
 * max_{j in doc c} <Qt[q_off[q] + i, :], D[j, :]>.  A query's score is a pure function of its own tokens and the
 * document (the token sum runs in an order fixed by the query's length alone), so it does not depend on the batch it is
 * scored in, on the kernel shape the plan picks, or on whether msim_fwd or msim_fwd_ragged computed it (for queries of up
 * to 128 tokens).  bf16 / f16 embeddings of width 128, a query of 0 .. 1280 tokens -- or of width 320 (ColQwen3,
 * models/qwen3/colqwen3/modeling_colqwen3.py:48: the flat panel kernel K1bPF), a query of 0 .. 512 tokens; MSIM_EUNSUPPORTED otherwise:
 * pad the queries to one length and call msim_fwd.  At width 320 a call of ONE query length and at most four 32-token tiles in all is
 * streamed by K1sP, whose token sum is a butterfly: there the bits depend on which kernel ran (values agree to fp32 summation
 * order); every other width-320 call of THIS entry is batch-independent like width 128.  (msim_fwd on a width-320 box whose Lq is a
 * multiple of 32 runs K1sP / K1bP -- butterfly sums as well: equal to this entry's result up to fp32 summation order, not bit for bit;
 * any other Lq <= 512 runs the flat kernel and returns this entry's bits.)  `q_off` is the device copy, `q_off_host` the host copy of the same
 * n_q + 1 numbers (read during the call only).  Workspace as msim_fwd's: msim_fwd_ragged_workspace_bytes() bytes or NULL.
 * Replaces the same reference lines as msim_fwd (processing_utils.py:172-179 with its pad_sequence of the query block).
 */
/* Which kernel shape a tuned forward call (bf16 / f16, width 128) takes for these queries -- host-only, no device work; the same
 * plan msim_fwd / msim_fwd_ragged make.  q_off_host = the n_q + 1 token offsets, or NULL for n_q uniform queries of Lq tokens
 * (more than 128 tokens: the 128-token pieces msim_fwd scores with a workspace).  out5 = { 0 = K1s (every wave holds all units:
 * HBM-bound regime) | 1 = K1b (waves share a document stream),  K1s: 16-token units per wave | K1b: waves per stream (2, 4, 8),
 * K1b: units a wave holds at most (8, 10),  query blocks (passes over a document range),  units of the heaviest wave }. */
int msim_fwd_plan(const int32_t *q_off_host, int n_q, int Lq, int32_t *out5);
size_t msim_fwd_ragged_workspace_bytes(int dtype, const int32_t *q_off_host, int n_q, int n_d, int dim);
int msim_fwd_ragged(int dtype, const void *Qt, const int32_t *q_off, const int32_t *q_off_host, int n_q,
                    const void *D, const int32_t *d_off, const uint8_t *d_clamp0,
                    int n_d, int dim,
                    float *scores, int64_t ld_scores,
                    uint32_t flags, void *workspace, void *stream);

/*
 * The same scores for two DENSE BOXES when the queries are long and the documents short -- the symmetric direction of the reference
 * trainer (trainer/contrastive_trainer.py:202-206, compute_symetric_loss: the pages as query_embeddings [B, 780, 128], the gathered
 * queries as doc_embeddings [C, 32, 128]; late_interaction_losses.py:297-298):
 *     scores[q, c] = sum_{i < Lq} max_{j < Ld} <Q[q, i, :], D[c, j, :]>
 * Q [n_q, Lq, 128], D [n_d, Ld, 128] (bf16 | f16, Ld <= 128; every row of a box takes part, zero padding rows included, as in the
 * reference), scores fp32 [n_q, ld_scores].  The long side streams, the short side is resident (kernel K1t, maxsim_batch_t.hip):
 * msim_fwd gives the same numbers up to fp32 summation order, 2-3 x slower on this shape (a 32-row document is one slab).
 * q_lengths: NULL, or int32 [n_q] that receives, as a by-product of streaming every query row, the number of rows of query q whose
 * first component is non-zero -- the `lengths` of late_interaction_losses.py:296, which msim_loss_epilogue accepts ready-made.
 */
int msim_fwd_transposed(int dtype, const void *Q, int n_q, int Lq, const void *D, int n_d, int Ld, int dim,
                        float *scores, int64_t ld_scores, int32_t *q_lengths, void *stream);

/*
 * The DENSE hard-max backward of that shape on the matrix cores (round 6; maxsim_dense_t.hip) -- what autograd derives for
 * late_interaction_losses.py:297-298 when EVERY (query, document) pair carries a gradient: ColbertLoss (:140-164, the trainer's
 * default loss, trainer/colmodel_training.py:33) and ColbertSigmoidLoss (:440-465) in the trainer's symmetric direction
 * (trainer/contrastive_trainer.py:202-206).  bf16 | f16, width 128, documents of at most 64 rows (msim_dense_t_supported; every
 * other shape keeps msim_pairs_bwd).
 *   msim_fwd_transposed_route   msim_fwd_transposed (bit-identical scores) that also leaves the ROUTING: route uint8
 *                               [n_q, n_d, Lq_pad] (Lq_pad = Lq rounded up to 64; msim_dense_t_route_bytes), route[q, c, i] = the row of
 *                               document c that won the max for row i of query q (the first maximal row on a tie); bytes of rows i >= Lq are unspecified
 *                               (they never reach a result: those rows of the query image are zero and no gradient is stored for them).
 *   msim_dense_t_bwd            dQ[q, i, :] = sum_c G[q, c] * g_scale * D[c, route[q, c, i], :]
 *                               dD[c, s, :] = sum_q sum_{i : route[q, c, i] = s} G[q, c] * g_scale * Q[q, i, :]
 *                               as two GEMMs against a G-scaled one-hot operand built in registers (v_mfma_f32_16x16x32), G rounded to
 *                               the embeddings' dtype per term, fp32 sums in a fixed order (no float atomics), dQ / dD written in the
 *                               embeddings' dtype.  G fp32 [n_q, ldg]; g_scale: NULL or one device scalar (dtype code g_scale_dtype).
 *                               workspace: msim_dense_t_bwd_workspace_bytes bytes, 16-byte aligned, contents irrelevant.
 */
int msim_dense_t_supported(int dtype, int n_q, int Lq, int n_d, int Ld, int dim);
size_t msim_dense_t_route_bytes(int n_q, int Lq, int n_d);
int msim_fwd_transposed_route(int dtype, const void *Q, int n_q, int Lq, const void *D, int n_d, int Ld, int dim,
                              float *scores, int64_t ld_scores, int32_t *q_lengths, uint8_t *route, void *stream);
size_t msim_dense_t_bwd_workspace_bytes(int n_q, int Lq, int n_d, int Ld, int dim);
int msim_dense_t_bwd(int dtype, const void *Q, int n_q, int Lq, const void *D, int n_d, int Ld, int dim,
                     const float *G, int64_t ldg, const void *g_scale, int g_scale_dtype, const uint8_t *route,
                     void *dQ, void *dD, void *workspace, void *stream);

/*
 * Packing queries into the flat layout: rows that are entirely zero (the model's padded positions,
 * modeling_colpali.py:72 / modeling_colqwen2.py:69; byte-wise test) are dropped, the others keep their order.
 *   msim_query_compact        device: box [n_q, Lq, row_bytes] (row_bytes a multiple of 16, Lq <= 4096).  counts != NULL:
 *                             counts[q] = rows of query q that are not all-zero.  out != NULL: those rows are copied to rows
 *                             q_off[q] .. of `out`.  (Call once for the counts, build q_off, call again to copy.)
 *   msim_host_count_nonzero_rows / msim_host_gather_nonzero_rows
 *                             host (native threads; synchronous): the same for a list of host buffers src[i] of rows[i] rows --
 *                             counts, then the copy of source i's non-zero rows to row dst_row[i] .. of `dst` (pinned staging).
 * Both forms are two passes over the same data (count, then copy to offsets built from the counts): the sources must not change in
 * between -- the copy trusts the offsets.
 */
int msim_query_compact(const void *box, int n_q, int Lq, int row_bytes, const int32_t *q_off, int32_t *counts, void *out,
                       void *stream);
int msim_host_count_nonzero_rows(const void *const *src, const int64_t *rows, int64_t row_bytes, int64_t n, int32_t *counts,
                                 int n_threads);
int msim_host_gather_nonzero_rows(void *dst, const void *const *src, const int64_t *rows, int64_t row_bytes,
                                  const int64_t *dst_row, int64_t n, int n_threads);

/*
 * MaxSim for an explicit list of (query, document) pairs, optionally reporting for every
 * (pair, query token) the document row (relative to the document) that attains the max;
 * -1 when the reference's zero padding row wins (d_clamp0).  First maximum wins on ties.
 * This is the routing autograd derives for amax in
 *   colpali_engine/loss/late_interaction_losses.py:298 -> :91 (scores_raw.amax(dim=dim_max)),
 * and the forward of the paired contractions "bnd,bsd->bns" / "bnd,blsd->blns" (:235-238, :381-384).
 * pairs: int32 [n_pairs, 2] = (query index, document index).
 * out_scores: fp32 [n_pairs] or NULL; out_argmax: int32 [n_pairs, Lq] or NULL.
 * max_doc_rows: an upper bound of the longest document's rows, or 0 = unknown.  It only selects the kernel: queries of more than
 * 128 tokens against documents of at most 128 rows (bf16 / f16, dim 128: the trainer's symmetric direction,
 * trainer/contrastive_trainer.py:202-206 -- pages as query_embeddings, queries as doc_embeddings) take the transposed pair
 * kernel (the long side streams, the short side is resident); with 0 they take the generic kernel (same results up to ties'
 * first-maximum rule, which both follow).  Short lists (<= 1024 pairs) are worked on by one workgroup per pair, long ones by one
 * wave per pair.
 */
int msim_pairs_argmax(int dtype, const void *Q, int n_q, int Lq,
                      const void *D, const int32_t *d_off, const uint8_t *d_clamp0,
                      int n_d, int dim, int max_doc_rows,
                      const int32_t *pairs, int n_pairs,
                      float *out_scores, int32_t *out_argmax, void *stream);

/*
 * The same for ALL n_q x n_d pairs (the row-major all-pairs list without the list): scores [n_q, ld_scores] and
 * out_argmax [(q * n_d + c), Lq] -- the forward of the in-batch losses whose upstream gradient is dense (ColbertLoss :152-164,
 * ColbertSigmoidLoss :444-465), which keep the routing for the backward.  A wave scores up to four queries (<= 128 tokens in all)
 * against one document, so a document is streamed once per GROUP of queries instead of once per pair.  bf16 / f16, dim 128, Lq <= 128
 * (MSIM_EUNSUPPORTED otherwise: list the pairs and call msim_pairs_argmax).  Either output may be NULL.
 */
int msim_allpairs_argmax(int dtype, const void *Q, int n_q, int Lq,
                         const void *D, const int32_t *d_off, const uint8_t *d_clamp0, int n_d, int dim,
                         float *out_scores, int64_t ld_scores, int32_t *out_argmax, void *stream);

/*
 * Backward of the contraction for a sparse set of (q, c) pairs with upstream gradient
 * g[p] = dLoss/dscores[q_p, c_p]:
 *     dQ[q, i, :]        = sum_p g[p] * D[d_off[c_p] + argmax[p, i], :]
 *     dD[d_off[c]+r, :]  = sum_{p, i : c_p = c, argmax[p, i] = r} g[p] * Q[q_p, i, :]
 * i.e. what autograd produces for einsum -> amax -> sum
 * (late_interaction_losses.py:297-298) restricted to the pairs whose upstream gradient is
 * non-zero (for ColbertPairwiseCELoss, :309-313, two per query).
 * dQ [n_q, Lq, dim] and dD [total_rows, dim] are fully overwritten (rows without a contribution are set to 0), as fp32
 * (out_dtype = MSIM_DTYPE_F32) or in the embeddings' own dtype (out_dtype = dtype: one rounding of the fp32 sum -- what autograd's
 * `.to(dtype)` would do in a launch of its own).  Deterministic: no floating-point atomics.
 * g_scale: NULL, or a DEVICE scalar (g_scale_dtype: bf16 / f16 / f32) every g[p] is multiplied by -- the loss' upstream gradient as
 * autograd hands it over (a 0-dim tensor), folded into the kernels instead of a `coef * grad` launch in front of them.
 * `pairs` must be sorted by query index; `order_by_doc` is a permutation of 0..n_pairs-1 that
 * sorts the pairs by document index (stable); `max_doc_rows` >= the longest document.
 * `workspace`: msim_pairs_bwd_workspace_bytes() bytes of 16-byte aligned device scratch (contents irrelevant), or NULL.  It is
 * non-zero when short documents (<= 64 rows) collect long entry lists -- the symmetric direction of the reference trainer
 * (trainer/contrastive_trainer.py:202-206: pages as query_embeddings, queries as doc_embeddings), dense upstream gradients or
 * queries of 256 tokens and more: every document's pair list
 * is then split over several workgroups whose partial sums are added in split order (still deterministic, still no atomics).
 * NULL is legal and only selects the one-workgroup-per-row-range form (the same values up to fp32 summation order).
 */
size_t msim_pairs_bwd_workspace_bytes(int n_q, int Lq, int n_d, int dim, int max_doc_rows, int n_pairs);
int msim_pairs_bwd(int dtype, const void *Q, int n_q, int Lq,
                   const void *D, const int32_t *d_off, int n_d, int dim, int max_doc_rows,
                   const int32_t *pairs, const int32_t *order_by_doc,
                   const float *g, const void *g_scale, int g_scale_dtype,
                   const int32_t *argmax, int n_pairs,
                   int out_dtype, void *dQ, void *dD, void *workspace, void *stream);

/*
 * Smooth-max late interaction (training losses constructed with use_smooth_max=True):
 *     scores[q, c] = sum_{i < Lq} tau * log sum_{j in doc c} exp(<Q[q,i,:], D[j,:]> / tau)
 * Replaces colpali_engine/loss/late_interaction_losses.py:40-44 (_smooth_max = tau * logsumexp(scores / tau)) applied
 * through :88-90 (_aggregate) to the einsum of :153 / :297 / :444, again without the [b, c, n, s] tensor.  Every row of a
 * document takes part (physical zero padding rows contribute exp(0), as in the reference); there is no d_clamp0 here
 * because score_multi_vector has no smooth-max mode.  dim * sizeof(element) must be a multiple of 32 bytes, <= 4096.
 *
 * msim_smooth_pairs: the same for an explicit pair list; out_lse [n_pairs, Lq] receives the natural-log
 *     lse[p, i] = log sum_j exp(<Q[q_p,i], D[j]> / tau)     (what the backward needs; either output may be NULL).
 * msim_smooth_pairs_bwd: autograd of the expression for the listed pairs with upstream gradient g[p]:
 *     w[p,i,j] = exp(<Q[q_p,i], D[j]> / tau - lse[p,i])     (softmax over the document's rows)
 *     dQ[q, i, :]       = sum_{p: q_p = q} g[p] * sum_j w[p,i,j] * D[d_off[c_p] + j, :]
 *     dD[d_off[c]+j, :] = sum_{p: c_p = c} g[p] * sum_i w[p,i,j] * Q[q_p, i, :]
 * Same conventions as msim_pairs_bwd (pairs sorted by query, order_by_doc, full overwrite, deterministic, no atomics).
 * The dQ pass splits a query's pair list over several workgroups and sums their partial results in a fixed order: it
 * needs msim_smooth_bwd_workspace_bytes(n_q, Lq, dim) bytes of scratch (16-byte aligned; 0 = none needed).
 */
int msim_smooth_fwd(int dtype, const void *Q, int n_q, int Lq,
                    const void *D, const int32_t *d_off, int n_d, int dim, float tau,
                    float *scores, int64_t ld_scores, void *stream);
int msim_smooth_pairs(int dtype, const void *Q, int n_q, int Lq,
                      const void *D, const int32_t *d_off, int n_d, int dim,
                      const int32_t *pairs, int n_pairs, float tau,
                      float *out_scores, float *out_lse, void *stream);
size_t msim_smooth_bwd_workspace_bytes(int n_q, int Lq, int dim);
int msim_smooth_pairs_bwd(int dtype, const void *Q, int n_q, int Lq,
                          const void *D, const int32_t *d_off, int n_d, int dim, int max_doc_rows,
                          const int32_t *pairs, const int32_t *order_by_doc,
                          const float *g, const float *lse, int n_pairs, float tau,
                          float *dQ, float *dD, void *workspace, void *stream);

/*
 * The [B, C]-sized epilogue of the in-batch losses, value and gradient with respect to the MaxSim scores in one launch
 * (no host synchronisation, hipGraph-capturable).  Replaces, for scores = the raw MaxSim matrix of msim_fwd / msim_pairs_argmax,
 *   colpali_engine/loss/late_interaction_losses.py:296 (lengths), :300-301 (normalisation), :303-307 (pos-aware filtering) and
 *   MSIM_LOSS_PAIRWISE  :309-313  softplus((hardest in-batch negative - positive) / T).mean()     (ColbertPairwiseCELoss)
 *   MSIM_LOSS_INFONCE   :164      cross_entropy(scores / T, pos_idx)                                (ColbertLoss)
 *   MSIM_LOSS_SIGMOID   :457-465  softplus(-scores.view(-1) / T * sign).mean(), sign +1 on the diagonal of the in-batch square,
 *                                 -1 elsewhere (C == B, offset == 0)                                (ColbertSigmoidLoss; round 6)
 * together with what autograd derives for them:
 *   PAIRWISE  pairs int32 [2B, 2] = the two (query, doc) entries per query that carry a gradient (positive, selected negative),
 *             sorted by (query, doc); coef fp32 [2B] = dLoss/dscore of each for a unit upstream gradient; order int32 [2B] = the
 *             stable by-document permutation msim_pairs_bwd wants.  Always exactly 2B pairs: nothing to read back.  C >= 2.
 *   INFONCE, SIGMOID   G fp32 [B, ld] = dLoss/dscores (dense).
 * Q [B, Lq, width] are the query embeddings (q_dtype as in msim_fwd): lengths[b] = number of tokens whose first component is
 * non-zero.  out fp32 [3] = loss, min and max of the normalised scores (the reference prints when they leave [-tol, 1+tol]).
 * q_lengths: NULL (the token counts are taken from Q here), or int32 [B] = lengths[b] ready-made (msim_fwd_transposed's by-product:
 * for 780-token "queries" -- the trainer's symmetric direction -- counting them here means one cache line per token on one CU).
 * loss_out: NULL, or one element of q_dtype that receives the loss rounded to the embeddings' dtype (what the reference's forward
 * returns: bf16 in -> bf16 scalar) -- no cast launch behind the kernel.
 * workspace: msim_loss_epilogue_workspace_bytes(B, C) bytes, 16-byte aligned, ZERO-FILLED once by the caller before its first use
 * (the call leaves it ready for the next one); one workspace per stream.  0 bytes (workspace may be NULL) for small batches --
 * B <= 1024 and B * C <= 262 144, BASELINE config 5's 32 x 256 among them: one workgroup reads the whole score matrix, one wave per
 * row, and keeps the per-row terms in LDS; larger ones run one workgroup per row and a ticket.  offset + B <= C.
 */
#define MSIM_LOSS_PAIRWISE 0
#define MSIM_LOSS_INFONCE 1
#define MSIM_LOSS_SIGMOID 2
size_t msim_loss_epilogue_workspace_bytes(int B, int C);
int msim_loss_epilogue(int mode, const float *scores, int64_t ld, int B, int C,
                       const void *Q, int q_dtype, int Lq, int width, int offset,
                       float temperature, int normalize, int filter, float filter_threshold, float filter_factor,
                       float *G, int32_t *pairs, float *coef, int32_t *order,
                       void *workspace, float *out, void *loss_out, const int32_t *q_lengths, void *stream);

/*
 * Embedding head: the last three lines of every Col* model forward, producing the scorer's corpus format.
 *     y   = X @ W^T + bias                       nn.Linear(hidden, 128)
 *     y   = y / ||y||_2                          (row-wise, rounding chain of the model dtype)
 *     out = y * mask
 * Replaces colpali_engine/models/paligemma/colpali/modeling_colpali.py:67-77 and
 * colpali_engine/models/qwen2/colqwen2/modeling_colqwen2.py:65-74 (identical in the other families), and the
 * unbind / .cpu() / pad_sequence / H2D round trip between the model and the scorer (README.md:121-126,
 * processing_utils.py:172-178): rows can be written straight into the packed corpus blob msim_fwd streams.
 *   X [M, H] bf16|f16 hidden states (row-major, H a multiple of 64), W [128, H], bias [128] or NULL (same dtype)
 *   row_map int32, ceil(M / 256) * 256 entries (the padding is never dereferenced for rows >= M but must be readable):
 *       v >= 0   write the normalised row m to out row v
 *       v == -1  drop row m (masked position of an unpadded corpus)
 *       v <= -2  write a row of (signed) zeros to out row -2 - v   (masked position kept in place: the dense model output)
 *   out [rows, ld_out] same dtype as X, ld_out >= 128 elements.
 */
int msim_embed_head(int dtype, const void *X, int64_t M, int H,
                    const void *W, const void *bias, int n_out,
                    const int32_t *row_map, void *out, int64_t ld_out, void *stream);

/*
 * Row map of the DENSE form of msim_embed_head (the model forward's own output layout): row m keeps its place,
 *   row_map[m] = (mask[m] != 0 && (extra == NULL || extra[m] != 0)) ? m : -2 - m,    -1 for the tile padding m in [M, ceil(M / 256) * 256)
 * i.e. `proj * attention_mask.unsqueeze(-1)` (modeling_colpali.py:72) and the optional `proj * image_mask` (:74-77) as one launch in
 * front of the head.  mask / extra: [M] elements of `kind` 0 = 1-byte (bool / uint8 / int8), 1 = int16, 2 = int32, 3 = int64, 4 = fp32,
 * 5 = bf16, 6 = fp16 (a floating zero of either sign counts as masked); row_map: int32 [ceil(M / 256) * 256].
 */
int msim_embed_head_row_map(const void *mask, int mask_kind, const void *extra, int extra_kind, int64_t M, int32_t *row_map, void *stream);

/*
 * Row map of the PACKING form of msim_embed_head (pages written back to back into the resident corpus, masked positions dropped:
 * what replaces README.md:121-126 `torch.unbind(embeddings.to("cpu"))` + the scorer's per-block pad_sequence):
 *   row_map[b * S + s] = keep(b, s) ? *rows_before + (kept positions of pages < b) + (kept positions of page b before s) : -1
 *   counts[b] = kept positions of page b (int64, device);  *rows_after = *rows_before + sum of counts;  tile padding of the map = -1.
 * mask / extra as in msim_embed_head_row_map, [B * S] elements; rows_before / rows_after: device int64; they must NOT alias (the page workgroups read *rows_before while another one writes *rows_after).
 */
int msim_embed_head_writer_map(const void *mask, int mask_kind, const void *extra, int extra_kind, int B, int S, const int64_t *rows_before,
                               int64_t *counts, int32_t *row_map, int64_t *rows_after, void *stream);

/*
 * Backward of the norm / mask tail of the embedding head -- what torch autograd derives for
 *   colpali_engine/models/paligemma/colpali/modeling_colpali.py:70  proj = proj / proj.norm(dim=-1, keepdim=True)
 *                                                                :72  proj = proj * attention_mask.unsqueeze(-1)      (+ :74-77)
 * with respect to the nn.Linear output (:67), for a model whose head runs inside the training graph
 * (trainer/contrastive_trainer.py:135-162 back-propagates through it):
 *   dproj[m, :] = row_map[m] >= 0 ? (g[m, :] - y <g[m, :], y>) / n : 0,    y = proj[m, :] / n,    n = ||proj[m, :]|| rounded to dtype
 *   proj [M, 128] = the Linear output, grad_out [M, 128] = the upstream gradient, dproj [M, 128]: dense rows, dtype bf16 | f16;
 *   row_map as in msim_embed_head (>= 0: the position was kept; < 0: masked, gradient exactly 0).
 * The GEMMs on either side (proj itself; dX = dproj W, dW = dproj^T X, db = sum dproj) are plain library GEMMs, left to the host.
 */
int msim_embed_head_bwd(int dtype, const void *proj, const void *grad_out, const int32_t *row_map, int64_t M, int n_out,
                        void *dproj, void *stream);

/*
 * Plain similarity matrix, no reduction:   out[i, j] = <A[i, :], B[j, :]>   (fp32 accumulate)
 * Replaces colpali_engine/utils/processing_utils.py:126  torch.einsum("bd,cd->bc", qs, ps)   (score_single_vector, the
 * bi-encoder scorer of the same processor class) and the contraction of
 * colpali_engine/interpretability/similarity_map_utils.py:50  torch.einsum("nk,ijk->nij", query, image_grid).
 *   A [n_a, dim], B [n_b, dim] (bf16 | f16 | f32, rows a multiple of 32 bytes, <= 4096), out fp32 [n_a, ld_out].
 *   MSIM_FLAG_REF_ROUNDING: round every dot product to the input dtype (what torch stores for 16-bit inputs).
 */
int msim_sim_matrix(int dtype, const void *A, int n_a, const void *B, int n_b, int dim,
                    float *out, int64_t ld_out, uint32_t flags, void *stream);

/*
 * Hierarchical token pooling (Ward clustering of a page's patch embeddings, mean-pooling every cluster):
 * replaces colpali_engine/compression/token_pooling/hierarchical_token_pooling.py:83-146, i.e. torch.mm + SciPy's
 * linkage(method="ward") on the rows of 1 - E E^T + fcluster(criterion="maxclust") + the per-cluster mean / normalize,
 * which the reference runs page by page on the CPU.
 *
 * msim_pool_cluster: labels[r0 + i] = 0-based flat cluster of row i of page c (r0 = d_off[c]), numbered as SciPy numbers
 *   them; n_clusters[c] = number of clusters (<= max(n_c / pool_factor, 1)).
 *   E [rows, dim] packed pages (bf16 | f16 | f32, rows a multiple of 32 bytes), d_off int32 [n_pages + 1];
 *   max_rows >= the longest page (<= 32768; pages of at most 2048 rows keep the clustering state in LDS, longer ones in their own
 *   region of X_ws, which is dead after the distance pass); ws_off int64 [n_pages + 1] = exclusive prefix sums of n_c * n_c (element offsets
 *   of page c in both workspaces); X_ws fp32 and D_ws fp64 each of ws_off[n_pages] elements.
 * msim_pool_reduce: out[out_off[c] + k, :] = normalize(mean of the rows of page c labelled k), k < out_off[c+1] - out_off[c],
 *   in E's dtype; `dim` logical columns, ld_in / ld_out = elements between consecutive rows of E / out.
 */
int msim_pool_cluster(int dtype, const void *E, const int32_t *d_off, int n_pages, int dim, int max_rows,
                      const int64_t *ws_off, int pool_factor, float *X_ws, double *D_ws,
                      int32_t *labels, int32_t *n_clusters, void *stream);
int msim_pool_reduce(int dtype, const void *E, const int32_t *d_off, int n_pages, int dim, int ld_in,
                     const int32_t *labels, const int32_t *out_off, void *out, int ld_out, void *stream);

/*
 * Host-side helper of the drop-in's upload path (no device work): copies n separate host buffers into one destination image,
 *     memcpy(dst + dst_off[i], src[i], nbytes[i])   for i < n,
 * with up to n_threads threads (contiguous runs of buffers of about equal bytes per thread).  The reference hands the scorer a
 * python list of per-page tensors (README.md:121-126) and re-pads it per block with pad_sequence (processing_utils.py:172-178);
 * here the pages are gathered once into a pinned staging buffer and uploaded.  Native because a thousand small memcpy calls
 * issued from python threads fight over the interpreter lock (70 ms stalls in a 10 ms call were measured).
 */
int msim_host_gather(void *dst, const void *const *src, const int64_t *dst_off, const int64_t *nbytes, int64_t n, int n_threads);
/* The same for a BYTE RANGE of the image: buffer i holds image bytes prefix[i] .. prefix[i+1]-1 (prefix: n + 1 non-decreasing numbers);
 * bytes [lo, hi) of the image are copied to dst (dst[0] = image byte lo), cut into equal byte shares over a persistent pool of native
 * threads (no thread is started per call).  The upload path sends the image through a bounded pinned staging buffer chunk by chunk:
 * one call per chunk, nothing per page on the Python side. */
int msim_host_gather_range(void *dst, const void *const *src, const int64_t *prefix, int64_t n, int64_t lo, int64_t hi, int n_threads);
/* The same gather in two calls: `begin` hands the request to a persistent native thread and returns at once, `wait` blocks until it is
 * done and returns its result.  One request in flight per process (a second `begin` before `wait` is MSIM_EINVAL); the buffers must
 * stay valid until `wait` returns.  The upload path gathers chunk k + 1 this way while the calling thread issues chunk k's H2D copy
 * and the MaxSim launches of the passages that have arrived (colpali_amd/corpus.py: upload_image). */
int msim_host_gather_range_begin(void *dst, const void *const *src, const int64_t *prefix, int64_t n, int64_t lo, int64_t hi, int n_threads);
int msim_host_gather_range_wait(void);
/* The library's host threads (the gather pool's workers and the driver thread above, present and future) run on the listed CPUs from
 * now on.  The upload path calls it when the caller's pages turn out to live on another NUMA node than the one the threads sit on
 * (colpali_amd/_lib.py: gather_cpus_for): a memcpy that READS across the socket link is the slower direction. */
int msim_host_threads_affinity(const int32_t *cpus, int n_cpus);

/*
 * Row-wise top-k of a score matrix with the deterministic order
 * (score descending, id ascending).
 *   scores fp32 [n_q, ld]; the candidates of row q are columns 0..n-1
 *   ids    int64 [n_q, ld] or NULL: id of column j (NULL: id = id_base + j)
 *   out_scores fp32 [n_q, k], out_ids int64 [n_q, k]; rows with fewer than k
 *   candidates are padded with (-inf, -1).
 * The reference only exposes top-k through the experimental
 *   colpali_engine/utils/processing_utils.py:189-219 (get_topk_plaid, k=10 default);
 * retrieval users otherwise call torch.topk on score_multi_vector's output.
 * Needs msim_topk_workspace_bytes(n_q, n, k) bytes of scratch.
 */
size_t msim_topk_workspace_bytes(int n_q, int64_t n, int k);
int msim_topk_f32(const float *scores, const int64_t *ids, int n_q, int64_t n, int64_t ld,
                  int k, int64_t id_base,
                  float *out_scores, int64_t *out_ids, void *workspace, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* COLPALI_AMD_MAXSIM_H */
