// Hard-max MaxSim backward for gfx950: what autograd derives for
//   colpali_engine/loss/late_interaction_losses.py:297-298 (+ :91)   einsum("bnd,csd->bcns").amax(3).sum(2)
// with an upstream gradient g[p] per (query, document) pair and the arg-max patch of every (pair, query token):
//   dQ[b, i, :]      = sum over the pairs p of query b      of  g[p] * D[c_p, argmax[p, i], :]
//   dD[c, s, :]      = sum over the pairs p of document c,
//                      tokens i with argmax[p, i] == s      of  g[p] * Q[b_p, i, :]
// Both are gathers of a few hundred KiB ... MiB: latency-bound, not bandwidth-bound.  The kernels are therefore built
// around loads in flight (independent gather chains, no per-element branches) and a deterministic summation order
// (no float atomics): dQ sums each query's pairs in pair-list order per lane group and folds the groups in a fixed
// tree; dD walks each document's (pair, token) entries in list order.
//
// Any width (rows a multiple of 32 bytes, up to 4 KiB) and dtype (DT: 0 bf16, 1 fp16, 2 fp32).
#pragma once
#include "maxsim_common.hpp"
#include "maxsim_pairs.hip"
#include "maxsim_generic.hip"

namespace msim {

template <int DT>
__device__ __forceinline__ void piece_to_floats(const uint4 v, float *f) {
    if constexpr (DT == kDtypeF32) {
        f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
    } else {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f[2 * k] = elem_to_float<DT == kDtypeF16>((uint16_t)(w[k] & 0xffffu));
            f[2 * k + 1] = elem_to_float<DT == kDtypeF16>((uint16_t)(w[k] >> 16));
        }
    }
}

// An optional device scalar every pair's gradient is multiplied by: the loss' upstream gradient (autograd hands the backward a
// 0-dim tensor in the loss' dtype) -- folded in here instead of a `coef * grad` launch in front of the kernels.
struct GScale {
    const void *p;     // null: 1
    int kind;          // 0 bf16, 1 fp16, 2 fp32
};
__device__ __forceinline__ float load_gscale(const GScale &gs) {
    if (gs.p == nullptr) return 1.0f;
    if (gs.kind == 2) return *static_cast<const float *>(gs.p);
    const uint16_t v = *static_cast<const uint16_t *>(gs.p);
    return gs.kind == 1 ? elem_to_float<true>(v) : elem_to_float<false>(v);
}

// gradients leave the kernels either as fp32 or in the embeddings' own dtype (OUT16: one rounding of the fp32 sum -- what the
// `.to(dtype)` launches behind the kernels used to do, 25 us and 150 MB of traffic per step at BASELINE config 5's shape)
template <int DT>
__device__ __forceinline__ uint16_t float_to_elem16(float v) {
    if constexpr (DT == kDtypeF16) return __builtin_bit_cast(uint16_t, (_Float16)v);
    else return __builtin_bit_cast(uint16_t, (__bf16)v);
}
template <int DT>
__device__ __forceinline__ uint4 pack8(const float *f) {
    uint4 o;
    o.x = (uint32_t)float_to_elem16<DT>(f[0]) | ((uint32_t)float_to_elem16<DT>(f[1]) << 16);
    o.y = (uint32_t)float_to_elem16<DT>(f[2]) | ((uint32_t)float_to_elem16<DT>(f[3]) << 16);
    o.z = (uint32_t)float_to_elem16<DT>(f[4]) | ((uint32_t)float_to_elem16<DT>(f[5]) << 16);
    o.w = (uint32_t)float_to_elem16<DT>(f[6]) | ((uint32_t)float_to_elem16<DT>(f[7]) << 16);
    return o;
}

// ---- dQ: one wave per (query, token).  A row is split into 16-byte pieces, one per lane; when a row needs fewer than
// 64 lanes the wave works on 64 / lanes-per-row pairs at once, and four such groups are unrolled, so 4 .. 128
// independent (argmax -> offset -> row) gather chains are in flight per wave.  `pairs` sorted by query index.
// `tpw` consecutive tokens of one query per wave (the range lookup in the pair list -- two wave-wide searches, dependent loads -- is
// done once per wave: with one token per wave it was most of the 16 us the trainer's symmetric direction spent here on 25 000 tokens).
template <int DT, bool OUT16>
__global__ __launch_bounds__(256) void maxsim_bwd_dq_kernel(const char *__restrict__ D, const int32_t *__restrict__ d_off,
                                                            const int32_t *__restrict__ pairs, const float *__restrict__ g,
                                                            const int32_t *__restrict__ argmax, void *__restrict__ dQ,
                                                            PairsArgs a, int row_bytes, int tpw, GScale gs) {
    constexpr int ES = elem_size<DT>();
    constexpr int EPP = 16 / ES;                         // elements per 16-byte piece
    static_assert(!OUT16 || DT != kDtypeF32, "OUT16 is the 16-bit embedding dtypes' own output");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int chunks = (a.Lq + 4 * tpw - 1) / (4 * tpw);
    const int b = blockIdx.x / chunks;
    const int i0 = (blockIdx.x - b * chunks) * 4 * tpw + wave * tpw;
    if (b >= a.n_q || i0 >= a.Lq) return;
    const float up = load_gscale(gs);
    const int s = lower_bound_wave(a.n_pairs, b, lane, [&](int k) { return pairs[2 * k]; });
    const int e = lower_bound_wave(a.n_pairs, b + 1, lane, [&](int k) { return pairs[2 * k]; });
    const int pieces = row_bytes >> 4;
    int pp_log = 1;
    while ((1 << pp_log) < pieces && pp_log < 6) ++pp_log;
    const int pp = 1 << pp_log;                          // lanes per pair
    const int G = 64 >> pp_log;                          // pairs per wave-step
    const int sub = lane >> pp_log, pc = lane & (pp - 1);
    const int i_end = i0 + tpw < a.Lq ? i0 + tpw : a.Lq;
    if (e - s <= 8 && G > 1) {
        // FEW pairs per query (the pairwise loss: two): the lane groups take different TOKENS instead of different pairs -- G tokens'
        // (argmax -> offset -> row) chains in flight per wave and no cross-group fold.  (With the groups on pairs, 14 of 16 gather
        // slots idled and a wave walked its tokens one dependent chain at a time: 17 us for the 25 000 tokens of the trainer's
        // symmetric direction.)
        for (int ib = i0; ib < i_end; ib += G) {
            const int i = ib + sub;
            const bool tok_ok = i < i_end;
            const int ic = tok_ok ? i : i_end - 1;
            const size_t tok = (size_t)b * a.Lq + ic;
            for (int c0 = 0; c0 < pieces; c0 += pp) {
                const int piece = c0 + pc;
                const bool col_ok = piece < pieces;
                const int boff = (col_ok ? piece : 0) << 4;
                float acc[EPP];
#pragma unroll
                for (int k = 0; k < EPP; ++k) acc[k] = 0.0f;
                for (int p0 = s; p0 < e; p0 += 4) {
                    int pj[4], arg[4], doc[4], off[4];
                    float w[4];
                    bool ok[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        ok[j] = p0 + j < e;
                        pj[j] = ok[j] ? p0 + j : e - 1;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        arg[j] = argmax[(size_t)pj[j] * a.Lq + ic];
                        doc[j] = pairs[2 * pj[j] + 1];
                        w[j] = g[pj[j]] * up;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) off[j] = d_off[doc[j]];
                    uint4 v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool use = ok[j] && arg[j] >= 0;
                        w[j] = use ? w[j] : 0.0f;
                        const int row = off[j] + (arg[j] >= 0 ? arg[j] : 0);
                        v[j] = *reinterpret_cast<const uint4 *>(D + (size_t)row * row_bytes + boff);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float f[EPP];
                        piece_to_floats<DT>(v[j], f);
#pragma unroll
                        for (int k = 0; k < EPP; ++k) acc[k] += w[j] * f[k];
                    }
                }
                if (tok_ok && col_ok) {
                    if constexpr (OUT16) {
                        *reinterpret_cast<uint4 *>(static_cast<char *>(dQ) + tok * row_bytes + ((size_t)piece << 4)) = pack8<DT>(acc);
                    } else {
                        float *o = static_cast<float *>(dQ) + tok * (row_bytes / ES) + piece * EPP;
#pragma unroll
                        for (int k = 0; k < EPP; k += 4) *reinterpret_cast<float4 *>(o + k) = make_float4(acc[k], acc[k + 1], acc[k + 2], acc[k + 3]);
                    }
                }
            }
        }
        return;
    }
    for (int i = i0; i < i_end; ++i) {
        const size_t tok = (size_t)b * a.Lq + i;
        for (int c0 = 0; c0 < pieces; c0 += 64) {            // more than one round only for rows wider than 1 KiB
            const int piece = c0 + pc;
            const bool col_ok = piece < pieces;
            const int boff = (col_ok ? piece : 0) << 4;
            float acc[EPP];
#pragma unroll
            for (int k = 0; k < EPP; ++k) acc[k] = 0.0f;
            for (int p0 = s; p0 < e; p0 += 4 * G) {
                int pj[4], arg[4], doc[4], off[4];
                float w[4];
                bool ok[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int p = p0 + j * G + sub;
                    ok[j] = p < e;
                    pj[j] = ok[j] ? p : e - 1;                                   // clamped: every load is a valid address
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    arg[j] = argmax[(size_t)pj[j] * a.Lq + i];
                    doc[j] = pairs[2 * pj[j] + 1];
                    w[j] = g[pj[j]] * up;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) off[j] = d_off[doc[j]];
                uint4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool use = ok[j] && arg[j] >= 0;                        // arg < 0: the zero padding row won the max
                    w[j] = use ? w[j] : 0.0f;                                     // the weight is selected, never the loaded row
                    const int row = off[j] + (arg[j] >= 0 ? arg[j] : 0);
                    v[j] = *reinterpret_cast<const uint4 *>(D + (size_t)row * row_bytes + boff);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float f[EPP];
                    piece_to_floats<DT>(v[j], f);
#pragma unroll
                    for (int k = 0; k < EPP; ++k) acc[k] += w[j] * f[k];
                }
            }
            for (int m = pp; m < 64; m <<= 1)
#pragma unroll
                for (int k = 0; k < EPP; ++k) acc[k] += __shfl_xor(acc[k], m);
            if (sub == 0 && col_ok) {
                if constexpr (OUT16) {
                    *reinterpret_cast<uint4 *>(static_cast<char *>(dQ) + tok * row_bytes + ((size_t)piece << 4)) = pack8<DT>(acc);
                } else {
                    float *o = static_cast<float *>(dQ) + tok * (row_bytes / ES) + piece * EPP;
#pragma unroll
                    for (int k = 0; k < EPP; k += 4) *reinterpret_cast<float4 *>(o + k) = make_float4(acc[k], acc[k + 1], acc[k + 2], acc[k + 3]);
                }
            }
        }
    }
}

// ---- dD: one workgroup per (document, group of 64-row ranges, 128-column chunk).  The document's pair range in the by-document
// order is looked up ONCE per workgroup (two wave-wide searches of dependent loads); a document without pairs -- in the pairwise
// loss all but 2B of them -- has its rows zero-filled with 16-byte stores and nothing else.  Otherwise, per row range: the
// document's (pair, token) entries are read 256 at a time with coalesced, independent loads; the ones that land in this row range are
// compacted (ballot + prefix, list order kept) into an LDS hit list; then every thread adds the hits of its rows to the LDS tile
// (thread t owns column t & 127 of the rows with parity t >> 7; four hits' query values are in flight at a time;
// hits of the other parity go to a dummy row instead of a branch).  `order_by_doc` lists pair indices sorted by document.
template <int DT, bool OUT16>
__global__ __launch_bounds__(256) void maxsim_bwd_dd_kernel(const char *__restrict__ Q, const int32_t *__restrict__ d_off,
                                                            const int32_t *__restrict__ pairs,
                                                            const int32_t *__restrict__ order_by_doc, const float *__restrict__ g,
                                                            const int32_t *__restrict__ argmax, void *__restrict__ dD,
                                                            PairsArgs a, int dim, GScale gs) {
    constexpr int ES = elem_size<DT>();
    constexpr int OES = OUT16 ? 2 : 4;                   // bytes per output element
    __shared__ float tile[kBwdRows + 2][128];            // + one dummy row per parity
    __shared__ int hit_r[256 + 4], hit_q[256 + 4];
    __shared__ float hit_g[256 + 4];
    __shared__ int wave_cnt[4];
    const int c = blockIdx.x;
    const int len = d_off[c + 1] - d_off[c];
    if ((int)blockIdx.y * kBwdRows >= len) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, lc = t & 127, half = t >> 7;
    const int col0 = blockIdx.z * 128;
    const int col = col0 + lc;
    const bool col_ok = col < dim;
    const int col_c = col_ok ? col : 0;
    const int ncol = dim - col0 < 128 ? dim - col0 : 128;   // columns of this chunk (a multiple of 8: rows are multiples of 32 bytes)
    auto doc_of = [&](int k) { return pairs[2 * order_by_doc[k] + 1]; };
    const int s = lower_bound_wave(a.n_pairs, c, lane, doc_of);
    const int e = lower_bound_wave(a.n_pairs, c + 1, lane, doc_of);
    const int n_ent = (e - s) * a.Lq;
    const float up = load_gscale(gs);
    char *out_doc = static_cast<char *>(dD) + (size_t)d_off[c] * dim * OES;
    const int groups = ncol >> 3;                        // 8-column groups per row of this chunk
    for (int r_lo = blockIdx.y * kBwdRows; r_lo < len; r_lo += gridDim.y * kBwdRows) {
        const int rows = (len - r_lo < kBwdRows) ? (len - r_lo) : kBwdRows;
        if (n_ent == 0) {                                 // nothing lands here: rows of zeros, 8 elements per store
            for (int idx = t; idx < rows * groups; idx += 256) {
                const int r = idx / groups, gcol = idx - r * groups;
                char *o = out_doc + ((size_t)(r_lo + r) * dim + col0 + gcol * 8) * OES;
                if constexpr (OUT16) {
                    *reinterpret_cast<uint4 *>(o) = make_uint4(0, 0, 0, 0);
                } else {
                    *reinterpret_cast<float4 *>(o) = make_float4(0.f, 0.f, 0.f, 0.f);
                    *reinterpret_cast<float4 *>(o + 16) = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            continue;
        }
        for (int r = half; r < kBwdRows + 2; r += 2) tile[r][lc] = 0.0f;
        for (int base = 0; base < n_ent; base += 256) {
            const int idx = base + t;
            bool hit = false;
            int r = 0, qrow = 0;
            float gp = 0.0f;
            if (idx < n_ent) {
                const int k = idx / a.Lq, i = idx - k * a.Lq;
                const int p = order_by_doc[s + k];
                r = argmax[(size_t)p * a.Lq + i] - r_lo;
                hit = r >= 0 && r < rows;
                qrow = pairs[2 * p] * a.Lq + i;
                gp = g[p] * up;
            }
            const unsigned long long m = __ballot(hit);
            const int before = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) wave_cnt[wave] = __popcll(m);
            __syncthreads();
            int woff = 0, total = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int cnt = wave_cnt[w];
                woff += w < wave ? cnt : 0;
                total += cnt;
            }
            if (hit) {
                hit_r[woff + before] = r;
                hit_q[woff + before] = qrow;
                hit_g[woff + before] = gp;
            }
            if (t < 4) {                                      // pad to a multiple of four: dummy hits (row -1, weight 0)
                hit_r[total + t] = -1;
                hit_q[total + t] = 0;
                hit_g[total + t] = 0.0f;
            }
            __syncthreads();
            for (int h = 0; h < total; h += 4) {
                int rr[4];
                float qv[4], gg[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int rj = hit_r[h + j];
                    const bool mine = rj >= 0 && (rj & 1) == half;
                    rr[j] = mine ? rj : kBwdRows + half;
                    gg[j] = hit_g[h + j];
                    qv[j] = load_elem<DT>(Q + ((size_t)hit_q[h + j] * dim + col_c) * ES);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) tile[rr[j]][lc] += gg[j] * qv[j];
            }
            __syncthreads();
        }
        // write-out: 8 consecutive columns of a row per thread (one 16-byte store in the 16-bit form); the barrier above has made
        // every row of the tile visible
        for (int idx = t; idx < rows * groups; idx += 256) {
            const int r = idx / groups, gcol = idx - r * groups;
            const float *src = &tile[r][gcol * 8];
            char *o = out_doc + ((size_t)(r_lo + r) * dim + col0 + gcol * 8) * OES;
            if constexpr (OUT16) {
                float f[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) f[k] = src[k];
                *reinterpret_cast<uint4 *>(o) = pack8<DT>(f);
            } else {
                *reinterpret_cast<float4 *>(o) = make_float4(src[0], src[1], src[2], src[3]);
                *reinterpret_cast<float4 *>(o + 16) = make_float4(src[4], src[5], src[6], src[7]);
            }
        }
        __syncthreads();                                   // the tile is re-zeroed for the next row range
    }
}

// (Round 6, measured and NOT kept: a "sorted" form of the kernel above -- the document's entries sorted once by (winning row, entry) with
// a bitonic sort in LDS, then every output row adding exactly its own entries -- one pass over the entries instead of one per 64-row
// range.  74.9 us against 61.3 us under ColbertLoss's dense gradient at config 5's shape and 59.7 against 14.4 us under the pairwise
// loss: 55 barrier-separated sort steps plus a chain of dependent LDS / L2 reads per row are worse than re-scanning 1024 entries.  Also measured and not kept: four hit lists by row class r & 3, one per wave, every lane two
// columns per hit (4-byte loads), eight hits in flight, the document's pair info cached in LDS -- 60.3 us against 61.8 (ColbertLoss) and 16.5
// against 14.4 (pairwise): the hit loop is not where this kernel's time goes.)

// ---- dD, dense form: SHORT documents (<= kBwdRows rows: one row range) with LONG entry lists -- the symmetric direction of the
// reference trainer (trainer/contrastive_trainer.py:202-206: pages as query_embeddings [B, 780, 128], queries as doc_embeddings
// [B, 32, 128]), where a document of 32 rows collects B x 780 (pair, token) entries and the launch above has n_d workgroups to
// spread them over (32 workgroups walking 25 000 entries each, every step a dependent gather: 2.4 ms of a 2.6 ms loss step
// against 0.7 ms for the reference's einsum autograd).  Here EVERY entry hits the one row range, so there is nothing to
// compact: workgroup (document c, split z) takes an even share of c's flattened (pair, token) entries; thread t owns column t & 127 and the tokens of
// parity t >> 7, eight tokens' loads in flight per thread, accumulating into its parity's LDS tile; the two tiles are summed
// (parity 0 first) into the split's partial, and maxsim_bwd_dd_sum_kernel adds the splits in split order: a fixed summation
// order, no float atomics.
constexpr int kBwdDenseUnroll = 8;

template <int DT>
__global__ __launch_bounds__(256) void maxsim_bwd_dd_dense_kernel(const char *__restrict__ Q, const int32_t *__restrict__ d_off,
                                                                  const int32_t *__restrict__ pairs,
                                                                  const int32_t *__restrict__ order_by_doc, const float *__restrict__ g,
                                                                  const int32_t *__restrict__ argmax, float *__restrict__ partial,
                                                                  PairsArgs a, int dim, int max_rows, int n_splits, GScale gs) {
    constexpr int ES = elem_size<DT>();
    extern __shared__ __attribute__((aligned(16))) char smem_dd[];
    float(*tile)[128] = reinterpret_cast<float(*)[128]>(smem_dd);           // [2 * max_rows][128]: parity-major
    const int c = blockIdx.x, z = blockIdx.y;
    const int len = d_off[c + 1] - d_off[c];
    const int t = threadIdx.x, lane = t & 63, lc = t & 127, half = t >> 7;
    const int col = blockIdx.z * 128 + lc;
    const bool col_ok = col < dim;
    const int col_c = col_ok ? col : 0;
    auto doc_of = [&](int k) { return pairs[2 * order_by_doc[k] + 1]; };
    const int s = lower_bound_wave(a.n_pairs, c, lane, doc_of);
    const int e = lower_bound_wave(a.n_pairs, c + 1, lane, doc_of);
    if (e > s)
        for (int r = 0; r < len; ++r) tile[half * max_rows + r][lc] = 0.0f;
    // split z takes an even share of the document's FLATTENED (pair, token) entries (pair-major): entries [e_lo, e_hi)
    const long long n_ent = (long long)(e - s) * a.Lq;
    if (n_ent == 0) return;                               // no pair names this document: the sum kernel writes its zeros without reading a partial
    const long long e_lo = (n_ent * z) / n_splits, e_hi = (n_ent * (z + 1)) / n_splits;
    const int k_first = s + (int)(e_lo / a.Lq), i_first = (int)(e_lo % a.Lq);
    const int k_last = e_hi > e_lo ? s + (int)((e_hi - 1) / a.Lq) : k_first - 1, i_last = e_hi > e_lo ? (int)((e_hi - 1) % a.Lq) + 1 : 0;
    float *my = &tile[half * max_rows][lc];
    const float up = load_gscale(gs);
    for (int k = k_first; k <= k_last; ++k) {
        const int p = order_by_doc[k];
        const float gp = g[p] * up;
        const int32_t *arg = argmax + (size_t)p * a.Lq;
        const char *qrow = Q + ((size_t)pairs[2 * p] * a.Lq * dim + col_c) * ES;
        const int i_begin = k == k_first ? i_first : 0, i_end = k == k_last ? i_last : a.Lq;
        for (int i0 = i_begin + half; i0 < i_end; i0 += 2 * kBwdDenseUnroll) {
            int r[kBwdDenseUnroll];
            float qv[kBwdDenseUnroll];
#pragma unroll
            for (int j = 0; j < kBwdDenseUnroll; ++j) {
                const int i = i0 + 2 * j;
                const bool ok = i < i_end;
                const int ic = ok ? i : i_begin;                               // clamped: every load is a valid address
                const int rj = arg[ic];
                r[j] = (ok && rj >= 0 && rj < len) ? rj : -1;                  // arg < 0: the zero padding row won the max
                qv[j] = load_elem<DT>(qrow + (size_t)ic * dim * ES);
            }
#pragma unroll
            for (int j = 0; j < kBwdDenseUnroll; ++j)
                if (r[j] >= 0) my[r[j] * 128] += gp * qv[j];                   // wave-uniform row: no divergence inside a wave
        }
    }
    __syncthreads();
    if (col_ok && half == 0) {
        float *out = partial + (((size_t)z * a.n_d + c) * max_rows) * dim + col;
        for (int r = 0; r < len; ++r) out[(size_t)r * dim] = tile[r][lc] + tile[max_rows + r][lc];
    }
}

// ---- dD, per-pair dense form: FEW pairs, each with a LONG entry list for its (short) document -- the pairwise loss in the trainer's
// symmetric direction: 2B = 64 pairs of 780 tokens over 256 documents of 32 rows.  The per-document form above launches a workgroup per
// (document, split) and every one of them starts with two wave-wide searches of the pair list (dependent loads) -- 4096 workgroups, 64
// of which have work.  Here the grid is (pair k in by-document order, split z): no search, every workgroup has work; partial tile
// [(k, z)] = the split's share of pair k's tokens; the sum kernel below adds a document's (pair, split) tiles in that order.
template <int DT>
__global__ __launch_bounds__(256) void maxsim_bwd_dd_pairs_kernel(const char *__restrict__ Q, const int32_t *__restrict__ d_off,
                                                                  const int32_t *__restrict__ pairs,
                                                                  const int32_t *__restrict__ order_by_doc, const float *__restrict__ g,
                                                                  const int32_t *__restrict__ argmax, float *__restrict__ partial,
                                                                  PairsArgs a, int dim, int max_rows, int n_splits, GScale gs) {
    constexpr int ES = elem_size<DT>();
    extern __shared__ __attribute__((aligned(16))) char smem_dd[];
    float(*tile)[128] = reinterpret_cast<float(*)[128]>(smem_dd);           // [2 * max_rows][128]: parity-major
    const int k = blockIdx.x, z = blockIdx.y;
    const int p = order_by_doc[k];
    const int c = pairs[2 * p + 1];
    int len = d_off[c + 1] - d_off[c];
    len = len < max_rows ? len : max_rows;
    const int t = threadIdx.x, lc = t & 127, half = t >> 7;
    const int col = blockIdx.z * 128 + lc;
    const bool col_ok = col < dim;
    const int col_c = col_ok ? col : 0;
    for (int r = 0; r < len; ++r) tile[half * max_rows + r][lc] = 0.0f;
    const int i_lo = (int)(((long long)a.Lq * z) / n_splits), i_hi = (int)(((long long)a.Lq * (z + 1)) / n_splits);
    const float gp = g[p] * load_gscale(gs);
    const int32_t *arg = argmax + (size_t)p * a.Lq;
    const char *qrow = Q + ((size_t)pairs[2 * p] * a.Lq * dim + col_c) * ES;
    float *my = &tile[half * max_rows][lc];
    for (int i0 = i_lo + half; i0 < i_hi; i0 += 2 * kBwdDenseUnroll) {
        int r[kBwdDenseUnroll];
        float qv[kBwdDenseUnroll];
#pragma unroll
        for (int j = 0; j < kBwdDenseUnroll; ++j) {
            const int i = i0 + 2 * j;
            const bool ok = i < i_hi;
            const int ic = ok ? i : i_lo;                                   // clamped: every load is a valid address
            const int rj = arg[ic];
            r[j] = (ok && rj >= 0 && rj < len) ? rj : -1;                  // arg < 0: the zero padding row won the max
            qv[j] = load_elem<DT>(qrow + (size_t)ic * dim * ES);
        }
#pragma unroll
        for (int j = 0; j < kBwdDenseUnroll; ++j)
            if (r[j] >= 0) my[r[j] * 128] += gp * qv[j];                   // wave-uniform row: no divergence inside a wave
    }
    __syncthreads();
    if (col_ok && half == 0) {
        float *out = partial + (((size_t)k * n_splits + z) * max_rows) * dim + col;
        for (int r = 0; r < len; ++r) out[(size_t)r * dim] = tile[r][lc] + tile[max_rows + r][lc];
    }
}

// dD[c, r, :] = sum over the document's pairs k (by-document order) and their splits z, in that order; one workgroup per
// (document, 512 / 256 elements): the pair range is looked up once, a document without pairs is zero-filled
template <int DT, bool OUT16>
__global__ __launch_bounds__(256) void maxsim_bwd_dd_pairsum_kernel(const float *__restrict__ partial, const int32_t *__restrict__ d_off,
                                                                    const int32_t *__restrict__ pairs, const int32_t *__restrict__ order_by_doc,
                                                                    void *__restrict__ dD, int n_pairs, int dim, int max_rows, int n_splits) {
    const int c = blockIdx.x;
    const int len = d_off[c + 1] - d_off[c];
    auto doc_of = [&](int k) { return pairs[2 * order_by_doc[k] + 1]; };
    const int lane = threadIdx.x & 63;
    const int s = lower_bound_wave(n_pairs, c, lane, doc_of);
    const int e = lower_bound_wave(n_pairs, c + 1, lane, doc_of);
    constexpr int EPT = OUT16 ? 2 : 1;                                      // elements per thread and step
    constexpr int kInFlight = 8;                                            // partial tiles read before the first add
    for (int idx = (blockIdx.y * 256 + threadIdx.x) * EPT; idx < len * dim; idx += gridDim.y * 256 * EPT) {
        const int r = idx / dim, col = idx - r * dim;
        float a0 = 0.0f, a1 = 0.0f;
        if (r < max_rows) {
            const int n_tiles = (e - s) * n_splits;                         // tiles (k, z) of this document, k-major: consecutive in `partial`
            const float *base = partial + ((size_t)s * n_splits * max_rows + r) * dim + col;
            const size_t tile_stride = (size_t)max_rows * dim;
            for (int t0 = 0; t0 < n_tiles; t0 += kInFlight) {
                float v0[kInFlight], v1[kInFlight];
#pragma unroll
                for (int j = 0; j < kInFlight; ++j) {
                    const float *src = base + (size_t)(t0 + j < n_tiles ? t0 + j : t0) * tile_stride;
                    if constexpr (OUT16) {
                        const float2 v = *reinterpret_cast<const float2 *>(src);
                        v0[j] = v.x;
                        v1[j] = v.y;
                    } else {
                        v0[j] = src[0];
                        v1[j] = 0.0f;
                    }
                }
#pragma unroll
                for (int j = 0; j < kInFlight; ++j)
                    if (t0 + j < n_tiles) {
                        a0 += v0[j];
                        a1 += v1[j];
                    }
            }
        }
        if constexpr (OUT16)
            *reinterpret_cast<uint32_t *>(static_cast<char *>(dD) + ((size_t)d_off[c] * dim + idx) * 2) =
                (uint32_t)float_to_elem16<DT>(a0) | ((uint32_t)float_to_elem16<DT>(a1) << 16);
        else
            static_cast<float *>(dD)[(size_t)d_off[c] * dim + idx] = a0;
    }
}

// dD[c, r, :] = sum over the splits, in split order; workgroup (c, y) owns elements 256 y .. 256 y + 255 of document c's rows
// (OUT16: every thread owns two neighbouring elements and stores them as one 4-byte word)
template <int DT, bool OUT16>
__global__ __launch_bounds__(256) void maxsim_bwd_dd_sum_kernel(const float *__restrict__ partial, const int32_t *__restrict__ d_off,
                                                                const int32_t *__restrict__ pairs, const int32_t *__restrict__ order_by_doc,
                                                                void *__restrict__ dD, int n_d, int n_pairs, int dim, int max_rows, int n_splits) {
    const int c = blockIdx.x;
    const int len = d_off[c + 1] - d_off[c];
    // a document no pair names has no partials (the dense kernel returned at once): its rows are zeros
    auto doc_of = [&](int k) { return pairs[2 * order_by_doc[k] + 1]; };
    const int lane = threadIdx.x & 63;
    const int s = lower_bound_wave(n_pairs, c, lane, doc_of);
    const int e = lower_bound_wave(n_pairs, c + 1, lane, doc_of);
    if (e == s) n_splits = 0;
    if constexpr (OUT16) {
        const int idx = (blockIdx.y * 256 + threadIdx.x) * 2;       // dim is even (rows are multiples of 32 bytes)
        if (idx >= len * dim) return;
        const int r = idx / dim, col = idx - r * dim;
        float a0 = 0.0f, a1 = 0.0f;
        for (int z = 0; z < n_splits; ++z) {
            const float2 v = *reinterpret_cast<const float2 *>(partial + (((size_t)z * n_d + c) * max_rows + r) * dim + col);
            a0 += v.x;
            a1 += v.y;
        }
        *reinterpret_cast<uint32_t *>(static_cast<char *>(dD) + ((size_t)d_off[c] * dim + idx) * 2) =
            (uint32_t)float_to_elem16<DT>(a0) | ((uint32_t)float_to_elem16<DT>(a1) << 16);
    } else {
        const int idx = blockIdx.y * 256 + threadIdx.x;
        if (idx >= len * dim) return;
        const int r = idx / dim, col = idx - r * dim;
        float acc = 0.0f;
        for (int z = 0; z < n_splits; ++z) acc += partial[(((size_t)z * n_d + c) * max_rows + r) * dim + col];
        static_cast<float *>(dD)[(size_t)d_off[c] * dim + idx] = acc;
    }
}

}  // namespace msim
