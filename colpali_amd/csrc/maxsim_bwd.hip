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
                                                            PairsArgs a, int row_bytes, int tpw, GScale gs, int psplit) {
    constexpr int ES = elem_size<DT>();
    constexpr int EPP = 16 / ES;                         // elements per 16-byte piece
    static_assert(!OUT16 || DT != kDtypeF32, "OUT16 is the 16-bit embedding dtypes' own output");
    // psplit (round 6): the workgroup's four waves take the SAME tokens and a quarter of the query's pairs each; their partial rows
    // meet in LDS and wave 0 adds them in wave order.  A dense gradient at config 5's shape is 1024 tokens x 256 pairs: with a wave per
    // token the chip holds one wave per SIMD and every wave walks 16 steps of three dependent loads (21 us); split four ways it holds
    // four waves per SIMD with 4 steps each.
    __shared__ float part[3][64][EPP];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int chunks = psplit ? (a.Lq + tpw - 1) / tpw : (a.Lq + 4 * tpw - 1) / (4 * tpw);
    const int b = blockIdx.x / chunks;
    const int i0 = psplit ? (blockIdx.x - b * chunks) * tpw : (blockIdx.x - b * chunks) * 4 * tpw + wave * tpw;
    if (b >= a.n_q || i0 >= a.Lq) return;
    const float up = load_gscale(gs);
    // the query's pair range: a guess four probes can prove (every query owns the same number of pairs -- any all-pairs or
    // explicit-negative list: q(s - 1) < b == q(s) == q(e - 1) < q(e) in a list sorted by query), else two wave-wide binary searches
    // (3 + 3 dependent round trips for 8192 pairs)
    int s = 0, e = 0;
    bool have = false;
    const int per_q = a.n_q > 0 ? a.n_pairs / a.n_q : 0;
    if (per_q > 0 && per_q * a.n_q == a.n_pairs) {
        const int gs0 = b * per_q, ge0 = gs0 + per_q;
        const int pos = lane == 0 ? gs0 - 1 : lane == 1 ? gs0 : lane == 2 ? ge0 - 1 : ge0;
        bool ok = true;
        if (lane < 4 && pos >= 0 && pos < a.n_pairs) {
            const int qv = pairs[2 * pos];
            ok = lane == 0 ? qv < b : lane == 3 ? qv > b : qv == b;
        }
        if (__ballot(!ok) == 0) { s = gs0; e = ge0; have = true; }
    }
    if (!have) {
        s = lower_bound_wave(a.n_pairs, b, lane, [&](int k) { return pairs[2 * k]; });
        e = lower_bound_wave(a.n_pairs, b + 1, lane, [&](int k) { return pairs[2 * k]; });
    }
    if (psplit) {                                        // this wave's quarter of the pairs (the host only asks for it on long lists)
        const int per = (e - s + 3) >> 2;
        const int ws = s + wave * per;
        e = ws + per < e ? ws + per : e;
        s = ws < e ? ws : e;
    }
    const int pieces = row_bytes >> 4;
    int pp_log = 1;
    while ((1 << pp_log) < pieces && pp_log < 6) ++pp_log;
    const int pp = 1 << pp_log;                          // lanes per pair
    const int G = 64 >> pp_log;                          // pairs per wave-step
    const int sub = lane >> pp_log, pc = lane & (pp - 1);
    const int i_end = i0 + tpw < a.Lq ? i0 + tpw : a.Lq;
    if (!psplit && e - s <= 8 && G > 1) {
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
            // U pairs per lane group and step: the step's three rounds of dependent loads (routing / pair -> row offset -> row) are
            // issued for all U pairs before any is used
            auto steps = [&](auto u_c) {
                constexpr int U = decltype(u_c)::value;
                for (int p0 = s; p0 < e; p0 += U * G) {
                    int pj[U], arg[U], doc[U], off[U];
                    float w[U];
                    bool ok[U];
#pragma unroll
                    for (int j = 0; j < U; ++j) {
                        const int p = p0 + j * G + sub;
                        ok[j] = p < e;
                        pj[j] = ok[j] ? p : e - 1;                                   // clamped: every load is a valid address
                    }
#pragma unroll
                    for (int j = 0; j < U; ++j) {
                        arg[j] = argmax[(size_t)pj[j] * a.Lq + i];
                        doc[j] = pairs[2 * pj[j] + 1];
                        w[j] = g[pj[j]] * up;
                    }
#pragma unroll
                    for (int j = 0; j < U; ++j) off[j] = d_off[doc[j]];
                    uint4 v[U];
#pragma unroll
                    for (int j = 0; j < U; ++j) {
                        const bool use = ok[j] && arg[j] >= 0;                        // arg < 0: the zero padding row won the max
                        w[j] = use ? w[j] : 0.0f;                                     // the weight is selected, never the loaded row
                        const int row = off[j] + (arg[j] >= 0 ? arg[j] : 0);
                        v[j] = *reinterpret_cast<const uint4 *>(D + (size_t)row * row_bytes + boff);
                    }
#pragma unroll
                    for (int j = 0; j < U; ++j) {
                        float f[EPP];
                        piece_to_floats<DT>(v[j], f);
#pragma unroll
                        for (int k = 0; k < EPP; ++k) acc[k] += w[j] * f[k];
                    }
                }
            };
            // (psplit with U = 16 -- a wave's quarter of 256 pairs as ONE step -- measured 19.1 us against 15.4 with U = 4: 262 144 scattered
            // 256-byte row reads from a 51 MB box are 67 MB in 15 us = 4.4 TB/s; the kernel is at the memory system, not at a latency chain)
            steps(std::integral_constant<int, 4>{});
            for (int m = pp; m < 64; m <<= 1)
#pragma unroll
                for (int k = 0; k < EPP; ++k) acc[k] += __shfl_xor(acc[k], m);
            if (psplit) {                                // (every wave of the workgroup is here: same tokens, same rounds)
                if (wave > 0 && sub == 0) {
#pragma unroll
                    for (int k = 0; k < EPP; ++k) part[wave - 1][pc][k] = acc[k];
                }
                __syncthreads();
                if (wave == 0 && sub == 0) {
#pragma unroll
                    for (int w2 = 0; w2 < 3; ++w2)
#pragma unroll
                        for (int k = 0; k < EPP; ++k) acc[k] += part[w2][pc][k];
                }
                __syncthreads();
                if (wave > 0) continue;
            }
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

// ---- dD, row-list form (round 6): LONG documents with SHORT entry lists -- the forward direction of the trainer (pages as documents:
// 780 rows; ColbertLoss's dense gradient gives every page 32 pairs x 32 tokens = 1024 entries, ~1.3 per row; the pairwise loss gives
// 2B pages 32 entries each and every other page none).  The kernel above re-scans a document's entries once per 64-row range (13 scans
// of 1024 entries, each a chain of dependent global loads, two barriers per 256 entries).  Here a workgroup reads the entries ONCE,
// buckets them by winning row with a counting sort in LDS (integer LDS atomics for the counts and the slots, then every row's short
// segment is put into entry order by ranking: the float sums see the entries in list order whatever order the atomics ran in), and
// then walks its rows -- 16 lanes per row, 8 columns per lane, 64 rows of the workgroup in flight -- adding each row's own entries and
// storing the row once (zeros where nothing landed: no tile, no zero-fill pass, no write-out pass).
// Grid: (document, row split, 128-column chunk).  Limits (the host checks the bound it can know, the kernel the real counts; a
// document beyond them takes the slow direct walk at the end): kRowsMaxEnt entries and kRowsMaxPairs pairs per document, kRowsMaxRows rows.
constexpr int kRowsThreads = 512;
constexpr int kRowsMaxEnt = 4096, kRowsMaxPairs = 1024, kRowsMaxRows = 1024;      // (rows: two per thread in the scan)
constexpr int kRowsCountPairs = 16384;     // pair lists up to this length: the document's pair range by ONE counting pass of the workgroup
#ifdef MSIM_AB
__device__ unsigned long long g_rows_trace[16];      // measurement builds: s_memtime stamps of workgroup (1, 0, 0)'s phases
#define ROWS_STAMP(i) do { if (blockIdx.x == 1 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) g_rows_trace[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ROWS_STAMP(i) do { } while (0)
#endif

template <int DT, bool OUT16>
__global__ __launch_bounds__(kRowsThreads) void maxsim_bwd_dd_rows_kernel(const char *__restrict__ Q, const int32_t *__restrict__ d_off,
                                                                          const int32_t *__restrict__ pairs,
                                                                          const int32_t *__restrict__ order_by_doc, const float *__restrict__ g,
                                                                          const int32_t *__restrict__ argmax, void *__restrict__ dD,
                                                                          PairsArgs a, int dim, GScale gs) {
    constexpr int ES = elem_size<DT>();
    constexpr int OES = OUT16 ? 2 : 4;
    constexpr int LPR = 4;                               // lanes per row
    constexpr int GPL = 4;                               // 8-column groups per lane: l4, l4 + 4, l4 + 8, l4 + 12
    __shared__ int16_t ent_row[kRowsMaxEnt];             // winning row of entry j (document-relative), -1: not in this workgroup's rows
    __shared__ int16_t slot[kRowsMaxEnt], seg[kRowsMaxEnt];   // entries bucketed by row: in atomic order, then in entry order
    __shared__ int cnt[kRowsMaxRows + 1];                // per row: count, then the scatter cursor
    __shared__ uint16_t start[kRowsMaxRows + 1];
    __shared__ int pq[kRowsMaxPairs], pp[kRowsMaxPairs];  // per pair of this document: query index, pair index
    __shared__ float pg[kRowsMaxPairs];
    __shared__ int wsum[kRowsThreads / 64], wsum2[kRowsThreads / 64];
    __shared__ int guess_ok;
    __shared__ int cls_cnt[8];
    __shared__ uint16_t row_pos[kRowsMaxRows], row_perm[kRowsMaxRows];
    const int c = blockIdx.x;
    const int len = d_off[c + 1] - d_off[c];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = (len + (int)gridDim.y - 1) / (int)gridDim.y;
    const int r_lo = (int)blockIdx.y * per;
    const int r_hi = r_lo + per < len ? r_lo + per : len;
    if (r_lo >= r_hi) return;
    const int n_rows = r_hi - r_lo;
    const int col0 = blockIdx.z * 128;
    const int ncol = dim - col0 < 128 ? dim - col0 : 128;
    const int groups = ncol >> 3;                        // 8-column groups per row of this chunk (<= 16)
    ROWS_STAMP(0);
    const float up = load_gscale(gs);
    auto doc_of = [&](int k) { return pairs[2 * order_by_doc[k] + 1]; };
    // ---- the document's range [s, e) in the by-document order.  First a GUESS that four probes can prove: when the list gives every
    // document the same number of pairs (all pairs: ColbertLoss; every explicit-negative list) document c owns positions c * per_doc ..;
    // the list is sorted by document, so doc(s - 1) < c == doc(s) == doc(e - 1) < doc(e) proves it.  The pairs of the guessed range are
    // loaded in the same round trip.  Otherwise one counting pass of the workgroup (short lists) or two wave-wide binary searches
    // (6 + 6 dependent round trips for 8192 pairs: ~8 us in front of everything else -- what the guess is for).
    int s = 0, e = 0;
    bool have = false;
    const int per_doc = a.n_d > 0 ? a.n_pairs / a.n_d : 0;
    if (per_doc > 0 && per_doc * a.n_d == a.n_pairs && per_doc <= kRowsMaxPairs) {
        const int gs0 = c * per_doc, ge0 = gs0 + per_doc;
        if (t == 0) guess_ok = 1;
        __syncthreads();
        if (t < 4) {
            const int pos = t == 0 ? gs0 - 1 : t == 1 ? gs0 : t == 2 ? ge0 - 1 : ge0;
            bool ok = true;
            if (pos >= 0 && pos < a.n_pairs) {
                const int d = doc_of(pos);
                ok = t == 0 ? d < c : t == 3 ? d > c : d == c;
            }
            if (!ok) guess_ok = 0;
        }
        for (int k = t; k < per_doc; k += kRowsThreads) {
            const int p = order_by_doc[gs0 + k];
            pp[k] = p;
            pq[k] = pairs[2 * p];
            pg[k] = g[p] * up;
        }
        __syncthreads();
        if (guess_ok) { s = gs0; e = ge0; have = true; }
        __syncthreads();
    }
    const bool pairs_loaded = have;
    if (!have && a.n_pairs <= kRowsCountPairs) {
        int below = 0, mine = 0;
        for (int k0 = t; k0 < a.n_pairs; k0 += 8 * kRowsThreads) {
            int d[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {                // eight independent (order -> pair) chains in flight
                const int k = k0 + u * kRowsThreads;
                d[u] = k < a.n_pairs ? doc_of(k) : 0x7fffffff;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                below += d[u] < c;
                mine += d[u] == c;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            below += __shfl_xor(below, o);
            mine += __shfl_xor(mine, o);
        }
        if (lane == 0) { wsum[wave] = below; wsum2[wave] = mine; }
        __syncthreads();
        int m = 0;
#pragma unroll
        for (int w = 0; w < kRowsThreads / 64; ++w) { s += wsum[w]; m += wsum2[w]; }
        e = s + m;
        have = true;
        __syncthreads();                                 // wsum is used again by the scan
    }
    if (!have) {
        s = lower_bound_wave(a.n_pairs, c, lane, doc_of);
        e = lower_bound_wave(a.n_pairs, c + 1, lane, doc_of);
    }
    ROWS_STAMP(1);
    const int n_pd = e - s;
    const int n_ent = n_pd * a.Lq;
    char *out_doc = static_cast<char *>(dD) + (size_t)d_off[c] * dim * OES;
    const int l4 = t & (LPR - 1);                        // this lane's 8-column groups: l4 + 4 * u
    auto store8 = [&](int r, int grp, const float *acc) {
        char *o = out_doc + ((size_t)r * dim + col0 + grp * 8) * OES;
        if constexpr (OUT16) {
            *reinterpret_cast<uint4 *>(o) = pack8<DT>(acc);
        } else {
            *reinterpret_cast<float4 *>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            *reinterpret_cast<float4 *>(o + 16) = make_float4(acc[4], acc[5], acc[6], acc[7]);
        }
    };
    auto store_row = [&](int r, const float *acc) {
#pragma unroll
        for (int u = 0; u < GPL; ++u)
            if (l4 + LPR * u < groups) store8(r, l4 + LPR * u, acc + 8 * u);
    };
    // 8 columns of a query row as floats
    auto load8 = [&](size_t qrow, int grp, float *f) {
        const char *p = Q + (qrow * dim + col0 + grp * 8) * ES;
        if constexpr (DT == kDtypeF32) {
            const float4 x = *reinterpret_cast<const float4 *>(p), y = *reinterpret_cast<const float4 *>(p + 16);
            f[0] = x.x; f[1] = x.y; f[2] = x.z; f[3] = x.w; f[4] = y.x; f[5] = y.y; f[6] = y.z; f[7] = y.w;
        } else {
            piece_to_floats<DT>(*reinterpret_cast<const uint4 *>(p), f);
        }
    };
    // the lane's 32 columns of a query row (groups beyond the chunk: zeros)
    auto load_row = [&](size_t qrow, float *f) {
#pragma unroll
        for (int u = 0; u < GPL; ++u) {
            if (l4 + LPR * u < groups) {
                load8(qrow, l4 + LPR * u, f + 8 * u);
            } else {
#pragma unroll
                for (int v = 0; v < 8; ++v) f[8 * u + v] = 0.0f;
            }
        }
    };
    if (n_ent == 0) {                                    // nothing lands in this document: rows of zeros
        float z[8 * GPL];
#pragma unroll
        for (int u = 0; u < 8 * GPL; ++u) z[u] = 0.0f;
        for (int r = r_lo + t / LPR; r < r_hi; r += kRowsThreads / LPR) store_row(r, z);
        return;
    }
    if (n_ent > kRowsMaxEnt || n_pd > kRowsMaxPairs || n_rows > kRowsMaxRows) {
        // beyond the LDS lists (a pair list with many duplicates; the host routes every list it can bound to the right kernel): the
        // direct walk, correct and slow -- every row scans the document's entries in list order
        for (int r = r_lo + t / LPR; r < r_hi; r += kRowsThreads / LPR) {
            float acc[8 * GPL];
#pragma unroll
            for (int u = 0; u < 8 * GPL; ++u) acc[u] = 0.0f;
            for (int k = 0; k < n_pd; ++k) {
                const int p = order_by_doc[s + k];
                const float gp = g[p] * up;
                const int qb = pairs[2 * p];
                for (int i = 0; i < a.Lq; ++i)
                    if (argmax[(size_t)p * a.Lq + i] == r) {
                        float f[8 * GPL];
                        load_row((size_t)qb * a.Lq + i, f);
#pragma unroll
                        for (int u = 0; u < 8 * GPL; ++u) acc[u] += gp * f[u];
                    }
            }
            store_row(r, acc);
        }
        return;
    }
    // ---- the document's pairs and counts
    if (!pairs_loaded) {
        for (int k = t; k < n_pd; k += kRowsThreads) {
            const int p = order_by_doc[s + k];
            pp[k] = p;
            pq[k] = pairs[2 * p];
            pg[k] = g[p] * up;
        }
    }
    for (int r = t; r <= n_rows; r += kRowsThreads) cnt[r] = 0;
    __syncthreads();
    ROWS_STAMP(2);
    // ---- entries: winning rows, histogram of this workgroup's rows
    for (int j = t; j < n_ent; j += kRowsThreads) {
        const int k = j / a.Lq, i = j - k * a.Lq;
        const int r = argmax[(size_t)pp[k] * a.Lq + i] - r_lo;
        const bool mine = r >= 0 && r < n_rows;
        ent_row[j] = mine ? (int16_t)r : (int16_t)-1;
        if (mine) atomicAdd(&cnt[r], 1);
    }
    __syncthreads();
    ROWS_STAMP(3);
    // ---- exclusive scan of the counts: thread t owns rows 2t, 2t + 1 (kRowsMaxRows = 2 * kRowsThreads)
    {
        const int c0 = 2 * t < n_rows ? cnt[2 * t] : 0, c1 = 2 * t + 1 < n_rows ? cnt[2 * t + 1] : 0;
        int v = c0 + c1;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o);
            if (lane >= o) incl += u;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        const int excl = base + incl - v;
        if (2 * t < n_rows) { start[2 * t] = (uint16_t)excl; cnt[2 * t] = excl; }
        if (2 * t + 1 < n_rows) { start[2 * t + 1] = (uint16_t)(excl + c0); cnt[2 * t + 1] = excl + c0; }
        if (2 * t == n_rows || 2 * t + 1 == n_rows) start[n_rows] = (uint16_t)(2 * t == n_rows ? excl : excl + c0);
    }
    __syncthreads();
    ROWS_STAMP(4);
    // ---- scatter into the rows' segments (atomic order), then every segment into entry order
    for (int j = t; j < n_ent; j += kRowsThreads) {
        const int r = ent_row[j];
        if (r >= 0) slot[atomicAdd(&cnt[r], 1)] = (int16_t)j;
    }
    __syncthreads();
    // ---- rows grouped by their number of entries (0, 1, 2, 3, 4+): the 16 rows a wave works on at a time then loop equally often -- with
    // the rows in natural order a wave's loop ran for its LONGEST row (4-5 entries where the average is 1.3) and the kernel, which is
    // bound by instruction issue (sixteen waves per CU, ~500 instructions per 16 rows), spent two thirds of its slots on masked lanes.
    // Which row of a class comes first is irrelevant to the result (rows are independent), so atomic slots are fine here.
    if (t < 8) cls_cnt[t] = 0;
    __syncthreads();
    for (int r = t; r < n_rows; r += kRowsThreads) {
        const int n = start[r + 1] - start[r];
        const int cl = n < 4 ? n : 4;
        row_pos[r] = (uint16_t)(atomicAdd(&cls_cnt[cl], 1) | (cl << 12));        // position inside the class, class in the top bits
    }
    __syncthreads();
    for (int r = t; r < n_rows; r += kRowsThreads) {
        const int cl = row_pos[r] >> 12;
        int base = 0;                                    // classes in descending order of work: 4+, 3, 2, 1, 0
#pragma unroll
        for (int k = 4; k > 0; --k) base += k > cl ? cls_cnt[k] : 0;
        row_perm[base + (row_pos[r] & 0xfff)] = (uint16_t)r;
    }
    __syncthreads();
    ROWS_STAMP(5);
    const bool lq_pow2 = (a.Lq & (a.Lq - 1)) == 0;
    const int lq_shift = 31 - __builtin_clz(a.Lq | 1);
    const float inv_lq = 1.0f / (float)a.Lq;
    auto split_entry = [&](int j, int &k, int &i) {      // j = k * Lq + i, j < 4096: a shift, or an exact float quotient with one correction
        if (lq_pow2) {
            k = j >> lq_shift;
        } else {
            k = (int)(((float)j + 0.5f) * inv_lq);
            k -= k * a.Lq > j;
            k += (k + 1) * a.Lq <= j;
        }
        i = j - k * a.Lq;
    };
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    for (int idx = t / LPR; idx < n_rows; idx += kRowsThreads / LPR) {
        const int r = row_perm[idx];
        const int b0 = start[r], n = start[r + 1] - b0;
        // rank every entry of the segment among the segment (entries are distinct): seg[b0 + rank] = entry; the lanes share the work
        if (n > 1) {
            for (int x = l4; x < n; x += LPR) {
                const int mine = slot[b0 + x];
                int rank = 0;
                for (int y = 0; y < n; ++y) rank += slot[b0 + y] < mine;
                seg[b0 + rank] = (int16_t)mine;
            }
        }
        // (the same wave wrote and reads: LDS operations of a wave complete in order)
        f32x2 acc[4 * GPL];
#pragma unroll
        for (int u = 0; u < 4 * GPL; ++u) acc[u] = f32x2{0.f, 0.f};
        for (int x = 0; x < n; ++x) {
            const int j = n > 1 ? seg[b0 + x] : slot[b0];
            int k, i;
            split_entry(j, k, i);
            const float gp = pg[k];
            const size_t qrow = (size_t)pq[k] * a.Lq + i;
            if constexpr (DT == kDtypeF32) {
                float f[8 * GPL];
                load_row(qrow, f);
#pragma unroll
                for (int u = 0; u < 4 * GPL; ++u) acc[u] += f32x2{gp, gp} * f32x2{f[2 * u], f[2 * u + 1]};
            } else {
                uint4 raw[GPL];
#pragma unroll
                for (int u = 0; u < GPL; ++u)
                    raw[u] = l4 + LPR * u < groups ? *reinterpret_cast<const uint4 *>(Q + (qrow * dim + col0 + (l4 + LPR * u) * 8) * ES)
                                                   : make_uint4(0, 0, 0, 0);
#pragma unroll
                for (int u = 0; u < GPL; ++u) {
                    const uint32_t w[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        f32x2 q2;
                        if constexpr (DT == kDtypeF16) {
                            q2 = f32x2{elem_to_float<true>((uint16_t)(w[v] & 0xffffu)), elem_to_float<true>((uint16_t)(w[v] >> 16))};
                        } else {
                            q2 = f32x2{__uint_as_float(w[v] << 16), __uint_as_float(w[v] & 0xffff0000u)};
                        }
                        acc[4 * u + v] += f32x2{gp, gp} * q2;
                    }
                }
            }
        }
        float out[8 * GPL];
#pragma unroll
        for (int u = 0; u < 4 * GPL; ++u) { out[2 * u] = acc[u].x; out[2 * u + 1] = acc[u].y; }
        store_row(r_lo + r, out);
        ROWS_STAMP(6 + (idx >= kRowsThreads / LPR ? 1 : 0));
    }
    ROWS_STAMP(8);
}

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
