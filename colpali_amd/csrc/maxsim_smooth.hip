// Smooth-max late interaction for gfx950 (training losses with use_smooth_max=True):
//
//   colpali_engine/loss/late_interaction_losses.py:40-44   _smooth_max = tau * logsumexp(scores / tau, dim)
//   :88-90  _aggregate(use_smooth_max=True): smooth-max over document rows, then the sum over query tokens
//
//     score[b,c] = sum_n tau * log sum_s exp(<Q[b,n], D[c,s]> / tau)
//
// and its backward: d score[b,c] / d sim[b,c,n,s] = softmax_s(sim[b,c,n,:] / tau) =: w[b,c,n,s]
//     dQ[b,n,:] = sum_c G[b,c] sum_s w[b,c,n,s] D[c,s,:]
//     dD[c,s,:] = sum_b G[b,c] sum_n w[b,c,n,s] Q[b,n,:]
// Nothing of size [B,C,Lq,Ld] is materialised: the forward is an online logsumexp on the MFMA accumulators,
// the backward recomputes the similarity tile, turns it into weights with the saved per-token logsumexp and
// feeds it straight back into a second MFMA as the A operand (no cross-lane movement: the C layout of the
// first product IS the A layout of the second when the contraction index is walked in accumulator-register
// order).  Every document row takes part (zero padding rows of a dense [C, Ld, dim] tensor contribute
// exp(0), exactly like in the reference, which reduces over the padded dimension).
//
// Dtype-generic and width-generic (row_bytes % 32 == 0, see maxsim_generic.hip).  The second product runs on the
// exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) for every input dtype: the weights stay fp32.
// Deterministic: fixed work split, fixed-order cross-wave reduction through LDS, no atomics.
#pragma once
#include "maxsim_common.hpp"
#include "maxsim_generic.hip"

namespace msim {

struct SmoothArgs {
    long long ld;
    int n_q, Lq, n_d;
    int row_bytes;
    float tau;
};

// running (max, sum of exp) update with the 16 accumulator values of one 32x32 tile, x = sim / tau
__device__ __forceinline__ void lse_update16(float &m, float &l, const f32x16 &x) {
    float mx = m;
    mx = fold_max16(mx, x);
    const float ref = mx == -INFINITY ? 0.0f : mx;   // every value masked so far: keep l = 0 without producing NaN
    float s = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += expf(x[r] - ref);
    l = l * expf(m - ref) + s;
    m = mx;
}

// combine the two lane halves (lane, lane ^ 32 hold different document rows of the same token): natural-log LSE
__device__ __forceinline__ float lse_finish(float m, float l) {
    const float m2 = __shfl_xor(m, 32), l2 = __shfl_xor(l, 32);
    const float M = fmaxf(m, m2);
    const float ref = M == -INFINITY ? 0.0f : M;
    const float L = l * expf(m - ref) + l2 * expf(m2 - ref);
    return ref + logf(L);
}

// ---------------------------------------------------------------------------------------------------------
// Dense forward: scores[q, c] for every (query, document).  Same blocking as maxsim_generic_kernel.
template <int DT, int T>
__global__ __launch_bounds__(kGenericWaves * 64) void maxsim_smooth_kernel(const char *__restrict__ Q,
                                                                           const char *__restrict__ D,
                                                                           const int32_t *__restrict__ d_off,
                                                                           float *__restrict__ scores, SmoothArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row_bytes = a.row_bytes;
    const int q_stride = row_bytes + 16;
    const int tile_bytes = kTokTile * q_stride;
    const int n_steps = row_bytes >> 5;
    const int n16 = row_bytes >> 4;
    const int tpq = (a.Lq + kTokTile - 1) / kTokTile;
    const bool whole = tpq <= T;
    const int qpg = whole ? T / tpq : 1;
    const int n_pass = whole ? 1 : (tpq + T - 1) / T;
    const int g = blockIdx.y;
    const int gw = blockIdx.x * kGenericWaves + wave, GW = gridDim.x * kGenericWaves;
    const int half_off = (lane >> 5) * 16;
    const float inv_tau = 1.0f / a.tau;

    for (int pass = 0; pass < n_pass; ++pass) {
        if (pass > 0) __syncthreads();
        for (int idx = threadIdx.x; idx < T * kTokTile * n16; idx += kGenericWaves * 64) {
            const int t = idx / (kTokTile * n16);
            const int rem = idx - t * (kTokTile * n16);
            const int r = rem / n16, p = rem - r * n16;
            const int q = whole ? g * qpg + t / tpq : g;
            const int tt = whole ? t % tpq : pass * T + t;
            const int tok = tt * kTokTile + r;
            const bool valid = (whole ? t < qpg * tpq : tt < tpq) && q < a.n_q && tok < a.Lq;
            i32x4 v = {0, 0, 0, 0};
            if (valid) v = *reinterpret_cast<const i32x4 *>(Q + ((size_t)q * a.Lq + tok) * row_bytes + p * 16);
            *reinterpret_cast<i32x4 *>(smem + t * tile_bytes + r * q_stride + p * 16) = v;
        }
        __syncthreads();
        const char *q_lds = smem + (lane & 31) * q_stride + half_off;

        for (int c = gw; c < a.n_d; c += GW) {
            const int r0 = d_off[c];
            const int len = d_off[c + 1] - r0;
            const char *doc = D + (size_t)r0 * row_bytes;
            float m[T], l[T];
#pragma unroll
            for (int t = 0; t < T; ++t) { m[t] = -INFINITY; l[t] = 0.0f; }
            for (int s0 = 0; s0 < len; s0 += kSlabRows) {
                int row = s0 + (lane & 31);
                row = row < len ? row : len - 1;
                const char *arow = doc + (size_t)row * row_bytes + half_off;
                f32x16 acc[T];
#pragma unroll
                for (int t = 0; t < T; ++t) acc[t] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                int j = 0;
#pragma unroll 1
                for (; j + 4 <= n_steps; j += 4) {
                    bf16x8 av[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) av[u] = *reinterpret_cast<const bf16x8 *>(arow + (j + u) * 32);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int t = 0; t < T; ++t) {
                            const bf16x8 bv = *reinterpret_cast<const bf16x8 *>(q_lds + t * tile_bytes + (j + u) * 32);
                            acc[t] = mfma_step<DT>(av[u], bv, acc[t]);
                        }
                }
#pragma unroll 1
                for (; j < n_steps; ++j) {
                    const bf16x8 av = *reinterpret_cast<const bf16x8 *>(arow + j * 32);
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        const bf16x8 bv = *reinterpret_cast<const bf16x8 *>(q_lds + t * tile_bytes + j * 32);
                        acc[t] = mfma_step<DT>(av, bv, acc[t]);
                    }
                }
                const int rows_left = len - s0;
#pragma unroll
                for (int t = 0; t < T; ++t) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[t][r] = (acc_row(r, lane) < rows_left) ? acc[t][r] * inv_tau : -INFINITY;
                    lse_update16(m[t], l[t], acc[t]);
                }
            }
            // token rows that do not exist (beyond Lq: tile rounding) must not contribute -- their LSE is log(len), not 0;
            // zero rows INSIDE the [n_q, Lq, dim] tensor do contribute, exactly as in the reference
            float tile_sum[T];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int tt = whole ? t % tpq : pass * T + t;
                const int tok = tt * kTokTile + (lane & 31);
                const float lf = lse_finish(m[t], l[t]);     // cross-lane: evaluated by every lane, selected afterwards
                tile_sum[t] = half_wave_sum((tok < a.Lq) ? a.tau * lf : 0.0f);
            }
            if (lane == 0) {
                if (whole) {
                    for (int qq = 0; qq < qpg; ++qq) {
                        const int q = g * qpg + qq;
                        if (q >= a.n_q) break;
                        float tot = 0.0f;
#pragma unroll
                        for (int t = 0; t < T; ++t)
                            if (t >= qq * tpq && t < (qq + 1) * tpq) tot += tile_sum[t];
                        scores[(size_t)q * a.ld + c] = tot;
                    }
                } else {
                    float *dst = scores + (size_t)g * a.ld + c;
                    float tot = pass > 0 ? *dst : 0.0f;
#pragma unroll
                    for (int t = 0; t < T; ++t)
                        if (pass * T + t < tpq) tot += tile_sum[t];
                    *dst = tot;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Pair-list forward: smooth score and per-token natural-log LSE (of sim / tau) for listed (query, document) pairs.
template <int DT>
__global__ __launch_bounds__(256) void maxsim_smooth_pairs_kernel(const char *__restrict__ Q, const char *__restrict__ D,
                                                                  const int32_t *__restrict__ d_off,
                                                                  const int32_t *__restrict__ pairs,
                                                                  float *__restrict__ out_scores,   // [n_pairs] or null
                                                                  float *__restrict__ out_lse,      // [n_pairs, Lq] or null
                                                                  PairsArgs a, int row_bytes, float tau) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gw = blockIdx.x * 4 + wave, GW = gridDim.x * 4;
    const int n_steps = row_bytes >> 5;
    const int tpq = (a.Lq + kTokTile - 1) / kTokTile;
    const int half_off = (lane >> 5) * 16;
    const float inv_tau = 1.0f / tau;

    for (int p = gw; p < a.n_pairs; p += GW) {
        const int q = pairs[2 * p], c = pairs[2 * p + 1];
        if (q < 0 || q >= a.n_q || c < 0 || c >= a.n_d) continue;
        const int r0 = d_off[c];
        const int len = d_off[c + 1] - r0;
        const char *doc = D + (size_t)r0 * row_bytes;
        float total = 0.0f;
        for (int tt = 0; tt < tpq; ++tt) {
            const int tok = tt * kTokTile + (lane & 31);
            const bool tok_valid = tok < a.Lq;
            const char *qrow = Q + ((size_t)q * a.Lq + (tok_valid ? tok : 0)) * row_bytes + half_off;
            float m = -INFINITY, l = 0.0f;
            for (int s0 = 0; s0 < len; s0 += kSlabRows) {
                int row = s0 + (lane & 31);
                row = row < len ? row : len - 1;
                const char *arow = doc + (size_t)row * row_bytes + half_off;
                f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
                for (int j = 0; j < n_steps; ++j) {
                    const bf16x8 av = *reinterpret_cast<const bf16x8 *>(arow + j * 32);
                    const bf16x8 bv = *reinterpret_cast<const bf16x8 *>(qrow + j * 32);
                    acc = mfma_step<DT>(av, bv, acc);
                }
                const int rows_left = len - s0;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = (acc_row(r, lane) < rows_left) ? acc[r] * inv_tau : -INFINITY;
                lse_update16(m, l, acc);
            }
            const float lse = lse_finish(m, l);
            if (out_lse != nullptr && lane < 32 && tok_valid) out_lse[(size_t)p * a.Lq + tok] = lse;
            total += half_wave_sum(tok_valid ? tau * lse : 0.0f);
        }
        if (out_scores != nullptr && lane == 0) out_scores[p] = total;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Backward.  One kernel body for both gradients; "owner" = the side whose gradient the workgroup produces (32 rows:
// a query token tile for dQ, a 32-row document slab for dD), "other" = the side that is reduced over.
//   S[o, t]  = <X_owner[o], X_other[t]>                        first MFMA (input dtype), owner -> lane column
//   W[o, t]  = g_p * exp(S / tau - lse[p, token])               token = o (dQ) or t (dD)
//   out[o,:] += sum_t W[o, t] * X_other[t, :]                   second MFMA (fp32, 32x32x2), 32 output columns per workgroup
// Workgroup = (owner tile, 32-column block); its 4 waves split the (pair, other tile) work list and are reduced in
// wave order through LDS.
struct SmoothBwdArgs {
    int n_q, Lq, n_d, n_pairs;
    int row_bytes, dim;     // dim = elements per row
    float tau;
};

// NW = waves per workgroup: 16 for dQ (few owner tiles, each with a long (pair, slab) list: more waves hide the dependent
// global loads of the recompute), 4 for dD (thousands of owner tiles).
constexpr int kSmoothWavesDQ = 16, kSmoothWavesDD = 4;

template <int DT, bool DQ>
__global__ __launch_bounds__(DQ ? kSmoothWavesDQ * 64 : kSmoothWavesDD * 64) void maxsim_smooth_bwd_kernel(const char *__restrict__ Q, const char *__restrict__ D,
                                                                const int32_t *__restrict__ d_off,
                                                                const int32_t *__restrict__ pairs,         // sorted by query
                                                                const int32_t *__restrict__ order_by_doc,  // pair ids sorted by doc
                                                                const float *__restrict__ g,               // [n_pairs]
                                                                const float *__restrict__ lse,             // [n_pairs, Lq]
                                                                float *__restrict__ out,                   // dQ or dD
                                                                SmoothBwdArgs a) {
    constexpr int ES = elem_size<DT>();
    constexpr int NW = DQ ? kSmoothWavesDQ : kSmoothWavesDD;
    __shared__ float red[NW - 1][16][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int half_off = half * 16;
    const int n_steps = a.row_bytes >> 5;
    const int row_bytes = a.row_bytes;
    const int col = blockIdx.z * 32 + l31;              // output column of this lane (second MFMA: B column, C column)
    const bool col_valid = col < a.dim;
    const float inv_tau = 1.0f / a.tau;

    // ---- owner tile
    const int own = blockIdx.x;                         // query index (DQ) or document index (DD)
    const int own_tile = blockIdx.y;                    // token tile (DQ) or 32-row slab (DD)
    int own_rows, own_len;                              // rows in the owner entity / valid rows of this tile
    const char *own_base;
    if constexpr (DQ) {
        own_len = a.Lq;
        own_base = Q + (size_t)own * a.Lq * row_bytes;
    } else {
        own_len = d_off[own + 1] - d_off[own];
        own_base = D + (size_t)d_off[own] * row_bytes;
    }
    own_rows = own_len - own_tile * 32;
    if (own_rows <= 0) return;
    const int orow = own_tile * 32 + (l31 < own_rows ? l31 : own_rows - 1);
    const char *own_frag = own_base + (size_t)orow * row_bytes + half_off;

    // ---- this owner's pair range
    int p_lo, p_hi;
    if constexpr (DQ) {
        p_lo = lower_bound_idx(a.n_pairs, own, [&](int k) { return pairs[2 * k]; });
        p_hi = lower_bound_idx(a.n_pairs, own + 1, [&](int k) { return pairs[2 * k]; });
    } else {
        auto doc_of = [&](int k) { return pairs[2 * order_by_doc[k] + 1]; };
        p_lo = lower_bound_idx(a.n_pairs, own, doc_of);
        p_hi = lower_bound_idx(a.n_pairs, own + 1, doc_of);
    }

    f32x16 acc2 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int item = 0;                                       // running index over (pair, other tile): wave w takes item % 4 == w
    for (int k = p_lo; k < p_hi; ++k) {
        const int p = DQ ? k : order_by_doc[k];
        const int oth = DQ ? pairs[2 * p + 1] : pairs[2 * p];      // the other entity: document (DQ) or query (DD)
        const float gp = g[p];
        int oth_len;
        const char *oth_base;
        if constexpr (DQ) {
            oth_len = d_off[oth + 1] - d_off[oth];
            oth_base = D + (size_t)d_off[oth] * row_bytes;
        } else {
            oth_len = a.Lq;
            oth_base = Q + (size_t)oth * a.Lq * row_bytes;
        }
        const int n_tiles = (oth_len + 31) >> 5;
        const float *lse_p = lse + (size_t)p * a.Lq;
        float lse_own = 0.0f;
        if constexpr (DQ) lse_own = lse_p[orow];        // token = owner row
        for (int ot = 0; ot < n_tiles; ++ot, ++item) {
            if (item % NW != wave) continue;
            const int t0 = ot * 32;
            const int t_rows = oth_len - t0;            // >= 1
            // ---- first product: A = other rows (-> accumulator registers), B = owner rows (-> lane column)
            const int trow = t0 + (l31 < t_rows ? l31 : t_rows - 1);
            const char *oth_frag = oth_base + (size_t)trow * row_bytes + half_off;
            f32x16 s = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
            for (int j = 0; j < n_steps; ++j) {
                const bf16x8 av = *reinterpret_cast<const bf16x8 *>(oth_frag + j * 32);
                const bf16x8 bv = *reinterpret_cast<const bf16x8 *>(own_frag + j * 32);
                s = mfma_step<DT>(av, bv, s);
            }
            // ---- weights (zero for rows that do not exist on either side) and the second product
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tr = acc_row(r, lane);                      // other row inside the tile held by this register
                const bool valid = tr < t_rows && l31 < own_rows;
                float ls = lse_own;
                if constexpr (!DQ) ls = lse_p[t0 + (tr < t_rows ? tr : 0)];   // token = other row
                const float w = valid ? gp * expf(s[r] * inv_tau - ls) : 0.0f;
                // B operand: X_other[t0 + row(r, half)][col]; the contraction index of this MFMA is k = half
                float b = 0.0f;
                if (col_valid && tr < t_rows) b = load_elem<DT>(oth_base + ((size_t)(t0 + tr) * a.dim + col) * ES);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b, acc2, 0, 0, 0);
            }
        }
    }

    // ---- fixed-order reduction over the 4 waves, then store: acc2[reg] of lane (col, half) = out[row(reg, half)][col]
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc2[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc2[r];
#pragma unroll
            for (int w = 0; w < NW - 1; ++w) v += red[w][r][lane];
            const int orow_out = acc_row(r, lane);
            if (col_valid && orow_out < own_rows) {
                const size_t base_row = DQ ? (size_t)own * a.Lq : (size_t)d_off[own];
                out[(base_row + own_tile * 32 + orow_out) * a.dim + col] = v;
            }
        }
    }
}

}  // namespace msim
