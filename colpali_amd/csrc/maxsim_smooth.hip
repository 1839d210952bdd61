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
#include "maxsim_pairs.hip"

namespace msim {

struct SmoothArgs {
    long long ld;
    int n_q, Lq, n_d;
    int row_bytes;
    float tau;
};

// exp through the hardware exp2 (v_exp_f32, ~1 ulp) instead of libm's range-reduced expf: the arguments are <= 0 differences to a
// running maximum, so the result is in [0, 1] and the extra rounding of x * log2(e) costs < 1e-6 relative on the terms that matter
// (the tests pin the scores to 1e-5 against a float64 logsumexp); 17 exponentials per 32 x 32 tile made expf the largest VALU item.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// running (max, sum of exp) update with the 16 accumulator values of one 32x32 tile, x = sim / tau
__device__ __forceinline__ void lse_update16(float &m, float &l, const f32x16 &x) {
    float mx = m;
    mx = fold_max16(mx, x);
    const float ref = mx == -INFINITY ? 0.0f : mx;   // every value masked so far: keep l = 0 without producing NaN
    float s = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += fast_exp(x[r] - ref);
    l = l * fast_exp(m - ref) + s;
    m = mx;
}

// combine the two lane halves (lane, lane ^ 32 hold different document rows of the same token): natural-log LSE
__device__ __forceinline__ float lse_finish(float m, float l) {
    const float m2 = __shfl_xor(m, 32), l2 = __shfl_xor(l, 32);
    const float M = fmaxf(m, m2);
    const float ref = M == -INFINITY ? 0.0f : M;
    const float L = l * fast_exp(m - ref) + l2 * fast_exp(m2 - ref);
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
                // 8 k-steps at a time: 16 independent loads in flight (clamped addresses, no select), then the MFMAs -- one memory
                // round trip per 8 steps instead of one per step
#pragma unroll 1
                for (int j0 = 0; j0 < n_steps; j0 += 8) {
                    bf16x8 av[8], bv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int j = j0 + u < n_steps ? j0 + u : n_steps - 1;
                        av[u] = *reinterpret_cast<const bf16x8 *>(arow + j * 32);
                        bv[u] = *reinterpret_cast<const bf16x8 *>(qrow + j * 32);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (j0 + u < n_steps) acc = mfma_step<DT>(av[u], bv[u], acc);
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
// The same pair-list forward for 128 x 16-bit rows on the K1s pipeline (one wave per pair, wave-private LDS ring filled by
// LDS-DMA, operands through ds_read): fragment-shaped global loads of the kernel above run at ~4 TB/s out of L2 / MALL,
// whole-row LDS-DMA at ~11 (measured on the arg-max twin of this kernel, maxsim_pairs_argmax_kernel).
template <int TPQ, bool F16>
__global__ __launch_bounds__(256, 2) void maxsim_smooth_pairs_stream_kernel(const uint16_t *__restrict__ Q,
                                                                         const uint16_t *__restrict__ D,
                                                                         const int32_t *__restrict__ d_off,
                                                                         const int32_t *__restrict__ pairs,
                                                                         float *__restrict__ out_scores,   // [n_pairs] or null
                                                                         float *__restrict__ out_lse,      // [n_pairs, Lq] or null
                                                                         PairsArgs a, float tau) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *ring = smem + wave * (kPairsRing * kSlabBytes);
    const int gw = blockIdx.x * 4 + wave;
    const int GW = gridDim.x * 4;
    const float inv_tau = 1.0f / tau;

    const int l16 = lane & 15, l4 = lane >> 4;
    int src_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) src_off[j] = l4 * kRowBytes + (((l16 ^ l4) ^ (j << 2)) << 4);
    int rd_off[kKSteps];
#pragma unroll
    for (int ks = 0; ks < kKSteps; ++ks) rd_off[ks] = slab_swizzled_off(lane & 31, 2 * ks + (lane >> 5));

    for (int p = gw; p < a.n_pairs; p += GW) {
        const int q = pairs[2 * p], c = pairs[2 * p + 1];
        if (q < 0 || q >= a.n_q || c < 0 || c >= a.n_d) continue;   // caller error: leave the outputs untouched
        bf16x8 qf[TPQ][kKSteps];
#pragma unroll
        for (int t = 0; t < TPQ; ++t) {
            const int row = t * kTokTile + (lane & 31);
            const bool valid = row < a.Lq;
            const uint16_t *qp = Q + ((size_t)q * a.Lq + (valid ? row : 0)) * kDim + (lane >> 5) * 8;
#pragma unroll
            for (int ks = 0; ks < kKSteps; ++ks) {
                bf16x8 v = *reinterpret_cast<const bf16x8 *>(qp + ks * 16);
                qf[t][ks] = valid ? v : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
        wait_vmcnt<0>();   // also retires every LDS-DMA / store of the previous pair
#pragma unroll
        for (int t = 0; t < TPQ; ++t)
#pragma unroll
            for (int ks = 0; ks < kKSteps; ++ks) asm volatile("" : "+v"(qf[t][ks]));

        const int r0 = d_off[c];
        const int len = d_off[c + 1] - r0;
        const int nslab = (len + kSlabRows - 1) / kSlabRows;
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void *)(D + (size_t)r0 * kDim), 0, len * kRowBytes, 0x00020000);
        int p_s = 0, p_slot = 0, c_slot = 0;
        auto produce = [&]() -> bool {
            if (p_s >= nslab) return false;
            char *dst = ring + p_slot * kSlabBytes;
            const int soff = p_s * kSlabBytes;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, MSIM_LDS(dst + i * 1024), 16, src_off[i & 3], soff + i * 1024, 0, 0);
            p_slot = (p_slot + 1 == kPairsRing) ? 0 : p_slot + 1;
            ++p_s;
            return true;
        };
#pragma unroll
        for (int i = 0; i < kPairsRing - 1; ++i) produce();

        float m[TPQ], l[TPQ];
#pragma unroll
        for (int t = 0; t < TPQ; ++t) { m[t] = -INFINITY; l[t] = 0.0f; }

        for (int s = 0; s < nslab; ++s) {
            if (produce()) wait_vmcnt<8 * (kPairsRing - 1)>(); else wait_vmcnt<0>();
            const char *src = ring + c_slot * kSlabBytes;
            c_slot = (c_slot + 1 == kPairsRing) ? 0 : c_slot + 1;
            bf16x8 af[kKSteps];
#pragma unroll
            for (int ks = 0; ks < kKSteps; ++ks) af[ks] = *reinterpret_cast<const bf16x8 *>(src + rd_off[ks]);
            const int rows_left = len - s * kSlabRows;
#pragma unroll
            for (int t = 0; t < TPQ; ++t) {
                f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < kKSteps; ++ks) acc = mfma32<F16>(af[ks], qf[t][ks], acc);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = (acc_row(r, lane) < rows_left) ? acc[r] * inv_tau : -INFINITY;
                lse_update16(m[t], l[t], acc);
            }
        }

        float total = 0.0f;
#pragma unroll
        for (int t = 0; t < TPQ; ++t) {
            const int tok = t * kTokTile + (lane & 31);
            const bool tok_valid = tok < a.Lq;
            const float lse = lse_finish(m[t], l[t]);
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
// round a FINITE fp32 value to the 16-bit dtype and back, without the NaN branch of bf16_round
template <int DT>
__device__ __forceinline__ float round_finite(float x) {
    if constexpr (DT == kDtypeF16) return (float)(_Float16)x;
    else {
        uint32_t u = __float_as_uint(x);
        u += 0x7fffu + ((u >> 16) & 1u);
        return __uint_as_float(u & 0xffff0000u);
    }
}

typedef __attribute__((ext_vector_type(2))) int i32x2;
template <int ES> struct BVec;
template <> struct BVec<4> { typedef i32x4 type; };
template <> struct BVec<2> { typedef i32x2 type; };

struct SmoothBwdArgs {
    int n_q, Lq, n_d, n_pairs;
    int row_bytes, dim;     // dim = elements per row
    float tau;
    int n_split;            // dQ: workgroups sharing one owner tile's pair list (partials reduced by smooth_reduce_kernel)
};

// NW = waves per workgroup: 8 for dQ (few owner tiles with long (pair, slab) lists), 4 for dD (thousands of owner tiles).
constexpr int kSmoothWavesDQ = 8, kSmoothWavesDD = 4;
constexpr int kSmoothCB = 4;                            // 32-column blocks per workgroup: the weights are computed once per 128 columns

// One workgroup = (owner tile, group of 128 output columns[, slice of the pair list]).  Lane l31 owns the 4 CONSECUTIVE
// columns 128*g + 4*l31 + {0,1,2,3} (column block cb of the second product = column 4*l31 + cb): one 8- or 16-byte load
// per other-row fetches its 4 operands, one 16-byte store per row writes its 4 results.
// HOIST: the rows are at most 8 k-steps (256 bytes) wide, so the owner tile's 8 operand fragments are loaded once per workgroup and
// kept in registers instead of being re-fetched for every (pair, other tile) item.
template <int DT, bool DQ, bool HOIST>
__global__ __launch_bounds__(DQ ? kSmoothWavesDQ * 64 : kSmoothWavesDD * 64) void maxsim_smooth_bwd_kernel(
    const char *__restrict__ Q, const char *__restrict__ D, const int32_t *__restrict__ d_off,
    const int32_t *__restrict__ pairs,         // sorted by query
    const int32_t *__restrict__ order_by_doc,  // pair ids sorted by doc
    const float *__restrict__ g,               // [n_pairs]
    const float *__restrict__ lse,             // [n_pairs, Lq]
    float *__restrict__ out,                   // dQ / dD, or the partial buffers [n_split][rows][dim] when n_split > 1
    SmoothBwdArgs a) {
    constexpr int ES = elem_size<DT>();
    constexpr int NW = DQ ? kSmoothWavesDQ : kSmoothWavesDD;
    __shared__ float red[NW - 1][kSmoothCB][16][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int half_off = half * 16;
    const int n_steps = a.row_bytes >> 5;
    const int row_bytes = a.row_bytes;
    const int col0 = blockIdx.z * (32 * kSmoothCB) + 4 * l31;      // first of this lane's 4 columns (dim % 4 == 0)
    const bool col_valid = col0 < a.dim;
    const int col_c = col_valid ? col0 : 0;
    const float inv_tau = 1.0f / a.tau;

    // ---- owner tile (and, for dQ, the slice of its pair list)
    const int own = DQ ? blockIdx.x / a.n_split : blockIdx.x;      // query index (DQ) or document index (DD)
    const int split = DQ ? blockIdx.x % a.n_split : 0;
    const int own_tile = blockIdx.y;                    // token tile (DQ) or 32-row slab (DD)
    int own_rows, own_len;
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

    bf16x8 own_reg[8];
    if constexpr (HOIST) {
#pragma unroll
        for (int u = 0; u < 8; ++u) own_reg[u] = *reinterpret_cast<const bf16x8 *>(own_frag + (u < n_steps ? u : n_steps - 1) * 32);
    }

    int p_lo, p_hi;
    if constexpr (DQ) {
        p_lo = lower_bound_idx(a.n_pairs, own, [&](int k) { return pairs[2 * k]; });
        p_hi = lower_bound_idx(a.n_pairs, own + 1, [&](int k) { return pairs[2 * k]; });
    } else {
        auto doc_of = [&](int k) { return pairs[2 * order_by_doc[k] + 1]; };
        p_lo = lower_bound_idx(a.n_pairs, own, doc_of);
        p_hi = lower_bound_idx(a.n_pairs, own + 1, doc_of);
    }

    f32x16 acc2[kSmoothCB];
#pragma unroll
    for (int cb = 0; cb < kSmoothCB; ++cb) acc2[cb] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int item = 0;                                       // running index over (pair, other tile): wave w takes item % NW == w
    for (int k = p_lo + split; k < p_hi; k += (DQ ? a.n_split : 1)) {
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
            // ---- the other operand of the second product first: 16 unconditional loads (4 columns each) from clamped,
            // always valid addresses and NO select on them (the weight of a row that does not exist is exactly 0, a column
            // beyond dim is never stored; a select would let the compiler sink each load under its own branch)
            typedef typename BVec<ES>::type bvec_t;     // 4 consecutive elements of a row: 16 B (fp32) or 8 B (16-bit)
            bvec_t bld[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tr = acc_row(r, lane);
                bld[r] = *reinterpret_cast<const bvec_t *>(oth_base + ((size_t)(t0 + (tr < t_rows ? tr : 0)) * a.dim + col_c) * ES);
            }
            // ---- first product: A = other rows (-> accumulator registers), B = owner rows (-> lane column)
            const int trow = t0 + (l31 < t_rows ? l31 : t_rows - 1);
            const char *oth_frag = oth_base + (size_t)trow * row_bytes + half_off;
            f32x16 s = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            // 8 k-steps at a time: their 16 operand loads are issued together with the 16 `bld` loads above (one memory round trip
            // per tile for 128 x 16-bit rows instead of nine)
#pragma unroll 1
            for (int j0 = 0; j0 < n_steps; j0 += 8) {
                bf16x8 av[8], bv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int j = j0 + u < n_steps ? j0 + u : n_steps - 1;
                    av[u] = *reinterpret_cast<const bf16x8 *>(oth_frag + j * 32);
                    if constexpr (HOIST) bv[u] = own_reg[u];
                    else bv[u] = *reinterpret_cast<const bf16x8 *>(own_frag + j * 32);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (j0 + u < n_steps) s = mfma_step<DT>(av[u], bv[u], s);
            }
            // ---- weights (zero for rows that do not exist on either side)
            float w[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tr = acc_row(r, lane);
                const bool valid = tr < t_rows && l31 < own_rows;
                float ls = lse_own;
                if constexpr (!DQ) ls = lse_p[t0 + (tr < t_rows ? tr : 0)];   // token = other row
                w[r] = valid ? gp * fast_exp(s[r] * inv_tau - ls) : 0.0f;
            }
            // ---- second product
            if constexpr (DT == kDtypeF32) {
                // fp32 embeddings: exact-fp32 MFMA, one k pair (rows row(r, 0), row(r, 1)) per instruction
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int cb = 0; cb < kSmoothCB; ++cb) {
                        const int bits = bld[r][cb];   // copy the element first: bit_cast applied to a vector-element lvalue reads element 0
                        acc2[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[r], __int_as_float(bits), acc2[cb], 0, 0, 0);
                    }
            } else {
                // 16-bit embeddings: the 16-bit MFMA (16x the fp32 rate).  The weights are split into a 16-bit head and a 16-bit
                // remainder (two MFMAs per k-step) so that they keep ~16 bits of mantissa; the other operand is the embedding
                // itself, exact.  Accumulator registers 0..7 / 8..15 of the first product are exactly the k-slices of k-step
                // 0 / 1 of the second one: element e of the slice <-> other row row(8*kk + e, half).
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    bf16x8 ah, al;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float hi = round_finite<DT>(w[8 * kk + e]);          // w is finite: branch-free rounding
                        const float lo = round_finite<DT>(w[8 * kk + e] - hi);
                        if constexpr (DT == kDtypeF16) {
                            ah[e] = __builtin_bit_cast(short, (_Float16)hi);
                            al[e] = __builtin_bit_cast(short, (_Float16)lo);
                        } else {
                            ah[e] = (short)(__float_as_uint(hi) >> 16);
                            al[e] = (short)(__float_as_uint(lo) >> 16);
                        }
                    }
#pragma unroll
                    for (int cb = 0; cb < kSmoothCB; ++cb) {
                        bf16x8 bo;                                                 // column 4*l31 + cb of the 8 rows of this k-slice
#pragma unroll
                        for (int e = 0; e < 8; ++e) bo[e] = (short)((uint32_t)bld[8 * kk + e][cb >> 1] >> ((cb & 1) * 16));
                        acc2[cb] = mfma32<DT == kDtypeF16>(ah, bo, acc2[cb]);
                        acc2[cb] = mfma32<DT == kDtypeF16>(al, bo, acc2[cb]);
                    }
                }
            }
        }
    }

    // ---- fixed-order reduction over the waves, then store: acc2[cb][reg] of lane (l31, half) = out[row(reg, half)][col0 + cb]
    if (wave > 0) {
#pragma unroll
        for (int cb = 0; cb < kSmoothCB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave - 1][cb][r][lane] = acc2[cb][r];
    }
    __syncthreads();
    if (wave == 0) {
        const size_t rows_total = DQ ? (size_t)a.n_q * a.Lq : (size_t)d_off[a.n_d];
        float *dst = out + (size_t)split * rows_total * a.dim;       // n_split == 1: the gradient itself
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            f32x4 v;
#pragma unroll
            for (int cb = 0; cb < kSmoothCB; ++cb) {
                float x = acc2[cb][r];
#pragma unroll
                for (int w2 = 0; w2 < NW - 1; ++w2) x += red[w2][cb][r][lane];
                v[cb] = x;
            }
            const int orow_out = acc_row(r, lane);
            if (col_valid && orow_out < own_rows) {
                const size_t base_row = DQ ? (size_t)own * a.Lq : (size_t)d_off[own];
                *reinterpret_cast<f32x4 *>(dst + (base_row + own_tile * 32 + orow_out) * a.dim + col0) = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// The same backward for 128 x 16-bit rows with the "other" tile STAGED: the generic kernel above fetches every other tile twice
// (once in MFMA-fragment form, 32-byte pieces of 32 rows, and once in column form) straight from L2; here the tile's 32 whole rows
// (8 KiB) arrive ONCE by LDS-DMA in a wave-private, double-buffered slot with the K1s swizzle, and both operand forms are read from
// LDS (ds_read_b128 for the first product, ds_read_b64 for the second).  The next item's tile and pair data are requested before the
// current item is computed.  Arithmetic, summation order and the cross-wave reduction are those of the generic kernel: the two are
// bit-identical.
constexpr int kSmoothStageBytes = 2 * kSlabBytes;      // per wave: two 8 KiB slots

template <bool F16, bool DQ>
__global__ __launch_bounds__(DQ ? kSmoothWavesDQ * 64 : kSmoothWavesDD * 64) void maxsim_smooth_bwd_staged_kernel(
    const char *__restrict__ Q, const char *__restrict__ D, const int32_t *__restrict__ d_off,
    const int32_t *__restrict__ pairs, const int32_t *__restrict__ order_by_doc, const float *__restrict__ g,
    const float *__restrict__ lse, float *__restrict__ out, SmoothBwdArgs a) {
    constexpr int DT = F16 ? kDtypeF16 : kDtypeBf16;
    constexpr int NW = DQ ? kSmoothWavesDQ : kSmoothWavesDD;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // NW x 16 KiB of tile slots, re-used for the final reduction
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int col0 = 4 * l31;                                       // dim = 128: one group of 128 columns, lane owns 4 of them
    const float inv_tau = 1.0f / a.tau;
    char *slots = smem + wave * kSmoothStageBytes;

    const int own = DQ ? blockIdx.x / a.n_split : blockIdx.x;
    const int split = DQ ? blockIdx.x % a.n_split : 0;
    const int own_tile = blockIdx.y;
    int own_len;
    const char *own_base;
    if constexpr (DQ) {
        own_len = a.Lq;
        own_base = Q + (size_t)own * a.Lq * kRowBytes;
    } else {
        own_len = d_off[own + 1] - d_off[own];
        own_base = D + (size_t)d_off[own] * kRowBytes;
    }
    const int own_rows = own_len - own_tile * 32;
    if (own_rows <= 0) return;
    const int orow = own_tile * 32 + (l31 < own_rows ? l31 : own_rows - 1);
    bf16x8 own_reg[kKSteps];
#pragma unroll
    for (int u = 0; u < kKSteps; ++u) own_reg[u] = *reinterpret_cast<const bf16x8 *>(own_base + (size_t)orow * kRowBytes + half * 16 + u * 32);

    int p_lo, p_hi;
    if constexpr (DQ) {
        p_lo = lower_bound_idx(a.n_pairs, own, [&](int k) { return pairs[2 * k]; });
        p_hi = lower_bound_idx(a.n_pairs, own + 1, [&](int k) { return pairs[2 * k]; });
    } else {
        auto doc_of = [&](int k) { return pairs[2 * order_by_doc[k] + 1]; };
        p_lo = lower_bound_idx(a.n_pairs, own, doc_of);
        p_hi = lower_bound_idx(a.n_pairs, own + 1, doc_of);
    }

    // ---- LDS addressing (K1s slab image: logical 16-byte chunk c of row r at chunk c ^ (r & 15))
    const int l16 = lane & 15, l4 = lane >> 4;
    int src_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) src_off[j] = l4 * kRowBytes + (((l16 ^ l4) ^ (j << 2)) << 4);
    int rd_off[kKSteps];
#pragma unroll
    for (int ks = 0; ks < kKSteps; ++ks) rd_off[ks] = slab_swizzled_off(l31, 2 * ks + half);

    // ---- this wave's items: (pair, other tile) with running index % NW == wave, in list order
    struct Item {
        const char *base;      // first row of the other entity
        const float *lse_p;    // per-token LSE of the pair
        float gp;
        int len, t0;
        bool valid;
    };
    int it_k = p_lo + split, it_ot = 0, it_ntiles = 0, it_item = 0;
    bool it_open = false;
    Item info{nullptr, nullptr, 0.0f, 0, 0, false};
    const int k_stride = DQ ? a.n_split : 1;
    auto next_item = [&]() -> Item {
        for (;;) {
            if (!it_open) {
                if (it_k >= p_hi) return Item{nullptr, nullptr, 0.0f, 0, 0, false};
                const int p = DQ ? it_k : order_by_doc[it_k];
                const int oth = DQ ? pairs[2 * p + 1] : pairs[2 * p];
                info.gp = g[p];
                info.lse_p = lse + (size_t)p * a.Lq;
                if constexpr (DQ) {
                    info.len = d_off[oth + 1] - d_off[oth];
                    info.base = D + (size_t)d_off[oth] * kRowBytes;
                } else {
                    info.len = a.Lq;
                    info.base = Q + (size_t)oth * a.Lq * kRowBytes;
                }
                it_ntiles = (info.len + 31) >> 5;
                it_ot = 0;
                it_open = true;
            }
            if (it_ot >= it_ntiles) {
                it_open = false;
                it_k += k_stride;
                continue;
            }
            const bool mine = (it_item % NW) == wave;
            Item r = info;
            r.t0 = it_ot * 32;
            r.valid = true;
            ++it_ot;
            ++it_item;
            if (mine) return r;
        }
    };
    auto issue = [&](const Item &it, int slot) {       // 8 LDS-DMA wave-instructions: rows t0 .. t0+31 (rows past the end read as zeros)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)it.base, 0, it.len * kRowBytes, 0x00020000);
        char *dst = slots + slot * kSlabBytes;
        const int soff = it.t0 * kRowBytes;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, MSIM_LDS(dst + i * 1024), 16, src_off[i & 3], soff + i * 1024, 0, 0);
    };

    f32x16 acc2[kSmoothCB];
#pragma unroll
    for (int cb = 0; cb < kSmoothCB; ++cb) acc2[cb] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    Item cur = next_item();
    if (cur.valid) issue(cur, 0);
    int slot = 0;
    while (cur.valid) {
        const int t_rows = cur.len - cur.t0;            // >= 1
        // ---- this item's LSE values first (ordinary loads), then the NEXT item's tile: waiting for all but the last 8 loads
        // leaves exactly that tile in flight under the arithmetic below
        float ls[16];
        if constexpr (DQ) {
            const float v = cur.lse_p[orow];
#pragma unroll
            for (int r = 0; r < 16; ++r) ls[r] = v;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tr = acc_row(r, lane);
                ls[r] = cur.lse_p[cur.t0 + (tr < t_rows ? tr : 0)];
            }
        }
        const Item nxt = next_item();
        if (nxt.valid) {
            issue(nxt, slot ^ 1);
            wait_vmcnt<8>();
        } else {
            wait_vmcnt<0>();
        }
        const char *tile = slots + slot * kSlabBytes;
        // ---- first product: A = other rows (from LDS), B = owner rows (registers)
        f32x16 sacc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < kKSteps; ++ks) {
            const bf16x8 av = *reinterpret_cast<const bf16x8 *>(tile + rd_off[ks]);
            sacc = mfma32<F16>(av, own_reg[ks], sacc);
        }
        float w[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int tr = acc_row(r, lane);
            const bool valid = tr < t_rows && l31 < own_rows;
            w[r] = valid ? cur.gp * fast_exp(sacc[r] * inv_tau - ls[r]) : 0.0f;
        }
        // ---- the other operand of the second product: columns 4*l31 .. +3 of other row row(r, half) = 8 bytes of logical chunk l31 >> 1
        i32x2 bld[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int tr = acc_row(r, lane);
            bld[r] = *reinterpret_cast<const i32x2 *>(tile + slab_swizzled_off(tr, l31 >> 1) + (l31 & 1) * 8);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 ah, al;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float hi = round_finite<DT>(w[8 * kk + e]);
                const float lo = round_finite<DT>(w[8 * kk + e] - hi);
                if constexpr (F16) {
                    ah[e] = __builtin_bit_cast(short, (_Float16)hi);
                    al[e] = __builtin_bit_cast(short, (_Float16)lo);
                } else {
                    ah[e] = (short)(__float_as_uint(hi) >> 16);
                    al[e] = (short)(__float_as_uint(lo) >> 16);
                }
            }
#pragma unroll
            for (int cb = 0; cb < kSmoothCB; ++cb) {
                bf16x8 bo;
#pragma unroll
                for (int e = 0; e < 8; ++e) bo[e] = (short)((uint32_t)bld[8 * kk + e][cb >> 1] >> ((cb & 1) * 16));
                acc2[cb] = mfma32<F16>(ah, bo, acc2[cb]);
                acc2[cb] = mfma32<F16>(al, bo, acc2[cb]);
            }
        }
        cur = nxt;
        slot ^= 1;
    }

    // ---- fixed-order reduction over the waves through the (now idle) tile slots, then store
    __syncthreads();
    float(*red)[kSmoothCB][16][64] = reinterpret_cast<float(*)[kSmoothCB][16][64]>(smem);
    if (wave > 0) {
#pragma unroll
        for (int cb = 0; cb < kSmoothCB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave - 1][cb][r][lane] = acc2[cb][r];
    }
    __syncthreads();
    if (wave == 0) {
        const size_t rows_total = DQ ? (size_t)a.n_q * a.Lq : (size_t)d_off[a.n_d];
        float *dst = out + (size_t)split * rows_total * kDim;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            f32x4 v;
#pragma unroll
            for (int cb = 0; cb < kSmoothCB; ++cb) {
                float x = acc2[cb][r];
#pragma unroll
                for (int w2 = 0; w2 < NW - 1; ++w2) x += red[w2][cb][r][lane];
                v[cb] = x;
            }
            const int orow_out = acc_row(r, lane);
            if (orow_out < own_rows) {
                const size_t base_row = DQ ? (size_t)own * a.Lq : (size_t)d_off[own];
                *reinterpret_cast<f32x4 *>(dst + (base_row + own_tile * 32 + orow_out) * kDim + col0) = v;
            }
        }
    }
}

// out[i] = sum_{s < n_split} partial[s][i] in split order (the deterministic second pass of the dQ pair-list split)
__global__ __launch_bounds__(256) void smooth_reduce_kernel(const float *__restrict__ partial, float *__restrict__ out, long long n,
                                                            int n_split) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    f32x4 v = *reinterpret_cast<const f32x4 *>(partial + i);
    for (int s2 = 1; s2 < n_split; ++s2) {
        const f32x4 u = *reinterpret_cast<const f32x4 *>(partial + (size_t)s2 * n + i);
        v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
    }
    *reinterpret_cast<f32x4 *>(out + i) = v;
}

}  // namespace msim
