// [B, C]-sized epilogue of the in-batch late-interaction losses, forward value AND the gradient with respect to the MaxSim
// scores in ONE launch, so that a training step has no host synchronisation and no chain of tiny launches:
//
//   colpali_engine/loss/late_interaction_losses.py
//     :296        lengths = (query_embeddings[:, :, 0] != 0).sum(dim=1)
//     :300-301    scores = scores / lengths[:, None]                      (_apply_normalization :46-71; its bound check only prints)
//     :303-307    pos-aware negative filtering: scores[b, c] *= filter_factor where scores[b, c] > filter_threshold * pos[b],
//                 c != pos_idx[b]                                           (_filter_high_negatives :93-107, in place)
//   ColbertPairwiseCELoss.forward :309-313
//                 pos = scores.diagonal(offset); top2 = scores.topk(2, dim=1).values
//                 neg = where(top2[:, 0] == pos, top2[:, 1], top2[:, 0]);  loss = softplus((neg - pos) / T).mean()
//   ColbertLoss.forward :164
//                 loss = cross_entropy(scores / T, pos_idx)
//   ColbertSigmoidLoss.forward :457-465   (round 6)
//                 sign = -1 everywhere, +1 at the flat positions pos_idx * (B + 1) of the [B * B] square;
//                 loss = softplus(-scores.view(-1) / T * sign).mean()          -- the square: C == B, hence offset == 0
//
// What autograd would derive is written out directly.  Pairwise: exactly two score entries per query carry a gradient (the
// positive and the selected negative): the kernel emits them as the pair list the backward kernels consume -- 2*B pairs, sorted
// by (query, doc), their coefficients dLoss/dscore for a unit upstream gradient, and the stable by-document permutation -- so the
// count never has to be read by the host (it is 2*B by construction; a tie that makes both entries the same element simply
// yields two pairs whose coefficients cancel).  InfoNCE: the dense G = (softmax - onehot) / (T * B), chained through the filter
// factor and the normalisation.  One workgroup per query row; the last workgroup to finish (ticket counter) folds the per-row
// terms in row order -- deterministic, no floating-point atomics -- and builds the by-document permutation.
// Ties between exactly equal scores: the lower document index ranks first (torch.topk leaves the choice open).
#pragma once
#include "maxsim_common.hpp"

namespace msim {

constexpr int kEpiThreads = 256;
constexpr int kEpiPairwise = 0, kEpiInfoNCE = 1, kEpiSigmoid = 2;

struct EpiArgs {
    long long ld;            // leading dimension of scores / G
    int B, C, Lq;
    int q_row_bytes;         // bytes between consecutive query tokens (width * element size)
    int q_elem_bytes;        // 2 or 4
    int q_is_f16;            // 2-byte elements: fp16 (1) or bf16 (0)
    int offset;
    int mode;
    int normalize, filter;
    float inv_T, filter_threshold, filter_factor;
};

__device__ __forceinline__ unsigned long long epi_key(float v, int idx) {   // larger key = better (value desc, index asc)
    uint32_t u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (uint32_t)(0x7fffffff - idx);
}

// reduction over the threads that share one row: the whole workgroup (WAVE = false: barriers + LDS) or one wave (WAVE = true:
// shuffles only, so different waves of a workgroup can work on different rows without meeting at a barrier)
template <bool WAVE, class T, class F>
__device__ __forceinline__ T epi_reduce(T v, F op, T *sh) {   // result valid in every thread of the group
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = op(v, (T)__shfl_xor(v, o));
    if constexpr (WAVE) {
        return v;
    } else {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        __syncthreads();
        if (lane == 0) sh[wave] = v;
        __syncthreads();
        T r = sh[0];
#pragma unroll
        for (int w = 1; w < kEpiThreads / 64; ++w) r = op(r, sh[w]);
        return r;
    }
}

struct EpiShared {
    unsigned long long u64[kEpiThreads / 64];
    float f[kEpiThreads / 64];
    int i[kEpiThreads / 64];
};

struct EpiRow {
    float loss, lo, hi;
    int doc0, doc1;        // pairwise: the documents of the row's two pair-list entries (2b, 2b + 1)
};

// the loss in the embeddings' own dtype (what the reference returns: bf16 in -> bf16 scalar), one rounding of the fp32 value
__device__ __forceinline__ void epi_store_loss(void *loss_out, int elem_bytes, bool f16, float v) {
    if (loss_out == nullptr) return;
    if (elem_bytes == 4) *static_cast<float *>(loss_out) = v;
    else if (f16) *static_cast<_Float16 *>(loss_out) = (_Float16)v;
    else *static_cast<__bf16 *>(loss_out) = (__bf16)v;
}

// One query row b, worked on by a group of `nthr` threads (this thread is number `tid` of it): lengths, bounds, the row's loss term,
// its gradient entries (pairwise: the two pair-list entries 2b, 2b + 1; InfoNCE: row b of G).
// :296 lengths: this thread's share of the query rows whose FIRST component is non-zero (-0.0 counts as zero, NaN as non-zero, like `!= 0`)
__device__ __forceinline__ int epi_count_tokens(int b, int tid, int nthr, const char *__restrict__ Q, const EpiArgs &a) {
    // eight independent loads in flight per thread: every token's first component is a cache line of its own, and a loop of
    // load -> test -> add is one memory round trip per iteration (13 of them per lane for a 780-token page: 26 us of a 38 us launch)
    constexpr int kInFlight = 8;
    const char *base = Q + (size_t)b * a.Lq * a.q_row_bytes;
    int cnt = 0;
    for (int n0 = tid; n0 < a.Lq; n0 += nthr * kInFlight) {
        uint32_t bits[kInFlight];
#pragma unroll
        for (int j = 0; j < kInFlight; ++j) {
            const int n = n0 + j * nthr;
            const char *p = base + (size_t)(n < a.Lq ? n : 0) * a.q_row_bytes;
            bits[j] = a.q_elem_bytes == 2 ? ((uint32_t) * reinterpret_cast<const uint16_t *>(p) & 0x7fffu)
                                          : (*reinterpret_cast<const uint32_t *>(p) & 0x7fffffffu);
        }
#pragma unroll
        for (int j = 0; j < kInFlight; ++j) cnt += (n0 + j * nthr < a.Lq) && bits[j] != 0;
    }
    return cnt;
}

// `srow`: the row's C raw scores (global memory, or the copy the small-batch kernel keeps in LDS); `length_in` >= 0: the row's token
// count if the caller has it already.
template <bool WAVE>
__device__ __forceinline__ EpiRow epi_row(int b, int tid, int nthr, const float *srow, int length_in, const char *__restrict__ Q,
                                          float *__restrict__ G, int32_t *__restrict__ pairs, float *__restrict__ coef,
                                          const EpiArgs &a, EpiShared *sh) {
    int length = length_in;
    if (length_in < 0) length = epi_reduce<WAVE, int>(epi_count_tokens(b, tid, nthr, Q, a), [](int x, int y) { return x + y; }, sh->i);
    const float len_f = (float)length;
    const int pos_idx = a.offset + b;
    auto norm = [&](float raw) { return a.normalize ? raw / len_f : raw; };
    const float pos = norm(srow[pos_idx]);
    const float limit = a.filter_threshold * pos;
    auto filtered = [&](int c, float s) { return a.filter && c != pos_idx && s > limit; };
    auto value = [&](int c) {               // the score the loss sees
        const float s = norm(srow[c]);
        return filtered(c, s) ? s * a.filter_factor : s;
    };

    // ---- bounds of the normalised scores (the reference prints when they leave [-tol, 1 + tol], :62-70)
    float lo = INFINITY, hi = -INFINITY;
    for (int c = tid; c < a.C; c += nthr) {
        const float s = norm(srow[c]);
        lo = fminf(lo, s);
        hi = fmaxf(hi, s);
    }
    lo = epi_reduce<WAVE, float>(lo, [](float x, float y) { return fminf(x, y); }, sh->f);
    hi = epi_reduce<WAVE, float>(hi, [](float x, float y) { return fmaxf(x, y); }, sh->f);

    float row_loss = 0.f;
    int doc0 = 0, doc1 = 0;
    const float inv_B = 1.0f / (float)a.B;
    if (a.mode == kEpiPairwise) {
        // top-2 of the row: (value desc, index asc)
        unsigned long long k1 = 0;
        for (int c = tid; c < a.C; c += nthr) {
            const unsigned long long k = epi_key(value(c), c);
            k1 = k > k1 ? k : k1;
        }
        k1 = epi_reduce<WAVE, unsigned long long>(k1, [](unsigned long long x, unsigned long long y) { return x > y ? x : y; }, sh->u64);
        const int i1 = 0x7fffffff - (int)(uint32_t)k1;
        unsigned long long k2 = 0;
        for (int c = tid; c < a.C; c += nthr) {
            if (c == i1) continue;
            const unsigned long long k = epi_key(value(c), c);
            k2 = k > k2 ? k : k2;
        }
        k2 = epi_reduce<WAVE, unsigned long long>(k2, [](unsigned long long x, unsigned long long y) { return x > y ? x : y; }, sh->u64);
        const int i2 = 0x7fffffff - (int)(uint32_t)k2;
        const float v1 = value(i1), v2 = value(i2);
        const bool first_is_pos = v1 == pos;                              // :311 exact float equality
        const int neg_idx = first_is_pos ? i2 : i1;
        const float neg = first_is_pos ? v2 : v1;
        const float x = (neg - pos) * a.inv_T;
        row_loss = x > 20.0f ? x : log1pf(expf(x));                       // F.softplus (beta 1, threshold 20)
        const float sig = x > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-x));
        const float up = sig * a.inv_T * inv_B;                           // dLoss / dneg = -dLoss / dpos
        float c_neg = up, c_pos = -up;
        if (filtered(neg_idx, norm(srow[neg_idx]))) c_neg *= a.filter_factor;
        if (a.normalize) { c_neg /= len_f; c_pos /= len_f; }
        const bool pos_first = pos_idx <= neg_idx;
        doc0 = pos_first ? pos_idx : neg_idx;
        doc1 = pos_first ? neg_idx : pos_idx;
        if (tid == 0) {
            const int e = 2 * b;
            pairs[2 * e] = b;
            pairs[2 * e + 1] = doc0;
            coef[e] = pos_first ? c_pos : c_neg;
            pairs[2 * e + 2] = b;
            pairs[2 * e + 3] = doc1;
            coef[e + 1] = pos_first ? c_neg : c_pos;
        }
    } else if (a.mode == kEpiSigmoid) {
        // :457-465: row b of the [B, B] square, +1 on its diagonal element (flat position pos_idx * (B + 1) with pos_idx = b + offset
        // and offset == 0 for a square), -1 elsewhere; the row's term is its share of the mean over B * C elements, times B (the fold
        // divides by B).  dLoss/dv = -sign / T * sigmoid(-v sign / T) / (B C), chained through the filter factor and the normalisation.
        const float inv_C = 1.0f / (float)a.C;
        float acc = 0.f;
        float *grow = G != nullptr ? G + (size_t)b * a.ld : nullptr;
        for (int c = tid; c < a.C; c += nthr) {
            const float s = norm(srow[c]);
            const bool f = filtered(c, s);
            const float v = f ? s * a.filter_factor : s;
            const float sign = c == pos_idx ? 1.0f : -1.0f;
            const float x = -v * a.inv_T * sign;
            acc += x > 20.0f ? x : log1pf(expf(x));                       // F.softplus (beta 1, threshold 20)
            if (grow != nullptr) {
                const float sig = x > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-x));
                float g = -sign * a.inv_T * sig * inv_B * inv_C;
                if (f) g *= a.filter_factor;
                if (a.normalize) g /= len_f;
                grow[c] = g;
            }
        }
        row_loss = epi_reduce<WAVE, float>(acc, [](float x, float y) { return x + y; }, sh->f) * inv_C;
    } else {
        // cross entropy of row b with target pos_idx: lse(v / T) - pos / T
        // Softmax without cancellation: d_c = logit_c - max (<= 0, the logit ONE rounded product everywhere: __fmul_rn keeps the
        // compiler from contracting it into the subtraction), p_c = exp(d_c) / sum.  The positive's gradient p_pos - 1 is formed as
        // -(sum over the OTHER documents) / sum: when the positive dominates (p_pos = 1 - 1e-10) that is still exact to fp32
        // relative precision, where exp(logit - lse) - 1 would return the rounding error of lse (~4e-6 at logits of 100).
        auto logit = [&](int c) { return __fmul_rn(value(c), a.inv_T); };
        float m = -INFINITY;
        for (int c = tid; c < a.C; c += nthr) m = fmaxf(m, logit(c));
        m = epi_reduce<WAVE, float>(m, [](float x, float y) { return fmaxf(x, y); }, sh->f);
        float se_others = 0.f;
        for (int c = tid; c < a.C; c += nthr)
            if (c != pos_idx) se_others += expf(logit(c) - m);
        se_others = epi_reduce<WAVE, float>(se_others, [](float x, float y) { return x + y; }, sh->f);
        const float d_pos = logit(pos_idx) - m;
        const float se = se_others + expf(d_pos);
        row_loss = logf(se) - d_pos;
        if (G != nullptr) {
            float *grow = G + (size_t)b * a.ld;
            const float scale = a.inv_T * inv_B / se;
            for (int c = tid; c < a.C; c += nthr) {
                const float s = norm(srow[c]);
                const bool f = filtered(c, s);
                float g = (c == pos_idx ? -se_others : expf(logit(c) - m)) * scale;
                if (f) g *= a.filter_factor;
                if (a.normalize) g /= len_f;
                grow[c] = g;
            }
        }
    }
    return EpiRow{row_loss, lo, hi, doc0, doc1};
}

__global__ __launch_bounds__(kEpiThreads) void loss_epilogue_kernel(const float *__restrict__ scores,      // [B, ld] raw MaxSim scores
                                                                    const char *__restrict__ Q,            // [B, Lq, width]
                                                                    float *__restrict__ G,                 // [B, ld] (InfoNCE) or null
                                                                    int32_t *__restrict__ pairs,           // [2B, 2] (pairwise)
                                                                    float *__restrict__ coef,              // [2B]
                                                                    int32_t *__restrict__ order,           // [2B]
                                                                    float *__restrict__ ws_rows,           // [3, B] scratch: loss, min, max
                                                                    unsigned int *__restrict__ ticket,     // zero before the first launch; left zero
                                                                    float *__restrict__ out,               // [3]: loss, min, max of the normalised scores
                                                                    void *__restrict__ loss_out,           // the loss in the embeddings' dtype, or null
                                                                    const int32_t *__restrict__ lengths,   // [B] token counts, or null
                                                                    EpiArgs a) {
    __shared__ EpiShared sh;
    __shared__ int sh_last;
    const int b = blockIdx.x, tid = threadIdx.x;
    const EpiRow r = epi_row<false>(b, tid, kEpiThreads, scores + (size_t)b * a.ld, lengths ? lengths[b] : -1, Q, G, pairs, coef, a, &sh);

    // ---- per-row terms -> scratch; the last workgroup to arrive folds them in row order
    if (tid == 0) {
        ws_rows[b] = r.loss;
        ws_rows[a.B + b] = r.lo;
        ws_rows[2 * a.B + b] = r.hi;
        __threadfence();
        sh_last = atomicAdd(ticket, 1u) == (unsigned)(a.B - 1);
    }
    __syncthreads();
    if (!sh_last) return;
    __threadfence();
    if (tid == 0) {
        float s = 0.f, mn = INFINITY, mx = -INFINITY;
        for (int i = 0; i < a.B; ++i) {
            s += __builtin_nontemporal_load(ws_rows + i);
            mn = fminf(mn, __builtin_nontemporal_load(ws_rows + a.B + i));
            mx = fmaxf(mx, __builtin_nontemporal_load(ws_rows + 2 * a.B + i));
        }
        const float inv_B = 1.0f / (float)a.B;
        out[0] = s * inv_B;
        out[1] = mn;
        out[2] = mx;
        epi_store_loss(loss_out, a.q_elem_bytes, a.q_is_f16 != 0, s * inv_B);
        *ticket = 0;                                                     // ready for the next launch
    }
    if (a.mode == kEpiPairwise) {
        // stable by-document permutation of the 2B pairs: rank[e] = #{e' : (doc[e'], e') < (doc[e], e)}
        const int n = 2 * a.B;
        for (int e = tid; e < n; e += kEpiThreads) {
            const int de = __builtin_nontemporal_load(pairs + 2 * e + 1);
            int rank = 0;
            for (int o = 0; o < n; ++o) {
                const int d_o = __builtin_nontemporal_load(pairs + 2 * o + 1);
                rank += (d_o < de) || (d_o == de && o < e);
            }
            order[rank] = e;
        }
    }
}

// The pairwise row on HALF a wave (32 lanes, the small-batch kernel: 32 rows at a time on 16 waves) in ONE pass over the row: every
// lane keeps the bounds and the best two keys of its elements, and one 5-step butterfly merges (min, max, top-2) of the 32 lanes --
// two sorted pairs merge as  k1 = max(a1, b1),  k2 = max(min(a1, b1), max(a2, b2)).  The same keys (value descending, index
// ascending), the same exact-equality rule and the same gradient entries as epi_row; three passes and four reductions fewer.
__device__ __forceinline__ EpiRow epi_row_pairwise_half(int b, int l32, const float *srow, int length, int32_t *__restrict__ pairs,
                                                        float *__restrict__ coef, const EpiArgs &a) {
    const float len_f = (float)length;
    const int pos_idx = a.offset + b;
    auto norm = [&](float raw) { return a.normalize ? raw / len_f : raw; };
    const float pos = norm(srow[pos_idx]);
    const float limit = a.filter_threshold * pos;
    auto filtered = [&](int c, float s) { return a.filter && c != pos_idx && s > limit; };
    float lo = INFINITY, hi = -INFINITY;
    unsigned long long k1 = 0, k2 = 0;
    for (int c = l32; c < a.C; c += 32) {
        const float s = norm(srow[c]);
        lo = fminf(lo, s);
        hi = fmaxf(hi, s);
        const unsigned long long k = epi_key(filtered(c, s) ? s * a.filter_factor : s, c);
        if (k > k1) { k2 = k1; k1 = k; }
        else if (k > k2) k2 = k;
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
        const unsigned long long o1 = __shfl_xor(k1, o), o2 = __shfl_xor(k2, o);
        const unsigned long long mx = k1 > o1 ? k1 : o1, mn = k1 > o1 ? o1 : k1, m2 = k2 > o2 ? k2 : o2;
        k1 = mx;
        k2 = mn > m2 ? mn : m2;
    }
    const int i1 = 0x7fffffff - (int)(uint32_t)k1, i2 = 0x7fffffff - (int)(uint32_t)k2;
    auto value = [&](int c) {
        const float s = norm(srow[c]);
        return filtered(c, s) ? s * a.filter_factor : s;
    };
    const float v1 = value(i1), v2 = value(i2);
    const bool first_is_pos = v1 == pos;                              // :311 exact float equality
    const int neg_idx = first_is_pos ? i2 : i1;
    const float neg = first_is_pos ? v2 : v1;
    const float x = (neg - pos) * a.inv_T;
    const float row_loss = x > 20.0f ? x : log1pf(expf(x));          // F.softplus (beta 1, threshold 20)
    const float sig = x > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-x));
    const float up = sig * a.inv_T / (float)a.B;                      // dLoss / dneg = -dLoss / dpos
    float c_neg = up, c_pos = -up;
    if (filtered(neg_idx, norm(srow[neg_idx]))) c_neg *= a.filter_factor;
    if (a.normalize) { c_neg /= len_f; c_pos /= len_f; }
    const bool pos_first = pos_idx <= neg_idx;
    const int doc0 = pos_first ? pos_idx : neg_idx, doc1 = pos_first ? neg_idx : pos_idx;
    if (l32 == 0) {
        const int e = 2 * b;
        pairs[2 * e] = b;
        pairs[2 * e + 1] = doc0;
        coef[e] = pos_first ? c_pos : c_neg;
        pairs[2 * e + 2] = b;
        pairs[2 * e + 3] = doc1;
        coef[e + 1] = pos_first ? c_neg : c_pos;
    }
    return EpiRow{row_loss, lo, hi, doc0, doc1};
}

// The dense-gradient rows (InfoNCE, sigmoid) on HALF a wave with the row's values kept in registers (round 6): one read of the row,
// one division / filter decision / exp per element instead of one per pass (epi_row re-derives `value(c)` in each of its five
// passes: ~1300 instructions per row and wave, 17 us for 32 x 256 scores on the small kernel's 16 waves), and two rows per wave at a
// time.  The same formulas as epi_row; sums run lane-strided over 32 lanes and a fixed 5-step butterfly.  C <= 32 * NE.
template <int NE>
__device__ __forceinline__ EpiRow epi_row_dense_half(int b, int l32, const float *srow, int length, float *__restrict__ G, const EpiArgs &a) {
    const float len_f = (float)length;
    const int pos_idx = a.offset + b;
    auto norm = [&](float raw) { return a.normalize ? raw / len_f : raw; };
    const float pos = norm(srow[pos_idx]);
    const float limit = a.filter_threshold * pos;
    const float inv_B = 1.0f / (float)a.B;
    float v[NE];
    unsigned fmask = 0;
    float lo = INFINITY, hi = -INFINITY;
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int c = l32 + 32 * u;
        const bool ok = c < a.C;
        const float s = norm(srow[ok ? c : pos_idx]);
        if (ok) { lo = fminf(lo, s); hi = fmaxf(hi, s); }
        const bool f = a.filter && c != pos_idx && s > limit;
        v[u] = f ? s * a.filter_factor : s;
        fmask |= (f ? 1u : 0u) << u;
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    float *grow = G != nullptr ? G + (size_t)b * a.ld : nullptr;
    float row_loss;
    if (a.mode == kEpiSigmoid) {
        const float inv_C = 1.0f / (float)a.C;
        float acc = 0.f;
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const int c = l32 + 32 * u;
            if (c < a.C) {
                const float sign = c == pos_idx ? 1.0f : -1.0f;
                const float x = -v[u] * a.inv_T * sign;
                acc += x > 20.0f ? x : log1pf(expf(x));                   // F.softplus (beta 1, threshold 20)
                if (grow != nullptr) {
                    const float sig = x > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-x));
                    float g = -sign * a.inv_T * sig * inv_B * inv_C;
                    if ((fmask >> u) & 1u) g *= a.filter_factor;
                    if (a.normalize) g /= len_f;
                    grow[c] = g;
                }
            }
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        row_loss = acc * inv_C;
    } else {
        float m = -INFINITY;
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            v[u] = __fmul_rn(v[u], a.inv_T);                              // the logit: ONE rounded product (see epi_row)
            if (l32 + 32 * u < a.C) m = fmaxf(m, v[u]);
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float se_others = 0.f;
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            const int c = l32 + 32 * u;
            v[u] = expf(v[u] - m);
            if (c < a.C && c != pos_idx) se_others += v[u];
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) se_others += __shfl_xor(se_others, o);
        const float d_pos = __fmul_rn(pos, a.inv_T) - m;                  // the positive is never filtered
        const float se = se_others + expf(d_pos);
        row_loss = logf(se) - d_pos;
        if (grow != nullptr) {
            const float scale = a.inv_T * inv_B / se;
#pragma unroll
            for (int u = 0; u < NE; ++u) {
                const int c = l32 + 32 * u;
                if (c < a.C) {
                    float g = (c == pos_idx ? -se_others : v[u]) * scale;
                    if ((fmask >> u) & 1u) g *= a.filter_factor;
                    if (a.normalize) g /= len_f;
                    grow[c] = g;
                }
            }
        }
    }
    return EpiRow{row_loss, lo, hi, 0, 0};
}

// ---- the same for SMALL batches (B <= kEpiSmallRows rows, B * C scores that one workgroup reads in a few microseconds -- BASELINE
// config 5: 32 x 256): ONE workgroup of 16 waves.  Phase 1, one round trip to memory: the whole score matrix is copied into LDS (when
// it fits kEpiStageFloats) and every wave counts the tokens of its rows.  Phase 2: one wave per row at a time (shuffle reductions
// only), every pass over the row out of LDS; the per-row terms stay in LDS, one barrier, and the fold / the by-document permutation
// read them from there.  No ticket, no scratch, no fence.  (History, rocprofv3 on the 150 us loss step of round 5: the multi-workgroup
// form spent 19 us here -- a lone thread walking the per-row terms and the pair list in global memory; the first one-workgroup form
// still 16 us -- each of its eight passes over a row was a dependent round trip to L2.)
constexpr int kEpiSmallThreads = 1024;
constexpr int kEpiSmallRows = 1024;
constexpr int kEpiStageFloats = 24576;          // 96 KiB of dynamic LDS

__global__ __launch_bounds__(kEpiSmallThreads) void loss_epilogue_small_kernel(const float *__restrict__ scores, const char *__restrict__ Q,
                                                                               const int32_t *__restrict__ lengths,   // [B] token counts, or null: counted here
                                                                               float *__restrict__ G, int32_t *__restrict__ pairs,
                                                                               float *__restrict__ coef, int32_t *__restrict__ order,
                                                                               float *__restrict__ out, void *__restrict__ loss_out, EpiArgs a,
                                                                               int staged) {
    extern __shared__ __attribute__((aligned(16))) float stage[];            // [B][C] when `staged`
    __shared__ float row_loss[kEpiSmallRows], row_lo[kEpiSmallRows], row_hi[kEpiSmallRows];
    __shared__ int pair_doc[2 * kEpiSmallRows];
    __shared__ int row_len[kEpiSmallRows];
    __shared__ EpiShared unused;                    // the wave-level reductions never touch it
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int kWaves = kEpiSmallThreads / 64;
    // Phase 1 is ONE round trip to memory when the matrix fits eight loads per thread (config 5: 32 x 256): the scores are loaded
    // into registers, the token counts' loads are issued behind them (a row per HALF wave: all rows of a 32-row batch at once -- with a
    // row per wave the second row's round trip followed the first's), and only then are the scores written to LDS.
    constexpr int kInFlight = 8;
    const int n = a.B * a.C;
    const bool one_round = staged && n <= kEpiSmallThreads * kInFlight;
    float v0[kInFlight];
    if (one_round) {
#pragma unroll
        for (int j = 0; j < kInFlight; ++j) {
            const int i = tid + j * kEpiSmallThreads;
            const int ic = i < n ? i : 0;
            const int r = ic / a.C, c = ic - r * a.C;
            v0[j] = scores[(size_t)r * a.ld + c];
        }
    } else if (staged) {
        // the whole matrix, up to eight loads in flight per thread before the first LDS write
        for (int i0 = tid; i0 < n; i0 += kEpiSmallThreads * kInFlight) {
            float v[kInFlight];
#pragma unroll
            for (int j = 0; j < kInFlight; ++j) {
                const int i = i0 + j * kEpiSmallThreads;
                const int ic = i < n ? i : 0;
                const int r = ic / a.C, c = ic - r * a.C;
                v[j] = scores[(size_t)r * a.ld + c];
            }
#pragma unroll
            for (int j = 0; j < kInFlight; ++j) {
                const int i = i0 + j * kEpiSmallThreads;
                if (i < n) stage[i] = v[j];
            }
        }
    }
    if (lengths != nullptr) {
        for (int b = tid; b < a.B; b += kEpiSmallThreads) row_len[b] = lengths[b];
    } else {
        const int l32 = lane & 31;
        for (int b = tid >> 5; b < a.B; b += kEpiSmallThreads / 32) {
            int cnt = epi_count_tokens(b, l32, 32, Q, a);
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
            if (l32 == 0) row_len[b] = cnt;
        }
    }
    if (one_round) {
#pragma unroll
        for (int j = 0; j < kInFlight; ++j) {
            const int i = tid + j * kEpiSmallThreads;
            if (i < n) stage[i] = v0[j];
        }
    }
    __syncthreads();
    if (a.mode == kEpiPairwise) {
        // pairwise: a row per HALF wave, one pass over the row (32 rows at a time)
        const int l32 = lane & 31;
        for (int b = tid >> 5; b < a.B; b += kEpiSmallThreads / 32) {
            const float *srow = staged ? stage + (size_t)b * a.C : scores + (size_t)b * a.ld;
            const EpiRow r = epi_row_pairwise_half(b, l32, srow, row_len[b], pairs, coef, a);
            if (l32 == 0) {
                row_loss[b] = r.loss;
                row_lo[b] = r.lo;
                row_hi[b] = r.hi;
                pair_doc[2 * b] = r.doc0;
                pair_doc[2 * b + 1] = r.doc1;
            }
        }
    } else if (a.C <= 512) {
        // dense gradients, rows of at most 512 documents: a row per HALF wave, its values in registers (32 rows at a time)
        const int l32 = lane & 31;
        for (int b = tid >> 5; b < a.B; b += kEpiSmallThreads / 32) {
            const float *srow = staged ? stage + (size_t)b * a.C : scores + (size_t)b * a.ld;
            const EpiRow r = a.C <= 256 ? epi_row_dense_half<8>(b, l32, srow, row_len[b], G, a) : epi_row_dense_half<16>(b, l32, srow, row_len[b], G, a);
            if (l32 == 0) {
                row_loss[b] = r.loss;
                row_lo[b] = r.lo;
                row_hi[b] = r.hi;
            }
        }
    } else {
        for (int b = wave; b < a.B; b += kWaves) {
            const float *srow = staged ? stage + (size_t)b * a.C : scores + (size_t)b * a.ld;
            const EpiRow r = epi_row<true>(b, lane, 64, srow, row_len[b], Q, G, pairs, coef, a, &unused);
            if (lane == 0) {
                row_loss[b] = r.loss;
                row_lo[b] = r.lo;
                row_hi[b] = r.hi;
            }
        }
    }
    __syncthreads();
    if (wave == 0) {
        // the mean over the rows: lane l adds rows l, l + 64, ... in that order, then a fixed butterfly -- an order that depends on B alone
        float s = 0.f, mn = INFINITY, mx = -INFINITY;
        for (int i = lane; i < a.B; i += 64) {
            s += row_loss[i];
            mn = fminf(mn, row_lo[i]);
            mx = fmaxf(mx, row_hi[i]);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s += __shfl_xor(s, o);
            mn = fminf(mn, __shfl_xor(mn, o));
            mx = fmaxf(mx, __shfl_xor(mx, o));
        }
        if (lane == 0) {
            const float inv_B = 1.0f / (float)a.B;
            out[0] = s * inv_B;
            out[1] = mn;
            out[2] = mx;
            epi_store_loss(loss_out, a.q_elem_bytes, a.q_is_f16 != 0, s * inv_B);
        }
    }
    if (a.mode == kEpiPairwise) {
        const int n = 2 * a.B;
        for (int e = tid; e < n; e += kEpiSmallThreads) {
            const int de = pair_doc[e];
            int rank = 0;
#pragma unroll 8
            for (int o = 0; o < n; ++o) {
                const int d_o = pair_doc[o];
                rank += (d_o < de) || (d_o == de && o < e);
            }
            order[rank] = e;
        }
    }
}

}  // namespace msim
