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
constexpr int kEpiPairwise = 0, kEpiInfoNCE = 1;

struct EpiArgs {
    long long ld;            // leading dimension of scores / G
    int B, C, Lq;
    int q_row_bytes;         // bytes between consecutive query tokens (width * element size)
    int q_elem_bytes;        // 2 or 4
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

template <class T, class F>
__device__ __forceinline__ T epi_block_reduce(T v, F op, T *sh) {   // result valid in every thread
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = op(v, (T)__shfl_xor(v, o));
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    T r = sh[0];
#pragma unroll
    for (int w = 1; w < kEpiThreads / 64; ++w) r = op(r, sh[w]);
    return r;
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
                                                                    EpiArgs a) {
    __shared__ unsigned long long sh_u64[kEpiThreads / 64];
    __shared__ float sh_f[kEpiThreads / 64];
    __shared__ int sh_i[kEpiThreads / 64];
    __shared__ int sh_last;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float *srow = scores + (size_t)b * a.ld;

    // ---- :296 lengths: query rows whose FIRST component is non-zero (-0.0 counts as zero, NaN as non-zero, like `!= 0`)
    int cnt = 0;
    for (int n = tid; n < a.Lq; n += kEpiThreads) {
        const char *p = Q + ((size_t)b * a.Lq + n) * a.q_row_bytes;
        const uint32_t bits = a.q_elem_bytes == 2 ? ((uint32_t) * reinterpret_cast<const uint16_t *>(p) & 0x7fffu)
                                                  : (*reinterpret_cast<const uint32_t *>(p) & 0x7fffffffu);
        cnt += bits != 0;
    }
    const int length = epi_block_reduce<int>(cnt, [](int x, int y) { return x + y; }, sh_i);
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
    for (int c = tid; c < a.C; c += kEpiThreads) {
        const float s = norm(srow[c]);
        lo = fminf(lo, s);
        hi = fmaxf(hi, s);
    }
    lo = epi_block_reduce<float>(lo, [](float x, float y) { return fminf(x, y); }, sh_f);
    hi = epi_block_reduce<float>(hi, [](float x, float y) { return fmaxf(x, y); }, sh_f);

    float row_loss = 0.f;
    const float inv_B = 1.0f / (float)a.B;
    if (a.mode == kEpiPairwise) {
        // top-2 of the row: (value desc, index asc)
        unsigned long long k1 = 0;
        for (int c = tid; c < a.C; c += kEpiThreads) {
            const unsigned long long k = epi_key(value(c), c);
            k1 = k > k1 ? k : k1;
        }
        k1 = epi_block_reduce<unsigned long long>(k1, [](unsigned long long x, unsigned long long y) { return x > y ? x : y; }, sh_u64);
        const int i1 = 0x7fffffff - (int)(uint32_t)k1;
        unsigned long long k2 = 0;
        for (int c = tid; c < a.C; c += kEpiThreads) {
            if (c == i1) continue;
            const unsigned long long k = epi_key(value(c), c);
            k2 = k > k2 ? k : k2;
        }
        k2 = epi_block_reduce<unsigned long long>(k2, [](unsigned long long x, unsigned long long y) { return x > y ? x : y; }, sh_u64);
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
        if (tid == 0) {
            const bool pos_first = pos_idx <= neg_idx;
            const int e = 2 * b;
            pairs[2 * e] = b;
            pairs[2 * e + 1] = pos_first ? pos_idx : neg_idx;
            coef[e] = pos_first ? c_pos : c_neg;
            pairs[2 * e + 2] = b;
            pairs[2 * e + 3] = pos_first ? neg_idx : pos_idx;
            coef[e + 1] = pos_first ? c_neg : c_pos;
        }
    } else {
        // cross entropy of row b with target pos_idx: lse(v / T) - pos / T
        // Softmax without cancellation: d_c = logit_c - max (<= 0, the logit ONE rounded product everywhere: __fmul_rn keeps the
        // compiler from contracting it into the subtraction), p_c = exp(d_c) / sum.  The positive's gradient p_pos - 1 is formed as
        // -(sum over the OTHER documents) / sum: when the positive dominates (p_pos = 1 - 1e-10) that is still exact to fp32
        // relative precision, where exp(logit - lse) - 1 would return the rounding error of lse (~4e-6 at logits of 100).
        auto logit = [&](int c) { return __fmul_rn(value(c), a.inv_T); };
        float m = -INFINITY;
        for (int c = tid; c < a.C; c += kEpiThreads) m = fmaxf(m, logit(c));
        m = epi_block_reduce<float>(m, [](float x, float y) { return fmaxf(x, y); }, sh_f);
        float se_others = 0.f;
        for (int c = tid; c < a.C; c += kEpiThreads)
            if (c != pos_idx) se_others += expf(logit(c) - m);
        se_others = epi_block_reduce<float>(se_others, [](float x, float y) { return x + y; }, sh_f);
        const float d_pos = logit(pos_idx) - m;
        const float se = se_others + expf(d_pos);
        row_loss = logf(se) - d_pos;
        if (G != nullptr) {
            float *grow = G + (size_t)b * a.ld;
            const float scale = a.inv_T * inv_B / se;
            for (int c = tid; c < a.C; c += kEpiThreads) {
                const float s = norm(srow[c]);
                const bool f = filtered(c, s);
                float g = (c == pos_idx ? -se_others : expf(logit(c) - m)) * scale;
                if (f) g *= a.filter_factor;
                if (a.normalize) g /= len_f;
                grow[c] = g;
            }
        }
    }

    // ---- per-row terms -> scratch; the last workgroup to arrive folds them in row order
    if (tid == 0) {
        ws_rows[b] = row_loss;
        ws_rows[a.B + b] = lo;
        ws_rows[2 * a.B + b] = hi;
        __threadfence();
        sh_last = atomicAdd(ticket, 1u) == (unsigned)(a.B - 1);
    }
    __syncthreads();
    if (!sh_last) return;
    __threadfence();
    if (tid == 0) {
        float s = 0.f, mn = INFINITY, mx = -INFINITY;
        for (int r = 0; r < a.B; ++r) {
            s += __builtin_nontemporal_load(ws_rows + r);
            mn = fminf(mn, __builtin_nontemporal_load(ws_rows + a.B + r));
            mx = fmaxf(mx, __builtin_nontemporal_load(ws_rows + 2 * a.B + r));
        }
        out[0] = s * inv_B;
        out[1] = mn;
        out[2] = mx;
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

}  // namespace msim
