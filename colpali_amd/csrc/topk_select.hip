// Row-wise top-k selection with the deterministic order (score descending, id ascending).
//
// The reference returns the full [n_q, n_p] score matrix on the CPU
// (colpali_engine/utils/processing_utils.py:180-186) and leaves ranking to the caller
// (torch.topk / argsort in the evaluators; k=10 default in the experimental
// get_topk_plaid, processing_utils.py:189-219).  For a sharded corpus the ranking has to be
// mergeable and bit-reproducible across shard counts, hence the total order on (score, id).
//
// One workgroup sorts one segment of <= SEG candidates of one row in LDS (bitonic network on
// a 32-bit order-preserving image of the score plus the 64-bit id) and emits the segment's
// best k; the host chains levels until one segment per row is left.  HBM traffic is 4 B per
// scored pair (vs 256 KiB/Bq per pair for the MaxSim kernel), so this is never the bottleneck.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace msim {

constexpr int kTopkSeg = 4096;      // most candidates one workgroup sorts (the host may choose a smaller power of two)
constexpr int kTopkThreads = 256;
constexpr int kTopkMaxK = 1024;     // k <= kTopkSeg / 4 keeps every level shrinking by >= 4x

// order-preserving map float -> uint32 (ascending); -0.0 is folded onto +0.0 so that equal
// floats always tie and the id decides
__device__ __forceinline__ uint32_t score_key(float s) {
    if (s == 0.0f) s = 0.0f;
    uint32_t u = __float_as_uint(s);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_score(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

// a ranks before b: higher score first, then lower id (ids compared unsigned so the -1 padding id ranks last)
__device__ __forceinline__ bool ranks_before(uint32_t ka, uint64_t ia, uint32_t kb, uint64_t ib) {
    return (ka > kb) || (ka == kb && ia < ib);
}

__global__ __launch_bounds__(kTopkThreads) void topk_segment_kernel(const float *__restrict__ scores,
                                                                    const int64_t *__restrict__ ids,  // or null
                                                                    long long n, long long ld, long long id_base, int k,
                                                                    int seg,   // candidates per workgroup: power of two, 4k <= seg <= kTopkSeg
                                                                    float *__restrict__ out_scores,
                                                                    int64_t *__restrict__ out_ids, long long out_ld) {
    __shared__ uint32_t skey[kTopkSeg];
    __shared__ uint64_t sid[kTopkSeg];
    const int tid = threadIdx.x;
    const long long row = blockIdx.y;
    const long long base = (long long)blockIdx.x * seg;
    const long long remaining = n - base;
    const int cnt = remaining < seg ? (int)remaining : seg;
    int npow2 = 64;                       // sort only the next power of two above the live candidates
    while (npow2 < cnt) npow2 <<= 1;
    const float *srow = scores + row * ld + base;
    const int64_t *irow = ids ? ids + row * ld + base : nullptr;

    for (int i = tid; i < npow2; i += kTopkThreads) {
        uint32_t key = 0u;                 // below every real score (even -inf and NaN images are > 0)
        uint64_t id = ~0ull;
        if (i < cnt) {
            key = score_key(srow[i]);
            id = irow ? (uint64_t)irow[i] : (uint64_t)(id_base + base + i);
            if (irow && irow[i] < 0) key = 0u;   // padding entries from a previous level / another shard
        }
        skey[i] = key;
        sid[i] = id;
    }

    for (int size = 2; size <= npow2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = tid; t < npow2 / 2; t += kTopkThreads) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool first_wins = (lo & size) == 0;   // this block is sorted best-first
                const uint32_t ka = skey[lo], kb = skey[hi];
                const uint64_t ia = sid[lo], ib = sid[hi];
                const bool a_first = ranks_before(ka, ia, kb, ib);
                if (a_first != first_wins) {
                    skey[lo] = kb; skey[hi] = ka;
                    sid[lo] = ib; sid[hi] = ia;
                }
            }
        }
    }
    __syncthreads();

    float *os = out_scores + row * out_ld + (long long)blockIdx.x * k;
    int64_t *oi = out_ids + row * out_ld + (long long)blockIdx.x * k;
    for (int j = tid; j < k; j += kTopkThreads) {
        const bool valid = j < cnt && skey[j] != 0u;
        os[j] = valid ? key_score(skey[j]) : -INFINITY;
        oi[j] = valid ? (int64_t)sid[j] : -1;
    }
}

}  // namespace msim
