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


// Level 0 for LONG rows of raw scores (ids implicit: id_base + column): a streaming threshold filter instead of a full sort.
// One workgroup owns kTopkFilterSeg = 16 384 consecutive scores of one row (64 KiB, sixteen 16-byte loads per thread, all issued
// up front).  Fast path: a threshold from the 256 per-thread maxima, one pass (see the kernel).  Slow path (ties): rounds of 1024 in column order.  A candidate list in LDS takes every score above the threshold T
// (T = 0 at first: everything); whenever the list holds a round's worth (1024) it is sorted, cut to its best k, and T becomes the
// k-th key.  From then on a score <= T cannot reach the top k: the list already holds k entries that rank before it -- higher
// key, or the same key and a SMALLER id, because rounds are taken in column order (which is why a strict `>` is exact for ties).
// On typical score rows (densely packed values) the first cut leaves a threshold that a few per cent of the later scores pass, so
// the kernel costs its reads: round 3's bitonic sort of every 4096-candidate segment ran at 94 GB/s (5.35 ms for 1000 x 125 000
// scores, 36 % of its LDS cycles in bank conflicts).  An adversarial row (ascending scores) degrades to one sort per round, never to
// a wrong answer.  Entries are one u64 = (score key << 32) | ~column, so "ranks before" is a single unsigned compare.
constexpr int kTopkFilterSeg = 16384;
constexpr int kTopkFilterRound = 1024;     // 256 threads x one 16-byte load
constexpr int kTopkFilterCap = 2048;       // candidate list (16 KiB of LDS)
constexpr int kTopkFilterMaxK = 256;       // k <= cap / 8: a cut always frees at least 7/8 of the list

// bitonic sort of sv[0 .. npow2) in DESCENDING order (npow2 a power of two <= kTopkFilterCap; all threads of the workgroup call it)
__device__ __noinline__ void topk_sort_desc(uint64_t *sv, int npow2, int tid) {
    for (int size = 2; size <= npow2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = tid; t < npow2 / 2; t += kTopkThreads) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const uint64_t a = sv[lo], b = sv[hi];
                if ((a > b) != desc) {
                    sv[lo] = b;
                    sv[hi] = a;
                }
            }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(kTopkThreads) void topk_filter_kernel(const float *__restrict__ scores, long long n, long long ld,
                                                                   long long id_base, int k, float *__restrict__ out_scores,
                                                                   int64_t *__restrict__ out_ids, long long out_ld) {
    __shared__ uint64_t cand[kTopkFilterCap];
    __shared__ int n_cand;
    const int tid = threadIdx.x;
    const long long row = blockIdx.y;
    const long long base = (long long)blockIdx.x * kTopkFilterSeg;
    const long long remaining = n - base;
    const int cnt = remaining < kTopkFilterSeg ? (int)remaining : kTopkFilterSeg;
    const float *srow = scores + row * ld + base;
    constexpr int kRounds = kTopkFilterSeg / kTopkFilterRound;

    // all of this thread's scores, requested before anything is looked at (a row start need not be 16-byte aligned: 4-byte loads
    // then), turned into order-preserving keys in place; columns past the row's end get key 0 (below every real score)
    uint32_t key[kRounds][4];
    const bool aligned = ((reinterpret_cast<uintptr_t>(srow) & 15) == 0);
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        const int c0 = r * kTopkFilterRound + tid * 4;
        float v[4];
        if (aligned && c0 + 3 < cnt) {
            const float4 q = *reinterpret_cast<const float4 *>(srow + c0);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = c0 + j < cnt ? srow[c0 + j] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) key[r][j] = c0 + j < cnt ? score_key(v[j]) : 0u;
    }

    // ---- fast path.  The k-th largest of the 256 per-thread maxima is a lower bound T0 of the row segment's k-th best key (k threads
    // hold a key >= T0 each), and on score rows as the scorer produces them the best k sit in k different threads, so T0 IS the k-th
    // best: one pass over the registers leaves k .. a few dozen candidates (every key >= T0: ties kept, the sort decides them by id).
    uint32_t mx = 0u;
#pragma unroll
    for (int r = 0; r < kRounds; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) mx = key[r][j] > mx ? key[r][j] : mx;
    cand[tid] = (uint64_t)mx << 32;
    if (tid == 0) n_cand = 0;
    topk_sort_desc(cand, kTopkThreads, tid);                 // (starts and ends with a barrier)
    const uint32_t t0 = (uint32_t)(cand[(k < kTopkThreads ? k : kTopkThreads) - 1] >> 32);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kRounds; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t ky = key[r][j];
            if (ky >= t0 && ky != 0u) {
                const int slot = atomicAdd(&n_cand, 1);
                if (slot < kTopkFilterCap) cand[slot] = ((uint64_t)ky << 32) | (uint32_t)(~(uint32_t)(r * kTopkFilterRound + tid * 4 + j));
            }
        }
    __syncthreads();
    int have = n_cand;
    __syncthreads();
    if (have > kTopkFilterCap) {
        // ---- slow path (massive ties at the threshold, e.g. a constant row): rounds of 1024 in column order.  The list takes every key
        // above a threshold T (0 at first); whenever it holds a round's worth it is sorted, cut to its best k, and T becomes the k-th
        // key: a later score <= T cannot reach the top k, the list already holds k entries that rank before it -- a higher key, or the
        // same key and a SMALLER id, because rounds are taken in column order (which is why the strict `>` is exact for ties).
        if (tid == 0) n_cand = 0;
        uint32_t thr = 0u;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
            if (r * kTopkFilterRound >= cnt) break;                          // (uniform)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (key[r][j] > thr) {
                    const int slot = atomicAdd(&n_cand, 1);
                    cand[slot] = ((uint64_t)key[r][j] << 32) | (uint32_t)(~(uint32_t)(r * kTopkFilterRound + tid * 4 + j));
                }
            }
            __syncthreads();
            const int cur = n_cand;                                          // (uniform after the barrier)
            __syncthreads();                                                 // ... and read by everyone before the next round adds to it
            const bool more = (r + 1) * kTopkFilterRound < cnt;
            if (more && cur >= kTopkFilterRound) {                           // cut: the next round (<= 1024 more) always fits the 2048 slots
                int np = 64;
                while (np < cur) np <<= 1;
                for (int i = cur + tid; i < np; i += kTopkThreads) cand[i] = 0ull;
                topk_sort_desc(cand, np, tid);
                if (cur >= k) thr = (uint32_t)(cand[k - 1] >> 32);
                if (tid == 0) n_cand = cur < k ? cur : k;
                __syncthreads();
            }
        }
        have = n_cand;
        __syncthreads();
    }
    int npow2 = 64;
    while (npow2 < have) npow2 <<= 1;
    for (int i = have + tid; i < npow2; i += kTopkThreads) cand[i] = 0ull;
    topk_sort_desc(cand, npow2, tid);

    float *os = out_scores + row * out_ld + (long long)blockIdx.x * k;
    int64_t *oi = out_ids + row * out_ld + (long long)blockIdx.x * k;
    for (int j = tid; j < k; j += kTopkThreads) {
        const uint64_t e = j < have ? cand[j] : 0ull;
        const uint32_t ky = (uint32_t)(e >> 32);
        const bool valid = ky != 0u;
        os[j] = valid ? key_score(ky) : -INFINITY;
        oi[j] = valid ? (int64_t)(id_base + base + (long long)(uint32_t)(~(uint32_t)e)) : -1;
    }
}

}  // namespace msim
