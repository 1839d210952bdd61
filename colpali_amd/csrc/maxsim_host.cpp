// msim_fwd_host -- the MaxSim scorer on the HOST cores (include/maxsim.h), for callers that name device="cpu" (or whose
// get_torch_device("auto") finds no GPU: colpali_engine/utils/torch_utils.py:12-31) through the reference's own signature
// (colpali_engine/utils/processing_utils.py:132-187 computes on whatever device it is given).  Product code: plain C++, compiled by
// the host compiler, no HIP, no torch; it never runs on behalf of a GPU request (colpali_amd/scoring.py dispatches on the device the
// caller asked for and nothing else).
//
// Arithmetic: every product and sum in fp32 (16-bit inputs are widened exactly), one fused multiply-add chain per (token, document
// row) in k order, max over rows, token sum in fp32 -- the "truth tier" of the GPU kernels (scores within 1e-5 of a float64 evaluation).
// MSIM_FLAG_REF_ROUNDING reproduces the reference's 16-bit rounding like the kernels do.
//
// Shape of the computation: the QUERIES are re-laid once per call -- blocks of whole queries of ~256 tokens, the tokens in the lanes of
// 16-float vectors, k-major -- so that the streaming side, the documents, needs no transposition: eight rows are widened to fp32 with
// contiguous loops and every (row, 16 tokens) runs acc += d[row][k] * Qt[k][:], a broadcast-FMA; the max over rows is an element-wise
// vector max, and nothing horizontal happens until a query's tokens are summed, once per (query, document).  Written on the compiler's
// generic vector type and cloned for AVX-512 / AVX2 / baseline x86-64 (resolved at load time); documents are dealt to a persistent pool
// of native threads in contiguous chunks.
#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <limits>
#include <mutex>
#include <new>
#include <thread>
#include <unistd.h>
#include <vector>

#include "../../include/maxsim.h"

// (sanitizer builds define MSIM_HOST_NO_CLONES: an ifunc resolver runs before ThreadSanitizer's runtime is up)
#ifdef MSIM_HOST_NO_CLONES
#define MSIM_HOST_CLONES
#else
#define MSIM_HOST_CLONES __attribute__((target_clones("avx512f", "avx2,fma", "default")))
#endif

namespace {

typedef float v16f __attribute__((vector_size(64)));

inline float bf16_to_f32(uint16_t v) {
    uint32_t u = (uint32_t)v << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

inline float f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1f, man = h & 0x3ffu, u;
    if (exp == 0) {
        if (man == 0) {
            u = sign;
        } else {                                   // subnormal: renormalise
            int e = -1;
            do {
                ++e;
                man <<= 1;
            } while (!(man & 0x400u));
            u = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) {
        u = sign | 0x7f800000u | (man << 13);
    } else {
        u = sign | ((exp + 112) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &u, 4);
    return f;
}

inline float round_bf16(float x) {                 // round to nearest even, like torch's float -> bfloat16
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return x;  // NaN stays NaN
    u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
    memcpy(&x, &u, 4);
    return x;
}

// float -> IEEE half (round to nearest even) -> float, in integer arithmetic (no _Float16 in every host compiler)
inline float round_f16(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint32_t sign = u & 0x80000000u;
    uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return x;                             // NaN
    if (a >= 0x477ff000u) {                                    // rounds to or beyond 65520: infinity
        a = 0x7f800000u;
    } else if (a < 0x38800000u) {                              // below the smallest normal half (2^-14): a multiple of 2^-24
        float f;
        memcpy(&f, &a, 4);
        f = f * 16777216.0f;                                   // exact scaling by 2^24
        const float r = std::nearbyint(f);                     // ties to even (default rounding mode)
        f = r / 16777216.0f;
        memcpy(&a, &f, 4);
    } else {
        a = (a + 0xfffu + ((a >> 13) & 1u)) & 0xffffe000u;     // 13 mantissa bits dropped
    }
    a |= sign;
    memcpy(&x, &a, 4);
    return x;
}

inline float widen(const void *base, int dtype, size_t idx) {
    switch (dtype) {
        case MSIM_DTYPE_BF16: return bf16_to_f32(static_cast<const uint16_t *>(base)[idx]);
        case MSIM_DTYPE_F16: return f16_to_f32(static_cast<const uint16_t *>(base)[idx]);
        default: return static_cast<const float *>(base)[idx];
    }
}

// one call, both sides as lists of row blocks: query q = q_len[q] rows at q_ptr[q], document c = d_len[c] rows at d_ptr[c] (the packed
// entry point fills the lists from its box / blob; the list entry point passes the caller's tensors through: nothing is copied)
struct HostCall {
    int dtype;
    const void *const *q_ptr;
    const int64_t *q_len;
    int n_q;
    const void *const *d_ptr;
    const int64_t *d_len;
    const uint8_t *clamp0;
    int n_d, dim;
    float *scores;
    int64_t ld;
    bool ref_round;
};

constexpr int kTok = 8;       // (msim_sim_matrix_host) A rows per register block
constexpr int kRows = 16;     // (msim_sim_matrix_host) B rows per vector
constexpr int kLanes = 16;    // query tokens per vector
constexpr int kGroup = 8;     // document rows per register block

// One block of whole queries: its tokens sit in the LANES of 16-float vectors, k-major (Qt[(tv * dim + k) * 16 + lane] = token
// 16 tv + lane, component k; lanes past the block's last token are zero) -- built once per call.  A document row then needs no
// transposition at all: acc[row][16 tokens] += d[row][k] (a scalar, broadcast) * Qt[k][16 tokens], eight rows per register block,
// and the max over rows is an element-wise vector max: no horizontal operation until a query's tokens are summed, once per
// (query, document).  Rows are widened to fp32 eight at a time with contiguous, vectorisable loops.
struct QueryBlock {
    int q0, nq;          // queries of the block
    int n_tv;            // token vectors
    size_t qt_off;       // offset of the block's Qt in floats
};

// documents [c_lo, c_hi) against every query block: `df` = scratch for 8 widened rows (8 * dim floats), `tmaxv` = scratch for the block's
// running maxima (n_tv vectors); `tok0[q]` = first token of query q inside its block.  A block's operands are ~128 KiB (256 tokens at
// dim 128: L2-resident) and the block walks all the documents of the range before the next one starts.
MSIM_HOST_CLONES
void score_range(const HostCall &c, const float *qt, const QueryBlock *blocks, int n_blocks, const int *tok0, float *df, v16f *tmaxv,
                 int c_lo, int c_hi) {
    const int dim = c.dim;
    const float ninf = -std::numeric_limits<float>::infinity();
    for (int b = 0; b < n_blocks; ++b) {
        const QueryBlock &qb = blocks[b];
        const float *qtb = qt + qb.qt_off;
        for (int doc = c_lo; doc < c_hi; ++doc) {
            const int len = (int)c.d_len[doc];
            for (int tv = 0; tv < qb.n_tv; ++tv)
                for (int l = 0; l < kLanes; ++l) tmaxv[tv][l] = ninf;
            for (int g = 0; g < len; g += kGroup) {
                const int nr = len - g < kGroup ? len - g : kGroup;
                for (int r = 0; r < nr; ++r) {                       // widen: contiguous in, contiguous out
                    const size_t base = (size_t)(g + r) * dim;
                    float *o = df + (size_t)r * dim;
                    if (c.dtype == MSIM_DTYPE_BF16) {
                        const uint16_t *p = static_cast<const uint16_t *>(c.d_ptr[doc]) + base;
                        for (int k = 0; k < dim; ++k) o[k] = bf16_to_f32(p[k]);
                    } else if (c.dtype == MSIM_DTYPE_F16) {
                        const uint16_t *p = static_cast<const uint16_t *>(c.d_ptr[doc]) + base;
                        for (int k = 0; k < dim; ++k) o[k] = f16_to_f32(p[k]);
                    } else {
                        memcpy(o, static_cast<const float *>(c.d_ptr[doc]) + base, (size_t)dim * sizeof(float));
                    }
                }
                for (int tv = 0; tv < qb.n_tv; ++tv) {
                    const float *qp = qtb + (size_t)tv * dim * kLanes;
                    v16f acc[kGroup];
                    for (int r = 0; r < kGroup; ++r) acc[r] = v16f{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                    if (nr == kGroup) {
                        for (int k = 0; k < dim; ++k) {
                            const v16f qv = *reinterpret_cast<const v16f *>(qp + (size_t)k * kLanes);
                            for (int r = 0; r < kGroup; ++r) acc[r] += df[(size_t)r * dim + k] * qv;
                        }
                    } else {
                        for (int k = 0; k < dim; ++k) {
                            const v16f qv = *reinterpret_cast<const v16f *>(qp + (size_t)k * kLanes);
                            for (int r = 0; r < nr; ++r) acc[r] += df[(size_t)r * dim + k] * qv;
                        }
                    }
                    v16f m = tmaxv[tv];
                    for (int r = 0; r < nr; ++r) {
                        v16f a = acc[r];
                        if (c.ref_round)               // the reference's 16-bit einsum rounds every similarity before the max
                            for (int l = 0; l < kLanes; ++l) a[l] = c.dtype == MSIM_DTYPE_F16 ? round_f16(a[l]) : round_bf16(a[l]);
                        m = (a > m) | (a != a) ? a : m;       // max that keeps a NaN similarity, like torch's (and like the clamp below)
                    }
                    tmaxv[tv] = m;
                }
            }
            const bool clamp = c.clamp0 != nullptr && c.clamp0[doc] != 0;
            const float *tm = reinterpret_cast<const float *>(tmaxv);
            for (int qi = 0; qi < qb.nq; ++qi) {
                const int q = qb.q0 + qi;
                const int t0 = tok0[q], t1 = t0 + (int)c.q_len[q];
                float tot = 0.0f;
                for (int t = t0; t < t1; ++t) {
                    float m = tm[t];
                    if (clamp && !(m > 0.0f)) m = m != m ? m : 0.0f;      // max(m, 0), NaN kept
                    tot += m;
                }
                if (c.ref_round) tot = c.dtype == MSIM_DTYPE_F16 ? round_f16(tot) : round_bf16(tot);
                c.scores[(size_t)q * c.ld + doc] = tot;
            }
        }
    }
}

// out[i, j] = <A_i, B_j> for the rows [b_lo, b_hi) of B: the same register blocking without the reduction (score_single_vector,
// processing_utils.py:126 einsum("bd,cd->bc"))
MSIM_HOST_CLONES
void sim_range(int dtype, const float *af, int n_a, const void *B, int dim, float *out, int64_t ld, bool ref_round, float *dt, int b_lo,
               int b_hi) {
    for (int g = b_lo; g < b_hi; g += kRows) {
        const int valid = b_hi - g < kRows ? b_hi - g : kRows;
        for (int r = 0; r < kRows; ++r) {
            if (r < valid) {
                const size_t base = (size_t)(g + r) * dim;
                for (int k = 0; k < dim; ++k) dt[k * kRows + r] = widen(B, dtype, base + k);
            } else {
                for (int k = 0; k < dim; ++k) dt[k * kRows + r] = 0.0f;
            }
        }
        for (int t0 = 0; t0 < n_a; t0 += kTok) {
            const int nt = n_a - t0 < kTok ? n_a - t0 : kTok;
            v16f acc[kTok];
            for (int i = 0; i < kTok; ++i) acc[i] = v16f{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            const float *q = af + (size_t)t0 * dim;
            for (int k = 0; k < dim; ++k) {
                const v16f d = *reinterpret_cast<const v16f *>(dt + k * kRows);
                for (int i = 0; i < nt; ++i) acc[i] += q[(size_t)i * dim + k] * d;
            }
            for (int i = 0; i < nt; ++i)
                for (int r = 0; r < valid; ++r) {
                    float v = acc[i][r];
                    if (ref_round) v = dtype == MSIM_DTYPE_F16 ? round_f16(v) : round_bf16(v);
                    out[(size_t)(t0 + i) * ld + g + r] = v;
                }
        }
    }
}

// A persistent pool of worker threads: a config-1 sized call (4 queries x 16 documents) is ~1 ms of arithmetic, and starting eight
// std::threads for it costs more than that on some hosts (containers with a user-space kernel: ~0.5 ms per thread).  Workers are
// started on first use, grow on demand, sleep on a condition variable between calls and are never joined (the pool object is
// leaked on purpose: no destructor runs under a worker's feet at exit).  One parallel region at a time; a second caller runs its
// region on its own thread.  A forked child starts with an empty pool.
class HostPool {
   public:
    static HostPool &get() {
        HostPool *p = instance().load(std::memory_order_acquire);
        if (!p) {
            std::lock_guard<std::mutex> g(init_mutex());
            p = instance().load(std::memory_order_acquire);
            if (!p) {
                p = new HostPool();
                static std::once_flag once;
                std::call_once(once, [] { pthread_atfork(nullptr, nullptr, [] { instance().store(nullptr, std::memory_order_release); }); });
                instance().store(p, std::memory_order_release);
            }
        }
        return *p;
    }

    // body(i) for i in [0, n), on up to `threads` threads including the caller's
    void run(int n, int threads, const std::function<void(int)> &body) {
        if (n <= 0) return;
        std::unique_lock<std::mutex> region(region_, std::try_to_lock);
        if (threads <= 1 || n == 1 || !region.owns_lock()) {
            for (int i = 0; i < n; ++i) body(i);
            return;
        }
        const int helpers = (threads < n ? threads : n) - 1;
        {
            std::unique_lock<std::mutex> lk(m_);
            while ((int)workers_ < helpers) {
                std::thread(&HostPool::worker, this).detach();
                ++workers_;
            }
            body_ = &body;
            n_ = n;
            next_.store(0, std::memory_order_relaxed);
            pending_ = helpers;
            wanted_ = helpers;
            ++epoch_;
        }
        cv_.notify_all();
        drain(body);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&] { return pending_ == 0; });
        body_ = nullptr;
    }

    // every worker, present and future, runs on `set` from now on (the drop-in moves its gather threads next to the caller's pages)
    void set_affinity(const cpu_set_t &set) {
        std::unique_lock<std::mutex> lk(m_);
        aff_ = set;
        has_aff_ = true;
        for (pthread_t h : handles_) pthread_setaffinity_np(h, sizeof(aff_), &aff_);
    }

   private:
    static std::atomic<HostPool *> &instance() {
        static std::atomic<HostPool *> p{nullptr};
        return p;
    }
    static std::mutex &init_mutex() {
        static std::mutex m;
        return m;
    }
    void drain(const std::function<void(int)> &body) {
        for (;;) {
            const int i = next_.fetch_add(1, std::memory_order_relaxed);
            if (i >= n_) break;
            body(i);
        }
    }
    void worker() {
        {
            std::unique_lock<std::mutex> lk(m_);
            handles_.push_back(pthread_self());
            if (has_aff_) pthread_setaffinity_np(pthread_self(), sizeof(aff_), &aff_);
        }
        unsigned long seen = 0;
        for (;;) {
            const std::function<void(int)> *body = nullptr;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return epoch_ != seen && wanted_ > 0; });
                seen = epoch_;
                --wanted_;
                body = body_;
            }
            drain(*body);
            {
                std::unique_lock<std::mutex> lk(m_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    std::vector<pthread_t> handles_;
    cpu_set_t aff_;
    bool has_aff_ = false;
    std::mutex region_, m_;
    std::condition_variable cv_, done_;
    const std::function<void(int)> *body_ = nullptr;
    std::atomic<int> next_{0};
    int n_ = 0, pending_ = 0, wanted_ = 0;
    unsigned workers_ = 0;
    unsigned long epoch_ = 0;
};

thread_local char g_host_err[256] = "";

}  // namespace

extern "C" {

const char *msim_host_last_error(void) { return g_host_err; }

}  // extern "C"

namespace {

int fail_host(int code, const char *msg) {
    strncpy(g_host_err, msg, sizeof(g_host_err) - 1);
    return code;
}

int fwd_host_lists(const HostCall &c, int n_threads) {
    const int n_q = c.n_q, n_d = c.n_d, dim = c.dim;
    // ---- the queries, once: blocks of whole queries of about 256 tokens, tokens in vector lanes, k-major
    std::vector<QueryBlock> blocks;
    std::vector<int> tok0(n_q);
    size_t qt_floats = 0;
    int max_tv = 1;
    long long total_tok = 0;
    for (int q0 = 0; q0 < n_q;) {
        int nq = 0, ntok = 0;
        while (q0 + nq < n_q && (nq == 0 || ntok + c.q_len[q0 + nq] <= 256)) {
            tok0[q0 + nq] = ntok;
            ntok += (int)c.q_len[q0 + nq];
            ++nq;
        }
        const int n_tv = (ntok + kLanes - 1) / kLanes;
        blocks.push_back(QueryBlock{q0, nq, n_tv, qt_floats});
        qt_floats += (size_t)n_tv * dim * kLanes;
        if (n_tv > max_tv) max_tv = n_tv;
        total_tok += ntok;
        q0 += nq;
    }
    std::vector<float> qt_store(qt_floats + 16, 0.0f);
    float *qt = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(qt_store.data()) + 63) & ~(uintptr_t)63);
    for (const QueryBlock &qb : blocks)
        for (int qi = 0; qi < qb.nq; ++qi) {
            const int q = qb.q0 + qi;
            for (int i = 0; i < (int)c.q_len[q]; ++i) {
                const int t = tok0[q] + i;
                float *o = qt + qb.qt_off + (size_t)(t / kLanes) * dim * kLanes + (t % kLanes);
                for (int k = 0; k < dim; ++k) o[(size_t)k * kLanes] = widen(c.q_ptr[q], c.dtype, (size_t)i * dim + k);
            }
        }
    int64_t rows = 0;
    for (int d = 0; d < n_d; ++d) rows += c.d_len[d];
    int nt = n_threads < 1 ? 1 : (n_threads > 256 ? 256 : n_threads);
    const double work = (double)rows * (double)total_tok * dim;         // multiply-adds
    const double by_work_d = work / 4e6 + 1.0;                          // a thread is worth waking for a few million of them
    const int by_work = by_work_d > 256.0 ? 256 : (int)by_work_d;        // clamped in double: the cast of a huge value is undefined
    if (nt > by_work) nt = by_work;
    if (nt > n_d) nt = n_d;
    // contiguous chunks of documents with about the same number of rows each, a few per thread (dynamic assignment evens out
    // what the clock and the other tenants of the host do)
    const int n_chunks = nt <= 1 ? 1 : (4 * nt < n_d ? 4 * nt : n_d);
    std::vector<int> cut(1, 0);
    {
        const int64_t per = (rows + n_chunks - 1) / n_chunks;
        int64_t acc = 0;
        for (int d = 0; d < n_d; ++d) {
            acc += c.d_len[d];
            if ((acc >= per && (int)cut.size() < n_chunks) || d + 1 == n_d) {
                cut.push_back(d + 1);
                acc = 0;
            }
        }
    }
    const std::function<void(int)> body = [&](int chunk) {
        thread_local std::vector<float> scratch;
        const size_t need = (size_t)kGroup * dim + (size_t)max_tv * kLanes + 64;
        if (scratch.size() < need) scratch.resize(need);
        float *al = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(scratch.data()) + 63) & ~(uintptr_t)63);
        v16f *tmaxv = reinterpret_cast<v16f *>(al);
        float *df = al + (size_t)max_tv * kLanes;
        score_range(c, qt, blocks.data(), (int)blocks.size(), tok0.data(), df, tmaxv, cut[chunk], cut[chunk + 1]);
    };
    HostPool::get().run((int)cut.size() - 1, nt, body);
    return MSIM_OK;
}

int check_host(int dtype, int n_q, int n_d, int dim, const float *scores, int64_t ld_scores, uint32_t flags) {
    if (n_q < 0 || n_d < 0 || dim <= 0) return fail_host(MSIM_EINVAL, "negative size");
    if (dtype != MSIM_DTYPE_BF16 && dtype != MSIM_DTYPE_F16 && dtype != MSIM_DTYPE_F32)
        return fail_host(MSIM_EUNSUPPORTED, "the host scorer takes bfloat16 (0), float16 (1) or float32 (2) embeddings");
    if (n_q > 0 && n_d > 0 && !scores) return fail_host(MSIM_EINVAL, "null pointer argument");
    if (ld_scores < n_d) return fail_host(MSIM_EINVAL, "ld_scores < n_d");
    if (flags & ~(MSIM_FLAG_REF_ROUNDING)) return fail_host(MSIM_EINVAL, "unknown flags");
    return MSIM_OK;
}

}  // namespace

extern "C" {

static int msim_fwd_host_impl(int dtype, const void *Q, int n_q, int Lq, const void *D, const int32_t *d_off, const uint8_t *d_clamp0, int n_d,
                  int dim, float *scores, int64_t ld_scores, uint32_t flags, int n_threads) {
    if (Lq < 0) return fail_host(MSIM_EINVAL, "negative size");
    if (int rc = check_host(dtype, n_q, n_d, dim, scores, ld_scores, flags)) return rc;
    if (n_q == 0 || n_d == 0) return MSIM_OK;
    if (!Q || !D || !d_off) return fail_host(MSIM_EINVAL, "null pointer argument");
    const size_t es = dtype == MSIM_DTYPE_F32 ? 4 : 2;
    std::vector<const void *> qp(n_q), dp(n_d);
    std::vector<int64_t> ql(n_q, Lq), dl(n_d);
    for (int q = 0; q < n_q; ++q) qp[q] = static_cast<const char *>(Q) + (size_t)q * Lq * dim * es;
    for (int d = 0; d < n_d; ++d) {
        if (d_off[d + 1] < d_off[d]) return fail_host(MSIM_EINVAL, "d_off must be non-decreasing");
        dp[d] = static_cast<const char *>(D) + (size_t)d_off[d] * dim * es;
        dl[d] = d_off[d + 1] - d_off[d];
    }
    const bool rr = (flags & MSIM_FLAG_REF_ROUNDING) != 0 && dtype != MSIM_DTYPE_F32;      // fp32 inputs: nothing is rounded
    return fwd_host_lists(HostCall{dtype, qp.data(), ql.data(), n_q, dp.data(), dl.data(), d_clamp0, n_d, dim, scores, ld_scores, rr},
                          n_threads);
}

static int msim_fwd_host_lists_impl(int dtype, const void *const *q_ptr, const int64_t *q_rows, int n_q, const void *const *d_ptr,
                        const int64_t *d_rows, const uint8_t *d_clamp0, int n_d, int dim, float *scores, int64_t ld_scores, uint32_t flags,
                        int n_threads) {
    if (int rc = check_host(dtype, n_q, n_d, dim, scores, ld_scores, flags)) return rc;
    if (n_q == 0 || n_d == 0) return MSIM_OK;
    if (!q_ptr || !q_rows || !d_ptr || !d_rows) return fail_host(MSIM_EINVAL, "null pointer argument");
    for (int q = 0; q < n_q; ++q)
        if (q_rows[q] < 0 || q_rows[q] > 0x7fffffff || (q_rows[q] > 0 && !q_ptr[q])) return fail_host(MSIM_EINVAL, "bad query buffer");
    for (int d = 0; d < n_d; ++d)
        if (d_rows[d] < 0 || d_rows[d] > 0x7fffffff || (d_rows[d] > 0 && !d_ptr[d])) return fail_host(MSIM_EINVAL, "bad document buffer");
    const bool rr = (flags & MSIM_FLAG_REF_ROUNDING) != 0 && dtype != MSIM_DTYPE_F32;
    return fwd_host_lists(HostCall{dtype, q_ptr, q_rows, n_q, d_ptr, d_rows, d_clamp0, n_d, dim, scores, ld_scores, rr}, n_threads);
}

static int msim_sim_matrix_host_impl(int dtype, const void *A, int n_a, const void *B, int n_b, int dim, float *out, int64_t ld_out, uint32_t flags,
                         int n_threads) {
    auto fail = [](int code, const char *msg) {
        strncpy(g_host_err, msg, sizeof(g_host_err) - 1);
        return code;
    };
    if (n_a < 0 || n_b < 0 || dim <= 0) return fail(MSIM_EINVAL, "negative size");
    if (n_a == 0 || n_b == 0) return MSIM_OK;
    if (!A || !B || !out) return fail(MSIM_EINVAL, "null pointer argument");
    if (dtype != MSIM_DTYPE_BF16 && dtype != MSIM_DTYPE_F16 && dtype != MSIM_DTYPE_F32)
        return fail(MSIM_EUNSUPPORTED, "msim_sim_matrix_host takes bfloat16 (0), float16 (1) or float32 (2) embeddings");
    if (ld_out < n_b) return fail(MSIM_EINVAL, "ld_out < n_b");
    if (flags & ~(MSIM_FLAG_REF_ROUNDING)) return fail(MSIM_EINVAL, "unknown flags");
    const bool ref_round = (flags & MSIM_FLAG_REF_ROUNDING) != 0 && dtype != MSIM_DTYPE_F32;
    std::vector<float> af((size_t)n_a * dim + 16);
    for (size_t i = 0; i < (size_t)n_a * dim; ++i) af[i] = widen(A, dtype, i);
    int nt = n_threads < 1 ? 1 : (n_threads > 256 ? 256 : n_threads);
    const double by_work_d = (double)n_a * n_b * dim / 4e6 + 1.0;
    const int by_work = by_work_d > 256.0 ? 256 : (int)by_work_d;
    if (nt > by_work) nt = by_work;
    const int groups = (n_b + kRows - 1) / kRows;
    const int n_chunks = nt <= 1 ? 1 : (4 * nt < groups ? 4 * nt : groups);
    const int per = ((groups + n_chunks - 1) / n_chunks) * kRows;
    const std::function<void(int)> body = [&](int chunk) {
        thread_local std::vector<float> dt;
        if (dt.size() < (size_t)dim * kRows + 16) dt.resize((size_t)dim * kRows + 16);
        float *dta = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(dt.data()) + 63) & ~(uintptr_t)63);
        const int lo = chunk * per, hi = lo + per < n_b ? lo + per : n_b;
        if (lo < hi) sim_range(dtype, af.data(), n_a, B, dim, out, ld_out, ref_round, dta, lo, hi);
    };
    HostPool::get().run(n_chunks, nt, body);
    return MSIM_OK;
}

}  // extern "C"

// Copies bytes [lo, hi) of the VIRTUAL concatenation of n host buffers -- buffer i holds image bytes prefix[i] .. prefix[i + 1] - 1 at
// src[i] -- to dst (dst[0] = image byte lo), on the persistent pool, the range cut into equal BYTE shares (not buffer shares).
// The drop-in's upload path (colpali_amd/corpus.py): a thousand per-page host tensors (README.md:121-126) go through a bounded pinned
// staging buffer chunk by chunk; one call per chunk, no per-call thread start (msim_host_gather spawned eight std::threads per
// 32 MiB chunk: about as long as the copy itself), no Python loop over the pages.
static int host_gather_range_impl(void *dst, const void *const *src, const int64_t *prefix, int64_t n, int64_t lo, int64_t hi, int n_threads) {
    if (n < 0 || lo < 0 || hi < lo) return fail_host(MSIM_EINVAL, "bad range");
    if (hi == lo || n == 0) return MSIM_OK;
    if (!dst || !src || !prefix) return fail_host(MSIM_EINVAL, "null pointer argument");
    if (hi > prefix[n]) return fail_host(MSIM_EINVAL, "range beyond the image");
    const int64_t total = hi - lo;
    int nt = n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads);
    const int64_t by_bytes = total >> 21;                                      // at least ~2 MiB per thread
    if (nt > by_bytes) nt = by_bytes < 1 ? 1 : (int)by_bytes;
    char *d = static_cast<char *>(dst);
    const std::function<void(int)> body = [&](int part) {
        const int64_t a = lo + total * part / nt, b = lo + total * (part + 1) / nt;
        if (a >= b) return;
        int64_t i0 = 0, i1 = n;                                                // the buffer that holds image byte a
        while (i0 < i1) {
            const int64_t mid = (i0 + i1) >> 1;
            if (prefix[mid + 1] <= a) i0 = mid + 1; else i1 = mid;
        }
        for (int64_t i = i0, pos = a; pos < b; ++i) {
            const int64_t end = prefix[i + 1] < b ? prefix[i + 1] : b;
            if (end > pos) memcpy(d + (pos - lo), static_cast<const char *>(src[i]) + (pos - prefix[i]), (size_t)(end - pos));
            pos = end > pos ? end : pos;
        }
    };
    HostPool::get().run(nt, nt, body);
    return MSIM_OK;
}

// The same gather, started and collected in two calls: the caller's thread (Python: the one that also issues the H2D copies and the
// kernel launches of the chunks that have arrived) is free while chunk k + 1 is gathered.  ONE persistent driver thread runs the
// requests in order (it is the "caller" of the pool's parallel region); one request may be in flight per process.  Failures are kept
// for `wait`, which reports them on the waiting thread's error slot.
class AsyncGather {
   public:
    static AsyncGather &get() {
        static AsyncGather *g = new AsyncGather();       // leaked on purpose, like the pool
        return *g;
    }
    int begin(void *dst, const void *const *src, const int64_t *prefix, int64_t n, int64_t lo, int64_t hi, int n_threads) {
        std::unique_lock<std::mutex> lk(m_);
        if (busy_) return fail_host(MSIM_EINVAL, "a gather is already in flight: call msim_host_gather_range_wait first");
        if (!started_ || pid_ != getpid()) {             // first use, or a forked child (threads do not survive a fork)
            std::thread(&AsyncGather::driver, this).detach();
            started_ = true;
            pid_ = getpid();
        }
        req_ = Req{dst, src, prefix, n, lo, hi, n_threads};
        busy_ = true;
        have_ = true;
        cv_.notify_all();
        return MSIM_OK;
    }
    void set_affinity(const cpu_set_t &set) {
        std::unique_lock<std::mutex> lk(m_);
        aff_ = set;
        has_aff_ = true;
        if (have_driver_ && pid_ == getpid()) pthread_setaffinity_np(driver_, sizeof(aff_), &aff_);
    }
    int wait() {
        std::unique_lock<std::mutex> lk(m_);
        if (!busy_) return fail_host(MSIM_EINVAL, "no gather in flight");
        done_.wait(lk, [&] { return !have_; });
        busy_ = false;
        if (rc_ != MSIM_OK) snprintf(g_host_err, sizeof(g_host_err), "%s", err_);
        return rc_;
    }

   private:
    struct Req {
        void *dst;
        const void *const *src;
        const int64_t *prefix;
        int64_t n, lo, hi;
        int n_threads;
    };
    void driver() {
        {
            std::unique_lock<std::mutex> lk(m_);
            driver_ = pthread_self();
            have_driver_ = true;
            if (has_aff_) pthread_setaffinity_np(driver_, sizeof(aff_), &aff_);
        }
        for (;;) {
            Req r;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return have_; });
                r = req_;
            }
            int rc;
            try {
                rc = host_gather_range_impl(r.dst, r.src, r.prefix, r.n, r.lo, r.hi, r.n_threads);
            } catch (...) {
                rc = fail_host(MSIM_ELAUNCH, "unexpected C++ exception in the gather thread");
            }
            std::unique_lock<std::mutex> lk(m_);
            rc_ = rc;
            if (rc != MSIM_OK) snprintf(err_, sizeof(err_), "%s", g_host_err);   // the driver's own thread-local message
            have_ = false;
            done_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_, done_;
    Req req_{};
    bool busy_ = false, have_ = false, started_ = false, have_driver_ = false, has_aff_ = false;
    pthread_t driver_{};
    cpu_set_t aff_;
    pid_t pid_ = 0;
    int rc_ = MSIM_OK;
    char err_[256] = "";
};

extern "C" {

// nothing may unwind across the C ABI (through ctypes that terminates the process): allocation failures become an error code
#define MSIM_HOST_GUARD(call)                                              \
    try {                                                                  \
        return call;                                                       \
    } catch (const std::bad_alloc &) {                                     \
        return fail_host(MSIM_ELAUNCH, "out of host memory");              \
    } catch (...) {                                                        \
        return fail_host(MSIM_ELAUNCH, "unexpected C++ exception");        \
    }

int msim_fwd_host(int dtype, const void *Q, int n_q, int Lq, const void *D, const int32_t *d_off, const uint8_t *d_clamp0, int n_d,
                  int dim, float *scores, int64_t ld_scores, uint32_t flags, int n_threads) {
    MSIM_HOST_GUARD(msim_fwd_host_impl(dtype, Q, n_q, Lq, D, d_off, d_clamp0, n_d, dim, scores, ld_scores, flags, n_threads))
}

int msim_fwd_host_lists(int dtype, const void *const *q_ptr, const int64_t *q_rows, int n_q, const void *const *d_ptr,
                        const int64_t *d_rows, const uint8_t *d_clamp0, int n_d, int dim, float *scores, int64_t ld_scores, uint32_t flags,
                        int n_threads) {
    MSIM_HOST_GUARD(msim_fwd_host_lists_impl(dtype, q_ptr, q_rows, n_q, d_ptr, d_rows, d_clamp0, n_d, dim, scores, ld_scores, flags, n_threads))
}

int msim_sim_matrix_host(int dtype, const void *A, int n_a, const void *B, int n_b, int dim, float *out, int64_t ld_out, uint32_t flags,
                         int n_threads) {
    MSIM_HOST_GUARD(msim_sim_matrix_host_impl(dtype, A, n_a, B, n_b, dim, out, ld_out, flags, n_threads))
}

int msim_host_gather_range(void *dst, const void *const *src, const int64_t *prefix, int64_t n, int64_t lo, int64_t hi, int n_threads) {
    MSIM_HOST_GUARD(host_gather_range_impl(dst, src, prefix, n, lo, hi, n_threads))
}

int msim_host_gather_range_begin(void *dst, const void *const *src, const int64_t *prefix, int64_t n, int64_t lo, int64_t hi, int n_threads) {
    MSIM_HOST_GUARD(AsyncGather::get().begin(dst, src, prefix, n, lo, hi, n_threads))
}

int msim_host_gather_range_wait(void) { MSIM_HOST_GUARD(AsyncGather::get().wait()) }

int msim_host_threads_affinity(const int32_t *cpus, int n_cpus) {
    if (!cpus || n_cpus <= 0) return fail_host(MSIM_EINVAL, "an empty CPU list");
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int i = 0; i < n_cpus; ++i) {
        if (cpus[i] < 0 || cpus[i] >= CPU_SETSIZE) return fail_host(MSIM_EINVAL, "CPU number out of range");
        CPU_SET(cpus[i], &set);
    }
    HostPool::get().set_affinity(set);
    AsyncGather::get().set_affinity(set);
    return MSIM_OK;
}

}  // extern "C"
