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
// Shape of the computation: a document is taken 16 rows at a time, widened to fp32 and transposed to k-major (Dt[k][16 rows], 8 KiB for
// dim 128: L1-resident); a block of 8 query tokens then runs acc[token][16 rows] += q[token][k] * Dt[k][:] -- per k one vector load and
// 8 broadcast-FMAs, no horizontal operation until the document ends.  Written on the compiler's generic vector type and cloned for
// AVX-512 / AVX2 / baseline x86-64 (resolved at load time); documents are dealt to std::threads in contiguous chunks.
#include <pthread.h>

#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <limits>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/maxsim.h"

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

struct HostCall {
    int dtype;
    const void *Q;
    int n_q, Lq;
    const void *D;
    const int32_t *d_off;
    const uint8_t *clamp0;
    int n_d, dim;
    float *scores;
    int64_t ld;
    bool ref_round;
};

constexpr int kTok = 8;       // query tokens per register block
constexpr int kRows = 16;     // document rows per vector

// documents [c_lo, c_hi): `qf` = all queries widened to fp32 [n_q * Lq, dim], `dt` = scratch for one 16-row group (dim * 16 floats),
// `tmax` = scratch [n_q * Lq] running maxima of the current document
__attribute__((target_clones("avx512f", "avx2,fma", "default")))
void score_range(const HostCall &c, const float *qf, float *dt, float *tmax, int c_lo, int c_hi) {
    const int dim = c.dim;
    const float ninf = -std::numeric_limits<float>::infinity();
    // Queries are taken in blocks of whole queries of about 256 tokens (128 KiB of fp32 operands at dim 128: L2-resident) and every
    // block walks ALL documents of the range before the next one starts: with the token loop innermost over the whole batch, a
    // 1000-query batch streamed 16 MB of query operands from the last-level cache for every 16-row group (measured on the 128-thread GPU host:
    // 5 GFLOP/s per thread, against 90 with the operands in L2).  The 16-row group is widened again per block -- 2048 conversions
    // against 0.5 M multiply-adds.
    const int q_per_blk = c.Lq > 0 ? (256 / c.Lq > 0 ? 256 / c.Lq : 1) : c.n_q;
    for (int q0 = 0; q0 < c.n_q; q0 += q_per_blk) {
        const int nq = c.n_q - q0 < q_per_blk ? c.n_q - q0 : q_per_blk;
        const int n_tok = nq * c.Lq;
        const float *qblk = qf + (size_t)q0 * c.Lq * dim;
        for (int doc = c_lo; doc < c_hi; ++doc) {
            const int r0 = c.d_off[doc], len = c.d_off[doc + 1] - r0;
            for (int t = 0; t < n_tok; ++t) tmax[t] = ninf;
            for (int g = 0; g < len; g += kRows) {
                const int valid = len - g < kRows ? len - g : kRows;
                // widen + transpose this group: dt[k * 16 + r] = D[r0 + g + r][k]; rows that do not exist are zero here and masked below
                for (int r = 0; r < kRows; ++r) {
                    if (r < valid) {
                        const size_t base = (size_t)(r0 + g + r) * dim;
                        for (int k = 0; k < dim; ++k) dt[k * kRows + r] = widen(c.D, c.dtype, base + k);
                    } else {
                        for (int k = 0; k < dim; ++k) dt[k * kRows + r] = 0.0f;
                    }
                }
                v16f lane_mask;                       // 0 for real rows, -inf for the others (added after the products: x + 0 = x)
                for (int r = 0; r < kRows; ++r) lane_mask[r] = r < valid ? 0.0f : ninf;
                for (int t0 = 0; t0 < n_tok; t0 += kTok) {
                    const int nt = n_tok - t0 < kTok ? n_tok - t0 : kTok;
                    v16f acc[kTok];
                    for (int i = 0; i < kTok; ++i) acc[i] = v16f{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                    const float *q = qblk + (size_t)t0 * dim;
                    if (nt == kTok) {
                        for (int k = 0; k < dim; ++k) {
                            const v16f d = *reinterpret_cast<const v16f *>(dt + k * kRows);
                            for (int i = 0; i < kTok; ++i) acc[i] += q[(size_t)i * dim + k] * d;
                        }
                    } else {
                        for (int k = 0; k < dim; ++k) {
                            const v16f d = *reinterpret_cast<const v16f *>(dt + k * kRows);
                            for (int i = 0; i < nt; ++i) acc[i] += q[(size_t)i * dim + k] * d;
                        }
                    }
                    for (int i = 0; i < nt; ++i) {
                        v16f a = acc[i];
                        if (c.ref_round) {             // the reference's 16-bit einsum rounds every similarity before the max
                            for (int r = 0; r < kRows; ++r) a[r] = c.dtype == MSIM_DTYPE_F16 ? round_f16(a[r]) : round_bf16(a[r]);
                        }
                        a += lane_mask;
                        float m = tmax[t0 + i];
                        for (int r = 0; r < kRows; ++r) m = a[r] > m ? a[r] : m;
                        tmax[t0 + i] = m;
                    }
                }
            }
            const bool clamp = c.clamp0 != nullptr && c.clamp0[doc] != 0;
            for (int qi = 0; qi < nq; ++qi) {
                float tot = 0.0f;
                for (int i = 0; i < c.Lq; ++i) {
                    float m = tmax[qi * c.Lq + i];
                    if (clamp && !(m > 0.0f)) m = m != m ? m : 0.0f;      // max(m, 0), NaN kept
                    tot += m;
                }
                if (c.ref_round) tot = c.dtype == MSIM_DTYPE_F16 ? round_f16(tot) : round_bf16(tot);
                c.scores[(size_t)(q0 + qi) * c.ld + doc] = tot;
            }
        }
    }
}

// out[i, j] = <A_i, B_j> for the rows [b_lo, b_hi) of B: the same register blocking without the reduction (score_single_vector,
// processing_utils.py:126 einsum("bd,cd->bc"))
__attribute__((target_clones("avx512f", "avx2,fma", "default")))
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

int msim_fwd_host(int dtype, const void *Q, int n_q, int Lq, const void *D, const int32_t *d_off, const uint8_t *d_clamp0, int n_d,
                  int dim, float *scores, int64_t ld_scores, uint32_t flags, int n_threads) {
    auto fail = [](int code, const char *msg) {
        strncpy(g_host_err, msg, sizeof(g_host_err) - 1);
        return code;
    };
    if (n_q < 0 || n_d < 0 || Lq < 0 || dim <= 0) return fail(MSIM_EINVAL, "negative size");
    if (n_q == 0 || n_d == 0) return MSIM_OK;
    if (!Q || !D || !d_off || !scores) return fail(MSIM_EINVAL, "null pointer argument");
    if (dtype != MSIM_DTYPE_BF16 && dtype != MSIM_DTYPE_F16 && dtype != MSIM_DTYPE_F32)
        return fail(MSIM_EUNSUPPORTED, "msim_fwd_host takes bfloat16 (0), float16 (1) or float32 (2) embeddings");
    if (ld_scores < n_d) return fail(MSIM_EINVAL, "ld_scores < n_d");
    if (flags & ~(MSIM_FLAG_REF_ROUNDING)) return fail(MSIM_EINVAL, "unknown flags");
    if ((flags & MSIM_FLAG_REF_ROUNDING) && dtype == MSIM_DTYPE_F32) flags &= ~MSIM_FLAG_REF_ROUNDING;   // fp32 inputs: nothing is rounded
    HostCall c{dtype, Q, n_q, Lq, D, d_off, d_clamp0, n_d, dim, scores, ld_scores, (flags & MSIM_FLAG_REF_ROUNDING) != 0};
    const size_t n_tok = (size_t)n_q * Lq;
    std::vector<float> qf(n_tok * dim + 16);
    for (size_t i = 0; i < n_tok * dim; ++i) qf[i] = widen(Q, dtype, i);
    int nt = n_threads < 1 ? 1 : (n_threads > 256 ? 256 : n_threads);
    const int64_t rows = d_off[n_d] - d_off[0];
    const double work = (double)rows * (double)n_tok * dim;            // multiply-adds
    const int by_work = (int)(work / 4e6) + 1;                          // a thread is worth starting for a few million of them
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
            acc += d_off[d + 1] - d_off[d];
            if ((acc >= per && (int)cut.size() < n_chunks) || d + 1 == n_d) {
                cut.push_back(d + 1);
                acc = 0;
            }
        }
    }
    const std::function<void(int)> body = [&](int chunk) {
        thread_local std::vector<float> dt, tmax;
        if (dt.size() < (size_t)dim * kRows + 16) dt.resize((size_t)dim * kRows + 16);
        if (tmax.size() < n_tok + 16) tmax.resize(n_tok + 16);
        float *dta = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(dt.data()) + 63) & ~(uintptr_t)63);
        score_range(c, qf.data(), dta, tmax.data(), cut[chunk], cut[chunk + 1]);
    };
    HostPool::get().run((int)cut.size() - 1, nt, body);
    return MSIM_OK;
}


int msim_sim_matrix_host(int dtype, const void *A, int n_a, const void *B, int n_b, int dim, float *out, int64_t ld_out, uint32_t flags,
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
    const int by_work = (int)((double)n_a * n_b * dim / 4e6) + 1;
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
