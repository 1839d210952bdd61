// C ABI of libmaxsim_gfx950.so (see include/maxsim.h).  Host-side dispatch only: argument
// validation, kernel selection and launch on the caller's stream.  Nothing here allocates,
// frees or synchronises, so every entry point is hipGraph-capturable.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/maxsim.h"
#include "maxsim_stream.hip"
#include "maxsim_batch.hip"
#if defined(MSIM_AB) || defined(MSIM_TRACE)
#include "maxsim_batch_packed.hip"     // K1bK: several short documents per chunk -- measured, slower than K1b, measurement builds only
#endif
#include "maxsim_batch_t.hip"
#include "maxsim_dense_t.hip"
#include "maxsim_pairs.hip"
#include "maxsim_generic.hip"
#include "maxsim_bwd.hip"
#include "maxsim_panels.hip"
#include "maxsim_smooth.hip"
#include "embed_head.hip"
#include "token_pooling.hip"
#include "loss_epilogue.hip"
#include "topk_select.hip"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// an integer A/B knob from the environment -- measurement builds only (maxsim_common.hpp: kAbBuild); the shipped library returns
// the default without looking
int ab_env(const char *name, int dflt) {
#if defined(MSIM_AB) || defined(MSIM_TRACE)
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

struct DeviceInfo {
    int cus = 0;
    int lds_per_cu = 0;
};

// once-initialised per-device cache (the only mutable global state of the library)
constexpr int kMaxDevices = 64;
DeviceInfo g_dev[kMaxDevices];
std::atomic<int> g_dev_ready[kMaxDevices];

int device_info(const DeviceInfo **out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(MSIM_ELAUNCH, "hipGetDevice: %s", hipGetErrorString(e));
    if (dev < 0 || dev >= kMaxDevices) return fail(MSIM_ELAUNCH, "device ordinal %d out of range", dev);
    if (!g_dev_ready[dev].load(std::memory_order_acquire)) {
        hipDeviceProp_t p;
        e = hipGetDeviceProperties(&p, dev);
        if (e != hipSuccess) return fail(MSIM_ELAUNCH, "hipGetDeviceProperties: %s", hipGetErrorString(e));
        if (strncmp(p.gcnArchName, "gfx950", 6) != 0)
            return fail(MSIM_EUNSUPPORTED, "libmaxsim_gfx950 is built for gfx950 (MI355X) only; device %d is %s", dev,
                        p.gcnArchName);
        g_dev[dev].cus = p.multiProcessorCount;
        g_dev[dev].lds_per_cu = 160 * 1024;
        g_dev_ready[dev].store(1, std::memory_order_release);
    }
    *out = &g_dev[dev];
    return MSIM_OK;
}

// kernels that ask for more than 64 KiB of dynamic LDS need the attribute raised once per (kernel, device)
template <class Kern>
int allow_lds(Kern kern, int bytes, std::atomic<int> *configured) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!configured[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return fail(MSIM_ELAUNCH, "hipFuncSetAttribute(%d B LDS): %s", bytes, hipGetErrorString(e));
        configured[dev].store(1, std::memory_order_release);
    }
    return MSIM_OK;
}

// tuned = the dim=128 16-bit kernels (K1s / K1b / pair-list); everything else goes to the generic kernels (K1g)
bool is_tuned(int dtype, int dim, int Lq) {
    return (dtype == MSIM_DTYPE_BF16 || dtype == MSIM_DTYPE_F16) && dim == msim::kDim &&
           (Lq + msim::kTokTile - 1) / msim::kTokTile <= 4;
}

// queries longer than 128 tokens in the tuned dtype / width: scored as 128-token segments on K1b (MaxSim is a sum over query
// tokens) when the caller passes scratch for the partial sums; otherwise (and for every other shape) the generic kernels take them
constexpr int kLongSegRows = 4 * msim::kTokTile;
bool is_long_tuned(int dtype, int dim, int Lq) {
    return (dtype == MSIM_DTYPE_BF16 || dtype == MSIM_DTYPE_F16) && dim == msim::kDim && Lq > kLongSegRows;
}
int long_segments(int Lq) { return (Lq + kLongSegRows - 1) / kLongSegRows; }

int elem_bytes(int dtype) { return dtype == MSIM_DTYPE_F32 ? 4 : 2; }

int check_common(const void *Q, const void *D, const int32_t *d_off, int dtype, int dim, int Lq) {
    if (!Q || !D || !d_off) return fail(MSIM_EINVAL, "null pointer argument");
    if (dtype != MSIM_DTYPE_BF16 && dtype != MSIM_DTYPE_F16 && dtype != MSIM_DTYPE_F32)
        return fail(MSIM_EUNSUPPORTED, "dtype code %d: the gfx950 kernels take bfloat16 (0), float16 (1) or float32 (2) embeddings",
                    dtype);
    if ((reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(D)) & 15)
        return fail(MSIM_EINVAL, "Q and D must be 16-byte aligned");
    if (dim <= 0) return fail(MSIM_EINVAL, "dim=%d", dim);
    if (!is_tuned(dtype, dim, Lq)) {
        const long long row_bytes = (long long)dim * elem_bytes(dtype);
        if (row_bytes % 32 != 0)
            return fail(MSIM_EUNSUPPORTED, "dim=%d: an embedding row must be a multiple of 32 bytes (pad the width with zero columns)", dim);
        if (row_bytes > msim::kGenericMaxRowBytes)
            return fail(MSIM_EUNSUPPORTED, "dim=%d: embedding rows above %d bytes are not supported", dim, msim::kGenericMaxRowBytes);
    }
    return MSIM_OK;
}

struct FwdCall {
    const uint16_t *Q, *D;       // Q: the [n_q, Lq, 128] box (uniform queries: also a flat token matrix) or the flat token matrix (q_off)
    const int32_t *d_off;
    const uint8_t *clamp0;
    float *scores;
    long long ld;
    int n_q, Lq, n_d;
    unsigned flags;
    const DeviceInfo *di;
    hipStream_t st;
    void *workspace = nullptr;   // msim_fwd_workspace_bytes() bytes or null
    // the flat token layout (maxsim_common.hpp)
    const int32_t *q_off = nullptr;        // device: token offsets [n_q + 1]; null = uniform queries of Lq tokens
    const int32_t *q_off_host = nullptr;   // the same numbers on the host: the plan below is made from them
    int seg = 0, n_seg = 1;                // uniform long queries: n_q counts PIECES of `seg` tokens (FlatQ::n_seg)
    int avg_rows = 0;                      // the caller's hint: average rows per document (MSIM_FLAG_AVG_ROWS), 0 = unknown
};

constexpr size_t kFwdWorkspaceBytes = 4096;   // K1b's convoy counters: n_ranges * n_qblocks <= 8 * 64 ints

constexpr int kStreamRing = 4;  // default slabs per wave-private ring: 4 waves x 4 x 8 KiB = 128 KiB per workgroup (launch_stream picks 2 for 5-8 units)

// host mirror of msim::flat_qoff
struct HostQ {
    const int32_t *off;
    int Lq, seg, n_seg;
    int at(int i) const {
        if (off) return off[i];
        if (n_seg <= 1) return i * Lq;
        const int r = i / n_seg, s = i - r * n_seg;
        const int o = s * seg;
        return r * Lq + (o < Lq ? o : Lq);
    }
};
HostQ host_q(const FwdCall &c) { return HostQ{c.q_off_host, c.Lq, c.seg, c.n_seg}; }
msim::FlatQ flat_q(const FwdCall &c) { return msim::FlatQ{c.q_off, c.Lq, c.seg, c.n_seg}; }

// Cache policy of K1s's document stream: every byte is read once by one CU, so the LDS-DMA loads carry `nt`
// (do not allocate in L2 / MALL).  Measured on MI355X, 16 GiB shard: 6.31 -> 7.02 TB/s at 1 query, 6.08 -> 6.45 TB/s
// at 4 queries.  MSIM_STREAM_NT=0 switches it off for A/B measurements (tuning knob, not part of the ABI).
int stream_nt() {
    static const int v = ab_env("MSIM_STREAM_NT", 1) != 0;
    return v;
}

template <int NU, bool F16, int AUX, bool IL, int RING = kStreamRing>
int launch_stream_aux(const FwdCall &c) {
    auto kern = msim::maxsim_stream_kernel<NU, RING, F16, AUX, IL>;
    constexpr int lds = 4 * (RING * msim::kSlabBytes + msim::kStreamTokBytes);
    static std::atomic<int> configured[kMaxDevices];
    if (int rc = allow_lds(kern, lds, configured)) return rc;
    msim::StreamArgs a;
    a.ld = c.ld;
    a.fq = flat_q(c);
    a.n_q = c.n_q;
    a.n_d = c.n_d;
    a.flags = c.flags;
    const int wg_needed = (c.n_d + 3) / 4;
    const int wg_cap = c.di->cus * (c.di->lds_per_cu / lds);
    hipLaunchKernelGGL(kern, dim3(wg_needed < wg_cap ? wg_needed : wg_cap), dim3(256), lds, c.st, c.Q, c.D, c.d_off,
                       c.clamp0, c.scores, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MSIM_ELAUNCH, "maxsim_stream_kernel<%d> launch: %s", NU, hipGetErrorString(e));
    return MSIM_OK;
}

// LDS-DMA pieces of the next slab issued between the MFMAs of the current one (default).  Interleaved A/B on a 16 GiB
// shard (profiles/r01_logs/ab_stream_il.log): 4 queries 6.37-6.43 -> 6.52-6.55 TB/s, 6-8 queries +1 %, 1-2 queries unchanged.
// MSIM_STREAM_IL=0 selects the block-issue variant (tuning knob, not part of the ABI).
int stream_il() {
    static const int v = ab_env("MSIM_STREAM_IL", 1) != 0;
    return v;
}

template <int NU, bool F16>
int launch_stream(const FwdCall &c) {
    // 5-8 units (3-4 token tiles of 32): a 2-slab ring (72 KiB per workgroup) lets TWO workgroups share a CU -- two waves per SIMD,
    // one covering the other's DMA issue / operand reads / max folds: 4 queries 6.51-6.55 -> 6.82 TB/s (85 % of spec); up to 4 units
    // are at the stream ceiling either way and keep the deeper ring.
    // MSIM_STREAM_RING=2|4 forces one of them (tuning knob, not part of the ABI).
    constexpr int ring_default = NU >= 5 ? 2 : kStreamRing;
#ifdef MSIM_AB
    {
        static const int ring_env = ab_env("MSIM_STREAM_RING", 0);
        const int ring = ring_env ? ring_env : ring_default;
        if (ring == 2) return launch_stream_aux<NU, F16, 2, true, 2>(c);
        if (stream_il() && stream_nt()) return launch_stream_aux<NU, F16, 2, true>(c);
        return stream_nt() ? launch_stream_aux<NU, F16, 2, false>(c) : launch_stream_aux<NU, F16, 0, false>(c);
    }
#endif
    return launch_stream_aux<NU, F16, 2, true, ring_default>(c);     // nt stream, interleaved DMA issue
}

// ---- the plan of one tuned forward call, made on the host from the queries' lengths -- ONE definition, shared by the dispatch and
// by msim_fwd_workspace_bytes (which must report scratch exactly when the launch would use it: round-2 advisor finding).
//   * up to 8 queries / 8 units (128 tokens) in total: K1s, every wave holds all units (HBM-bound regime, one pass over the corpus,
//     no barriers at all);
//   * more: K1b, whose query blocks hold WHOLE queries -- at most nw * maxu units of tokens and nw * 8 queries (one 8-lane group of the
//     workgroup per query in the reduction).  One block if the batch fits one of the six shapes
//         units <= 16: pair (2 waves) | <= 20: 4 waves x 5 units, three workgroups per CU | <= 32: 4 waves | <= 40: 4 waves x 10 | <= 64: 8 waves | <= 80: 8 x 10
//     (the measured ladder of rounds 2-3 in 16-token units: profiles/r02_logs/ab_ridge.log, r03_logs/ab_batch_t5.log,
//     ab_batch8_final.log), else several blocks on the 8-wave form, filled greedily in query order and then re-cut evenly; eight or
//     ten units per wave by cost: a block's pace is set by its heaviest wave, so a plan costs (blocks) x (units of the heaviest
//     wave), and the ten-unit bodies run ~10 % behind the eight-unit ones per step.
struct FlatPlan {
    bool stream = false;
    int nu = 0;                   // K1s: units per wave
    int nw = 0, maxu = 0;         // K1b
    std::vector<int> blk_q0;      // K1b: first query of every block, + n_q
    int n_blocks() const { return (int)blk_q0.size() - 1; }
};

// greedy fill in query order under (token, query) capacities; false if one query alone exceeds a block
bool fill_blocks(const HostQ &hq, int n_q, int nw, int maxu, std::vector<int> &blk) {
    const int cap_tok = nw * maxu * msim::kUnitTok, cap_q = nw * 8;
    blk.clear();
    blk.push_back(0);
    int cur_tok = 0, cur_q = 0;
    for (int i = 0; i < n_q; ++i) {
        const int len = hq.at(i + 1) - hq.at(i);
        if (len > cap_tok) return false;
        if (cur_q == cap_q || cur_tok + len > cap_tok) {
            blk.push_back(i);
            cur_tok = 0;
            cur_q = 0;
        }
        cur_tok += len;
        ++cur_q;
    }
    blk.push_back(n_q);
    return true;
}

// the same number of blocks, cut evenly by tokens (a block's pace is its heaviest wave: 31 + 32 + 31 + ... beats 32 + 32 + ... + 8);
// kept only if every block still fits
void balance_blocks(const HostQ &hq, int n_q, int nw, int maxu, std::vector<int> &blk) {
    const int nb = (int)blk.size() - 1;
    if (nb <= 1) return;
    const int cap_tok = nw * maxu * msim::kUnitTok, cap_q = nw * 8;
    const long long base = hq.at(0), total = (long long)hq.at(n_q) - base;
    std::vector<int> cut(1, 0);
    int q = 0;
    for (int b = 1; b < nb; ++b) {
        const long long want = (total * b + nb - 1) / nb;         // tokens in front of block b
        while (q < n_q && (long long)hq.at(q) - base < want) ++q;
        if (q <= cut.back()) q = cut.back() + 1;
        if (q >= n_q) return;
        cut.push_back(q);
    }
    cut.push_back(n_q);
    for (int b = 0; b < nb; ++b)
        if (hq.at(cut[b + 1]) - hq.at(cut[b]) > cap_tok || cut[b + 1] - cut[b] > cap_q) return;
    blk.swap(cut);
}

int heaviest_wave_units(const HostQ &hq, const std::vector<int> &blk, int nw) {
    int worst = 0;
    for (size_t b = 0; b + 1 < blk.size(); ++b) {
        const int units = (hq.at(blk[b + 1]) - hq.at(blk[b]) + msim::kUnitTok - 1) / msim::kUnitTok;
        const int w = (units + nw - 1) / nw;
        if (w > worst) worst = w;
    }
    return worst;
}

// MSIM_BATCH_NW=2|4|8 forces the number of waves that share a document stream in K1b (tuning knob for A/B measurements, not
// part of the ABI)
int batch_nw_override() {
    static const int v = [] {
        const int x = ab_env("MSIM_BATCH_NW", 0);
        return (x == 2 || x == 4 || x == 8) ? x : 0;
    }();
    return v;
}

int flat_plan(const HostQ &hq, int n_q, FlatPlan &p) {
    const long long tokens = (long long)hq.at(n_q) - hq.at(0);
    if (tokens < 0) return fail(MSIM_EINVAL, "query token offsets are not non-decreasing");
    const long long units = (tokens + msim::kUnitTok - 1) / msim::kUnitTok;
    p = FlatPlan{};
    if (n_q <= 8 && units <= 8) {
        p.stream = true;
        p.nu = units > 0 ? (int)units : 1;
        return MSIM_OK;
    }
    static const int shapes[7][2] = {{2, 8}, {4, 5}, {2, 10}, {4, 8}, {4, 10}, {8, 8}, {8, 10}};
    // round 4 re-measured the ladder in units (profiles/r04_logs/ab_plan_ladder.log, 16 GiB shard, random rows / zero-filled shard):
    // 18 units: pair x 10 (9 + 9) 4.12 ms / 2.97 vs 4 waves (5/5/4/4) 4.23 / 2.99; 20 units: 4 waves x 5 units 4.38 / 3.04 vs the
    // pair 4.42-4.45 / 3.19 -- four evenly loaded waves beat two ten-unit ones once the units divide by four.  Then the FIVE-unit
    // form of the 4-wave shape: a kernel that never holds more than five units needs 168 registers, so THREE workgroups share a CU
    // (three waves per SIMD; 2-chunk ring, 37.5 KiB of LDS each) -- 17..20 units: 9 queries 4.02 -> 3.98 ms, 10 queries 4.25 -> 4.13
    // (zero shard 0.726 -> 0.747, 0.731 -> 0.740), bit-identical (profiles/r04_logs/ab_five_units_3wg.log; with the 3-chunk ring the
    // third workgroup does not fit the LDS and it is slower than the two-workgroup form).  The pair x 10 shape is a measurement-build
    // shape since
    const int forced = batch_nw_override();
    static const int forced_maxu = ab_env("MSIM_BATCH_MAXU", 0);      // 8 | 10 (measurement builds)
#ifdef MSIM_AB
    // measurement builds only (next lead): the five-unit instantiation of the PAIR form -- six pairs per CU
    if (forced == 2 && forced_maxu == 5 && units <= 10 && n_q <= 16 && fill_blocks(hq, n_q, 2, 5, p.blk_q0) && p.n_blocks() == 1) {
        p.nw = 2;
        p.maxu = 5;
        return MSIM_OK;
    }
#endif
    for (const auto &sh : shapes) {
        if (forced && sh[0] != forced) continue;
        if (forced_maxu && sh[1] != forced_maxu) continue;
        if (units > sh[0] * sh[1] || n_q > sh[0] * 8) continue;
        if (sh[0] == 2 && sh[1] == 10 && !(forced == 2 || forced_maxu == 10)) continue;       // superseded by {4, 5}: measurement builds only
        if (!fill_blocks(hq, n_q, sh[0], sh[1], p.blk_q0) || p.n_blocks() != 1) continue;
        p.nw = sh[0];
        p.maxu = sh[1];
        return MSIM_OK;
    }
#ifdef MSIM_AB
    // measurement builds only (round 5, short documents): several blocks on the FOUR-wave form -- two workgroups per CU, 64-row chunks,
    // one covering the other's per-document barriers -- instead of the eight-wave form (MSIM_BATCH_NW=4 with more than 40 units)
    if (forced == 4) {
        std::vector<int> b4;
        if (fill_blocks(hq, n_q, 4, 8, b4)) {
            balance_blocks(hq, n_q, 4, 8, b4);
            p.nw = 4;
            p.maxu = 8;
            p.blk_q0.swap(b4);
            return MSIM_OK;
        }
    }
#endif
    std::vector<int> b8, b10;
    if (!fill_blocks(hq, n_q, 8, 8, b8)) {
        if (!fill_blocks(hq, n_q, 8, 10, b10))
            return fail(MSIM_EUNSUPPORTED, "a query of more than %d tokens does not fit one query block", 8 * 10 * msim::kUnitTok);
        balance_blocks(hq, n_q, 8, 10, b10);
        p.nw = 8;
        p.maxu = 10;
        p.blk_q0.swap(b10);
        return MSIM_OK;
    }
    balance_blocks(hq, n_q, 8, 8, b8);
    p.nw = 8;
    p.maxu = 8;
    if (fill_blocks(hq, n_q, 8, 10, b10)) {
        balance_blocks(hq, n_q, 8, 10, b10);
        const long long c8 = (long long)(b8.size() - 1) * heaviest_wave_units(hq, b8, 8);
        const long long c10 = (long long)(b10.size() - 1) * heaviest_wave_units(hq, b10, 8);
        // ... or no slower AND the only one whose blocks are all resident at once (32 CUs per XCD on MI355X: one document range per
        // XCD, held together by the convoy, every corpus byte read from HBM once).  1000 queries x 40 tokens: 40 blocks of 64 units
        // run as four ranges x five rounds and read the corpus TWICE (PMC: 66.5 GB for 33.3 GB); 32 blocks of 80 units read it once
        // in the same time (99.3 vs 99.4-99.8 ms, profiles/r06_logs/ab_short_docs_packed.log section 7)
        // (a one-round plan takes as long as its heaviest wave whatever the number of blocks: only when the round is full, 31 or 32
        // blocks -- with 28 ten-unit blocks four CUs per XCD idle and the eight-unit plan's sub-ranges win by 13 %)
        const bool single_round10 = b10.size() - 1 <= 32 && b10.size() - 1 >= 31 && b8.size() - 1 > 32;
        if (11 * c10 < 10 * c8 || (c10 <= c8 && single_round10)) {
            p.maxu = 10;
            p.blk_q0.swap(b10);
            return MSIM_OK;
        }
    }
    p.blk_q0.swap(b8);
    return MSIM_OK;
}

// document ranges per XCD of a K1b / K1bPF launch (the rule is explained where launch_batch calls it): one block -> one range per
// resident workgroup; several blocks -> the smallest number of ranges whose last round of resident workgroups is >= 97 % full, else the
// fullest; never ranges of fewer than `min_docs` documents (~2000 rows by the caller's MSIM_FLAG_AVG_ROWS hint, 64 documents without it)
int ranges_per_xcd(int n_qblocks, int cus_per_xcd, int n_d, int avg_rows) {
    int min_docs = 64;
    if (avg_rows > 0) {
        min_docs = (2048 + avg_rows - 1) / avg_rows;
        min_docs = min_docs < 2 ? 2 : (min_docs > 64 ? 64 : min_docs);
    }
    int sub = n_qblocks >= cus_per_xcd ? 1 : cus_per_xcd / n_qblocks;
    if (n_qblocks > 1) {
        double best = 0.0;
        int best_sub = 1;
        for (int s = 1; s <= 32; ++s) {
            if (s > 1 && (long long)8 * s * min_docs > n_d) break;
            const long long slots_x = (long long)n_qblocks * s;
            const long long rounds = (slots_x + cus_per_xcd - 1) / cus_per_xcd;
            const double eff = (double)slots_x / (double)(rounds * cus_per_xcd);
            if (eff > best + 1e-9) {
                best = eff;
                best_sub = s;
            }
            if (eff >= (s == 1 ? 0.93 : 0.97)) break;    // one range per XCD keeps every block of a range resident at once: the convoy
                                                        // holds them together and the range is fetched from HBM once (30 blocks on 32
                                                        // CUs: 33 GB per launch instead of 354 GB in sixteen ranges, for the same time)
        }
        sub = best_sub;
    }
    return sub;
}

// PACKED (measurement builds): K1bK (maxsim_batch_packed.hip), the eight-wave form whose chunks hold several short documents
template <bool F16, int NW, int RING, int AUX, int MAXU, bool PACKED>
auto batch_kernel_ptr() {
#if defined(MSIM_AB) || defined(MSIM_TRACE)
    if constexpr (PACKED) return msim::maxsim_batch_packed_kernel<F16, AUX, MAXU>;
    else
#endif
    return msim::maxsim_batch_kernel<F16, NW, RING, AUX, MAXU>;
}

template <bool F16, int NW, int RING = 3, int AUX = 0, int MAXU = 8, bool PACKED = false>
int launch_batch(const FwdCall &c, const FlatPlan &plan) {
    static_assert(!PACKED || (NW == 8 && RING == 3), "K1bK is the eight-wave form");
    auto kern = batch_kernel_ptr<F16, NW, RING, AUX, MAXU, PACKED>();
    // ring + the per-token max table(s) + the queries' token ranges
    constexpr int lds = RING * (NW / 2) * msim::kSlabBytes +
                        (PACKED ? 2 * (NW / 2) * NW * MAXU * msim::kUnitTok * 4 : (NW > 2 ? 2 : 1) * NW * MAXU * msim::kUnitTok * 16) +
                        NW * 8 * 8;
    constexpr int wg_per_cu = MAXU == 5 ? 12 / NW : 8 / NW;  // the five-unit form: 168 registers, three waves per SIMD (three 4-wave workgroups per CU)
    static std::atomic<int> configured[kMaxDevices];
    if (int rc = allow_lds(kern, lds, configured)) return rc;
    // blockIdx -> (XCD = b % 8, slot = b / 8): the CUs of one XCD share a document range through its L2
    const int cus_per_xcd = (c.di->cus / 8 > 0 ? c.di->cus / 8 : 1) * wg_per_cu;   // resident workgroups per XCD
    static const int over_env = ab_env("MSIM_BATCH_OVER", NW == 4 ? 8 : 1);
    static const bool convoy_off = ab_env("MSIM_BATCH_CONVOY", 1) == 0;
    const int total_blocks = plan.n_blocks();
    for (int b0 = 0; b0 < total_blocks; b0 += msim::kMaxQBlocks) {       // the block table travels in the kernel arguments
        msim::BatchArgs a{};
        a.ld = c.ld;
        a.fq = flat_q(c);
        a.n_d = c.n_d;
        a.flags = c.flags;
        a.n_qblocks = total_blocks - b0 < msim::kMaxQBlocks ? total_blocks - b0 : msim::kMaxQBlocks;
        for (int b = 0; b <= a.n_qblocks; ++b) a.blk_q0[b] = plan.blk_q0[b0 + b];
        // Ranges per XCD.  An XCD's workgroups are handed out in blockIdx order, range-major: (range 0, all query blocks), (range 1, ...),
        // so the slots of one XCD are a queue its CUs drain.  One block: one range per resident workgroup.  Several blocks: the number
        // of slots (blocks x ranges) should fill whole rounds of the XCD's resident workgroups -- 40 blocks on 32 CUs in ONE range each
        // are two rounds, the second a quarter full (round 4, 1000 queries x 40 tokens: 1125 ms; four ranges = 160 slots = five full
        // rounds); 20 blocks in one range leave 12 of 32 CUs idle.  Smallest number of ranges whose last round is >= 97 % full, else
        // the fullest; never ranges of fewer than two documents.
        // a range costs its workgroup a prologue (the block's 64 units: 256 KiB out of L2) whatever its size, so it should hold ~2000
        // rows.  The host does not see the row offsets: without the caller's hint (MSIM_FLAG_AVG_ROWS) that is taken as 64 documents.
        // (Round 5: with that floor alone a 1000-page corpus of 1030-row pages -- the drop-in call of BASELINE config 2 -- got ONE range
        // per XCD: 4 query blocks x 8 ranges = 32 workgroups on 256 CUs.)
        int sub = ranges_per_xcd(a.n_qblocks, cus_per_xcd, c.n_d, c.avg_rows);
        // Two workgroups share a CU in the 4-wave form, and the matrix pipe serves the OLDER wave first: of two workgroups that start
        // together one finishes after ~2/3 of the launch and the other runs its last third alone, one wave per SIMD, which cannot fill
        // the pipe (tools/trace_batch.py: workgroup 0 busy for 68 % of the launch; SQ_WAVE_CYCLES: 83 % occupancy).  With 8 x more, smaller
        // ranges than resident workgroups a finished workgroup is replaced at once and the lone phase shrinks to the last range: +3-5 %
        // at 9..16 queries (profiles/r02_logs/ab_batch_over.log; the pair form does not gain and keeps one range per resident
        // workgroup).  Only when one query block streams the corpus (no L2 sharing between blocks to preserve), and never down
        // to ranges of fewer than ~16 documents.
        if (NW < 8 && a.n_qblocks == 1 && over_env > 1) {
            int over = over_env;
            while (over > 1 && (long long)8 * sub * over * 16 > c.n_d) over >>= 1;
            sub *= over;
        }
        a.n_ranges = 8 * sub;
        const int slots = sub > 1 ? a.n_qblocks * sub : a.n_qblocks;
        // convoy (maxsim_batch.hip): only when several query blocks share a range AND all of them are resident at once
        a.convoy = nullptr;
        if (c.workspace && !convoy_off && a.n_qblocks > 1 && (long long)a.n_qblocks * sub <= cus_per_xcd && a.n_qblocks <= 64 &&
            (size_t)a.n_ranges * a.n_qblocks * sizeof(int) <= kFwdWorkspaceBytes) {
            a.convoy = static_cast<int *>(c.workspace);
            if (hipMemsetAsync(a.convoy, 0, (size_t)a.n_ranges * a.n_qblocks * sizeof(int), c.st) != hipSuccess)
                return fail(MSIM_ELAUNCH, "hipMemsetAsync(convoy counters) failed");
        }
        a.trace = nullptr;
#ifdef MSIM_TRACE
        if (const char *tp = getenv("MSIM_BATCH_TRACE_PTR")) a.trace = reinterpret_cast<unsigned long long *>(strtoull(tp, nullptr, 0));
#endif
        hipLaunchKernelGGL(kern, dim3(8 * slots), dim3(NW * 64), lds, c.st, c.Q, c.D, c.d_off, c.clamp0, c.scores, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(MSIM_ELAUNCH, "maxsim_batch%s_kernel<%d,%d> launch: %s", PACKED ? "_packed" : "", NW, MAXU, hipGetErrorString(e));
    }
    return MSIM_OK;
}

template <bool F16>
int fwd_dispatch(const FwdCall &c) {
    thread_local FlatPlan plan;
    if (int rc = flat_plan(host_q(c), c.n_q, plan)) return rc;
    if (plan.stream) {
        switch (plan.nu) {
            case 1: return launch_stream<1, F16>(c);
            case 2: return launch_stream<2, F16>(c);
            case 3: return launch_stream<3, F16>(c);
            case 4: return launch_stream<4, F16>(c);
            case 5: return launch_stream<5, F16>(c);
            case 6: return launch_stream<6, F16>(c);
            case 7: return launch_stream<7, F16>(c);
            default: return launch_stream<8, F16>(c);
        }
    }
    // one query block = every corpus byte is read once by one workgroup: stream it past L2 / MALL (nt), like K1s does.
    // MSIM_BATCH_NT=0 switches that off for A/B measurements (tuning knob, not part of the ABI).
    static const bool nt_off = ab_env("MSIM_BATCH_NT", 1) == 0;
    const bool single = plan.n_blocks() == 1 && !nt_off;
    if (plan.nw == 4 && plan.maxu == 5) {
#ifdef MSIM_AB
        static const int ring5 = ab_env("MSIM_BATCH_RING5", 2);
        if (ring5 == 3) return launch_batch<F16, 4, 3, 2, 5>(c, plan);
#endif
        return launch_batch<F16, 4, 2, 2, 5>(c, plan);
    }
#ifdef MSIM_AB
    if (plan.nw == 2 && plan.maxu == 10) return launch_batch<F16, 2, 4, 2, 10>(c, plan);
    if (plan.nw == 2 && plan.maxu == 5) {
        static const int ring5p = ab_env("MSIM_BATCH_RING5", 2);
        return ring5p == 3 ? launch_batch<F16, 2, 3, 2, 5>(c, plan) : launch_batch<F16, 2, 2, 2, 5>(c, plan);
    }
#endif
    if (plan.nw == 2) return launch_batch<F16, 2, 4, 2, 8>(c, plan);
#ifdef MSIM_AB
    if (plan.nw == 4 && plan.maxu == 8 && !single) return launch_batch<F16, 4, 3, 0, 8>(c, plan);
#endif
    if (plan.nw == 4) return plan.maxu == 10 ? launch_batch<F16, 4, 3, 2, 10>(c, plan) : launch_batch<F16, 4, 3, 2, 8>(c, plan);
#if defined(MSIM_AB) || defined(MSIM_TRACE)
    // measurement builds only (round 6, short documents): MSIM_BATCH_PACKED=1 -- chunks that hold several documents (K1bK); the same
    // bits, 4-9 % slower than K1b at 64 / 343 rows (profiles/r06_logs/ab_short_docs_packed.log)
    static const int packed_env = ab_env("MSIM_BATCH_PACKED", 0);
    if (packed_env != 0) {
        if (plan.maxu == 10) return single ? launch_batch<F16, 8, 3, 2, 10, true>(c, plan) : launch_batch<F16, 8, 3, 0, 10, true>(c, plan);
        return single ? launch_batch<F16, 8, 3, 2, 8, true>(c, plan) : launch_batch<F16, 8, 3, 0, 8, true>(c, plan);
    }
#endif
    if (plan.maxu == 10) return single ? launch_batch<F16, 8, 3, 2, 10>(c, plan) : launch_batch<F16, 8, 3, 0, 10>(c, plan);
    return single ? launch_batch<F16, 8, 3, 2, 8>(c, plan) : launch_batch<F16, 8, 3, 0, 8>(c, plan);
}

// scratch of a tuned forward call: K1b's convoy counters, needed once several query blocks stream a document range
size_t flat_workspace_bytes(const HostQ &hq, int n_q) {
    thread_local FlatPlan plan;
    if (flat_plan(hq, n_q, plan) != MSIM_OK) return 0;
    return (!plan.stream && plan.n_blocks() > 1) ? kFwdWorkspaceBytes : 0;
}

template <int TPQ, bool F16, int WPP, int RING = msim::kPairsRing>
int launch_pairs_argmax(const uint16_t *Q, const uint16_t *D, const int32_t *d_off, const uint8_t *clamp0,
                        const int32_t *pairs, float *out_scores, int32_t *out_argmax, const msim::PairsArgs &a,
                        const DeviceInfo &di, hipStream_t st) {
    auto kern = msim::maxsim_pairs_argmax_kernel<TPQ, F16, WPP, RING>;
    constexpr int lds = 4 * RING * msim::kSlabBytes + (WPP > 1 ? 4 * TPQ * msim::kTokTile * 8 : 0);
    static std::atomic<int> configured[kMaxDevices];
    if (int rc = allow_lds(kern, lds, configured)) return rc;
    const int wg_needed = WPP > 1 ? a.n_pairs : (a.n_pairs + 3) / 4;
    const int wg_cap = di.cus * (di.lds_per_cu / lds);
    hipLaunchKernelGGL(kern, dim3(wg_needed < wg_cap ? wg_needed : wg_cap), dim3(256), lds, st, Q, D, d_off, clamp0, pairs,
                       out_scores, out_argmax, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MSIM_ELAUNCH, "maxsim_pairs_argmax_kernel<%d,%d> launch: %s", TPQ, WPP, hipGetErrorString(e));
    return MSIM_OK;
}

template <int TPQ1, int GQ, bool F16>
int launch_allpairs_argmax(const uint16_t *Q, const uint16_t *D, const int32_t *d_off, const uint8_t *clamp0, float *out_scores,
                           long long ld, int32_t *out_argmax, const msim::PairsArgs &a, const DeviceInfo &di, hipStream_t st) {
    auto kern = msim::maxsim_allpairs_argmax_kernel<TPQ1, GQ, F16>;
    constexpr int lds = 4 * msim::kPairsRing * msim::kSlabBytes;
    static std::atomic<int> configured[kMaxDevices];
    if (int rc = allow_lds(kern, lds, configured)) return rc;
    const long long work = (long long)((a.n_q + GQ - 1) / GQ) * a.n_d;
    const long long wg_needed = (work + 3) / 4;
    const int wg_cap = di.cus * (di.lds_per_cu / lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)(wg_needed < wg_cap ? wg_needed : wg_cap)), dim3(256), lds, st, Q, D, d_off, clamp0, out_scores, ld,
                       out_argmax, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MSIM_ELAUNCH, "maxsim_allpairs_argmax_kernel<%d,%d> launch: %s", TPQ1, GQ, hipGetErrorString(e));
    return MSIM_OK;
}

template <bool F16>
int allpairs_argmax_dispatch(int tpq, const uint16_t *Q, const uint16_t *D, const int32_t *d_off, const uint8_t *clamp0, float *out_scores,
                             long long ld, int32_t *out_argmax, const msim::PairsArgs &a, const DeviceInfo &di, hipStream_t st) {
    switch (tpq) {
        case 1: return launch_allpairs_argmax<1, 4, F16>(Q, D, d_off, clamp0, out_scores, ld, out_argmax, a, di, st);
        case 2: return launch_allpairs_argmax<2, 2, F16>(Q, D, d_off, clamp0, out_scores, ld, out_argmax, a, di, st);
        case 3: return launch_allpairs_argmax<3, 1, F16>(Q, D, d_off, clamp0, out_scores, ld, out_argmax, a, di, st);
        default: return launch_allpairs_argmax<4, 1, F16>(Q, D, d_off, clamp0, out_scores, ld, out_argmax, a, di, st);
    }
}

// short pair lists (the 2B pairs of the pairwise loss): one workgroup per pair, four waves sharing the document (latency);
// long lists: one wave per pair (throughput)
constexpr int kPairsSplitMax = 1024;

template <bool F16>
int pairs_argmax_dispatch(int tpq, const uint16_t *Q, const uint16_t *D, const int32_t *d_off, const uint8_t *clamp0,
                          const int32_t *pairs, float *out_scores, int32_t *out_argmax, const msim::PairsArgs &a,
                          const DeviceInfo &di, hipStream_t st) {
    if (a.n_pairs <= di.cus) {       // every pair's workgroup is resident at once: a deep ring (three slabs in flight per wave) costs nothing
        switch (tpq) {
            case 1: return launch_pairs_argmax<1, F16, 4, 4>(Q, D, d_off, clamp0, pairs, out_scores, out_argmax, a, di, st);
            case 2: return launch_pairs_argmax<2, F16, 4, 4>(Q, D, d_off, clamp0, pairs, out_scores, out_argmax, a, di, st);
            case 3: return launch_pairs_argmax<3, F16, 4, 4>(Q, D, d_off, clamp0, pairs, out_scores, out_argmax, a, di, st);
            default: return launch_pairs_argmax<4, F16, 4, 4>(Q, D, d_off, clamp0, pairs, out_scores, out_argmax, a, di, st);
        }
    }
    if (a.n_pairs <= kPairsSplitMax) {
        switch (tpq) {
            case 1: return launch_pairs_argmax<1, F16, 4>(Q, D, d_off, clamp0, pairs, out_scores, out_argmax, a, di, st);
            case 2: return launch_pairs_argmax<2, F16, 4>(Q, D, d_off, clamp0, pairs, out_scores, out_argmax, a, di, st);
            case 3: return launch_pairs_argmax<3, F16, 4>(Q, D, d_off, clamp0, pairs, out_scores, out_argmax, a, di, st);
            default: return launch_pairs_argmax<4, F16, 4>(Q, D, d_off, clamp0, pairs, out_scores, out_argmax, a, di, st);
        }
    }
    switch (tpq) {
        case 1: return launch_pairs_argmax<1, F16, 1>(Q, D, d_off, clamp0, pairs, out_scores, out_argmax, a, di, st);
        case 2: return launch_pairs_argmax<2, F16, 1>(Q, D, d_off, clamp0, pairs, out_scores, out_argmax, a, di, st);
        case 3: return launch_pairs_argmax<3, F16, 1>(Q, D, d_off, clamp0, pairs, out_scores, out_argmax, a, di, st);
        default: return launch_pairs_argmax<4, F16, 1>(Q, D, d_off, clamp0, pairs, out_scores, out_argmax, a, di, st);
    }
}

// long queries against short documents (the trainer's symmetric direction): the transposed pair kernel, one workgroup per pair
template <int TPD, bool F16, int RING>
int launch_pairs_argmax_t(const uint16_t *Q, const uint16_t *D, const int32_t *d_off, const uint8_t *clamp0,
                          const int32_t *pairs, float *out_scores, int32_t *out_argmax, const msim::PairsArgs &a,
                          const DeviceInfo &di, hipStream_t st) {
    auto kern = msim::maxsim_pairs_argmax_t_kernel<TPD, F16, RING>;
    constexpr int lds = 4 * RING * msim::kSlabBytes + 16;
    static std::atomic<int> configured[kMaxDevices];
    if (int rc = allow_lds(kern, lds, configured)) return rc;
    const int wg_cap = 4 * di.cus * (di.lds_per_cu / lds);        // a few rounds of resident workgroups; the kernel strides beyond
    hipLaunchKernelGGL(kern, dim3(a.n_pairs < wg_cap ? a.n_pairs : wg_cap), dim3(256), lds, st, Q, D, d_off, clamp0, pairs,
                       out_scores, out_argmax, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MSIM_ELAUNCH, "maxsim_pairs_argmax_t_kernel<%d> launch: %s", TPD, hipGetErrorString(e));
    return MSIM_OK;
}

template <bool F16>
int pairs_argmax_t_dispatch(int tpd, const uint16_t *Q, const uint16_t *D, const int32_t *d_off, const uint8_t *clamp0,
                            const int32_t *pairs, float *out_scores, int32_t *out_argmax, const msim::PairsArgs &a,
                            const DeviceInfo &di, hipStream_t st) {
    // few pairs (the pairwise loss' 2B): every workgroup resident at once, a 4-slab ring per wave hides the LDS-DMA round trips;
    // many (dense upstream gradients: B x C pairs): the 2-slab ring keeps two workgroups on a CU
    if (a.n_pairs <= di.cus) {
        switch (tpd) {
            case 1: return launch_pairs_argmax_t<1, F16, 4>(Q, D, d_off, clamp0, pairs, out_scores, out_argmax, a, di, st);
            case 2: return launch_pairs_argmax_t<2, F16, 4>(Q, D, d_off, clamp0, pairs, out_scores, out_argmax, a, di, st);
            default: return launch_pairs_argmax_t<4, F16, 4>(Q, D, d_off, clamp0, pairs, out_scores, out_argmax, a, di, st);
        }
    }
    switch (tpd) {
        case 1: return launch_pairs_argmax_t<1, F16, 2>(Q, D, d_off, clamp0, pairs, out_scores, out_argmax, a, di, st);
        case 2: return launch_pairs_argmax_t<2, F16, 2>(Q, D, d_off, clamp0, pairs, out_scores, out_argmax, a, di, st);
        default: return launch_pairs_argmax_t<4, F16, 2>(Q, D, d_off, clamp0, pairs, out_scores, out_argmax, a, di, st);
    }
}

// dD for SHORT documents with LONG entry lists (the trainer's symmetric direction: maxsim_bwd.hip, dense form): number of splits of
// every document's pair list, 0 = use the row-range kernel.  A function of the sizes alone (host side, no device read).
struct DdPlan {
    int mode = 0;          // 0: the row-range kernel; 1: dense, per (document, split); 2: dense, per (pair, split)
    int splits = 0;
    size_t bytes = 0;      // scratch
};

DdPlan dd_plan(int n_pairs, int Lq, int n_d, int dim, int max_doc_rows, int cus) {
    DdPlan pl;
    if (n_d <= 0 || n_pairs <= 0 || max_doc_rows <= 0 || max_doc_rows > msim::kBwdRows || dim <= 0) return pl;
    const long long entries_per_doc = (long long)n_pairs * Lq / n_d;
    if (entries_per_doc >= 1024) {                                // long lists on average: the dense upstream gradient of ColbertLoss
        int splits = (4 * cus + n_d - 1) / n_d;                   // ~4 workgroups per CU (16-32 KiB of LDS each)
        const long long by_work = entries_per_doc / 256;          // at least 256 (pair, token) entries per split
        if (splits > by_work) splits = (int)by_work;
        if (splits > 64) splits = 64;
        pl.mode = 1;
        pl.splits = splits < 1 ? 1 : splits;
        pl.bytes = (size_t)pl.splits * n_d * max_doc_rows * dim * sizeof(float);
    } else if (Lq >= 256) {
        // few pairs, but each brings a long list to ITS document (the pairwise loss in the symmetric direction: 2B pairs of 780 tokens
        // over 256 documents -- 195 entries per document on average, 780 or more for the <= 2B documents that have any): the row-range
        // kernel walked those 780 entries as three rounds of dependent gathers on ONE workgroup per document, 159 us of a 370 us step.
        // One workgroup per (pair, split of ~64 tokens): every step of the walk is a dependent gather, so few of them per workgroup
        pl.mode = 2;
        pl.splits = Lq / 64 > 16 ? 16 : Lq / 64;
        pl.bytes = (size_t)pl.splits * n_pairs * max_doc_rows * dim * sizeof(float);
    }
    if (pl.bytes > ((size_t)256 << 20)) pl = DdPlan{};            // scratch stays bounded: the row-range kernel serves the rest
    return pl;
}

// A side stream and a few events per device, created on first use (round 6).  The two GEMM kernels of msim_dense_t_bwd (dP, dR) are
// independent of each other and each alone keeps the matrix cores ~27 % busy (latency chains, one workgroup per CU): the call forks
// them onto two streams and joins before it returns to the caller's stream, so they share the CUs (LDS 67 + 70 KiB, 4 waves per
// SIMD) and cover each other's stalls (ColbertLoss, both directions at config 5's shape: 0.468 -> 0.390 ms).  Fork / join with
// events is the capturable pattern: a hipGraph of the step gets two parallel branches.
struct SideStream {
    hipStream_t st = nullptr;
    hipEvent_t ev[8] = {};
    std::atomic<int> ready{0};
    std::atomic<unsigned> next{0};
};
SideStream g_side[kMaxDevices];

int side_stream(SideStream **out) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= kMaxDevices) return fail(MSIM_ELAUNCH, "device ordinal %d out of range", dev);
    SideStream &s = g_side[dev];
    if (!s.ready.load(std::memory_order_acquire)) {
        static std::mutex mu;
        std::lock_guard<std::mutex> lock(mu);
        if (!s.ready.load(std::memory_order_relaxed)) {
            hipError_t e = hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking);
            if (e != hipSuccess) return fail(MSIM_ELAUNCH, "hipStreamCreateWithFlags: %s", hipGetErrorString(e));
            for (auto &ev : s.ev) {
                e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
                if (e != hipSuccess) return fail(MSIM_ELAUNCH, "hipEventCreateWithFlags: %s", hipGetErrorString(e));
            }
            s.ready.store(1, std::memory_order_release);
        }
    }
    *out = &s;
    return MSIM_OK;
}

// every gradient kernel of one msim_pairs_bwd call; OUT16: dQ / dD in the embeddings' own 16-bit dtype
template <int DT, bool OUT16>
void launch_bwd_kernels(const char *Q, const char *D, const int32_t *d_off, int max_doc_rows, const int32_t *pairs,
                        const int32_t *order_by_doc, const float *g, const int32_t *argmax, void *dQ, void *dD,
                        const msim::PairsArgs &a, int dim, int cus, hipStream_t st, float *partial, const DdPlan &pl, msim::GScale gs) {
    const int row_bytes = dim * msim::elem_size<DT>();
    // (round 6, measured and NOT kept: dQ on a second stream beside dD, as msim_dense_t_bwd does with its two GEMM kernels -- the
    // fork / join edges cost more than the 5-20 us of overlap they buy here: pairwise loss 0.102 -> 0.123 ms, ColbertLoss 0.188 -> 0.209)
    if (a.n_q > 0 && a.Lq > 0) {
        // tokens per wave: the pair-range lookup is per wave, so few waves per query once there are more tokens than the chip has waves
        const long long tokens = (long long)a.n_q * a.Lq;
        int tpw = (int)((tokens + 4095) / 4096);
        tpw = tpw < 1 ? 1 : (tpw > 16 ? 16 : tpw);
        // few tokens with long pair lists (a dense gradient: 1024 tokens x 256 pairs at config 5's shape): the four waves of a workgroup
        // share the tokens and split the pairs (maxsim_bwd.hip: psplit)
        const int psplit = (tokens <= 2048 && (long long)a.n_pairs >= 64LL * a.n_q) ? 1 : 0;
        if (psplit) tpw = 1;
        const int chunks = psplit ? a.Lq : (a.Lq + 4 * tpw - 1) / (4 * tpw);
        hipLaunchKernelGGL((msim::maxsim_bwd_dq_kernel<DT, OUT16>), dim3((unsigned)a.n_q * chunks), dim3(256), 0, st, D, d_off, pairs, g,
                           argmax, dQ, a, row_bytes, tpw, gs, psplit);
    }
    const int ry = (max_doc_rows + msim::kBwdRows - 1) / msim::kBwdRows;
    const int zc = (dim + 127) / 128;
    if (a.n_d <= 0 || ry <= 0) return;
    const int splits = pl.splits;
    if (pl.mode == 2 && partial) {
        const int lds = 2 * max_doc_rows * 128 * (int)sizeof(float);  // <= 64 KiB
        hipLaunchKernelGGL(msim::maxsim_bwd_dd_pairs_kernel<DT>, dim3(a.n_pairs, splits, zc), dim3(256), lds, st, Q, d_off, pairs, order_by_doc,
                           g, argmax, partial, a, dim, max_doc_rows, splits, gs);
        const int per_wg = 256 * (OUT16 ? 2 : 1);                     // one step per thread
        hipLaunchKernelGGL((msim::maxsim_bwd_dd_pairsum_kernel<DT, OUT16>), dim3(a.n_d, (max_doc_rows * dim + per_wg - 1) / per_wg), dim3(256), 0,
                           st, partial, d_off, pairs, order_by_doc, dD, a.n_pairs, dim, max_doc_rows, splits);
        return;
    }
    if (pl.mode == 1 && partial) {
        const int lds = 2 * max_doc_rows * 128 * (int)sizeof(float);  // <= 64 KiB
        hipLaunchKernelGGL(msim::maxsim_bwd_dd_dense_kernel<DT>, dim3(a.n_d, splits, zc), dim3(256), lds, st, Q, d_off, pairs, order_by_doc,
                           g, argmax, partial, a, dim, max_doc_rows, splits, gs);
        const int per_thread = OUT16 ? 2 : 1;
        hipLaunchKernelGGL((msim::maxsim_bwd_dd_sum_kernel<DT, OUT16>), dim3(a.n_d, (max_doc_rows * dim + 256 * per_thread - 1) / (256 * per_thread)),
                           dim3(256), 0, st, partial, d_off, pairs, order_by_doc, dD, a.n_d, a.n_pairs, dim, max_doc_rows, splits);
        return;
    }
    // round 6: documents whose entry lists fit the LDS lists of the row-list kernel (a bound the host can know: a document meets every
    // query at most twice in the lists the losses make) are bucketed by row once instead of re-scanned per 64-row range
    static const bool rows_off = ab_env("MSIM_DD_ROWS", 1) == 0;          // tuning knob (A/B), not part of the ABI
    const long long pairs_per_doc = std::min<long long>(a.n_pairs, 2LL * a.n_q);
    // (a list with a handful of entries per document -- the pairwise loss: 2B pairs over C documents -- stays with the kernel below,
    // whose eight small workgroups per CU zero-fill the untouched documents faster: 13.5 against 18 us at config 5's shape)
    const bool dense_enough = (long long)a.n_pairs * a.Lq >= 64LL * a.n_d;
    if (!rows_off && dense_enough && pairs_per_doc <= msim::kRowsMaxPairs && pairs_per_doc * a.Lq <= msim::kRowsMaxEnt) {
        int sy = (2 * cus + a.n_d - 1) / a.n_d;                            // about two 512-thread workgroups per CU
        const int by_rows = (max_doc_rows + 63) / 64, need = (max_doc_rows + msim::kRowsMaxRows - 1) / msim::kRowsMaxRows;
        sy = sy > by_rows ? by_rows : sy;
        sy = sy < need ? need : (sy < 1 ? 1 : sy);
        hipLaunchKernelGGL((msim::maxsim_bwd_dd_rows_kernel<DT, OUT16>), dim3(a.n_d, sy, zc), dim3(msim::kRowsThreads), 0, st, Q, d_off, pairs,
                           order_by_doc, g, argmax, dD, a, dim, gs);
        return;
    }
    // row ranges per workgroup: about eight workgroups per CU in total (each looks its document's pair range up once)
    int gy = (8 * cus + a.n_d - 1) / a.n_d;
    gy = gy < 1 ? 1 : (gy > ry ? ry : gy);
    hipLaunchKernelGGL((msim::maxsim_bwd_dd_kernel<DT, OUT16>), dim3(a.n_d, gy, zc), dim3(256), 0, st, Q, d_off, pairs, order_by_doc, g,
                       argmax, dD, a, dim, gs);
}

// ---------------------------------------------------------------- generic kernels (K1g)
struct GenericCall {
    const char *Q, *D;
    const int32_t *d_off;
    const uint8_t *clamp0;
    float *scores;
    long long ld;
    int n_q, Lq, n_d, row_bytes;
    unsigned flags;
    const DeviceInfo *di;
    hipStream_t st;
};

template <int DT, int T>
int launch_generic(const GenericCall &c) {
    auto kern = msim::maxsim_generic_kernel<DT, T>;
    const int lds = T * msim::kTokTile * (c.row_bytes + 16);
    static std::atomic<int> configured[kMaxDevices];
    if (int rc = allow_lds(kern, 160 * 1024, configured)) return rc;
    msim::GenericArgs a;
    a.ld = c.ld;
    a.n_q = c.n_q;
    a.Lq = c.Lq;
    a.n_d = c.n_d;
    a.row_bytes = c.row_bytes;
    a.flags = c.flags;
    const int tpq = (c.Lq + msim::kTokTile - 1) / msim::kTokTile;
    const int groups = tpq <= T ? (c.n_q + (T / tpq) - 1) / (T / tpq) : c.n_q;
    if (groups > 65535) return fail(MSIM_EUNSUPPORTED, "too many query groups (%d) for one launch", groups);
    const int wg_needed = (c.n_d + msim::kGenericWaves - 1) / msim::kGenericWaves;
    int per_cu = c.di->lds_per_cu / lds;
    per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
    const int wg_cap = c.di->cus * per_cu;
    hipLaunchKernelGGL(kern, dim3(wg_needed < wg_cap ? wg_needed : wg_cap, groups), dim3(msim::kGenericWaves * 64), lds, c.st,
                       c.Q, c.D, c.d_off, c.clamp0, c.scores, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MSIM_ELAUNCH, "maxsim_generic_kernel<%d,%d> launch: %s", DT, T, hipGetErrorString(e));
    return MSIM_OK;
}

template <int DT>
int generic_dispatch(const GenericCall &c) {
    const int tpq = (c.Lq + msim::kTokTile - 1) / msim::kTokTile;
    const long long tiles = (long long)c.n_q * tpq;
    const int tile_lds = msim::kTokTile * (c.row_bytes + 16);
    int T = 4;                                   // resident token tiles: as many as fit 96 KiB of LDS
    while (T > 1 && (T * tile_lds > 96 * 1024 || T / 2 >= tiles)) T >>= 1;
    switch (T) {
        case 4: return launch_generic<DT, 4>(c);
        case 2: return launch_generic<DT, 2>(c);
        default: return launch_generic<DT, 1>(c);
    }
}

int generic_fwd(int dtype, const GenericCall &c) {
    switch (dtype) {
        case MSIM_DTYPE_F32: return generic_dispatch<msim::kDtypeF32>(c);
        case MSIM_DTYPE_F16: return generic_dispatch<msim::kDtypeF16>(c);
        default: return generic_dispatch<msim::kDtypeBf16>(c);
    }
}

template <int DT>
int generic_pairs_argmax(const char *Q, const char *D, const int32_t *d_off, const uint8_t *clamp0, const int32_t *pairs,
                         float *out_scores, int32_t *out_argmax, const msim::PairsArgs &a, int row_bytes, const DeviceInfo &di,
                         hipStream_t st) {
    const int wg_needed = (a.n_pairs + 3) / 4;
    const int wg_cap = di.cus * 8;
    hipLaunchKernelGGL(msim::maxsim_generic_pairs_argmax_kernel<DT>, dim3(wg_needed < wg_cap ? wg_needed : wg_cap), dim3(256), 0,
                       st, Q, D, d_off, clamp0, pairs, out_scores, out_argmax, a, row_bytes);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MSIM_ELAUNCH, "maxsim_generic_pairs_argmax_kernel launch: %s", hipGetErrorString(e));
    return MSIM_OK;
}


// ---------------------------------------------------------------- smooth-max (tau * logsumexp) kernels
int check_smooth(const void *Q, const void *D, const int32_t *d_off, int dtype, int dim, int Lq, float tau) {
    if (!Q || !D || !d_off) return fail(MSIM_EINVAL, "null pointer argument");
    if (dtype != MSIM_DTYPE_BF16 && dtype != MSIM_DTYPE_F16 && dtype != MSIM_DTYPE_F32)
        return fail(MSIM_EUNSUPPORTED, "dtype code %d", dtype);
    if ((reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(D)) & 15)
        return fail(MSIM_EINVAL, "Q and D must be 16-byte aligned");
    if (!(tau > 0.0f)) return fail(MSIM_EINVAL, "tau must be positive");
    if (dim <= 0 || Lq <= 0) return fail(MSIM_EINVAL, "bad size (dim=%d Lq=%d)", dim, Lq);
    const long long row_bytes = (long long)dim * elem_bytes(dtype);
    if (row_bytes % 32 != 0)
        return fail(MSIM_EUNSUPPORTED, "dim=%d: an embedding row must be a multiple of 32 bytes (pad the width with zero columns)", dim);
    if (row_bytes > msim::kGenericMaxRowBytes)
        return fail(MSIM_EUNSUPPORTED, "dim=%d: embedding rows above %d bytes are not supported", dim, msim::kGenericMaxRowBytes);
    return MSIM_OK;
}

template <int DT, int T>
int launch_smooth(const GenericCall &c, float tau) {
    auto kern = msim::maxsim_smooth_kernel<DT, T>;
    const int lds = T * msim::kTokTile * (c.row_bytes + 16);
    static std::atomic<int> configured[kMaxDevices];
    if (int rc = allow_lds(kern, 160 * 1024, configured)) return rc;
    msim::SmoothArgs a;
    a.ld = c.ld;
    a.n_q = c.n_q;
    a.Lq = c.Lq;
    a.n_d = c.n_d;
    a.row_bytes = c.row_bytes;
    a.tau = tau;
    const int tpq = (c.Lq + msim::kTokTile - 1) / msim::kTokTile;
    const int groups = tpq <= T ? (c.n_q + (T / tpq) - 1) / (T / tpq) : c.n_q;
    if (groups > 65535) return fail(MSIM_EUNSUPPORTED, "too many query groups (%d) for one launch", groups);
    const int wg_needed = (c.n_d + msim::kGenericWaves - 1) / msim::kGenericWaves;
    int per_cu = c.di->lds_per_cu / lds;
    per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
    const int wg_cap = c.di->cus * per_cu;
    hipLaunchKernelGGL(kern, dim3(wg_needed < wg_cap ? wg_needed : wg_cap, groups), dim3(msim::kGenericWaves * 64), lds, c.st,
                       c.Q, c.D, c.d_off, c.scores, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MSIM_ELAUNCH, "maxsim_smooth_kernel<%d,%d> launch: %s", DT, T, hipGetErrorString(e));
    return MSIM_OK;
}

template <int DT>
int smooth_dispatch(const GenericCall &c, float tau) {
    const int tpq = (c.Lq + msim::kTokTile - 1) / msim::kTokTile;
    const long long tiles = (long long)c.n_q * tpq;
    const int tile_lds = msim::kTokTile * (c.row_bytes + 16);
    int T = 2;                                   // (max, sum) state per tile on top of the accumulators: two tiles per wave
    while (T > 1 && (T * tile_lds > 80 * 1024 || T / 2 >= tiles)) T >>= 1;
    return T == 2 ? launch_smooth<DT, 2>(c, tau) : launch_smooth<DT, 1>(c, tau);
}

template <int DT>
int smooth_pairs(const char *Q, const char *D, const int32_t *d_off, const int32_t *pairs, float *out_scores, float *out_lse,
                 const msim::PairsArgs &a, int row_bytes, float tau, const DeviceInfo &di, hipStream_t st) {
    const int wg_needed = (a.n_pairs + 3) / 4;
    const int wg_cap = di.cus * 8;
    hipLaunchKernelGGL(msim::maxsim_smooth_pairs_kernel<DT>, dim3(wg_needed < wg_cap ? wg_needed : wg_cap), dim3(256), 0, st, Q, D,
                       d_off, pairs, out_scores, out_lse, a, row_bytes, tau);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MSIM_ELAUNCH, "maxsim_smooth_pairs_kernel launch: %s", hipGetErrorString(e));
    return MSIM_OK;
}

template <int DT>
int smooth_bwd(const char *Q, const char *D, const int32_t *d_off, int max_doc_rows, const int32_t *pairs,
               const int32_t *order_by_doc, const float *g, const float *lse, float *dQ, float *dD, float *workspace,
               msim::SmoothBwdArgs a, int n_split, hipStream_t st) {
    const int tpq = (a.Lq + msim::kTokTile - 1) / msim::kTokTile;
    const int cg = (a.dim + 32 * msim::kSmoothCB - 1) / (32 * msim::kSmoothCB);
    const bool hoist = a.row_bytes <= 256;                       // the owner tile's fragments fit 8 registers quads
    static const bool no_stage = ab_env("MSIM_SMOOTH_NO_STAGE", 0) != 0;   // A/B knob (measurement builds)
    const int slabs = (max_doc_rows + 31) / 32;
    if constexpr (DT != msim::kDtypeF32) {
        if (a.row_bytes == msim::kRowBytes && a.dim == msim::kDim && !no_stage) {   // 128 x 16-bit rows: staged "other" tiles
            constexpr bool F16 = DT == msim::kDtypeF16;
            if (a.n_q > 0) {
                a.n_split = n_split;
                auto kern = msim::maxsim_smooth_bwd_staged_kernel<F16, true>;
                constexpr int lds = msim::kSmoothWavesDQ * msim::kSmoothStageBytes;
                static std::atomic<int> configured[kMaxDevices];
                if (int rc = allow_lds(kern, lds, configured)) return rc;
                hipLaunchKernelGGL(kern, dim3(a.n_q * n_split, tpq, 1), dim3(msim::kSmoothWavesDQ * 64), lds, st, Q, D, d_off, pairs,
                                   order_by_doc, g, lse, n_split > 1 ? workspace : dQ, a);
                if (n_split > 1) {
                    const long long n = (long long)a.n_q * a.Lq * a.dim;
                    hipLaunchKernelGGL(msim::smooth_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, workspace, dQ, n,
                                       n_split);
                }
            }
            if (a.n_d > 0 && slabs > 0) {
                a.n_split = 1;
                auto kern = msim::maxsim_smooth_bwd_staged_kernel<F16, false>;
                constexpr int lds = msim::kSmoothWavesDD * msim::kSmoothStageBytes;
                static std::atomic<int> configured[kMaxDevices];
                if (int rc = allow_lds(kern, lds, configured)) return rc;
                hipLaunchKernelGGL(kern, dim3(a.n_d, slabs, 1), dim3(msim::kSmoothWavesDD * 64), lds, st, Q, D, d_off, pairs, order_by_doc, g,
                                   lse, dD, a);
            }
            hipError_t es = hipGetLastError();
            if (es != hipSuccess) return fail(MSIM_ELAUNCH, "maxsim_smooth_bwd_staged_kernel launch: %s", hipGetErrorString(es));
            return MSIM_OK;
        }
    }
    if (a.n_q > 0) {
        a.n_split = n_split;
        const dim3 grid(a.n_q * n_split, tpq, cg), block(msim::kSmoothWavesDQ * 64);
        float *dst = n_split > 1 ? workspace : dQ;
        if (hoist)
            hipLaunchKernelGGL((msim::maxsim_smooth_bwd_kernel<DT, true, true>), grid, block, 0, st, Q, D, d_off, pairs, order_by_doc, g, lse, dst, a);
        else
            hipLaunchKernelGGL((msim::maxsim_smooth_bwd_kernel<DT, true, false>), grid, block, 0, st, Q, D, d_off, pairs, order_by_doc, g, lse, dst, a);
        if (n_split > 1) {
            const long long n = (long long)a.n_q * a.Lq * a.dim;
            hipLaunchKernelGGL(msim::smooth_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, workspace, dQ, n, n_split);
        }
    }
    if (a.n_d > 0 && slabs > 0) {
        a.n_split = 1;
        const dim3 grid(a.n_d, slabs, cg), block(msim::kSmoothWavesDD * 64);
        if (hoist)
            hipLaunchKernelGGL((msim::maxsim_smooth_bwd_kernel<DT, false, true>), grid, block, 0, st, Q, D, d_off, pairs, order_by_doc, g, lse, dD, a);
        else
            hipLaunchKernelGGL((msim::maxsim_smooth_bwd_kernel<DT, false, false>), grid, block, 0, st, Q, D, d_off, pairs, order_by_doc, g, lse, dD, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MSIM_ELAUNCH, "maxsim_smooth_bwd_kernel launch: %s", hipGetErrorString(e));
    return MSIM_OK;
}

// number of workgroups that share one owner tile's pair list in the dQ pass (enough to fill the chip twice)
int smooth_dq_split(int n_q, int Lq, const DeviceInfo &di) {
    const int tiles = n_q * ((Lq + msim::kTokTile - 1) / msim::kTokTile);
    int s = (2 * di.cus + tiles - 1) / (tiles > 0 ? tiles : 1);
    return s < 1 ? 1 : (s > 32 ? 32 : s);
}

// ---------------------------------------------------------------- plain similarity matrix
template <int DT, int T>
int launch_sim(const char *A, const char *B, float *out, const msim::SimArgs &a, const DeviceInfo &di, hipStream_t st) {
    auto kern = msim::sim_matrix_kernel<DT, T>;
    const int lds = T * msim::kTokTile * (a.row_bytes + 16);
    static std::atomic<int> configured[kMaxDevices];
    if (int rc = allow_lds(kern, 160 * 1024, configured)) return rc;
    const int groups = (a.n_a + T * msim::kTokTile - 1) / (T * msim::kTokTile);
    if (groups > 65535) return fail(MSIM_EUNSUPPORTED, "too many row groups (%d) for one launch", groups);
    const int slabs = (a.n_b + msim::kSlabRows - 1) / msim::kSlabRows;
    const int wg_needed = (slabs + msim::kGenericWaves - 1) / msim::kGenericWaves;
    int per_cu = di.lds_per_cu / lds;
    per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
    const int wg_cap = di.cus * per_cu;
    hipLaunchKernelGGL(kern, dim3(wg_needed < wg_cap ? wg_needed : wg_cap, groups), dim3(msim::kGenericWaves * 64), lds, st, A, B,
                       out, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MSIM_ELAUNCH, "sim_matrix_kernel<%d,%d> launch: %s", DT, T, hipGetErrorString(e));
    return MSIM_OK;
}

template <int DT>
int sim_dispatch(const char *A, const char *B, float *out, const msim::SimArgs &a, const DeviceInfo &di, hipStream_t st) {
    const int tile_lds = msim::kTokTile * (a.row_bytes + 16);
    int T = 4;
    while (T > 1 && (T * tile_lds > 80 * 1024 || (T / 2) * msim::kTokTile >= a.n_a)) T >>= 1;
    switch (T) {
        case 4: return launch_sim<DT, 4>(A, B, out, a, di, st);
        case 2: return launch_sim<DT, 2>(A, B, out, a, di, st);
        default: return launch_sim<DT, 1>(A, B, out, a, di, st);
    }
}

// ---------------------------------------------------------------- panel kernels (K1sP / K1bP): 16-bit, dim 320
constexpr int kPanels320 = 3, kLast320 = 4;     // 320 = (2 * 8 + 4) * 16

bool is_panels(int dtype, int dim, long long tiles, int tpq) {
    if (!(dtype == MSIM_DTYPE_BF16 || dtype == MSIM_DTYPE_F16) || dim != 320 || tpq > 4) return false;
    return tiles <= 4 || tpq <= 2;               // K1sP holds <= 4 token tiles, K1bP whole queries of <= 2 tiles per wave
}

template <int QT, int TPQ, bool F16>
int launch_stream_panels(const FwdCall &c) {
    auto kern = msim::maxsim_stream_panels_kernel<QT, TPQ, kPanels320, kLast320, F16, 2>;
    constexpr int lds = 4 * msim::kPanelRing * msim::kSlabBytes;
    static std::atomic<int> configured[kMaxDevices];
    if (int rc = allow_lds(kern, lds, configured)) return rc;
    msim::PanelStreamArgs a;
    a.ld = c.ld;
    a.n_q = c.n_q;
    a.Lq = c.Lq;
    a.n_d = c.n_d;
    a.flags = c.flags;
    const int wg_needed = (c.n_d + 3) / 4;
    const int wg_cap = c.di->cus * (c.di->lds_per_cu / lds);
    hipLaunchKernelGGL(kern, dim3(wg_needed < wg_cap ? wg_needed : wg_cap), dim3(256), lds, c.st, c.Q, c.D, c.d_off, c.clamp0,
                       c.scores, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MSIM_ELAUNCH, "maxsim_stream_panels_kernel<%d,%d> launch: %s", QT, TPQ, hipGetErrorString(e));
    return MSIM_OK;
}

template <int NT, int TPQ, bool F16>
int launch_batch_panels(const FwdCall &c) {
    auto kern = msim::maxsim_batch_panels_kernel<NT, TPQ, kPanels320, kLast320, F16>;
    constexpr int lds = msim::kPanelStages * kPanels320 * msim::kSlabBytes;
    static std::atomic<int> configured[kMaxDevices];
    if (int rc = allow_lds(kern, lds, configured)) return rc;
    msim::PanelBatchArgs a{};
    a.ld = c.ld;
    a.n_q = c.n_q;
    a.Lq = c.Lq;
    a.n_d = c.n_d;
    a.flags = c.flags;
    const int q_per_block = msim::kBatchWaves * NT / TPQ;
    a.n_qblocks = (c.n_q + q_per_block - 1) / q_per_block;
    const int cus_per_xcd = c.di->cus / 8 > 0 ? c.di->cus / 8 : 1;
    const int sub = a.n_qblocks >= cus_per_xcd ? 1 : cus_per_xcd / a.n_qblocks;
    a.n_ranges = 8 * sub;
    const int slots = sub > 1 ? a.n_qblocks * sub : a.n_qblocks;
    hipLaunchKernelGGL(kern, dim3(8 * slots), dim3(512), lds, c.st, c.Q, c.D, c.d_off, c.clamp0, c.scores, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MSIM_ELAUNCH, "maxsim_batch_panels_kernel<%d,%d> launch: %s", NT, TPQ, hipGetErrorString(e));
    return MSIM_OK;
}

template <bool F16>
int panels_dispatch(const FwdCall &c) {
    const int tpq = (c.Lq + msim::kTokTile - 1) / msim::kTokTile;
    if ((long long)c.n_q * tpq <= 4) {
        switch (c.n_q * 10 + tpq) {
            case 11: return launch_stream_panels<1, 1, F16>(c);
            case 21: return launch_stream_panels<2, 1, F16>(c);
            case 31: return launch_stream_panels<3, 1, F16>(c);
            case 41: return launch_stream_panels<4, 1, F16>(c);
            case 12: return launch_stream_panels<2, 2, F16>(c);
            case 22: return launch_stream_panels<4, 2, F16>(c);
            case 13: return launch_stream_panels<3, 3, F16>(c);
            default: return launch_stream_panels<4, 4, F16>(c);
        }
    }
    if (tpq == 2) return launch_batch_panels<2, 2, F16>(c);
    return c.n_q <= 8 ? launch_batch_panels<1, 1, F16>(c) : launch_batch_panels<2, 1, F16>(c);
}


// ---- K1bPF: the flat token layout at width 320 (maxsim_panels.hip).  8 waves x <= 4 units (round 6: or ONE block of 2 / 4 waves for small batches), query blocks of whole queries
// (<= 512 tokens, <= 64 queries), filled greedily in query order and re-cut evenly like K1b's.
constexpr int kPanelsFlatMaxU = 4;
constexpr int kPanelsFlatMaxTokens = msim::kBatchWaves * kPanelsFlatMaxU * msim::kUnitTok;      // 512

int panels_flat_plan(const HostQ &hq, int n_q, FlatPlan &p) {
    if ((long long)hq.at(n_q) - hq.at(0) < 0) return fail(MSIM_EINVAL, "query token offsets are not non-decreasing");
    p = FlatPlan{};
    // the ladder (round 6): a batch that fits ONE block of two waves (<= 8 units, <= 16 queries) or four waves (<= 16 units, <= 32
    // queries) takes that shape -- up to four units per wave behind a narrower barrier, two workgroups per CU; everything else the
    // 8-wave shape (one or several blocks)
    const long long tokens = (long long)hq.at(n_q) - hq.at(0);
    const long long units = (tokens + msim::kUnitTok - 1) / msim::kUnitTok;
    p.nw = (units <= 2 * kPanelsFlatMaxU && n_q <= 16) ? 2 : (units <= 4 * kPanelsFlatMaxU && n_q <= 32) ? 4 : msim::kBatchWaves;
    p.maxu = kPanelsFlatMaxU;
    if (!fill_blocks(hq, n_q, p.nw, p.maxu, p.blk_q0) || (p.nw != msim::kBatchWaves && p.n_blocks() != 1)) {
        p.nw = msim::kBatchWaves;                     // (unit padding at query borders cannot overflow: units counts whole tokens)
        p.blk_q0.clear();
    }
    if (p.blk_q0.empty() && !fill_blocks(hq, n_q, p.nw, p.maxu, p.blk_q0))
        return fail(MSIM_EUNSUPPORTED, "a query of more than %d tokens does not fit one query block of the width-320 kernels: pad the "
                    "queries to one length and call msim_fwd", kPanelsFlatMaxTokens);
    balance_blocks(hq, n_q, p.nw, p.maxu, p.blk_q0);
    return MSIM_OK;
}

template <bool F16, int NW, int STAGES>
int launch_batch_panels_flat_nw(const FwdCall &c, const FlatPlan &plan) {
    auto kern = msim::maxsim_batch_panels_flat_kernel<F16, kPanels320, kLast320, kPanelsFlatMaxU, NW, STAGES>;
    constexpr int lds = STAGES * kPanels320 * msim::kSlabBytes + NW * kPanelsFlatMaxU * msim::kUnitTok * 16 +
                        NW * 8 * 8;                      // stage ring + the per-token max table + the queries' token ranges
    static std::atomic<int> configured[kMaxDevices];
    if (int rc = allow_lds(kern, lds, configured)) return rc;
    const int cus_per_xcd = (c.di->cus / 8 > 0 ? c.di->cus / 8 : 1) * (NW == msim::kBatchWaves ? 1 : 2);   // workgroups resident per XCD
    const int total_blocks = plan.n_blocks();
    for (int b0 = 0; b0 < total_blocks; b0 += msim::kMaxQBlocks) {
        msim::BatchArgs a{};
        a.ld = c.ld;
        a.fq = flat_q(c);
        a.n_d = c.n_d;
        a.flags = c.flags;
        a.n_qblocks = total_blocks - b0 < msim::kMaxQBlocks ? total_blocks - b0 : msim::kMaxQBlocks;
        for (int b = 0; b <= a.n_qblocks; ++b) a.blk_q0[b] = plan.blk_q0[b0 + b];
        const int sub = ranges_per_xcd(a.n_qblocks, cus_per_xcd, c.n_d, c.avg_rows);
        a.n_ranges = 8 * sub;
        const int slots = sub > 1 ? a.n_qblocks * sub : a.n_qblocks;
        a.convoy = nullptr;
        a.trace = nullptr;
        hipLaunchKernelGGL(kern, dim3(8 * slots), dim3(NW * 64), lds, c.st, c.Q, c.D, c.d_off, c.clamp0, c.scores, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(MSIM_ELAUNCH, "maxsim_batch_panels_flat_kernel<%d> launch: %s", NW, hipGetErrorString(e));
    }
    return MSIM_OK;
}

template <bool F16>
int launch_batch_panels_flat(const FwdCall &c) {
    thread_local FlatPlan plan;
    if (int rc = panels_flat_plan(host_q(c), c.n_q, plan)) return rc;
    if (plan.nw == 2) return launch_batch_panels_flat_nw<F16, 2, 3>(c, plan);
    if (plan.nw == 4) return launch_batch_panels_flat_nw<F16, 4, 3>(c, plan);
    return launch_batch_panels_flat_nw<F16, msim::kBatchWaves, msim::kPanelStages>(c, plan);
}

// a uniform box [n_q, Lq, 320] on the flat kernel?  K1sP keeps the calls of <= 4 token tiles (HBM-bound, no barriers); K1bP keeps whole
// queries of 32 or 64 tokens (no padding to win back); everything else up to 512 tokens is scored as 16-token units of the token matrix,
// which cross query borders (1000 x Lq 40: 2500 units of 16 instead of 2000 tiles of 32) -- and three and more tiles per query, which
// K1bP does not take, are just more units.  MSIM_PANELS_FLAT=0|1 forces the choice (A/B knob of the measurement builds, not ABI).
bool box_on_panels_flat(int dtype, int dim, int n_q, int Lq) {
    if (!(dtype == MSIM_DTYPE_BF16 || dtype == MSIM_DTYPE_F16) || dim != 320 || Lq > kPanelsFlatMaxTokens) return false;
    const int tpq = (Lq + msim::kTokTile - 1) / msim::kTokTile;
    if ((long long)n_q * tpq <= 4) return false;
    static const int forced = ab_env("MSIM_PANELS_FLAT", -1);
    if (forced == 0) return tpq > 2;
    if (forced == 1) return true;
    return tpq > 2 || (Lq % msim::kTokTile) != 0;
}

}  // namespace

extern "C" {

int msim_abi_version(void) { return MSIM_ABI_VERSION; }
#ifdef MSIM_AB
int msim_ab_rows_trace(unsigned long long *out16) {      // measurement builds only
    return hipMemcpyFromSymbol(out16, HIP_SYMBOL(msim::g_rows_trace), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : -3;
}
#endif


const char *msim_last_error(void) { return g_err; }

size_t msim_fwd_workspace_bytes(int dtype, int n_q, int Lq, int n_d, int dim) {
    // the only scratch msim_fwd uses: the progress counters of K1b's convoy, needed once more than one query block streams a
    // document range (bf16 / fp16, width 128).  Passing NULL instead only switches the convoy off; a non-null workspace must hold
    // at least the bytes reported here (4096 whenever it is non-zero).
    if (n_q <= 0 || n_d <= 0 || Lq <= 0) return 0;
    if (is_long_tuned(dtype, dim, Lq))                                              // pieces on K1b: counters + the partial sums
        return kFwdWorkspaceBytes + (size_t)n_q * long_segments(Lq) * n_d * sizeof(float);
    if (!is_tuned(dtype, dim, Lq)) return 0;
    return flat_workspace_bytes(HostQ{nullptr, Lq, 0, 1}, n_q);                     // exactly launch_batch's condition
}

int msim_fwd_plan(const int32_t *q_off_host, int n_q, int Lq, int32_t *out5) {
    if (!out5 || n_q <= 0 || (!q_off_host && Lq <= 0)) return fail(MSIM_EINVAL, "bad arguments");
    thread_local FlatPlan plan;
    const HostQ hq = q_off_host ? HostQ{q_off_host, 0, 0, 1}
                                : (Lq > kLongSegRows ? HostQ{nullptr, Lq, kLongSegRows, long_segments(Lq)} : HostQ{nullptr, Lq, 0, 1});
    if (!q_off_host && Lq > kLongSegRows && (long long)n_q * long_segments(Lq) > 0x7fffffff / 8)
        return fail(MSIM_EUNSUPPORTED, "too many 128-token pieces (%d queries x %d)", n_q, long_segments(Lq));
    const int n = (!q_off_host && Lq > kLongSegRows) ? n_q * long_segments(Lq) : n_q;
    if (int rc = flat_plan(hq, n, plan)) return rc;
    out5[0] = plan.stream ? 0 : 1;
    out5[1] = plan.stream ? plan.nu : plan.nw;
    out5[2] = plan.stream ? 0 : plan.maxu;
    out5[3] = plan.stream ? 1 : plan.n_blocks();
    out5[4] = plan.stream ? plan.nu : heaviest_wave_units(hq, plan.blk_q0, plan.nw);
    return MSIM_OK;
}

size_t msim_fwd_ragged_workspace_bytes(int dtype, const int32_t *q_off_host, int n_q, int n_d, int dim) {
    if (n_q <= 0 || n_d <= 0 || !q_off_host) return 0;
    if (!(dtype == MSIM_DTYPE_BF16 || dtype == MSIM_DTYPE_F16) || dim != msim::kDim) return 0;
    return flat_workspace_bytes(HostQ{q_off_host, 0, 0, 1}, n_q);
}

int msim_fwd_ragged(int dtype, const void *Qt, const int32_t *q_off, const int32_t *q_off_host, int n_q, const void *D,
                    const int32_t *d_off, const uint8_t *d_clamp0, int n_d, int dim, float *scores, int64_t ld_scores,
                    uint32_t flags, void *workspace, void *stream) {
    if (n_q < 0 || n_d < 0) return fail(MSIM_EINVAL, "negative size (n_q=%d n_d=%d)", n_q, n_d);
    if (n_q == 0 || n_d == 0) return MSIM_OK;
    if (!scores || !Qt || !D || !d_off || !q_off || !q_off_host) return fail(MSIM_EINVAL, "null pointer argument");
    if (!(dtype == MSIM_DTYPE_BF16 || dtype == MSIM_DTYPE_F16) || !(dim == msim::kDim || dim == 320))
        return fail(MSIM_EUNSUPPORTED, "msim_fwd_ragged takes bfloat16 / float16 embeddings of width %d or 320 (dtype code %d, dim %d): "
                    "pad the queries to one length and call msim_fwd", msim::kDim, dtype, dim);
    if ((reinterpret_cast<uintptr_t>(Qt) | reinterpret_cast<uintptr_t>(D)) & 15) return fail(MSIM_EINVAL, "Qt and D must be 16-byte aligned");
    if (workspace && (reinterpret_cast<uintptr_t>(workspace) & 15)) return fail(MSIM_EINVAL, "workspace must be 16-byte aligned");
    if (ld_scores < n_d) return fail(MSIM_EINVAL, "ld_scores=%lld < n_d=%d", (long long)ld_scores, n_d);
    const int avg_rows = (int)((flags >> 8) & 0xffffu);          // MSIM_FLAG_AVG_ROWS(n): a launch-shape hint, never a result
    flags &= ~(0xffffu << 8);
    if (flags & ~(MSIM_FLAG_REF_ROUNDING)) return fail(MSIM_EINVAL, "unknown flags 0x%x", flags);
    if (q_off_host[0] != 0) return fail(MSIM_EINVAL, "q_off[0] must be 0");
    for (int i = 0; i < n_q; ++i)
        if (q_off_host[i + 1] < q_off_host[i]) return fail(MSIM_EINVAL, "q_off must be non-decreasing (query %d)", i);
    FwdCall c;
    if (int rc = device_info(&c.di)) return rc;
    c.Q = static_cast<const uint16_t *>(Qt);
    c.D = static_cast<const uint16_t *>(D);
    c.d_off = d_off;
    c.clamp0 = d_clamp0;
    c.scores = scores;
    c.ld = ld_scores;
    c.n_q = n_q;
    c.Lq = 0;
    c.n_d = n_d;
    c.flags = flags;
    c.st = static_cast<hipStream_t>(stream);
    c.workspace = workspace;
    c.q_off = q_off;
    c.q_off_host = q_off_host;
    c.avg_rows = avg_rows;
    if (dim == 320) {
        // width 320 (ColQwen3): queries of ONE length and at most four 32-token tiles in all are a box K1sP streams without a barrier;
        // everything else goes to the flat panel kernel
        const int L0 = q_off_host[1];
        bool uniform = true;
        for (int i = 1; i < n_q && uniform; ++i) uniform = q_off_host[i + 1] - q_off_host[i] == L0;
        const int tpq = (L0 + msim::kTokTile - 1) / msim::kTokTile;
        if (uniform && L0 > 0 && is_panels(dtype, dim, (long long)n_q * tpq, tpq) && (long long)n_q * tpq <= 4) {
            c.Lq = L0;
            c.q_off = nullptr;
            c.q_off_host = nullptr;
            return dtype == MSIM_DTYPE_F16 ? panels_dispatch<true>(c) : panels_dispatch<false>(c);
        }
        return dtype == MSIM_DTYPE_F16 ? launch_batch_panels_flat<true>(c) : launch_batch_panels_flat<false>(c);
    }
    return dtype == MSIM_DTYPE_F16 ? fwd_dispatch<true>(c) : fwd_dispatch<false>(c);
}

int msim_fwd(int dtype, const void *Q, int n_q, int Lq, const void *D, const int32_t *d_off, const uint8_t *d_clamp0,
             int n_d, int dim, float *scores, int64_t ld_scores, uint32_t flags, void *workspace, void *stream) {
    if (n_q < 0 || n_d < 0 || Lq <= 0) return fail(MSIM_EINVAL, "negative size (n_q=%d n_d=%d Lq=%d)", n_q, n_d, Lq);
    if (n_q == 0 || n_d == 0) return MSIM_OK;
    if (!scores) return fail(MSIM_EINVAL, "null pointer argument");
    if (int rc = check_common(Q, D, d_off, dtype, dim, Lq)) return rc;
    if (ld_scores < n_d) return fail(MSIM_EINVAL, "ld_scores=%lld < n_d=%d", (long long)ld_scores, n_d);
    const int avg_rows = (int)((flags >> 8) & 0xffffu);          // MSIM_FLAG_AVG_ROWS(n): a launch-shape hint, never a result
    flags &= ~(0xffffu << 8);
    if (flags & ~(MSIM_FLAG_REF_ROUNDING)) return fail(MSIM_EINVAL, "unknown flags 0x%x", flags);
    {
        const int tpq = (Lq + msim::kTokTile - 1) / msim::kTokTile;
        const bool on_flat = box_on_panels_flat(dtype, dim, n_q, Lq);
        if (on_flat || is_panels(dtype, dim, (long long)n_q * tpq, tpq)) {
            FwdCall c;
            if (int rc = device_info(&c.di)) return rc;
            c.Q = static_cast<const uint16_t *>(Q);
            c.D = static_cast<const uint16_t *>(D);
            c.d_off = d_off;
            c.clamp0 = d_clamp0;
            c.scores = scores;
            c.ld = ld_scores;
            c.n_q = n_q;
            c.Lq = Lq;
            c.n_d = n_d;
            c.flags = flags;
            c.avg_rows = avg_rows;
            c.st = static_cast<hipStream_t>(stream);
            if (on_flat)      // the box IS a flat token matrix with uniform offsets (FlatQ: q_off null, Lq)
                return dtype == MSIM_DTYPE_F16 ? launch_batch_panels_flat<true>(c) : launch_batch_panels_flat<false>(c);
            return dtype == MSIM_DTYPE_F16 ? panels_dispatch<true>(c) : panels_dispatch<false>(c);
        }
    }
    // long queries in the tuned dtype / width (pages as queries, image-to-image retrieval, the trainer's symmetric direction): 128-token
    // PIECES on K1b -- MaxSim is a sum over query tokens, and in the flat token layout a piece is nothing but another pair of token
    // offsets -- partial token sums into the scratch, added in piece order: 3-4 x the generic kernels' rate on a large corpus.
    // The piece rows are reduced 65 535 queries at a time (grid.y of segment_sum_kernel).
    if (is_long_tuned(dtype, dim, Lq) && workspace != nullptr && (long long)n_q * long_segments(Lq) <= 0x7fffffff / 8) {
        if (reinterpret_cast<uintptr_t>(workspace) & 15) return fail(MSIM_EINVAL, "workspace must be 16-byte aligned");
        const int n_seg = long_segments(Lq);
        float *partial = reinterpret_cast<float *>(static_cast<char *>(workspace) + kFwdWorkspaceBytes);
        FwdCall c;
        if (int rc = device_info(&c.di)) return rc;
        c.Q = static_cast<const uint16_t *>(Q);
        c.D = static_cast<const uint16_t *>(D);
        c.d_off = d_off;
        c.clamp0 = d_clamp0;
        c.scores = partial;
        c.ld = n_d;
        c.n_q = n_q * n_seg;
        c.Lq = Lq;
        c.n_d = n_d;
        c.flags = flags | msim::kFlagPartial;
        c.avg_rows = avg_rows;
        c.st = static_cast<hipStream_t>(stream);
        c.workspace = workspace;
        c.seg = kLongSegRows;
        c.n_seg = n_seg;
        const bool f16 = dtype == MSIM_DTYPE_F16;
        if (int rc = f16 ? fwd_dispatch<true>(c) : fwd_dispatch<false>(c)) return rc;
        const bool round_total = (flags & MSIM_FLAG_REF_ROUNDING) != 0;
        for (int q0 = 0; q0 < n_q; q0 += 65535) {
            const int nq = n_q - q0 < 65535 ? n_q - q0 : 65535;
            const dim3 grid((n_d + 255) / 256, nq);
            const float *part = partial + (size_t)q0 * n_seg * n_d;
            float *out = scores + (size_t)q0 * ld_scores;
            if (f16)
                hipLaunchKernelGGL(msim::segment_sum_kernel<true>, grid, dim3(256), 0, c.st, part, (long long)n_d, n_seg, n_d, out,
                                   (long long)ld_scores, round_total);
            else
                hipLaunchKernelGGL(msim::segment_sum_kernel<false>, grid, dim3(256), 0, c.st, part, (long long)n_d, n_seg, n_d, out,
                                   (long long)ld_scores, round_total);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(MSIM_ELAUNCH, "segment_sum_kernel launch: %s", hipGetErrorString(e));
        return MSIM_OK;
    }
    if (!is_tuned(dtype, dim, Lq)) {
        GenericCall c;
        if (int rc = device_info(&c.di)) return rc;
        c.Q = static_cast<const char *>(Q);
        c.D = static_cast<const char *>(D);
        c.d_off = d_off;
        c.clamp0 = d_clamp0;
        c.scores = scores;
        c.ld = ld_scores;
        c.n_q = n_q;
        c.Lq = Lq;
        c.n_d = n_d;
        c.row_bytes = dim * elem_bytes(dtype);
        c.flags = flags;
        c.st = static_cast<hipStream_t>(stream);
        return generic_fwd(dtype, c);
    }
    FwdCall c;
    if (int rc = device_info(&c.di)) return rc;
    c.Q = static_cast<const uint16_t *>(Q);
    c.D = static_cast<const uint16_t *>(D);
    c.d_off = d_off;
    c.clamp0 = d_clamp0;
    c.scores = scores;
    c.ld = ld_scores;
    c.n_q = n_q;
    c.Lq = Lq;
    c.n_d = n_d;
    c.flags = flags;
    c.avg_rows = avg_rows;
    c.st = static_cast<hipStream_t>(stream);
    c.workspace = workspace;
    return dtype == MSIM_DTYPE_F16 ? fwd_dispatch<true>(c) : fwd_dispatch<false>(c);
}

// ---------------------------------------------------------------- K1t: long queries x short documents, all pairs
}  // extern "C"

namespace {
template <bool F16, int U, int DPW, bool ROUTE>
int launch_batch_t(const uint16_t *Q, const uint16_t *D, float *scores, int32_t *q_lengths, uint8_t *route, msim::BatchTArgs a,
                   const DeviceInfo &di, hipStream_t st) {
    auto kern = msim::maxsim_batch_t_kernel<F16, U, DPW, ROUTE>;
    constexpr int lds = 3 * 4 * msim::kSlabBytes;                  // 96 KiB ring
    static std::atomic<int> configured[kMaxDevices];
    if (int rc = allow_lds(kern, lds, configured)) return rc;
    a.n_blocks = (a.n_d + 8 * DPW - 1) / (8 * DPW);
    // page slots per XCD: every page its own slot until the launch holds ~4 workgroups per CU, then the workgroups walk pages
    int slots_p = (a.n_q + 7) / 8;
    const int cap = (4 * di.cus / 8 + a.n_blocks - 1) / a.n_blocks;
    if (slots_p > cap) slots_p = cap < 1 ? 1 : cap;
    a.slots_p = slots_p;
    a.n_slots = slots_p * a.n_blocks;
    hipLaunchKernelGGL(kern, dim3(8 * a.n_slots), dim3(512), lds, st, Q, D, scores, q_lengths, route, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MSIM_ELAUNCH, "maxsim_batch_t_kernel<%d,%d> launch: %s", U, DPW, hipGetErrorString(e));
    return MSIM_OK;
}

template <bool F16>
int batch_t_dispatch(int units, const uint16_t *Q, const uint16_t *D, float *scores, int32_t *ql, uint8_t *route,
                     const msim::BatchTArgs &a, const DeviceInfo &di, hipStream_t st) {
    if (route) {        // with the routing bytes of the dense backward (documents of at most 64 rows)
        if (units <= 1) return launch_batch_t<F16, 1, 8, true>(Q, D, scores, ql, route, a, di, st);
        if (units == 2) return launch_batch_t<F16, 2, 4, true>(Q, D, scores, ql, route, a, di, st);
        if (units == 3) return launch_batch_t<F16, 3, 2, true>(Q, D, scores, ql, route, a, di, st);
        return launch_batch_t<F16, 4, 2, true>(Q, D, scores, ql, route, a, di, st);
    }
    if (units <= 1) return launch_batch_t<F16, 1, 8, false>(Q, D, scores, ql, nullptr, a, di, st);
    if (units == 2) return launch_batch_t<F16, 2, 4, false>(Q, D, scores, ql, nullptr, a, di, st);
    if (units == 3) return launch_batch_t<F16, 3, 2, false>(Q, D, scores, ql, nullptr, a, di, st);
    if (units == 4) return launch_batch_t<F16, 4, 2, false>(Q, D, scores, ql, nullptr, a, di, st);
    return launch_batch_t<F16, 8, 1, false>(Q, D, scores, ql, nullptr, a, di, st);
}

int fwd_transposed(int dtype, const void *Q, int n_q, int Lq, const void *D, int n_d, int Ld, int dim, float *scores,
                   int64_t ld_scores, int32_t *q_lengths, uint8_t *route, void *stream) {
    if (n_q < 0 || n_d < 0 || Lq <= 0 || Ld <= 0) return fail(MSIM_EINVAL, "negative or empty size");
    if (n_q == 0 || n_d == 0) return MSIM_OK;
    if (!Q || !D || !scores) return fail(MSIM_EINVAL, "null pointer argument");
    if ((dtype != MSIM_DTYPE_BF16 && dtype != MSIM_DTYPE_F16) || dim != msim::kDim)
        return fail(MSIM_EUNSUPPORTED, "msim_fwd_transposed takes bf16 / f16 embeddings of width %d", msim::kDim);
    if (Ld > 8 * msim::kUnitTok) return fail(MSIM_EUNSUPPORTED, "resident documents of at most %d rows (got %d)", 8 * msim::kUnitTok, Ld);
    if (route && Ld > msim::kDenseTMaxLd)
        return fail(MSIM_EUNSUPPORTED, "the routing is kept for resident documents of at most %d rows (got %d)", msim::kDenseTMaxLd, Ld);
    if ((long long)Lq * msim::kRowBytes >= (1ll << 31)) return fail(MSIM_EUNSUPPORTED, "queries of %d rows", Lq);
    if (ld_scores < n_d) return fail(MSIM_EINVAL, "ld_scores < n_d");
    if ((reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(D)) & 15) return fail(MSIM_EINVAL, "embeddings must be 16-byte aligned");
    const DeviceInfo *di = nullptr;
    if (int rc = device_info(&di)) return rc;
    msim::BatchTArgs a{};
    a.ld = ld_scores;
    a.n_q = n_q;
    a.Lq = Lq;
    a.n_d = n_d;
    a.Ld = Ld;
    a.Lq_pad = msim::dense_t_lq_pad(Lq);
    const int units = (Ld + msim::kUnitTok - 1) / msim::kUnitTok;
    const uint16_t *q = static_cast<const uint16_t *>(Q), *d = static_cast<const uint16_t *>(D);
    hipStream_t st = static_cast<hipStream_t>(stream);
    return dtype == MSIM_DTYPE_F16 ? batch_t_dispatch<true>(units, q, d, scores, q_lengths, route, a, *di, st)
                                   : batch_t_dispatch<false>(units, q, d, scores, q_lengths, route, a, *di, st);
}

// ---- the dense hard-max backward of the transposed shape (maxsim_dense_t.hip)
constexpr int kDenseTMaxDocs = 4096;        // dP keeps one weight pair per document of the page in LDS
constexpr int kDenseTMaxPagesPer = 256;     // dR keeps one weight pair per (page of its split, document of the workgroup) in LDS
struct DenseTPlan {
    size_t rimg, pimg, partial, bytes;   // byte offsets of the two operand images and the page-split partials; total
    int ks, nsb, nc, n_split, pages_per, doc_groups;
};
static inline size_t up16(size_t x) { return (x + 15) & ~(size_t)15; }
DenseTPlan dense_t_plan(int n_q, int Lq, int n_d, int Ld, int cus) {
    DenseTPlan p{};
    p.ks = (Ld + 31) / 32;
    p.nsb = Ld <= 16 ? 1 : Ld <= 32 ? 2 : 4;     // 16-row blocks per document: NSB * NC = 4 combinations per wave pair
    p.nc = 4 / p.nsb;
    p.doc_groups = (n_d + 4 * p.nc - 1) / (4 * p.nc);
    int split = (cus + p.doc_groups - 1) / (p.doc_groups > 0 ? p.doc_groups : 1);
    const int min_split = (n_q + kDenseTMaxPagesPer - 1) / kDenseTMaxPagesPer;
    split = split < min_split ? min_split : split;
    split = split < 1 ? 1 : split > n_q ? n_q : split;
    p.pages_per = split > 0 ? (n_q + split - 1) / split : 1;
    p.n_split = p.pages_per > 0 ? (n_q + p.pages_per - 1) / p.pages_per : 0;
    const size_t ksp = msim::dense_t_lq_pad(Lq) / 32;
    p.rimg = 0;
    p.pimg = up16((size_t)n_d * p.ks * msim::kKStepBytes);
    p.partial = p.pimg + up16((size_t)n_q * ksp * msim::kKStepBytes);
    p.bytes = p.partial + up16((size_t)p.n_split * n_d * Ld * msim::kDim * sizeof(float));
    return p;
}
bool dense_t_supported(int dtype, int n_q, int Lq, int n_d, int Ld, int dim) {
    if ((dtype != MSIM_DTYPE_BF16 && dtype != MSIM_DTYPE_F16) || dim != msim::kDim) return false;
    if (Ld <= 0 || Ld > msim::kDenseTMaxLd || Lq <= 0 || n_q <= 0 || n_d <= 0 || n_d > kDenseTMaxDocs) return false;
    const long long lq_pad = msim::dense_t_lq_pad(Lq);
    return (long long)n_q * lq_pad * msim::kRowBytes < (1ll << 31) && (long long)n_q * n_d * lq_pad < (1ll << 31) &&
           (long long)n_d * 64 * msim::kRowBytes < (1ll << 31);
}

template <bool F16>
int dense_t_bwd_launch(const uint16_t *Q, const uint16_t *D, const float *G, msim::GScale gs, const uint8_t *route, uint16_t *dQ,
                       uint16_t *dD, char *ws, const DenseTPlan &pl, msim::DenseTArgs a, hipStream_t st) {
    uint16_t *rimg = reinterpret_cast<uint16_t *>(ws + pl.rimg), *pimg = reinterpret_cast<uint16_t *>(ws + pl.pimg);
    float *partial = reinterpret_cast<float *>(ws + pl.partial);
    SideStream *side = nullptr;
    if (int rc = side_stream(&side)) return rc;
    const unsigned e0 = side->next.fetch_add(2) % 8;      // two events of the pool per call (fork, join)
    hipEvent_t fork = side->ev[e0], join = side->ev[(e0 + 1) % 8];
    // fork: the side stream takes the page image and dR (+ its split sum), the caller's stream the document image and dP
    hipError_t e = hipEventRecord(fork, st);
    if (e == hipSuccess) e = hipStreamWaitEvent(side->st, fork, 0);
    if (e != hipSuccess) return fail(MSIM_ELAUNCH, "msim_dense_t_bwd fork: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(msim::dense_t_image_kernel, dim3(a.n_d * pl.ks), dim3(256), 0, st, D, rimg, a.n_d, a.Ld, pl.ks);
    hipLaunchKernelGGL(msim::dense_t_image_kernel, dim3(a.n_q * a.ksp), dim3(256), 0, side->st, Q, pimg, a.n_q, a.Lq, a.ksp);
    static std::atomic<int> conf_long[2][kMaxDevices], conf_short[4][kMaxDevices];
    // LDS: the operand ring + the routing bytes + the W patterns / weight pairs (dP: one per document of the page, padded to whole
    // stages; dR: one per (page of the split, document of the workgroup)); the attribute is raised once to what dense_t_supported admits
    constexpr int kLongRing = msim::kDenseTLongRing, kShortRing = msim::kDenseTShortRing;
    constexpr int kLongStage = msim::kDenseTLongSteps * (8192 + 128 + 144);      // image + routing bytes + W patterns (KS = 1: one document per step)
    constexpr int lds_long_max = kLongRing * kLongStage + 4 * (kDenseTMaxDocs + 4), lds_short_max = kShortRing * (16384 + 1024) + 4 * 16 * kDenseTMaxPagesPer;
    const int lds_long = kLongRing * kLongStage + 4 * ((a.n_d + 3) / 4 * 4 + 4);
    const int lds_short = kShortRing * (16384 + 1024) + 4 * 16 * pl.pages_per;
    const dim3 grid_long((a.Lq + 127) / 128, a.n_q);
    if (pl.ks == 1) {
        auto k = msim::dense_t_bwd_long_kernel<F16, 1>;
        if (int rc = allow_lds(k, lds_long_max, conf_long[0])) return rc;
        hipLaunchKernelGGL(k, grid_long, dim3(512), lds_long, st, rimg, route, G, gs, dQ, a);
    } else {
        auto k = msim::dense_t_bwd_long_kernel<F16, 2>;
        if (int rc = allow_lds(k, lds_long_max, conf_long[1])) return rc;
        hipLaunchKernelGGL(k, grid_long, dim3(512), lds_long, st, rimg, route, G, gs, dQ, a);
    }
    const dim3 grid_short(pl.doc_groups, pl.n_split);
#define MSIM_SHORT(NSB, NC, SLOT)                                                                         \
    {                                                                                                     \
        auto k = msim::dense_t_bwd_short_kernel<F16, NSB, NC>;                                            \
        if (int rc = allow_lds(k, lds_short_max, conf_short[SLOT])) return rc;                            \
        hipLaunchKernelGGL(k, grid_short, dim3(512), lds_short, side->st, pimg, route, G, gs, partial, a); \
    }
    if (pl.nsb <= 1) MSIM_SHORT(1, 4, 0)
    else if (pl.nsb == 2) MSIM_SHORT(2, 2, 1)
    else MSIM_SHORT(4, 1, 2)
#undef MSIM_SHORT
    const long long n_elems = (long long)a.n_d * a.Ld * msim::kDim;
    hipLaunchKernelGGL(msim::dense_t_bwd_short_sum_kernel<F16>, dim3((unsigned)((n_elems / 4 + 255) / 256)), dim3(256), 0, side->st, partial, dD,
                       n_elems, pl.n_split);
    // join: the caller's stream continues when both halves are done
    e = hipEventRecord(join, side->st);
    if (e == hipSuccess) e = hipStreamWaitEvent(st, join, 0);
    if (e != hipSuccess) return fail(MSIM_ELAUNCH, "msim_dense_t_bwd join: %s", hipGetErrorString(e));
    e = hipGetLastError();
    if (e != hipSuccess) return fail(MSIM_ELAUNCH, "msim_dense_t_bwd launch: %s", hipGetErrorString(e));
    return MSIM_OK;
}
}  // namespace

extern "C" {

int msim_fwd_transposed(int dtype, const void *Q, int n_q, int Lq, const void *D, int n_d, int Ld, int dim, float *scores,
                        int64_t ld_scores, int32_t *q_lengths, void *stream) {
    return fwd_transposed(dtype, Q, n_q, Lq, D, n_d, Ld, dim, scores, ld_scores, q_lengths, nullptr, stream);
}

size_t msim_dense_t_route_bytes(int n_q, int Lq, int n_d) {
    if (n_q <= 0 || Lq <= 0 || n_d <= 0) return 0;
    return (size_t)n_q * n_d * msim::dense_t_lq_pad(Lq);
}

int msim_dense_t_supported(int dtype, int n_q, int Lq, int n_d, int Ld, int dim) {
    return dense_t_supported(dtype, n_q, Lq, n_d, Ld, dim) ? 1 : 0;
}

int msim_fwd_transposed_route(int dtype, const void *Q, int n_q, int Lq, const void *D, int n_d, int Ld, int dim, float *scores,
                              int64_t ld_scores, int32_t *q_lengths, uint8_t *route, void *stream) {
    if (!route) return fail(MSIM_EINVAL, "null routing buffer");
    if (n_q > 0 && n_d > 0 && !dense_t_supported(dtype, n_q, Lq, n_d, Ld, dim))
        return fail(MSIM_EUNSUPPORTED, "msim_fwd_transposed_route: bf16 / f16, width %d, documents of at most %d rows, sizes below 2^31 bytes",
                    msim::kDim, msim::kDenseTMaxLd);
    return fwd_transposed(dtype, Q, n_q, Lq, D, n_d, Ld, dim, scores, ld_scores, q_lengths, route, stream);
}

size_t msim_dense_t_bwd_workspace_bytes(int n_q, int Lq, int n_d, int Ld, int dim) {
    (void)dim;
    if (n_q <= 0 || n_d <= 0 || Lq <= 0 || Ld <= 0) return 0;
    const DeviceInfo *di = nullptr;
    const int cus = device_info(&di) == MSIM_OK ? di->cus : 256;            // the plan only has to be the same in both calls
    return dense_t_plan(n_q, Lq, n_d, Ld, cus).bytes;
}

int msim_dense_t_bwd(int dtype, const void *Q, int n_q, int Lq, const void *D, int n_d, int Ld, int dim, const float *G, int64_t ldg,
                     const void *g_scale, int g_scale_dtype, const uint8_t *route, void *dQ, void *dD, void *workspace, void *stream) {
    if (n_q < 0 || n_d < 0 || Lq <= 0 || Ld <= 0) return fail(MSIM_EINVAL, "negative or empty size");
    if (n_q == 0 || n_d == 0) return MSIM_OK;
    if (!Q || !D || !G || !route || !dQ || !dD || !workspace) return fail(MSIM_EINVAL, "null pointer argument");
    if (!dense_t_supported(dtype, n_q, Lq, n_d, Ld, dim))
        return fail(MSIM_EUNSUPPORTED, "msim_dense_t_bwd: bf16 / f16, width %d, documents of at most %d rows, sizes below 2^31 bytes",
                    msim::kDim, msim::kDenseTMaxLd);
    if (ldg < n_d) return fail(MSIM_EINVAL, "ldg < n_d");
    if (g_scale && g_scale_dtype != MSIM_DTYPE_BF16 && g_scale_dtype != MSIM_DTYPE_F16 && g_scale_dtype != MSIM_DTYPE_F32)
        return fail(MSIM_EINVAL, "g_scale dtype code %d", g_scale_dtype);
    if ((reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(D) | reinterpret_cast<uintptr_t>(dQ) | reinterpret_cast<uintptr_t>(dD) |
         reinterpret_cast<uintptr_t>(workspace)) & 15)
        return fail(MSIM_EINVAL, "embeddings, gradients and workspace must be 16-byte aligned");
    const DeviceInfo *di = nullptr;
    if (int rc = device_info(&di)) return rc;
    const DenseTPlan pl = dense_t_plan(n_q, Lq, n_d, Ld, di->cus);
    msim::DenseTArgs a{};
    a.ldg = ldg;
    a.n_q = n_q;
    a.Lq = Lq;
    a.n_d = n_d;
    a.Ld = Ld;
    a.Lq_pad = msim::dense_t_lq_pad(Lq);
    a.ksp = a.Lq_pad / 32;
    a.n_split = pl.n_split;
    a.pages_per = pl.pages_per;
    if (msim::kAbBuild) {
        const char *e = getenv("MSIM_DENSE_T_DBG");
        a.dbg = e ? atoi(e) : 0;
        const char *o = getenv("MSIM_DENSE_T_DBG_OUT");
        a.dbg_out = o ? reinterpret_cast<unsigned long long *>(strtoull(o, nullptr, 0)) : nullptr;
    }
    const msim::GScale gs{g_scale, g_scale_dtype};
    const uint16_t *q = static_cast<const uint16_t *>(Q), *d = static_cast<const uint16_t *>(D);
    hipStream_t st = static_cast<hipStream_t>(stream);
    return dtype == MSIM_DTYPE_F16
               ? dense_t_bwd_launch<true>(q, d, G, gs, route, static_cast<uint16_t *>(dQ), static_cast<uint16_t *>(dD),
                                          static_cast<char *>(workspace), pl, a, st)
               : dense_t_bwd_launch<false>(q, d, G, gs, route, static_cast<uint16_t *>(dQ), static_cast<uint16_t *>(dD),
                                           static_cast<char *>(workspace), pl, a, st);
}

// ---------------------------------------------------------------- pair lists (training losses)
int msim_pairs_argmax(int dtype, const void *Q, int n_q, int Lq, const void *D, const int32_t *d_off,
                      const uint8_t *d_clamp0, int n_d, int dim, int max_doc_rows, const int32_t *pairs, int n_pairs,
                      float *out_scores, int32_t *out_argmax, void *stream) {
    if (n_q < 0 || n_d < 0 || Lq <= 0 || n_pairs < 0 || max_doc_rows < 0) return fail(MSIM_EINVAL, "negative size");
    if (n_pairs == 0) return MSIM_OK;
    if (!pairs) return fail(MSIM_EINVAL, "null pointer argument");
    if (int rc = check_common(Q, D, d_off, dtype, dim, Lq)) return rc;
    const DeviceInfo *di = nullptr;
    if (int rc = device_info(&di)) return rc;
    const int tpq = (Lq + msim::kTokTile - 1) / msim::kTokTile;
    msim::PairsArgs a{n_q, Lq, n_d, n_pairs};
    hipStream_t st = static_cast<hipStream_t>(stream);
    const uint16_t *q = static_cast<const uint16_t *>(Q), *d = static_cast<const uint16_t *>(D);
    // long queries x short documents in the tuned dtype / width (pages as queries, the trainer's symmetric direction): transposed form
    if (dtype != MSIM_DTYPE_F32 && dim == msim::kDim && Lq > kLongSegRows && max_doc_rows > 0 && max_doc_rows <= 4 * msim::kTokTile &&
        (long long)Lq * msim::kRowBytes < (1ll << 31)) {
        const int tpd = (max_doc_rows + msim::kTokTile - 1) / msim::kTokTile;
        return dtype == MSIM_DTYPE_F16
                   ? pairs_argmax_t_dispatch<true>(tpd, q, d, d_off, d_clamp0, pairs, out_scores, out_argmax, a, *di, st)
                   : pairs_argmax_t_dispatch<false>(tpd, q, d, d_off, d_clamp0, pairs, out_scores, out_argmax, a, *di, st);
    }
    if (!is_tuned(dtype, dim, Lq)) {
        const char *qc = static_cast<const char *>(Q), *dc = static_cast<const char *>(D);
        const int rb = dim * elem_bytes(dtype);
        switch (dtype) {
            case MSIM_DTYPE_F32:
                return generic_pairs_argmax<msim::kDtypeF32>(qc, dc, d_off, d_clamp0, pairs, out_scores, out_argmax, a, rb, *di, st);
            case MSIM_DTYPE_F16:
                return generic_pairs_argmax<msim::kDtypeF16>(qc, dc, d_off, d_clamp0, pairs, out_scores, out_argmax, a, rb, *di, st);
            default:
                return generic_pairs_argmax<msim::kDtypeBf16>(qc, dc, d_off, d_clamp0, pairs, out_scores, out_argmax, a, rb, *di, st);
        }
    }
    return dtype == MSIM_DTYPE_F16
               ? pairs_argmax_dispatch<true>(tpq, q, d, d_off, d_clamp0, pairs, out_scores, out_argmax, a, *di, st)
               : pairs_argmax_dispatch<false>(tpq, q, d, d_off, d_clamp0, pairs, out_scores, out_argmax, a, *di, st);
}

int msim_allpairs_argmax(int dtype, const void *Q, int n_q, int Lq, const void *D, const int32_t *d_off, const uint8_t *d_clamp0,
                         int n_d, int dim, float *out_scores, int64_t ld_scores, int32_t *out_argmax, void *stream) {
    if (n_q < 0 || n_d < 0 || Lq <= 0) return fail(MSIM_EINVAL, "negative size");
    if (n_q == 0 || n_d == 0) return MSIM_OK;
    if (!out_scores && !out_argmax) return fail(MSIM_EINVAL, "nothing to compute");
    if (out_scores && ld_scores < n_d) return fail(MSIM_EINVAL, "ld_scores < n_d");
    if (int rc = check_common(Q, D, d_off, dtype, dim, Lq)) return rc;
    if (!is_tuned(dtype, dim, Lq))
        return fail(MSIM_EUNSUPPORTED, "msim_allpairs_argmax takes bf16 / f16 embeddings of width %d and queries of at most %d tokens "
                    "(list the pairs and call msim_pairs_argmax otherwise)", msim::kDim, 4 * msim::kTokTile);
    if ((long long)n_q * n_d > 0x7fffffffLL) return fail(MSIM_EUNSUPPORTED, "more than 2^31 pairs");
    const DeviceInfo *di = nullptr;
    if (int rc = device_info(&di)) return rc;
    const int tpq = (Lq + msim::kTokTile - 1) / msim::kTokTile;
    msim::PairsArgs a{n_q, Lq, n_d, n_q * n_d};
    hipStream_t st = static_cast<hipStream_t>(stream);
    const uint16_t *q = static_cast<const uint16_t *>(Q), *d = static_cast<const uint16_t *>(D);
    return dtype == MSIM_DTYPE_F16 ? allpairs_argmax_dispatch<true>(tpq, q, d, d_off, d_clamp0, out_scores, ld_scores, out_argmax, a, *di, st)
                                   : allpairs_argmax_dispatch<false>(tpq, q, d, d_off, d_clamp0, out_scores, ld_scores, out_argmax, a, *di, st);
}

size_t msim_pairs_bwd_workspace_bytes(int n_q, int Lq, int n_d, int dim, int max_doc_rows, int n_pairs) {
    (void)n_q;
    const DeviceInfo *di = nullptr;
    const int cus = device_info(&di) == MSIM_OK ? di->cus : 256;            // the plan only has to be the same in both calls
    return dd_plan(n_pairs, Lq, n_d, dim, max_doc_rows, cus).bytes;
}

int msim_pairs_bwd(int dtype, const void *Q, int n_q, int Lq, const void *D, const int32_t *d_off, int n_d, int dim,
                   int max_doc_rows, const int32_t *pairs, const int32_t *order_by_doc, const float *g,
                   const void *g_scale, int g_scale_dtype, const int32_t *argmax, int n_pairs, int out_dtype, void *dQ, void *dD,
                   void *workspace, void *stream) {
    if (n_q < 0 || n_d < 0 || Lq <= 0 || n_pairs < 0 || max_doc_rows < 0) return fail(MSIM_EINVAL, "negative size");
    if (!dQ || !dD) return fail(MSIM_EINVAL, "null pointer argument");
    if (n_pairs > 0 && (!pairs || !order_by_doc || !g || !argmax)) return fail(MSIM_EINVAL, "null pair-list argument");
    if (int rc = check_common(Q, D, d_off, dtype, dim, Lq)) return rc;
    if ((max_doc_rows + msim::kBwdRows - 1) / msim::kBwdRows > 65535)
        return fail(MSIM_EUNSUPPORTED, "max_doc_rows=%d too large", max_doc_rows);
    if (n_d > 0x7fffffff / 2) return fail(MSIM_EUNSUPPORTED, "too many documents");
    if (out_dtype != MSIM_DTYPE_F32 && out_dtype != dtype)
        return fail(MSIM_EINVAL, "gradients come out as fp32 or in the embeddings' own dtype (out_dtype %d, dtype %d)", out_dtype, dtype);
    if (g_scale && g_scale_dtype != MSIM_DTYPE_BF16 && g_scale_dtype != MSIM_DTYPE_F16 && g_scale_dtype != MSIM_DTYPE_F32)
        return fail(MSIM_EINVAL, "g_scale dtype code %d", g_scale_dtype);
    const DeviceInfo *di = nullptr;
    if (int rc = device_info(&di)) return rc;
    msim::PairsArgs a{n_q, Lq, n_d, n_pairs};
    hipStream_t st = static_cast<hipStream_t>(stream);
    // short documents with long entry lists (the trainer's symmetric direction): the dense dD form, through the caller's scratch
    DdPlan pl;
    float *partial = static_cast<float *>(workspace);
    if (partial) {
        pl = dd_plan(n_pairs, Lq, n_d, dim, max_doc_rows, di->cus);
        if (reinterpret_cast<uintptr_t>(workspace) & 15) return fail(MSIM_EINVAL, "workspace must be 16-byte aligned");
    }
    const msim::GScale gs{g_scale, g_scale_dtype};
    const char *qc = static_cast<const char *>(Q), *dc = static_cast<const char *>(D);
    const bool out16 = out_dtype != MSIM_DTYPE_F32;
#define MSIM_BWD(DT, O16) \
    launch_bwd_kernels<DT, O16>(qc, dc, d_off, max_doc_rows, pairs, order_by_doc, g, argmax, dQ, dD, a, dim, di->cus, st, partial, pl, gs)
    if (dtype == MSIM_DTYPE_F32) MSIM_BWD(msim::kDtypeF32, false);
    else if (dtype == MSIM_DTYPE_F16) { if (out16) MSIM_BWD(msim::kDtypeF16, true); else MSIM_BWD(msim::kDtypeF16, false); }
    else { if (out16) MSIM_BWD(msim::kDtypeBf16, true); else MSIM_BWD(msim::kDtypeBf16, false); }
#undef MSIM_BWD
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MSIM_ELAUNCH, "maxsim_pairs_bwd launch: %s", hipGetErrorString(e));
    return MSIM_OK;
}

// ---------------------------------------------------------------- loss epilogue
// one workgroup reads the whole score matrix when it is small (loss_epilogue_small_kernel): no scratch, no ticket
static bool epilogue_is_small(int B, int C) { return B > 0 && B <= msim::kEpiSmallRows && (long long)B * C <= (1 << 18); }

size_t msim_loss_epilogue_workspace_bytes(int B, int C) {
    if (epilogue_is_small(B, C)) return 0;
    return B > 0 ? 16 + (size_t)3 * B * sizeof(float) : 16;
}

int msim_loss_epilogue(int mode, const float *scores, int64_t ld, int B, int C, const void *Q, int q_dtype, int Lq, int width,
                       int offset, float temperature, int normalize, int filter, float filter_threshold, float filter_factor,
                       float *G, int32_t *pairs, float *coef, int32_t *order, void *workspace, float *out, void *loss_out,
                       const int32_t *q_lengths, void *stream) {
    if (B < 0 || C < 0 || Lq < 0 || width <= 0) return fail(MSIM_EINVAL, "negative size");
    if (mode != MSIM_LOSS_PAIRWISE && mode != MSIM_LOSS_INFONCE && mode != MSIM_LOSS_SIGMOID) return fail(MSIM_EINVAL, "unknown loss mode %d", mode);
    if (mode == MSIM_LOSS_SIGMOID && C != B)
        return fail(MSIM_EINVAL, "the sigmoid loss is defined on the in-batch square: %d queries, %d documents", B, C);
    if (!scores || !Q || !out) return fail(MSIM_EINVAL, "null pointer argument");
    if (q_dtype != MSIM_DTYPE_BF16 && q_dtype != MSIM_DTYPE_F16 && q_dtype != MSIM_DTYPE_F32)
        return fail(MSIM_EUNSUPPORTED, "dtype code %d", q_dtype);
    if (B == 0) return fail(MSIM_EINVAL, "empty batch");
    if (offset < 0 || (long long)offset + B > C) return fail(MSIM_EINVAL, "offset %d + batch %d exceeds the %d documents", offset, B, C);
    if (ld < C) return fail(MSIM_EINVAL, "ld=%lld < C=%d", (long long)ld, C);
    if (temperature == 0.0f) return fail(MSIM_EINVAL, "temperature must be non-zero");
    if (mode == MSIM_LOSS_PAIRWISE) {
        if (C < 2) return fail(MSIM_EINVAL, "the pairwise loss needs at least 2 documents (topk(2))");
        if (!pairs || !coef || !order) return fail(MSIM_EINVAL, "null pair-list output");
    }
    const bool small = epilogue_is_small(B, C);
    if (!small && !workspace) return fail(MSIM_EINVAL, "this batch needs msim_loss_epilogue_workspace_bytes(B, C) bytes of zero-filled scratch");
    if (workspace && (reinterpret_cast<uintptr_t>(workspace) & 15)) return fail(MSIM_EINVAL, "workspace must be 16-byte aligned");
    msim::EpiArgs a;
    a.ld = ld;
    a.B = B;
    a.C = C;
    a.Lq = Lq;
    a.q_elem_bytes = elem_bytes(q_dtype);
    a.q_is_f16 = q_dtype == MSIM_DTYPE_F16;
    a.q_row_bytes = width * a.q_elem_bytes;
    a.offset = offset;
    a.mode = mode == MSIM_LOSS_PAIRWISE ? msim::kEpiPairwise : mode == MSIM_LOSS_SIGMOID ? msim::kEpiSigmoid : msim::kEpiInfoNCE;
    a.normalize = normalize != 0;
    a.filter = filter != 0;
    a.inv_T = 1.0f / temperature;
    a.filter_threshold = filter_threshold;
    a.filter_factor = filter_factor;
    if (small) {
        const int staged = (long long)B * C <= msim::kEpiStageFloats;
        const int lds = staged ? B * C * (int)sizeof(float) : 0;
        static std::atomic<int> configured[kMaxDevices];
        if (int rc = allow_lds(msim::loss_epilogue_small_kernel, msim::kEpiStageFloats * (int)sizeof(float), configured)) return rc;
        hipLaunchKernelGGL(msim::loss_epilogue_small_kernel, dim3(1), dim3(msim::kEpiSmallThreads), lds, static_cast<hipStream_t>(stream),
                           scores, static_cast<const char *>(Q), q_lengths, G, pairs, coef, order, out, loss_out, a, staged);
    } else {
        char *ws = static_cast<char *>(workspace);
        hipLaunchKernelGGL(msim::loss_epilogue_kernel, dim3(B), dim3(msim::kEpiThreads), 0, static_cast<hipStream_t>(stream), scores,
                           static_cast<const char *>(Q), G, pairs, coef, order, reinterpret_cast<float *>(ws + 16),
                           reinterpret_cast<unsigned int *>(ws), out, loss_out, q_lengths, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MSIM_ELAUNCH, "loss_epilogue_kernel launch: %s", hipGetErrorString(e));
    return MSIM_OK;
}

// ---------------------------------------------------------------- smooth-max entry points
int msim_smooth_fwd(int dtype, const void *Q, int n_q, int Lq, const void *D, const int32_t *d_off, int n_d, int dim, float tau,
                    float *scores, int64_t ld_scores, void *stream) {
    if (n_q < 0 || n_d < 0) return fail(MSIM_EINVAL, "negative size (n_q=%d n_d=%d)", n_q, n_d);
    if (n_q == 0 || n_d == 0) return MSIM_OK;
    if (!scores) return fail(MSIM_EINVAL, "null pointer argument");
    if (int rc = check_smooth(Q, D, d_off, dtype, dim, Lq, tau)) return rc;
    if (ld_scores < n_d) return fail(MSIM_EINVAL, "ld_scores=%lld < n_d=%d", (long long)ld_scores, n_d);
    GenericCall c;
    if (int rc = device_info(&c.di)) return rc;
    c.Q = static_cast<const char *>(Q);
    c.D = static_cast<const char *>(D);
    c.d_off = d_off;
    c.clamp0 = nullptr;
    c.scores = scores;
    c.ld = ld_scores;
    c.n_q = n_q;
    c.Lq = Lq;
    c.n_d = n_d;
    c.row_bytes = dim * elem_bytes(dtype);
    c.flags = 0;
    c.st = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case MSIM_DTYPE_F32: return smooth_dispatch<msim::kDtypeF32>(c, tau);
        case MSIM_DTYPE_F16: return smooth_dispatch<msim::kDtypeF16>(c, tau);
        default: return smooth_dispatch<msim::kDtypeBf16>(c, tau);
    }
}

}  // extern "C"

namespace {
template <int TPQ, bool F16>
int launch_smooth_pairs_stream(const uint16_t *Q, const uint16_t *D, const int32_t *d_off, const int32_t *pairs, float *out_scores,
                               float *out_lse, const msim::PairsArgs &a, float tau, const DeviceInfo &di, hipStream_t st) {
    auto kern = msim::maxsim_smooth_pairs_stream_kernel<TPQ, F16>;
    constexpr int lds = 4 * msim::kPairsRing * msim::kSlabBytes;
    static std::atomic<int> configured[kMaxDevices];
    if (int rc = allow_lds(kern, lds, configured)) return rc;
    const int wg_needed = (a.n_pairs + 3) / 4;
    const int wg_cap = di.cus * (di.lds_per_cu / lds);
    hipLaunchKernelGGL(kern, dim3(wg_needed < wg_cap ? wg_needed : wg_cap), dim3(256), lds, st, Q, D, d_off, pairs, out_scores, out_lse,
                       a, tau);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MSIM_ELAUNCH, "maxsim_smooth_pairs_stream_kernel<%d> launch: %s", TPQ, hipGetErrorString(e));
    return MSIM_OK;
}
template <bool F16>
int smooth_pairs_stream_dispatch(int tpq, const uint16_t *Q, const uint16_t *D, const int32_t *d_off, const int32_t *pairs,
                                 float *out_scores, float *out_lse, const msim::PairsArgs &a, float tau, const DeviceInfo &di,
                                 hipStream_t st) {
    switch (tpq) {
        case 1: return launch_smooth_pairs_stream<1, F16>(Q, D, d_off, pairs, out_scores, out_lse, a, tau, di, st);
        case 2: return launch_smooth_pairs_stream<2, F16>(Q, D, d_off, pairs, out_scores, out_lse, a, tau, di, st);
        case 3: return launch_smooth_pairs_stream<3, F16>(Q, D, d_off, pairs, out_scores, out_lse, a, tau, di, st);
        default: return launch_smooth_pairs_stream<4, F16>(Q, D, d_off, pairs, out_scores, out_lse, a, tau, di, st);
    }
}
}  // namespace

extern "C" {

int msim_smooth_pairs(int dtype, const void *Q, int n_q, int Lq, const void *D, const int32_t *d_off, int n_d, int dim,
                      const int32_t *pairs, int n_pairs, float tau, float *out_scores, float *out_lse, void *stream) {
    if (n_q < 0 || n_d < 0 || n_pairs < 0) return fail(MSIM_EINVAL, "negative size");
    if (n_pairs == 0) return MSIM_OK;
    if (!pairs) return fail(MSIM_EINVAL, "null pointer argument");
    if (int rc = check_smooth(Q, D, d_off, dtype, dim, Lq, tau)) return rc;
    const DeviceInfo *di = nullptr;
    if (int rc = device_info(&di)) return rc;
    msim::PairsArgs a{n_q, Lq, n_d, n_pairs};
    const char *qc = static_cast<const char *>(Q), *dc = static_cast<const char *>(D);
    const int rb = dim * elem_bytes(dtype);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int tpq = (Lq + msim::kTokTile - 1) / msim::kTokTile;
    if (dim == msim::kDim && dtype != MSIM_DTYPE_F32 && tpq <= 4) {   // 128 x 16-bit rows: the LDS-DMA pipeline
        const uint16_t *q16 = static_cast<const uint16_t *>(Q), *d16 = static_cast<const uint16_t *>(D);
        return dtype == MSIM_DTYPE_F16 ? smooth_pairs_stream_dispatch<true>(tpq, q16, d16, d_off, pairs, out_scores, out_lse, a, tau, *di, st)
                                       : smooth_pairs_stream_dispatch<false>(tpq, q16, d16, d_off, pairs, out_scores, out_lse, a, tau, *di, st);
    }
    switch (dtype) {
        case MSIM_DTYPE_F32: return smooth_pairs<msim::kDtypeF32>(qc, dc, d_off, pairs, out_scores, out_lse, a, rb, tau, *di, st);
        case MSIM_DTYPE_F16: return smooth_pairs<msim::kDtypeF16>(qc, dc, d_off, pairs, out_scores, out_lse, a, rb, tau, *di, st);
        default: return smooth_pairs<msim::kDtypeBf16>(qc, dc, d_off, pairs, out_scores, out_lse, a, rb, tau, *di, st);
    }
}

size_t msim_smooth_bwd_workspace_bytes(int n_q, int Lq, int dim) {
    const DeviceInfo *di = nullptr;
    if (n_q <= 0 || Lq <= 0 || dim <= 0 || device_info(&di)) return 0;
    const int ns = smooth_dq_split(n_q, Lq, *di);
    return ns > 1 ? (size_t)ns * n_q * Lq * dim * sizeof(float) : 0;
}

int msim_smooth_pairs_bwd(int dtype, const void *Q, int n_q, int Lq, const void *D, const int32_t *d_off, int n_d, int dim,
                          int max_doc_rows, const int32_t *pairs, const int32_t *order_by_doc, const float *g, const float *lse,
                          int n_pairs, float tau, float *dQ, float *dD, void *workspace, void *stream) {
    if (n_q < 0 || n_d < 0 || n_pairs < 0 || max_doc_rows < 0) return fail(MSIM_EINVAL, "negative size");
    if (!dQ || !dD) return fail(MSIM_EINVAL, "null pointer argument");
    if (n_pairs > 0 && (!pairs || !order_by_doc || !g || !lse)) return fail(MSIM_EINVAL, "null pair-list argument");
    if (int rc = check_smooth(Q, D, d_off, dtype, dim, Lq, tau)) return rc;
    if ((max_doc_rows + 31) / 32 > 65535) return fail(MSIM_EUNSUPPORTED, "max_doc_rows=%d too large", max_doc_rows);
    if ((Lq + 31) / 32 > 65535) return fail(MSIM_EUNSUPPORTED, "Lq=%d too large", Lq);
    const DeviceInfo *di = nullptr;
    if (int rc = device_info(&di)) return rc;
    const int ns = smooth_dq_split(n_q, Lq, *di);
    if (ns > 1 && !workspace) return fail(MSIM_EINVAL, "workspace required (msim_smooth_bwd_workspace_bytes)");
    if ((reinterpret_cast<uintptr_t>(dQ) | reinterpret_cast<uintptr_t>(dD) | reinterpret_cast<uintptr_t>(workspace)) & 15)
        return fail(MSIM_EINVAL, "dQ, dD and the workspace must be 16-byte aligned");
    msim::SmoothBwdArgs a{n_q, Lq, n_d, n_pairs, dim * elem_bytes(dtype), dim, tau, 1};
    const char *qc = static_cast<const char *>(Q), *dc = static_cast<const char *>(D);
    float *ws = static_cast<float *>(workspace);
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case MSIM_DTYPE_F32: return smooth_bwd<msim::kDtypeF32>(qc, dc, d_off, max_doc_rows, pairs, order_by_doc, g, lse, dQ, dD, ws, a, ns, st);
        case MSIM_DTYPE_F16: return smooth_bwd<msim::kDtypeF16>(qc, dc, d_off, max_doc_rows, pairs, order_by_doc, g, lse, dQ, dD, ws, a, ns, st);
        default: return smooth_bwd<msim::kDtypeBf16>(qc, dc, d_off, max_doc_rows, pairs, order_by_doc, g, lse, dQ, dD, ws, a, ns, st);
    }
}

// ---------------------------------------------------------------- embedding head (the producer of the corpus format)
int msim_embed_head_row_map(const void *mask, int mask_kind, const void *extra, int extra_kind, int64_t M, int32_t *row_map, void *stream) {
    if (M < 0) return fail(MSIM_EINVAL, "bad size (M=%lld)", (long long)M);
    if (M == 0) return MSIM_OK;
    if (!mask || !row_map) return fail(MSIM_EINVAL, "null pointer argument");
    if (mask_kind < 0 || mask_kind > 6 || (extra && (extra_kind < 0 || extra_kind > 6))) return fail(MSIM_EINVAL, "unknown mask kind");
    if (M > 0x7ffffffdLL) return fail(MSIM_EUNSUPPORTED, "too many rows for an int32 row map");
    const long long padded = (M + msim::kHeadBM - 1) / msim::kHeadBM * msim::kHeadBM;
    hipLaunchKernelGGL(msim::head_row_map_kernel, dim3((unsigned)((padded + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), mask,
                       mask_kind, extra, extra_kind, (long long)M, padded, row_map);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MSIM_ELAUNCH, "head_row_map_kernel launch: %s", hipGetErrorString(e));
    return MSIM_OK;
}

int msim_embed_head_writer_map(const void *mask, int mask_kind, const void *extra, int extra_kind, int B, int S, const int64_t *rows_before,
                               int64_t *counts, int32_t *row_map, int64_t *rows_after, void *stream) {
    if (B < 0 || S <= 0) return fail(MSIM_EINVAL, "bad size (B=%d S=%d)", B, S);
    if (B == 0) return MSIM_OK;
    if (!mask || !rows_before || !counts || !row_map || !rows_after) return fail(MSIM_EINVAL, "null pointer argument");
    if (mask_kind < 0 || mask_kind > 6 || (extra && (extra_kind < 0 || extra_kind > 6))) return fail(MSIM_EINVAL, "unknown mask kind");
    const long long M = (long long)B * S;
    const long long padded = (M + msim::kHeadBM - 1) / msim::kHeadBM * msim::kHeadBM;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(msim::head_page_count_kernel, dim3(B), dim3(256), 0, st, mask, mask_kind, extra, extra_kind, S,
                       reinterpret_cast<long long *>(counts));
    hipLaunchKernelGGL(msim::head_writer_map_kernel, dim3(B + 1), dim3(256), 0, st, mask, mask_kind, extra, extra_kind, B, S,
                       reinterpret_cast<const long long *>(rows_before), reinterpret_cast<const long long *>(counts), padded, row_map,
                       reinterpret_cast<long long *>(rows_after));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MSIM_ELAUNCH, "head_writer_map_kernel launch: %s", hipGetErrorString(e));
    return MSIM_OK;
}

int msim_embed_head_bwd(int dtype, const void *proj, const void *grad_out, const int32_t *row_map, int64_t M, int n_out,
                        void *dproj, void *stream) {
    if (M < 0) return fail(MSIM_EINVAL, "bad size (M=%lld)", (long long)M);
    if (M == 0) return MSIM_OK;
    if (!proj || !grad_out || !row_map || !dproj) return fail(MSIM_EINVAL, "null pointer argument");
    if (dtype != MSIM_DTYPE_BF16 && dtype != MSIM_DTYPE_F16)
        return fail(MSIM_EUNSUPPORTED, "dtype code %d: the embedding head takes bfloat16 (0) or float16 (1)", dtype);
    if (n_out != msim::kHeadN) return fail(MSIM_EUNSUPPORTED, "n_out=%d: the embedding head is built for 128 output columns", n_out);
    if ((reinterpret_cast<uintptr_t>(proj) | reinterpret_cast<uintptr_t>(grad_out) | reinterpret_cast<uintptr_t>(dproj)) & 15)
        return fail(MSIM_EINVAL, "proj, grad_out and dproj must be 16-byte aligned");
    const DeviceInfo *di = nullptr;
    if (int rc = device_info(&di)) return rc;
    const long long blocks = (M + 15) / 16;
    const int grid = (int)(blocks < (long long)di->cus * 16 ? blocks : (long long)di->cus * 16);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const uint16_t *p = static_cast<const uint16_t *>(proj), *g = static_cast<const uint16_t *>(grad_out);
    uint16_t *o = static_cast<uint16_t *>(dproj);
    if (dtype == MSIM_DTYPE_F16)
        hipLaunchKernelGGL(msim::embed_head_bwd_rows_kernel<true>, dim3(grid), dim3(256), 0, st, p, g, row_map, (long long)M, o);
    else
        hipLaunchKernelGGL(msim::embed_head_bwd_rows_kernel<false>, dim3(grid), dim3(256), 0, st, p, g, row_map, (long long)M, o);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MSIM_ELAUNCH, "embed_head_bwd_rows_kernel launch: %s", hipGetErrorString(e));
    return MSIM_OK;
}

int msim_embed_head(int dtype, const void *X, int64_t M, int H, const void *W, const void *bias, int n_out,
                    const int32_t *row_map, void *out, int64_t ld_out, void *stream) {
    if (M < 0 || H <= 0) return fail(MSIM_EINVAL, "bad size (M=%lld H=%d)", (long long)M, H);
    if (M == 0) return MSIM_OK;
    if (!X || !W || !row_map || !out) return fail(MSIM_EINVAL, "null pointer argument");
    if (dtype != MSIM_DTYPE_BF16 && dtype != MSIM_DTYPE_F16)
        return fail(MSIM_EUNSUPPORTED, "dtype code %d: the embedding head takes bfloat16 (0) or float16 (1) hidden states", dtype);
    if (n_out != msim::kHeadN) return fail(MSIM_EUNSUPPORTED, "n_out=%d: the embedding head is built for 128 output columns", n_out);
    if (H % msim::kHeadBK != 0 || H > 16384) return fail(MSIM_EUNSUPPORTED, "H=%d: hidden size must be a multiple of 64, <= 16384", H);
    if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(W)) & 15) return fail(MSIM_EINVAL, "X and W must be 16-byte aligned");
    if (ld_out < msim::kHeadN) return fail(MSIM_EINVAL, "ld_out=%lld < 128", (long long)ld_out);
    if (M > (int64_t)0x7fffffff * 128) return fail(MSIM_EUNSUPPORTED, "too many rows");
    const DeviceInfo *di = nullptr;
    if (int rc = device_info(&di)) return rc;
    msim::HeadArgs a;
    a.M = M;
    a.H = H;
    a.ld_out = ld_out;
    a.trace = nullptr;
    a.stagger = ab_env("MSIM_HEAD_STAGGER", 0);
    a.stagger_sleep = ab_env("MSIM_HEAD_STAGGER_SLEEP", 1);
#ifdef MSIM_TRACE                                            // `make trace` only: device address of 9 x 8 uint64 (tools/trace_head.py)
    if (const char *tp = getenv("MSIM_HEAD_TRACE_PTR")) a.trace = reinterpret_cast<unsigned long long *>(strtoull(tp, nullptr, 0));
#endif
    const long long tiles = (M + msim::kHeadBM - 1) / msim::kHeadBM;
    const int grid = tiles < di->cus ? (int)tiles : di->cus;
    const long long tiles_h = (M + 127) / 128;                               // HALF variant: 128-row tiles, two workgroups per CU
    const int grid_h = tiles_h < 2 * di->cus ? (int)tiles_h : 2 * di->cus;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const uint16_t *x = static_cast<const uint16_t *>(X), *w = static_cast<const uint16_t *>(W), *b = static_cast<const uint16_t *>(bias);
    uint16_t *o = static_cast<uint16_t *>(out);
    auto go = [&](auto kern, std::atomic<int> *configured, int lds) -> int {
        if (int rc = allow_lds(kern, lds, configured)) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(msim::kHeadThreads), lds, st, x, w, b, row_map, o, a);
        return MSIM_OK;
    };
    auto go_half = [&](auto kern, std::atomic<int> *configured) -> int {
        constexpr int lds = 3 * 128 * 128 + 2 * msim::kHeadWBytes;             // 48 + 32 KiB
        if (int rc = allow_lds(kern, lds, configured)) return rc;
        hipLaunchKernelGGL(kern, dim3(grid_h), dim3(320), lds, st, x, w, b, row_map, o, a);
        return MSIM_OK;
    };
    int rc;
    const bool f16 = dtype == MSIM_DTYPE_F16;
    // Shipped: loader two weight chunks ahead (rings 3 + 3), output rows staged through LDS and stored as whole rows with the
    // streaming policy (needs 16-byte aligned output rows; the 2-byte-store form of the same kernel otherwise).  Every other
    // variant of embed_head_kernel was measured and not kept; they are compiled into measurement builds only
    // (preprocessor, not `if constexpr`: in a non-template function a discarded branch is still instantiated and code-generated).
#if !defined(MSIM_AB) && !defined(MSIM_TRACE)
    {
        const bool whole_rows = ld_out % 8 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
        static std::atomic<int> cfg[4][kMaxDevices];
        if (whole_rows)
            rc = f16 ? go(msim::embed_head_kernel<true, false, false, false, true, false, false, true>, cfg[0], msim::kHeadFLds)
                     : go(msim::embed_head_kernel<false, false, false, false, true, false, false, true>, cfg[1], msim::kHeadFLds);
        else
            rc = f16 ? go(msim::embed_head_kernel<true, false, false, false, true>, cfg[2], msim::kHeadFLds)
                     : go(msim::embed_head_kernel<false, false, false, false, true>, cfg[3], msim::kHeadFLds);
        (void)go_half;
    }
#else
    {
        static std::atomic<int> configured[12][kMaxDevices];
        // MSIM_HEAD_VARIANT = bit 0: flag-synchronised weight ring instead of one s_barrier per K chunk; bit 1: hand-pipelined operand
        // fetch; bit 2: swapped MFMA roles + per-row epilogue; bit 3 (default): loader two weight chunks ahead, rings 3 + 3; bit 4: two half-size workgroups per CU; bit 5: DMA pieces between the
        // MFMAs; bit 6 (default): whole-row output stores through LDS
        // (tuning knob for A/B measurements, not part of the ABI; profiles/r02_logs/ab_head_variants.log)
        static const int variant = ab_env("MSIM_HEAD_VARIANT", 72) & 127;
        const bool epi2 = (variant & 4) && ld_out % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0 &&
                          (bias == nullptr || (reinterpret_cast<uintptr_t>(bias) & 3) == 0);   // 8-byte stores, 4-byte bias loads
        // bit 3: loader two weight chunks ahead (rings 3 + 3)
        // bit 6 (default): output rows staged through LDS and stored as whole rows; needs 16-byte aligned rows, else the 2-byte form
        const bool epi3 = (variant & 64) && ld_out % 8 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
        static const int pair_env = ab_env("MSIM_HEAD_PAIR", 0);       // round 3: chunks requested two at a time (rings 4 + 2, whole-row stores)
        if (pair_env && epi3) {
            static std::atomic<int> configured6[2][kMaxDevices];
            rc = f16 ? go(msim::embed_head_kernel<true, false, false, false, false, false, false, true, true>, configured6[0], msim::kHeadLds)
                     : go(msim::embed_head_kernel<false, false, false, false, false, false, false, true, true>, configured6[1], msim::kHeadLds);
        } else if (epi3) {
            static std::atomic<int> configured5[2][kMaxDevices];
            rc = f16 ? go(msim::embed_head_kernel<true, false, false, false, true, false, false, true>, configured5[0], msim::kHeadFLds)
                     : go(msim::embed_head_kernel<false, false, false, false, true, false, false, true>, configured5[1], msim::kHeadFLds);
        } else if (variant & 32) {           // bit 5: hidden-state DMA pieces issued between the k-steps' MFMAs (+ bit 3 rings, + bit 2 epilogue)
            static std::atomic<int> configured4[6][kMaxDevices];
            if ((variant & 8) && (variant & 4) && epi2)
                rc = f16 ? go(msim::embed_head_kernel<true, false, false, true, true, false, true>, configured4[0], msim::kHeadFLds)
                         : go(msim::embed_head_kernel<false, false, false, true, true, false, true>, configured4[1], msim::kHeadFLds);
            else if (variant & 8)
                rc = f16 ? go(msim::embed_head_kernel<true, false, false, false, true, false, true>, configured4[2], msim::kHeadFLds)
                         : go(msim::embed_head_kernel<false, false, false, false, true, false, true>, configured4[3], msim::kHeadFLds);
            else
                rc = f16 ? go(msim::embed_head_kernel<true, false, false, false, false, false, true>, configured4[4], msim::kHeadLds)
                         : go(msim::embed_head_kernel<false, false, false, false, false, false, true>, configured4[5], msim::kHeadLds);
        } else if (variant & 16) {           // bit 4: two half-size workgroups per CU
            static std::atomic<int> configured3[2][kMaxDevices];
            rc = f16 ? go_half(msim::embed_head_kernel<true, false, false, false, false, true>, configured3[0])
                     : go_half(msim::embed_head_kernel<false, false, false, false, false, true>, configured3[1]);
        } else if ((variant & 8) && (variant & 4) && epi2) {
            static std::atomic<int> configured2[2][kMaxDevices];
            rc = f16 ? go(msim::embed_head_kernel<true, false, false, true, true>, configured2[0], msim::kHeadFLds)
                     : go(msim::embed_head_kernel<false, false, false, true, true>, configured2[1], msim::kHeadFLds);
        } else if (variant & 8) {
            rc = f16 ? go(msim::embed_head_kernel<true, false, false, false, true>, configured[10], msim::kHeadFLds)
                     : go(msim::embed_head_kernel<false, false, false, false, true>, configured[11], msim::kHeadFLds);
        } else
        switch (epi2 ? 4 : (variant & 3)) {
            case 1: rc = f16 ? go(msim::embed_head_kernel<true, true, false>, configured[0], msim::kHeadFLds)
                             : go(msim::embed_head_kernel<false, true, false>, configured[1], msim::kHeadFLds); break;
            case 2: rc = f16 ? go(msim::embed_head_kernel<true, false, true>, configured[2], msim::kHeadLds)
                             : go(msim::embed_head_kernel<false, false, true>, configured[3], msim::kHeadLds); break;
            case 3: rc = f16 ? go(msim::embed_head_kernel<true, true, true>, configured[4], msim::kHeadFLds)
                             : go(msim::embed_head_kernel<false, true, true>, configured[5], msim::kHeadFLds); break;
            case 4: rc = f16 ? go(msim::embed_head_kernel<true, false, false, true>, configured[8], msim::kHeadLds)
                             : go(msim::embed_head_kernel<false, false, false, true>, configured[9], msim::kHeadLds); break;
            default: rc = f16 ? go(msim::embed_head_kernel<true, false, false>, configured[6], msim::kHeadLds)
                              : go(msim::embed_head_kernel<false, false, false>, configured[7], msim::kHeadLds); break;
        }
    }
#endif
    if (rc) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MSIM_ELAUNCH, "embed_head_kernel launch: %s", hipGetErrorString(e));
    return MSIM_OK;
}

// ---------------------------------------------------------------- plain similarity matrix entry point
int msim_sim_matrix(int dtype, const void *A, int n_a, const void *B, int n_b, int dim, float *out, int64_t ld_out,
                    uint32_t flags, void *stream) {
    if (n_a < 0 || n_b < 0) return fail(MSIM_EINVAL, "negative size (n_a=%d n_b=%d)", n_a, n_b);
    if (n_a == 0 || n_b == 0) return MSIM_OK;
    if (!out) return fail(MSIM_EINVAL, "null pointer argument");
    static const int32_t dummy_off[2] = {0, 0};
    if (int rc = check_smooth(A, B, dummy_off, dtype, dim, 1, 1.0f)) return rc;     // same row-layout contract as the generic kernels
    if (ld_out < n_b) return fail(MSIM_EINVAL, "ld_out=%lld < n_b=%d", (long long)ld_out, n_b);
    if (flags & ~(MSIM_FLAG_REF_ROUNDING)) return fail(MSIM_EINVAL, "unknown flags 0x%x", flags);
    const DeviceInfo *di = nullptr;
    if (int rc = device_info(&di)) return rc;
    msim::SimArgs a{ld_out, n_a, n_b, dim * elem_bytes(dtype), flags};
    const char *ac = static_cast<const char *>(A), *bc = static_cast<const char *>(B);
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case MSIM_DTYPE_F32: return sim_dispatch<msim::kDtypeF32>(ac, bc, out, a, *di, st);
        case MSIM_DTYPE_F16: return sim_dispatch<msim::kDtypeF16>(ac, bc, out, a, *di, st);
        default: return sim_dispatch<msim::kDtypeBf16>(ac, bc, out, a, *di, st);
    }
}

// ---------------------------------------------------------------- hierarchical token pooling
int msim_pool_cluster(int dtype, const void *E, const int32_t *d_off, int n_pages, int dim, int max_rows,
                      const int64_t *ws_off, int pool_factor, float *X_ws, double *D_ws, int32_t *labels,
                      int32_t *n_clusters, void *stream) {
    if (n_pages < 0 || max_rows < 0) return fail(MSIM_EINVAL, "negative size");
    if (n_pages == 0) return MSIM_OK;
    if (!E || !d_off || !ws_off || !X_ws || !D_ws || !labels || !n_clusters) return fail(MSIM_EINVAL, "null pointer argument");
    if (pool_factor < 1) return fail(MSIM_EINVAL, "pool_factor must be >= 1");
    static const int32_t dummy_off[2] = {0, 0};
    if (int rc = check_smooth(E, E, dummy_off, dtype, dim, 1, 1.0f)) return rc;      // row layout contract of the generic kernels
    if (max_rows > msim::kPoolMaxRows) return fail(MSIM_EUNSUPPORTED, "a page of %d rows: at most %d are supported", max_rows, msim::kPoolMaxRows);
    if (n_pages > 65535) return fail(MSIM_EUNSUPPORTED, "at most 65535 pages per call");
    hipStream_t st = static_cast<hipStream_t>(stream);
    msim::PoolArgs a{n_pages, dim, dim * elem_bytes(dtype), pool_factor};
    const char *e = static_cast<const char *>(E);
    if (max_rows > 0) {
        const dim3 ggrid((max_rows + 127) / 128, (max_rows + 31) / 32, n_pages);
        switch (dtype) {
            case MSIM_DTYPE_F32: hipLaunchKernelGGL(msim::pool_gram_kernel<msim::kDtypeF32>, ggrid, dim3(256), 0, st, e, d_off, ws_off, X_ws, a); break;
            case MSIM_DTYPE_F16: hipLaunchKernelGGL(msim::pool_gram_kernel<msim::kDtypeF16>, ggrid, dim3(256), 0, st, e, d_off, ws_off, X_ws, a); break;
            default: hipLaunchKernelGGL(msim::pool_gram_kernel<msim::kDtypeBf16>, ggrid, dim3(256), 0, st, e, d_off, ws_off, X_ws, a); break;
        }
        const int tiles = (max_rows + 15) / 16;
        hipLaunchKernelGGL(msim::pool_pdist_kernel, dim3(tiles, tiles, n_pages), dim3(256), 0, st, d_off, ws_off, X_ws, D_ws);
    }
    // pages of at most kPoolMaxN rows keep the clustering state in LDS; a call with a longer page runs the variant whose long pages
    // keep it in their (by then dead) region of X_ws
    if (max_rows > msim::kPoolMaxN) {
        static std::atomic<int> configured_big[kMaxDevices];
        if (int rc = allow_lds(msim::pool_cluster_kernel<true>, (int)sizeof(msim::PoolLds), configured_big)) return rc;
        hipLaunchKernelGGL(msim::pool_cluster_kernel<true>, dim3(n_pages), dim3(msim::kPoolThreads), sizeof(msim::PoolLds), st, d_off, ws_off,
                           X_ws, D_ws, labels, n_clusters, pool_factor);
    } else {
        static std::atomic<int> configured[kMaxDevices];
        if (int rc = allow_lds(msim::pool_cluster_kernel<false>, (int)sizeof(msim::PoolLds), configured)) return rc;
        hipLaunchKernelGGL(msim::pool_cluster_kernel<false>, dim3(n_pages), dim3(msim::kPoolThreads), sizeof(msim::PoolLds), st, d_off, ws_off,
                           X_ws, D_ws, labels, n_clusters, pool_factor);
    }
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(MSIM_ELAUNCH, "token pooling launch: %s", hipGetErrorString(err));
    return MSIM_OK;
}

int msim_pool_reduce(int dtype, const void *E, const int32_t *d_off, int n_pages, int dim, int ld_in, const int32_t *labels,
                     const int32_t *out_off, void *out, int ld_out, void *stream) {
    if (n_pages < 0) return fail(MSIM_EINVAL, "negative size");
    if (n_pages == 0) return MSIM_OK;
    if (!E || !d_off || !labels || !out_off || !out) return fail(MSIM_EINVAL, "null pointer argument");
    if (dtype != MSIM_DTYPE_BF16 && dtype != MSIM_DTYPE_F16 && dtype != MSIM_DTYPE_F32) return fail(MSIM_EUNSUPPORTED, "dtype code %d", dtype);
    if (dim <= 0 || dim > 2048 || ld_in < dim || ld_out < dim) return fail(MSIM_EUNSUPPORTED, "dim=%d (ld_in=%d ld_out=%d)", dim, ld_in, ld_out);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const char *e = static_cast<const char *>(E);
    char *o = static_cast<char *>(out);
    const int es = elem_bytes(dtype);
    switch (dtype) {
        case MSIM_DTYPE_F32: hipLaunchKernelGGL(msim::pool_reduce_kernel<msim::kDtypeF32>, dim3(n_pages), dim3(256), 0, st, e, d_off, labels, out_off, o, dim, ld_in * es, ld_out * es); break;
        case MSIM_DTYPE_F16: hipLaunchKernelGGL(msim::pool_reduce_kernel<msim::kDtypeF16>, dim3(n_pages), dim3(256), 0, st, e, d_off, labels, out_off, o, dim, ld_in * es, ld_out * es); break;
        default: hipLaunchKernelGGL(msim::pool_reduce_kernel<msim::kDtypeBf16>, dim3(n_pages), dim3(256), 0, st, e, d_off, labels, out_off, o, dim, ld_in * es, ld_out * es); break;
    }
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(MSIM_ELAUNCH, "pool_reduce_kernel launch: %s", hipGetErrorString(err));
    return MSIM_OK;
}

int msim_host_gather(void *dst, const void *const *src, const int64_t *dst_off, const int64_t *nbytes, int64_t n, int n_threads) {
    if (n < 0) return fail(MSIM_EINVAL, "negative count");
    if (n == 0) return MSIM_OK;
    if (!dst || !src || !dst_off || !nbytes) return fail(MSIM_EINVAL, "null pointer argument");
    int64_t total = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (nbytes[i] < 0 || dst_off[i] < 0 || (nbytes[i] > 0 && !src[i])) return fail(MSIM_EINVAL, "bad buffer %lld", (long long)i);
        total += nbytes[i];
    }
    char *d = static_cast<char *>(dst);
    auto run = [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i)
            if (nbytes[i]) memcpy(d + dst_off[i], src[i], (size_t)nbytes[i]);
    };
    int nt = n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads);
    if (total < (int64_t)(4 << 20) * nt) nt = (int)(total >> 22) < 1 ? 1 : (int)(total >> 22);   // below ~4 MiB per thread: fewer
    if (nt <= 1 || n < 2) {
        run(0, n);
        return MSIM_OK;
    }
    std::vector<std::thread> pool;
    pool.reserve(nt);
    const int64_t per = (total + nt - 1) / nt;
    int64_t lo = 0, acc = 0;
    for (int64_t i = 0; i < n; ++i) {
        acc += nbytes[i];
        if (acc >= per || i + 1 == n) {
            pool.emplace_back(run, lo, i + 1);
            lo = i + 1;
            acc = 0;
        }
    }
    for (auto &t : pool) t.join();
    return MSIM_OK;
}

}  // extern "C"

namespace {
inline bool row_is_zero(const char *p, int64_t row_bytes) {
    int64_t i = 0;
    uint64_t acc = 0;
    for (; i + 8 <= row_bytes; i += 8) {
        uint64_t v;
        memcpy(&v, p + i, 8);
        acc |= v;
    }
    for (; i < row_bytes; ++i) acc |= (unsigned char)p[i];
    return acc == 0;
}

template <class F>
void host_parallel(int64_t n, int n_threads, int64_t work_bytes, F &&body) {
    int nt = n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads);
    if (work_bytes < (int64_t)(4 << 20) * nt) nt = (int)(work_bytes >> 22) < 1 ? 1 : (int)(work_bytes >> 22);
    if (nt <= 1 || n < 2) {
        body(0, n);
        return;
    }
    std::vector<std::thread> pool;
    pool.reserve(nt);
    const int64_t per = (n + nt - 1) / nt;
    for (int64_t lo = 0; lo < n; lo += per) pool.emplace_back(body, lo, lo + per < n ? lo + per : n);
    for (auto &t : pool) t.join();
}
}  // namespace

extern "C" {

int msim_host_count_nonzero_rows(const void *const *src, const int64_t *rows, int64_t row_bytes, int64_t n, int32_t *counts,
                                 int n_threads) {
    if (n < 0 || row_bytes <= 0) return fail(MSIM_EINVAL, "bad size");
    if (n == 0) return MSIM_OK;
    if (!src || !rows || !counts) return fail(MSIM_EINVAL, "null pointer argument");
    int64_t total = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (rows[i] < 0 || rows[i] > 0x7fffffff || (rows[i] > 0 && !src[i])) return fail(MSIM_EINVAL, "bad buffer %lld", (long long)i);
        total += rows[i] * row_bytes;
    }
    host_parallel(n, n_threads, total, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            const char *p = static_cast<const char *>(src[i]);
            int32_t c = 0;
            for (int64_t r = 0; r < rows[i]; ++r) c += row_is_zero(p + r * row_bytes, row_bytes) ? 0 : 1;
            counts[i] = c;
        }
    });
    return MSIM_OK;
}

int msim_host_gather_nonzero_rows(void *dst, const void *const *src, const int64_t *rows, int64_t row_bytes, const int64_t *dst_row,
                                  int64_t n, int n_threads) {
    if (n < 0 || row_bytes <= 0) return fail(MSIM_EINVAL, "bad size");
    if (n == 0) return MSIM_OK;
    if (!dst || !src || !rows || !dst_row) return fail(MSIM_EINVAL, "null pointer argument");
    int64_t total = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (rows[i] < 0 || dst_row[i] < 0 || (rows[i] > 0 && !src[i])) return fail(MSIM_EINVAL, "bad buffer %lld", (long long)i);
        total += rows[i] * row_bytes;
    }
    char *d = static_cast<char *>(dst);
    host_parallel(n, n_threads, total, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            const char *p = static_cast<const char *>(src[i]);
            char *o = d + dst_row[i] * row_bytes;
            for (int64_t r = 0; r < rows[i]; ++r) {
                if (row_is_zero(p + r * row_bytes, row_bytes)) continue;
                memcpy(o, p + r * row_bytes, (size_t)row_bytes);
                o += row_bytes;
            }
        }
    });
    return MSIM_OK;
}

int msim_query_compact(const void *box, int n_q, int Lq, int row_bytes, const int32_t *q_off, int32_t *counts, void *out,
                       void *stream) {
    if (n_q < 0 || Lq < 0 || row_bytes <= 0 || (row_bytes & 15)) return fail(MSIM_EINVAL, "bad size (row bytes must be a multiple of 16)");
    if (n_q == 0 || Lq == 0) return MSIM_OK;
    if (!box || (!counts && !(q_off && out))) return fail(MSIM_EINVAL, "null pointer argument");
    if (Lq > msim::kCompactMaxRows) return fail(MSIM_EUNSUPPORTED, "query boxes of more than %d rows are not compacted", msim::kCompactMaxRows);
    if (reinterpret_cast<uintptr_t>(box) & 15 || reinterpret_cast<uintptr_t>(out) & 15) return fail(MSIM_EINVAL, "buffers must be 16-byte aligned");
    hipLaunchKernelGGL(msim::query_compact_kernel, dim3(n_q), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const char *>(box), Lq, row_bytes, q_off, counts, static_cast<char *>(out));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MSIM_ELAUNCH, "query_compact_kernel launch: %s", hipGetErrorString(e));
    return MSIM_OK;
}

// ---------------------------------------------------------------- top-k selection
// Level plan: level 0 splits each row into segments of `seg0` candidates (a power of two chosen so that the
// launch has enough workgroups to fill the chip even for a single row); later levels use full segments.
static inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

// level 0 as the streaming threshold filter (topk_select.hip: topk_filter_kernel): rows of raw scores long enough for it, few
// enough winners per row, and enough (row, segment) workgroups to fill the chip -- the many-query regime, where the bitonic level
// was the one kernel of the timed step two orders off its roof
static bool topk_use_filter(const int64_t *ids, int n_q, long long n, int k) {
    if (ids != nullptr || k > msim::kTopkFilterMaxK || n < (long long)msim::kTopkFilterSeg) return false;
    return (long long)n_q * ((n + msim::kTopkFilterSeg - 1) / msim::kTopkFilterSeg) >= 512;      // two workgroups per CU at least
}

static int topk_first_segment(int n_q, long long n, int k) {
    int seg = 512;
    while (seg < 4 * k) seg <<= 1;                       // every level must shrink its input at least 4x
    while (seg < msim::kTopkSeg && (long long)n_q * ((n + seg - 1) / seg) > 1024) seg <<= 1;   // ~4 workgroups per CU is plenty
    return seg;
}
static inline long long topk_level_out(long long n, int seg, int k) { return ((n + seg - 1) / seg) * (long long)k; }
// Later levels: the smallest legal segment.  A bitonic sort of s candidates costs ~log2(s)^2 / 2 barrier-separated stages of s / 512
// passes each, so two levels of 512 (45 stages + a tiny final sort) beat one 4096-candidate sort (78 stages x 8 passes) several
// times over -- the single-workgroup last level was 157 us of a 200 us top-k at 4 queries x 125 000 documents.
static int topk_later_segment(int k) {
    int seg = 512;
    while (seg < 4 * k) seg <<= 1;
    return seg;
}
// one more workgroup-per-segment level only while the row is longer than two segments; otherwise one workgroup finishes the row
static inline bool topk_is_last(long long n, int seg) { return n <= 2LL * seg && n <= msim::kTopkSeg; }

static size_t topk_plan_bytes(int n_q, long long n, int k, int seg0) {
    if (topk_is_last(n, seg0)) return 0;
    const int seg1 = topk_later_segment(k);
    const long long na = topk_level_out(n, seg0, k);
    const long long nb = topk_is_last(na, seg1) ? 0 : topk_level_out(na, seg1, k);
    return align16((size_t)n_q * na * 4) + align16((size_t)n_q * na * 8) + align16((size_t)n_q * nb * 4) +
           align16((size_t)n_q * nb * 8);
}

size_t msim_topk_workspace_bytes(int n_q, int64_t n, int k) {
    if (n_q <= 0 || k <= 0 || k > msim::kTopkMaxK) return 0;
    // the same problem takes the filter level without explicit ids and the plain level with them: room for either
    size_t need = topk_plan_bytes(n_q, n, k, topk_first_segment(n_q, n, k));
    if (topk_use_filter(nullptr, n_q, n, k)) {
        const size_t f = topk_plan_bytes(n_q, n, k, msim::kTopkFilterSeg);
        if (f > need) need = f;
    }
    return need;
}

int msim_topk_f32(const float *scores, const int64_t *ids, int n_q, int64_t n, int64_t ld, int k, int64_t id_base,
                  float *out_scores, int64_t *out_ids, void *workspace, void *stream) {
    if (n_q < 0 || n < 0 || k <= 0) return fail(MSIM_EINVAL, "bad size (n_q=%d n=%lld k=%d)", n_q, (long long)n, k);
    if (n_q == 0) return MSIM_OK;
    if (!out_scores || !out_ids || (n > 0 && !scores)) return fail(MSIM_EINVAL, "null pointer argument");
    if (k > msim::kTopkMaxK) return fail(MSIM_EUNSUPPORTED, "k=%d > %d", k, msim::kTopkMaxK);
    if (ld < n) return fail(MSIM_EINVAL, "ld=%lld < n=%lld", (long long)ld, (long long)n);
    const bool filter0 = topk_use_filter(ids, n_q, n, k);
    const int seg0 = filter0 ? msim::kTopkFilterSeg : topk_first_segment(n_q, n, k);
    const int seg1 = topk_later_segment(k);
    if (!topk_is_last(n, seg0) && !workspace) return fail(MSIM_EINVAL, "workspace required (msim_topk_workspace_bytes)");
    hipStream_t st = static_cast<hipStream_t>(stream);

    const long long na = topk_is_last(n, seg0) ? 0 : topk_level_out(n, seg0, k);
    const long long nb = (na == 0 || topk_is_last(na, seg1)) ? 0 : topk_level_out(na, seg1, k);
    char *w = static_cast<char *>(workspace);
    float *bufs_s[2];
    int64_t *bufs_i[2];
    long long bufs_ld[2] = {na, nb};
    bufs_s[0] = reinterpret_cast<float *>(w);
    w += align16((size_t)n_q * na * 4);
    bufs_i[0] = reinterpret_cast<int64_t *>(w);
    w += align16((size_t)n_q * na * 8);
    bufs_s[1] = reinterpret_cast<float *>(w);
    w += align16((size_t)n_q * nb * 4);
    bufs_i[1] = reinterpret_cast<int64_t *>(w);

    const float *in_s = scores;
    const int64_t *in_i = ids;
    long long in_n = n, in_ld = ld, in_base = id_base;
    int which = 0, seg = seg0;
    for (;;) {
        const bool last = topk_is_last(in_n, seg);
        const long long segs = last ? 1 : (in_n + seg - 1) / seg;
        float *o_s = last ? out_scores : bufs_s[which];
        int64_t *o_i = last ? out_ids : bufs_i[which];
        const long long o_ld = last ? k : segs * k;
        if (!last && o_ld > bufs_ld[which]) return fail(MSIM_ELAUNCH, "internal: top-k level does not fit its buffer");
        for (int r0 = 0; r0 < n_q; r0 += 65535) {   // grid.y limit
            const int rows = (n_q - r0 < 65535) ? (n_q - r0) : 65535;
            if (filter0 && in_s == scores) {        // level 0: the streaming filter
                hipLaunchKernelGGL(msim::topk_filter_kernel, dim3((unsigned)segs, (unsigned)rows), dim3(msim::kTopkThreads), 0, st,
                                   in_s + (size_t)r0 * in_ld, in_n, in_ld, in_base, k, o_s + (size_t)r0 * o_ld, o_i + (size_t)r0 * o_ld, o_ld);
                continue;
            }
            hipLaunchKernelGGL(msim::topk_segment_kernel, dim3((unsigned)segs, (unsigned)rows), dim3(msim::kTopkThreads), 0, st,
                               in_s + (size_t)r0 * in_ld, in_i ? in_i + (size_t)r0 * in_ld : nullptr, in_n, in_ld, in_base, k,
                               last ? msim::kTopkSeg : seg, o_s + (size_t)r0 * o_ld, o_i + (size_t)r0 * o_ld, o_ld);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(MSIM_ELAUNCH, "topk_segment_kernel launch: %s", hipGetErrorString(e));
        if (last) break;
        in_s = o_s;
        in_i = o_i;
        in_n = o_ld;
        in_ld = o_ld;
        in_base = 0;
        which ^= 1;
        seg = seg1;
    }
    return MSIM_OK;
}

}  // extern "C"
