// K1bK -- K1b's eight-wave form for SHORT documents (pooled pages, 64-row documents): several documents per chunk.
// STATUS (round 6): built, bit-identical to K1b (tests/test_gpu_short_docs.py against a measurement build with MSIM_BATCH_PACKED=1),
// and 4-9 % SLOWER than K1b at 64 / 343 rows -- compiled into measurement builds only (make ab / make trace), never shipped.
// Its phase trace (tools/trace_batch.py, profiles/r06_logs/ab_short_docs_packed.log): the chunk barrier it saves is worth ~350 cycles
// per document; a document end costs ~1000 (lane groups combined on the VALU), the requests of a full chunk 1200-1600.
// Same arithmetic and the same bits as K1b (colpali_engine/utils/processing_utils.py:179; the maxima are exact whatever the order,
// the token sums run in reduce_query_tokens' order).
//
// K1b gives every document chunks of its own: a 64-row document fills half of a 128-row chunk, and the chunk's fixed cost -- the
// barrier, the four LDS-DMA instructions per wave (half of them fetching rows that do not exist), the counted wait -- is paid per
// 128 MFMAs of a wave instead of per 256 (tools/trace_batch.py at 64 rows: ~1100 of 6500 cycles per document).  Here the documents
// of a range are one sequence of 32-row SLABS (a document = ceil(rows / 32) of them, its last one padded with zeros by the
// descriptor's bounds check and masked), and a chunk is the next four slabs of that sequence whatever documents they belong to:
//   * producer: wave w fills half (w & 1) of slab slot (w >> 1) of every chunk, i.e. it walks the slab sequence with a cursor of
//     its own in steps of four (descriptor of the slot's document, rebuilt per chunk from scalar values);
//   * consumer: every wave walks the whole sequence; where a document ends it combines its four lane groups' maxima on the VALU
//     (v_permlane16/32_swap) and writes ONE float per token to a table -- up to four documents end in a chunk, and a table is read
//     behind the next barrier, so there are 2 x 4 tables of 4 B per token (K1b: 2 tables of 16 B per token);
//   * the row offsets are read through a 64-document window held in the lanes of one register (v_readlane with a scalar index:
//     no scalar-memory round trip per document, neither in the producer nor in the consumer).
#pragma once
#include "maxsim_batch.hip"
#include "maxsim_batch_t.hip"

namespace msim {

constexpr int kPackMaxEnds = kChunkSlabs;            // documents that can end in one chunk
constexpr int kPackWindow = 64;                      // documents in the offset window

// tokens [s, e) of a 4-byte-per-token table, added in reduce_query_tokens' order: lane i of the query's 8 lanes adds tokens s+i,
// s+i+8, ... in that order (four loads in flight; a token past the end adds +0.0, which changes no bit of a sum that starts at
// +0.0), then xor 4, 2, 1
template <bool F16>
__device__ __forceinline__ float reduce_query_tokens4(const char *tab, int s, int e, int i, bool clamp, bool ref_round) {
    float acc = 0.0f;
    for (int t = s + i; t < e; t += 32) {
        float x[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = *reinterpret_cast<const float *>(tab + ((t + 8 * k < e ? t + 8 * k : s) << 2));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = x[k];
            if (clamp) v = fmaxf(v, 0.0f);
            if (ref_round) v = round_to_input<F16>(v);
            if (t + 8 * k < e) acc += v;
        }
    }
    acc += __shfl_xor(acc, 4);
    acc += __shfl_xor(acc, 2);
    acc += __shfl_xor(acc, 1);
    return acc;
}

template <bool F16, int AUX = 0, int MAXU = 8>
__global__ __launch_bounds__(512, 2) void maxsim_batch_packed_kernel(const uint16_t *__restrict__ Qt, const uint16_t *__restrict__ D,
                                                                     const int32_t *__restrict__ d_off,
                                                                     const uint8_t *__restrict__ clamp0, float *__restrict__ scores,
                                                                     BatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = 8;
    constexpr int kRing = 3;
    static_assert(MAXU == 8 || MAXU == 10, "eight or ten units per wave");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int kTab4Bytes = NW * MAXU * kUnitTok * 4;            // one table: a float per token of the block
    char *const tokmax = smem + kRing * kChunkBytes;                // 2 generations x kPackMaxEnds tables
    int *const rtab = reinterpret_cast<int *>(tokmax + 2 * kPackMaxEnds * kTab4Bytes);

    // ---- (query block, document range) of this workgroup: as in K1b
    const int sub = a.n_ranges >> 3;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int qblock, range;
    if (sub > 1) { qblock = slot % a.n_qblocks; range = xcd * sub + slot / a.n_qblocks; }
    else         { qblock = slot;               range = xcd; }
    if (qblock >= a.n_qblocks || range >= a.n_ranges) return;
    const long long row0 = d_off[0], total_rows = (long long)d_off[a.n_d] - row0;
    const int want_lo = (int)(row0 + (total_rows * range) / a.n_ranges), want_hi = (int)(row0 + (total_rows * (range + 1)) / a.n_ranges);
    const int d_lo = lower_bound_wave(a.n_d, want_lo, lane, [&](int k) { return d_off[k]; });
    const int d_hi = (range + 1 == a.n_ranges) ? a.n_d : lower_bound_wave(a.n_d, want_hi, lane, [&](int k) { return d_off[k]; });
    int *const my_prog = a.convoy ? a.convoy + (size_t)range * a.n_qblocks : nullptr;
    if (d_lo >= d_hi) {
        if (my_prog && threadIdx.x == 0) __hip_atomic_store(my_prog + qblock, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    constexpr int kConvoyWindow = 24;
    bool convoy_on = my_prog != nullptr;
    int g_chunk = 0;

    // ---- the offset window: lane j holds d_off[wbase + j] (and the clamp0 byte of document wbase + j)
    int wbase = d_lo;
    int win = 0;
    unsigned cwin = 0;
    auto window_load = [&]() {
        const int k = wbase + lane;
        win = d_off[k < a.n_d ? k : a.n_d];
        if (clamp0 != nullptr) cwin = clamp0[k < a.n_d ? k : a.n_d - 1];
        asm volatile("" : "+v"(win), "+v"(cwin));      // the compiler's wait for these loads sits HERE, not at the uses inside the chunk loop
    };
    window_load();
    auto off_at = [&](int i) -> int {
        const unsigned j = (unsigned)(i - wbase);
        return j < (unsigned)kPackWindow ? __builtin_amdgcn_readlane(win, (int)j) : d_off[i];
    };
    auto clamp_at = [&](int i) -> bool {
        if (clamp0 == nullptr) return false;
        const unsigned j = (unsigned)(i - wbase);
        if (j < (unsigned)kPackWindow) return __builtin_amdgcn_readlane((int)cwin, (int)j) != 0;
        const uint64_t addr = reinterpret_cast<uint64_t>(clamp0) + (uint64_t)i;
        return ((scalar_load_u32(addr & ~3ull) >> ((addr & 3) * 8)) & 0xffu) != 0;
    };

    // ---- this block's tokens (K1b's dealing: block-local unit u lives in wave u % NW)
    const int qb0 = a.blk_q0[qblock];
    const int qb_n = a.blk_q0[qblock + 1] - qb0;
    const int tok0 = flat_qoff(a.fq, qb0);
    const int n_tok = flat_qoff(a.fq, qb0 + qb_n) - tok0;
    const int n_units = (n_tok + kUnitTok - 1) / kUnitTok;
    const int my_nu = wave < n_units ? (n_units - 1 - wave) / NW + 1 : 0;
    QueryUnit qu[MAXU];
    const int l16 = lane & 15, l4 = lane >> 4;
    int src_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) src_off[j] = l4 * kRowBytes + (((l16 ^ l4) ^ (j << 2)) << 4);
    int rd_off[2][kKSteps16];
    slab_rd_offsets16(lane, rd_off);
    const int my_lds_off = (wave >> 1) * kSlabBytes + (wave & 1) * 4096;

    // ---- producer: this wave's cursor over the slab sequence -- slab p_sl of document p_idx is the next one of slot (wave >> 1)
    int p_idx = d_lo, p_sl = wave >> 1, p_r0 = 0, p_len = 0;
    auto p_norm = [&]() {
        while (p_idx < d_hi) {
            const int r0 = off_at(p_idx), len = off_at(p_idx + 1) - r0;
            const int ns = (len + kSlabRows - 1) / kSlabRows;
            if (p_sl < ns) { p_r0 = r0; p_len = len; return; }
            p_sl -= ns;
            ++p_idx;
        }
    };
    p_norm();
    int p_slot = 0;
    auto produce = [&]() {
        if (p_idx >= d_hi) return;
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void *)(D + (size_t)p_r0 * kDim), 0, p_len * kRowBytes, 0x00020000);
        char *dst = smem + p_slot * kChunkBytes + my_lds_off;
        const int soff = (p_sl * kSlabRows + (wave & 1) * 16) * kRowBytes;      // rows past the document end read as zeros
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, MSIM_LDS(dst + j * 1024), 16, src_off[j], soff + j * 1024, 0, AUX);
        p_sl += kChunkSlabs;
        p_norm();
    };
    auto produce_step = [&]() {              // every wave steps its ring slot per chunk, whether or not it had a slab to fetch
        produce();
        p_slot = (p_slot + 1 == kRing) ? 0 : p_slot + 1;
    };
#pragma unroll
    for (int i = 0; i < kRing - 1; ++i) produce_step();

#pragma unroll
    for (int t = 0; t < MAXU; ++t)
        load_query_unit(qu[t], Qt + (size_t)tok0 * kDim, (wave + NW * t) * kUnitTok, n_tok, lane, t < my_nu);
    {
        const int rq = threadIdx.x >> 3;
        if (rq < qb_n) {
            const int s = flat_qoff(a.fq, qb0 + rq) - tok0, e = flat_qoff(a.fq, qb0 + rq + 1) - tok0;
            rtab[2 * rq] = s;
            rtab[2 * rq + 1] = e;
        }
    }
    wait_vmcnt<0>();
#pragma unroll
    for (int t = 0; t < MAXU; ++t)
#pragma unroll
        for (int ks = 0; ks < kKSteps16; ++ks) {
            if (MAXU > 8 && t >= 2) asm volatile("" : "+a"(qu[t].f[ks]));
            else asm volatile("" : "+v"(qu[t].f[ks]));
        }

    const bool ref_bf16 = (a.flags & kFlagRefBf16) != 0;
    const bool round_total = ref_bf16 && !(a.flags & kFlagPartial);
    int c_slot = 0;
    const bool tracing = kTraceBuild && a.trace != nullptr && blockIdx.x == 0;      // `make trace` builds only (tools/trace_batch.py)
    unsigned long long tr[7] = {0, 0, 0, 0, 0, 0, 0};     // vmcnt + convoy, barrier, DMA issue, token sums, slabs, document ends, chunks

    auto reduce_doc = [&](int doc, bool clamp, int tab) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int rq = tid >> 3, ri = tid & 7;
        if (rq < qb_n) {
            float tot = reduce_query_tokens4<F16>(tokmax + tab * kTab4Bytes, rtab[2 * rq], rtab[2 * rq + 1], ri, clamp, ref_bf16);
            if (round_total) tot = round_to_input<F16>(tot);
            if (ri == 0) scores[(size_t)(qb0 + rq) * a.ld + doc] = tot;
        }
    };
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    auto run = [&](auto nu_c) {
    constexpr int NU = decltype(nu_c)::value;
    constexpr bool wave_has_units = NU > 0;
    constexpr int NUA = NU > 0 ? NU : 1;
    // consumer cursor: slab c_sl of document c_idx (non-empty, or c_idx == d_hi); documents without rows are scored on the way
    int c_idx = d_lo, c_sl = 0, c_len = 0;
    auto c_open = [&]() {
        while (c_idx < d_hi) {
            c_len = off_at(c_idx + 1) - off_at(c_idx);
            if (c_len > 0) return;
            const bool cl = clamp_at(c_idx);
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            const int rq = tid >> 3;
            if (rq < qb_n && (tid & 7) == 0)
                scores[(size_t)(qb0 + rq) * a.ld + c_idx] = (rtab[2 * rq + 1] > rtab[2 * rq] && !cl) ? -INFINITY : 0.0f;
            ++c_idx;
        }
    };
    c_open();
    // documents whose maxima wait in the tables of generation gen ^ 1
    int pend_n = 0, pend_doc[kPackMaxEnds] = {0, 0, 0, 0};
    unsigned pend_clamp = 0;
    int gen = 0;
    float m[NUA];
#pragma unroll
    for (int t = 0; t < NUA; ++t) m[t] = -INFINITY;

    while (c_idx < d_hi) {
        // keep the window around both cursors: the producers run at most (ring - 1) chunks = 8 + 3 slabs ahead of the consumer
        if (c_idx - wbase >= kPackWindow / 2) {
            wbase = c_idx;
            window_load();
        }
        const unsigned long long t0 = tracing ? __builtin_amdgcn_s_memtime() : 0;
        if (p_idx < d_hi) wait_vmcnt<4 * (kRing - 2)>(); else wait_vmcnt<0>();
        if (convoy_on && wave == 0 && (g_chunk & (kConvoyEvery - 1)) == 0) {
            if (lane == 0) __hip_atomic_store(my_prog + qblock, g_chunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spin = 0;
            for (; spin < kConvoySpins; ++spin) {
                int v = lane < a.n_qblocks ? __hip_atomic_load(my_prog + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x7fffffff;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
                if (g_chunk - v <= kConvoyWindow) break;
                __builtin_amdgcn_s_sleep(8);
            }
            if (spin == kConvoySpins) convoy_on = false;
        }
        ++g_chunk;
        const unsigned long long t1 = tracing ? __builtin_amdgcn_s_memtime() : 0;
        lds_barrier();
        const unsigned long long t2 = tracing ? __builtin_amdgcn_s_memtime() : 0;
        produce_step();
        const unsigned long long t3 = tracing ? __builtin_amdgcn_s_memtime() : 0;
#pragma unroll
        for (int e = 0; e < kPackMaxEnds; ++e)
            if (e < pend_n) reduce_doc(pend_doc[e], (pend_clamp >> e) & 1u, (gen ^ 1) * kPackMaxEnds + e);
        const unsigned long long t4 = tracing ? __builtin_amdgcn_s_memtime() : 0;
        unsigned long long t_ends = 0;

        const int cbuf = c_slot * kChunkBytes;
        c_slot = (c_slot + 1 == kRing) ? 0 : c_slot + 1;
        int cur_n = 0, cur_doc[kPackMaxEnds] = {0, 0, 0, 0};
        unsigned cur_clamp = 0;
#pragma unroll 1
        for (int sl = 0; sl < kChunkSlabs && c_idx < d_hi; ++sl) {
            const int rows_left = c_len - c_sl * kSlabRows;        // >= 1
            if constexpr (wave_has_units) {
                const int src_lds = cbuf + sl * kSlabBytes;
                bf16x8 af[2][kKSteps16];
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int ks = 0; ks < kKSteps16; ++ks) af[g][ks] = *reinterpret_cast<const bf16x8 *>(smem + src_lds + rd_off[g][ks]);
                if (rows_left >= kSlabRows) slab_units<F16, NUA, false, false>(m, af, qu, kSlabRows, lane, [](int) {});
                else slab_units<F16, NUA, true, false>(m, af, qu, rows_left, lane, [](int) {});
            }
            ++c_sl;
            if (rows_left <= kSlabRows) {                          // the document ends with this slab
                const unsigned long long te0 = tracing ? __builtin_amdgcn_s_memtime() : 0;
                if constexpr (wave_has_units) {
                    char *tab = tokmax + (gen * kPackMaxEnds + cur_n) * kTab4Bytes;
#pragma unroll
                    for (int t = 0; t < NUA; ++t) m[t] = group_max4(m[t]);
                    if (lane < kUnitTok) {
#pragma unroll
                        for (int t = 0; t < NUA; ++t) *reinterpret_cast<float *>(tab + (((wave + NW * t) * kUnitTok + lane) << 2)) = m[t];
                    }
#pragma unroll
                    for (int t = 0; t < NUA; ++t) m[t] = -INFINITY;
                }
                const unsigned cl = clamp_at(c_idx) ? 1u : 0u;
#pragma unroll
                for (int e = 0; e < kPackMaxEnds; ++e) cur_doc[e] = (cur_n == e) ? c_idx : cur_doc[e];
                cur_clamp |= cl << cur_n;
                ++cur_n;
                ++c_idx;
                c_sl = 0;
                c_open();
                if (tracing) t_ends += __builtin_amdgcn_s_memtime() - te0;
            }
        }
        if (tracing) {
            if constexpr (wave_has_units) {
                float sink = 0.f;
#pragma unroll
                for (int t = 0; t < NUA; ++t) sink += m[t];
                asm volatile("" ::"v"(sink));
            }
            const unsigned long long t5 = __builtin_amdgcn_s_memtime();
            tr[0] += t1 - t0; tr[1] += t2 - t1; tr[2] += t3 - t2; tr[3] += t4 - t3; tr[4] += t5 - t4 - t_ends; tr[5] += t_ends; tr[6] += 1;
        }
        pend_n = cur_n;
        pend_clamp = cur_clamp;
#pragma unroll
        for (int e = 0; e < kPackMaxEnds; ++e) pend_doc[e] = cur_doc[e];
        gen ^= 1;
    }
    if (pend_n > 0) {
        lds_barrier();
#pragma unroll
        for (int e = 0; e < kPackMaxEnds; ++e)
            if (e < pend_n) reduce_doc(pend_doc[e], (pend_clamp >> e) & 1u, (gen ^ 1) * kPackMaxEnds + e);
    }
    if (tracing && lane == 0) {
#pragma unroll
        for (int i = 0; i < 7; ++i) a.trace[wave * 8 + i] = tr[i];
    }
    if (my_prog && threadIdx.x == 0)
        __hip_atomic_store(my_prog + qblock, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };   // run

    switch (my_nu) {
        case 0: run(std::integral_constant<int, 0>{}); break;
        case 1: run(std::integral_constant<int, 1>{}); break;
        case 2: run(std::integral_constant<int, 2>{}); break;
        case 3: run(std::integral_constant<int, 3>{}); break;
        case 4: run(std::integral_constant<int, 4>{}); break;
        case 5: run(std::integral_constant<int, 5>{}); break;
        case 6: run(std::integral_constant<int, 6>{}); break;
        case 7: run(std::integral_constant<int, 7>{}); break;
        case 8: run(std::integral_constant<int, 8>{}); break;
        case 9: if constexpr (MAXU >= 9) run(std::integral_constant<int, 9>{}); break;
        default: if constexpr (MAXU >= 10) run(std::integral_constant<int, 10>{}); break;
    }
}

}  // namespace msim
