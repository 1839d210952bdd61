// K1b -- "batch" MaxSim kernel for gfx950 (MI355X): more than 8 units (of 16 query tokens) or more than 8 queries in total, i.e. the
// ridge and the MFMA-bound regime.  Same arithmetic as K1s (colpali_engine/utils/processing_utils.py:179 and
// colpali_engine/loss/late_interaction_losses.py:297-298), different blocking.
//
// Structure
//   * the queries are ONE flat token matrix (maxsim_common.hpp: the flat layout); a query BLOCK is a run of whole queries (the host
//     plan, maxsim_abi.hip: flat_plan, cuts them: <= NW * MAXU units of tokens, <= NW * 8 queries) and its 16-token units are dealt to
//     the waves round-robin, whatever queries they belong to; every wave keeps its units (16 VGPRs each, up to 8; 10 with eight of them
//     in AGPRs) in registers as MFMA B operands and runs the loop body compiled for the number it actually holds;
//   * NW waves share ONE document stream through an LDS ring:
//       NW = 2 ("pair", <= 16 / 20 units):  four pairs per CU -- two waves per SIMD at <= 256 registers; one 32-row slab per barrier,
//               and the barrier spans two waves only;
//       NW = 4 (<= 32 / 40 units):          two workgroups per CU, 64-row chunks: one computes while the other sits at its barrier;
//       NW = 8 (more; several blocks):      one workgroup per CU, 128-row chunks -- the fewest passes over the corpus;
//   * a chunk = NW/2 slabs; each wave issues 4 of the chunk's LDS-DMA wave-instructions (buffer_load_dwordx4 ... lds,
//     per-document bounds-checked descriptor, XOR-swizzled source); ONE raw s_barrier per chunk, LDS-DMA stays in flight across it
//     (counted vmcnt, never 0);
//   * per slab a wave reads the 8 operand fragments ONCE (ds_read_b128, conflict free), then runs 8 MFMAs per unit and folds the
//     16 -> 1 max; per-token running max in registers.  (A first version processed the units in two passes and re-read the fragments
//     for the second: the one-pass body is 1-5 % faster everywhere -- this regime is POWER-bound on real data, see DESIGN.md, and an
//     LDS read costs energy.)
//   * end of a document: every wave writes its per-token maxima to the workgroup's token table in LDS; behind the next chunk barrier
//     (the one that exists anyway) 8 lanes per query add that query's tokens -- in an order fixed by the query's length alone -- and
//     one lane stores the score;
//   * grid: blockIdx -> (XCD, slot); the workgroups resident on one XCD stream the SAME document range for different query
//     blocks, so a document is pulled from HBM once per XCD and served to the other CUs from that XCD's L2 (placement only
//     affects speed, never results).
#pragma once
#include <type_traits>

#include "maxsim_common.hpp"
#include "maxsim_stream.hip"

namespace msim {

constexpr int kBatchWaves = 8;                           // K1bP (maxsim_panels.hip); K1b takes its wave count as a template parameter
constexpr int kChunkSlabs = 4;
constexpr int kChunkRows = kChunkSlabs * kSlabRows;      // 128 patches
constexpr int kChunkBytes = kChunkSlabs * kSlabBytes;    // 32 KiB
constexpr int kBatchRing = 3;
constexpr int kBatchLds = kBatchRing * kChunkBytes;      // 96 KiB

constexpr int kMaxQBlocks = 256;     // query blocks per launch (the table travels in the kernel arguments; msim_fwd loops beyond that)

struct BatchArgs {
    long long ld;        // leading dimension of scores
    FlatQ fq;            // where the queries sit in the flat token matrix
    int n_d;
    int n_qblocks;       // query blocks: block b holds the WHOLE queries blk_q0[b] .. blk_q0[b+1]-1 -- at most NW * MAXU * 16 tokens and
                         // NW * 8 queries (host plan: maxsim_abi.hip flat_plan); its 16-token units are dealt to the waves round-robin
    int n_ranges;        // document ranges (multiple of 8: XCD x owns ranges x*sub .. x*sub+sub-1)
    unsigned flags;
    int *convoy;         // [n_ranges, n_qblocks] progress counters (zeroed before the launch) or null: see the convoy below
    unsigned long long *trace;   // debug (MSIM_BATCH_TRACE_PTR): per wave of workgroup 0, s_memtime ticks per phase of the chunk loop; null in production
    int blk_q0[kMaxQBlocks + 1];
};

// Convoy: the workgroups of one XCD that stream the SAME document range for different query blocks share that stream through
// the XCD's 4 MiB L2 -- as long as they stay within a few hundred KiB of each other.  Nothing forces that: they drift apart over a
// long launch and every straggler re-fetches from HBM what the leader already pulled (measured at 1000 queries: 66-125 GB of
// HBM reads for a 32.8 GB shard, different in every run).  Every kConvoyEvery chunks wave 0 publishes the workgroup's chunk
// count and waits (bounded) until it is no more than kConvoyWindow chunks ahead of the slowest workgroup of its range.  Purely
// a speed / traffic device: a workgroup that is not there (not resident, different XCD) only costs one time-out, after
// which the waiting workgroup stops looking; results never depend on it.
constexpr int kConvoyEvery = 8;
constexpr int kConvoySpins = 4096;

// first document whose start row is >= row (d_off is non-decreasing)
__device__ __forceinline__ int lower_bound_doc(const int32_t *__restrict__ d_off, int n_d, long long row) {
    int lo = 0, hi = n_d;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((long long)d_off[mid] < row) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// NW : waves per workgroup.  8 = one workgroup per CU, 128-row chunks (the MFMA-bound end: fewest query blocks per corpus pass);
//      4 = two workgroups per CU with 64-row chunks and half the tokens each: while one sits at its chunk barrier the other
//      computes (+5 % at 9..16 queries of 32 tokens, -2..-4 % from 64 up, where the doubled number of query blocks costs more);
//      2 = the pair form, four pairs per CU.
// RING: chunks in the shared LDS ring (4 for the pair form: 32 KiB per workgroup, 3 otherwise: 48 / 96 KiB).
// AUX : cache policy of the LDS-DMA loads: 2 = nt (no L2 / MALL allocation) when ONE query block streams the corpus, i.e. every byte
//       is read exactly once, as in K1s; 0 = default when several query blocks share a range through the XCD's L2.
// MAXU: 16-token units a wave holds at most: 8, or 10 (round 3's five-tile form): 160 B-operand registers, 128 of them pinned to
//       AGPRs, still two waves per SIMD.  How many it really holds is a run-time, wave-uniform number that selects the compiled body.
template <bool F16, int NW, int RING = 3, int AUX = 0, int MAXU = 8>
__global__ __launch_bounds__(NW * 64, (MAXU <= 5 ? 3 : 2)) void maxsim_batch_kernel(const uint16_t *__restrict__ Qt,
                                                               const uint16_t *__restrict__ D,
                                                               const int32_t *__restrict__ d_off,
                                                               const uint8_t *__restrict__ clamp0,
                                                               float *__restrict__ scores, BatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int kB1Waves = NW;
    constexpr int kBatchRing = RING;                       // (shadows the namespace constant)
    constexpr int kChunkSlabs = NW / 2;                    // every wave fills half a slab of each chunk (shadows the 8-wave constants)
    constexpr int kChunkRows = kChunkSlabs * kSlabRows;
    constexpr int kChunkBytes = kChunkSlabs * kSlabBytes;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // per-token max table of the workgroup: NW * MAXU units x 16 tokens x 16 B.  TWO of them (round 6), used by alternate documents:
    // with one table a one-chunk document needs a barrier of its own between the previous document's sums and its writes -- for a
    // corpus of 64-row documents (pooled pages) that is a second barrier per 128 MFMAs.  The pair form keeps one table (four pairs
    // per CU: a second table would push it over 160 KiB).
    constexpr bool kTwoTables = NW > 2;
    constexpr int kTableBytes = kB1Waves * MAXU * kUnitTok * 16;
    char *const tokmax = smem + kBatchRing * kChunkBytes;

    // ---- which (query block, document range) is this workgroup?
    const int sub = a.n_ranges >> 3;                       // ranges per XCD
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int qblock, range;
    if (sub > 1) { qblock = slot % a.n_qblocks; range = xcd * sub + slot / a.n_qblocks; }
    else         { qblock = slot;               range = xcd; }
    if (qblock >= a.n_qblocks || range >= a.n_ranges) return;
    // the range's documents: two searches of the row offsets, by the whole wave (64 probes per round: 2 dependent loads for 256
    // documents, 3 for 125 000, instead of 8 / 17 -- per workgroup, in front of everything else: part of the fixed cost of a
    // one-page-per-workgroup launch, tools/fixed_overhead.py)
    // (d_off may be a SLICE of a larger corpus' absolute offsets -- the drop-in's pipelined sub-ranges -- so the cut points are
    // taken between d_off[0] and d_off[n_d], not between 0 and d_off[n_d]: with the latter most ranges of a late slice were empty
    // and a quarter of the workgroups did all the work, 0.52 ms instead of 0.16 for 232 pages)
    const long long row0 = d_off[0], total_rows = (long long)d_off[a.n_d] - row0;
    const int want_lo = (int)(row0 + (total_rows * range) / a.n_ranges), want_hi = (int)(row0 + (total_rows * (range + 1)) / a.n_ranges);
    const int d_lo = lower_bound_wave(a.n_d, want_lo, lane, [&](int k) { return d_off[k]; });
    const int d_hi = (range + 1 == a.n_ranges) ? a.n_d : lower_bound_wave(a.n_d, want_hi, lane, [&](int k) { return d_off[k]; });
    int *const my_prog = a.convoy ? a.convoy + (size_t)range * a.n_qblocks : nullptr;    // this range's counters, one per query block
    if (d_lo >= d_hi) {
        if (my_prog && threadIdx.x == 0) __hip_atomic_store(my_prog + qblock, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    constexpr int kConvoyWindow = (NW == 8 ? 24 : 48);      // chunks a workgroup may lead by: 768 KiB of the shared stream
    bool convoy_on = my_prog != nullptr;
    int g_chunk = 0;                                        // chunks consumed by this workgroup (wave-uniform)

    // ---- this block's tokens; block-local unit u lives in wave u % NW (slot u / NW)
    static_assert(MAXU == 5 || MAXU == 8 || MAXU == 10, "a wave holds up to 8 units (10: the AGPR form; 5: three workgroups per CU)");
    const int qb0 = a.blk_q0[qblock];                                       // first query of this block
    const int qb_n = a.blk_q0[qblock + 1] - qb0;                            // queries in this block (<= 8 * NW)
    const int tok0 = flat_qoff(a.fq, qb0);
    const int n_tok = flat_qoff(a.fq, qb0 + qb_n) - tok0;                   // tokens in this block (<= 16 * NW * MAXU)
    const int n_units = (n_tok + kUnitTok - 1) / kUnitTok;
    const int my_nu = wave < n_units ? (n_units - 1 - wave) / kB1Waves + 1 : 0;   // units of this wave (wave-uniform)
    QueryUnit qu[MAXU];
    // ---- per-lane address constants (same slab image as K1s)
    const int l16 = lane & 15, l4 = lane >> 4;
    int src_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) src_off[j] = l4 * kRowBytes + (((l16 ^ l4) ^ (j << 2)) << 4);
    int rd_off[2][kKSteps16];
    slab_rd_offsets16(lane, rd_off);
    // this wave fills rows 16*(wave&1) .. +15 of slab (wave>>1) of every chunk: 4 wave-instructions of 4 rows
    const int my_lds_off = (wave >> 1) * kSlabBytes + (wave & 1) * 4096;
    const int my_row_off = (wave >> 1) * kSlabRows + (wave & 1) * 16;

    // ---- producer cursor over the flattened (document, chunk) sequence of [d_lo, d_hi)
    int p_idx = d_lo, p_row = 0, p_len = 0;
    __amdgpu_buffer_rsrc_t p_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)D, 0, 0, 0x00020000);
    auto p_open = [&]() {
        while (p_idx < d_hi) {
            const int r0 = d_off[p_idx], r1 = d_off[p_idx + 1];
            p_len = r1 - r0;
            if (p_len > 0) {
                p_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(D + (size_t)r0 * kDim), 0, p_len * kRowBytes,
                                                           0x00020000);
                p_row = 0;
                return;
            }
            ++p_idx;
        }
    };
    p_open();
    int p_slot = 0;
    // A share of a chunk that lies wholly past the document's end is NOT requested (round 6): an LDS-DMA instruction costs its wave
    // ~100 cycles whether or not its rows exist -- with 64-row documents waves 4..7 spent 700 cycles per document fetching zeros that
    // nobody reads (slabs past the end are never computed, and the rows of a tail slab past the end are masked after the MFMA: what
    // the ring holds there does not matter)
    unsigned p_hist = 0;                                     // bit k: the produce() k calls ago made its four requests (wave-uniform)
    auto produce = [&]() -> bool {
        p_hist <<= 1;
        if (p_idx >= d_hi) return false;
        char *dst = smem + p_slot * kChunkBytes + my_lds_off;
        const int soff = (p_row + my_row_off) * kRowBytes;   // rows past the document end read as zeros (bounds check)
        if (p_row + my_row_off < p_len) {
            p_hist |= 1u;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(p_rsrc, MSIM_LDS(dst + j * 1024), 16, src_off[j], soff + j * 1024, 0, AUX);
        }
        p_slot = (p_slot + 1 == kBatchRing) ? 0 : p_slot + 1;
        p_row += kChunkRows;
        if (p_row >= p_len) {
            ++p_idx;
            p_open();
        }
        return true;
    };

#pragma unroll
    for (int i = 0; i < kBatchRing - 1; ++i) produce();

    // ---- the block's units, loaded BEHIND the first chunks' LDS-DMA requests (both in flight together: the ring fill used to wait
    // for the block's 256 KiB of query operands to arrive first)
#pragma unroll
    for (int t = 0; t < MAXU; ++t)
        load_query_unit(qu[t], Qt + (size_t)tok0 * kDim, (wave + kB1Waves * t) * kUnitTok, n_tok, lane, t < my_nu);
    // the reduction after every document: 8 lanes per query, query j of the block in the workgroup's lanes 504 - 8j .. 511 - 8j (reduce_doc)
    // (the token range of the query waits in LDS next to the table, written by the lanes that read it back: the slab loop of the
    // ten-unit form has no two registers to spare)
    int *const rtab = reinterpret_cast<int *>(tokmax + (kTwoTables ? 2 : 1) * kTableBytes);
    {
        const int rq = (kB1Waves * 64 - 1 - (int)threadIdx.x) >> 3;
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
            // ten units: eight of them live in AGPRs (MFMA srcB reads either file); left alone hipcc keeps 128 VGPRs and shuffles
            // the rest through v_accvgpr copies inside the slab loop
            if (MAXU > 8 && t >= 2) asm volatile("" : "+a"(qu[t].f[ks]));
            else asm volatile("" : "+v"(qu[t].f[ks]));
        }


    const bool ref_bf16 = (a.flags & kFlagRefBf16) != 0;
    const bool round_total = ref_bf16 && !(a.flags & kFlagPartial);
    int c_slot = 0;
    // MSIM_TRACE builds only (make trace -> tools/_ab/libmaxsim_trace.so): even a never-taken tracing branch in the chunk loop costs
    // the product kernel ~5 % (registers and schedule), so the stamps are compiled out of the shipped library
    const bool tracing = kTraceBuild && a.trace != nullptr && blockIdx.x == 0;
    unsigned long long tr[7] = {0, 0, 0, 0, 0, 0, 0};     // vmcnt, convoy, barrier, DMA issue, slabs, document epilogue, chunks

    // ---- the token sums of the document whose maxima are in the table (all waves have passed a barrier since they were written)
    auto reduce_doc = [&](int doc, bool clamp, int tab) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));        // opaque: nothing derived from it (row pointers, table addresses) is hoisted into registers that
        // query j of the block is summed by the workgroup's LAST lanes (512 - 8j - 8 ..): the first waves are the ones the matrix pipe
        // serves first and, on short documents, the ones that issue the chunk's LDS-DMA requests; the sums go to their SIMD partners
        const int rq = (kB1Waves * 64 - 1 - tid) >> 3, ri = tid & 7;   // stay live across the slab loop (the ten-unit form spills otherwise)
        if (rq < qb_n) {
            float tot = reduce_query_tokens_lane0<F16>(tokmax + tab * kTableBytes, rtab[2 * rq], rtab[2 * rq + 1], ri, clamp, ref_bf16);
            if (round_total) tot = round_to_input<F16>(tot);
            if (ri == 0) scores[(size_t)(qb0 + rq) * a.ld + doc] = tot;
        }
    };
    // a barrier that also orders this wave's table writes (ds_write retires out of sight of s_barrier)
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    // the walk over [d_lo, d_hi) for a wave that holds NU units (NU = 0: it only feeds the ring, keeps the barriers and reduces)
    auto run = [&](auto nu_c) {
    constexpr int NU = decltype(nu_c)::value;
    constexpr bool wave_has_units = NU > 0;
    constexpr int NUA = NU > 0 ? NU : 1;                   // array extents (no zero-length arrays)
    int pend_doc = -1;                                      // document whose maxima wait in the table (wave-uniform, the same in all waves)
    bool pend_clamp = false;
    int pend_tab = 0, tab = 0;                              // the table they wait in; the table the current document writes
    for (int c_idx = d_lo; c_idx < d_hi; ++c_idx) {
        const int len = d_off[c_idx + 1] - d_off[c_idx];
        const int nchunk = (len + kChunkRows - 1) / kChunkRows;
        bool clamp = false;
        if (clamp0 != nullptr) {
            const uint64_t addr = reinterpret_cast<uint64_t>(clamp0) + (uint64_t)c_idx;
            clamp = ((scalar_load_u32(addr & ~3ull) >> ((addr & 3) * 8)) & 0xffu) != 0;
        }
        if (nchunk == 0) {          // a document without rows never enters the ring: every token's max is over nothing (-inf, or 0 under clamp0)
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            const int rq = (kB1Waves * 64 - 1 - tid) >> 3;
            if (rq < qb_n && (tid & 7) == 0)
                scores[(size_t)(qb0 + rq) * a.ld + c_idx] = (rtab[2 * rq + 1] > rtab[2 * rq] && !clamp) ? -INFINITY : 0.0f;
            continue;
        }
        float m[NUA];
#pragma unroll
        for (int t = 0; t < NUA; ++t) m[t] = -INFINITY;
        auto slab = [&](int src_lds, auto tail, int rows_left) {   // src_lds: LDS byte address of the slab (wave-uniform)
            constexpr bool kTail = decltype(tail)::value;
            bf16x8 af[2][kKSteps16];
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int ks = 0; ks < kKSteps16; ++ks) af[g][ks] = *reinterpret_cast<const bf16x8 *>(smem + src_lds + rd_off[g][ks]);
            slab_units<F16, NUA, kTail, false>(m, af, qu, rows_left, lane, [](int) {});
        };

        for (int ch = 0; ch < nchunk; ++ch) {
            const unsigned long long t0 = tracing ? __builtin_amdgcn_s_memtime() : 0;
            // my share of chunk `ch` has landed once nothing but the requests of the (ring-2) chunks produced after it is in flight:
            // 4 each if they were made at all
            if constexpr (kBatchRing == 2) wait_vmcnt<0>();
            else if constexpr (kBatchRing == 3) { if (p_hist & 1u) wait_vmcnt<4>(); else wait_vmcnt<0>(); }
            else {
                static_assert(kBatchRing <= 4, "the request history below covers two chunks");
                const int later = __builtin_popcount(p_hist & 3u);
                if (later == 2) wait_vmcnt<8>(); else if (later == 1) wait_vmcnt<4>(); else wait_vmcnt<0>();
            }
            const unsigned long long t1 = tracing ? __builtin_amdgcn_s_memtime() : 0;
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
                if (spin == kConvoySpins) convoy_on = false;   // somebody is not there: stop waiting for good
            }
            ++g_chunk;
            const unsigned long long t2 = tracing ? __builtin_amdgcn_s_memtime() : 0;
            lds_barrier();                  // everyone's share landed; everyone is done reading the previous chunk (and has written its maxima)
            const unsigned long long t3 = tracing ? __builtin_amdgcn_s_memtime() : 0;
            produce();                      // refill the buffer that was read in the previous iteration
            const unsigned long long t4 = tracing ? __builtin_amdgcn_s_memtime() : 0;
            if (ch == 0 && pend_doc >= 0) reduce_doc(pend_doc, pend_clamp, pend_tab);   // the previous document's token sums, behind its barrier

            const int cbuf = c_slot * kChunkBytes;
            c_slot = (c_slot + 1 == kBatchRing) ? 0 : c_slot + 1;
            const int rows_in_chunk = len - ch * kChunkRows;   // >= 1
            if constexpr (wave_has_units) {
                const int n_full = rows_in_chunk >= kChunkRows ? kChunkSlabs : rows_in_chunk / kSlabRows;
#pragma unroll 1
                for (int sl = 0; sl < n_full; ++sl) slab(cbuf + sl * kSlabBytes, std::false_type{}, kSlabRows);
                const int rem = rows_in_chunk - n_full * kSlabRows;
                if (n_full < kChunkSlabs && rem > 0) slab(cbuf + n_full * kSlabBytes, std::true_type{}, rem);
            }
            if (tracing) {
                if constexpr (wave_has_units) {     // the MFMAs have been issued, not retired: the running maxima must be readable
                    float sink = 0.f;
#pragma unroll
                    for (int t = 0; t < NUA; ++t) sink += m[t];
                    asm volatile("" ::"v"(sink));
                }
                const unsigned long long t5 = __builtin_amdgcn_s_memtime();
                tr[0] += t1 - t0; tr[1] += t2 - t1; tr[2] += t3 - t2; tr[3] += t4 - t3; tr[4] += t5 - t4; tr[6] += 1;
            }
        }
        const unsigned long long te0 = tracing ? __builtin_amdgcn_s_memtime() : 0;
        // ---- document epilogue: this wave's maxima into the workgroup's table; the sums are taken behind the next barrier.  With ONE
        // table a one-chunk document has no barrier between the previous document's sums and these writes, so it gets one; with two
        // tables the previous document's sums read the other one, and the table written here was last read two barriers ago
        if (nchunk == 1 && !kTwoTables) lds_barrier();
        if constexpr (wave_has_units) {
#pragma unroll
            for (int t = 0; t < NUA; ++t) store_token_max(tokmax + tab * kTableBytes, wave + kB1Waves * t, m[t], lane);
        }
        pend_doc = c_idx;
        pend_clamp = clamp;
        pend_tab = tab;
        if constexpr (kTwoTables) tab ^= 1;
        if (tracing) tr[5] += __builtin_amdgcn_s_memtime() - te0;
    }
    if (pend_doc >= 0) {
        lds_barrier();
        reduce_doc(pend_doc, pend_clamp, pend_tab);
    }
    if (tracing && lane == 0) {
#pragma unroll
        for (int i = 0; i < 7; ++i) a.trace[wave * 8 + i] = tr[i];
    }
    if (my_prog && threadIdx.x == 0)        // finished: never hold anybody back
        __hip_atomic_store(my_prog + qblock, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };   // run

    switch (my_nu) {                                       // wave-uniform; every body executes the same barriers
        case 0: run(std::integral_constant<int, 0>{}); break;
        case 1: run(std::integral_constant<int, 1>{}); break;
        case 2: run(std::integral_constant<int, 2>{}); break;
        case 3: run(std::integral_constant<int, 3>{}); break;
        case 4: run(std::integral_constant<int, 4>{}); break;
        case 5: run(std::integral_constant<int, 5>{}); break;
        case 6: if constexpr (MAXU >= 6) run(std::integral_constant<int, 6>{}); break;
        case 7: if constexpr (MAXU >= 7) run(std::integral_constant<int, 7>{}); break;
        case 8: if constexpr (MAXU >= 8) run(std::integral_constant<int, 8>{}); break;
        case 9: if constexpr (MAXU >= 9) run(std::integral_constant<int, 9>{}); break;
        default: if constexpr (MAXU >= 10) run(std::integral_constant<int, 10>{}); break;
    }
}

// scores[q, c] = sum over the segments s of partial[q * n_seg + s, c], in segment order (long queries: msim_fwd scores them as
// 128-token segments on K1b and adds the partial token sums here); `round_f16` / `round_bf16`: the literal tier's rounding of the
// token sum to the embedding dtype, applied once, to the total
template <bool F16>
__global__ __launch_bounds__(256) void segment_sum_kernel(const float *__restrict__ partial, long long ld_part, int n_seg, int n_d,
                                                          float *__restrict__ scores, long long ld, bool round_total) {
    const int q = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n_d) return;
    const float *p = partial + (size_t)q * n_seg * ld_part + c;
    float acc = 0.0f;
    for (int s = 0; s < n_seg; ++s) acc += p[(size_t)s * ld_part];
    if (round_total) acc = round_to_input<F16>(acc);
    scores[(size_t)q * ld + c] = acc;
}


// Zero rows of a [n_q, Lq, dim] query box add exactly 0 to every score (the model multiplies padded positions by the attention mask:
// modeling_colpali.py:72, modeling_colqwen2.py:69): the flat layout drops them.  One workgroup per query.  counts != null: the number
// of rows that are not all-zero goes to counts[q].  out != null: those rows are copied, in order, to rows q_off[q] .. of `out`.
constexpr int kCompactMaxRows = 4096;
__global__ __launch_bounds__(256) void query_compact_kernel(const char *__restrict__ box, int Lq, int row_bytes,
                                                            const int32_t *__restrict__ q_off, int32_t *__restrict__ counts,
                                                            char *__restrict__ out) {
    __shared__ int pos[kCompactMaxRows + 1];
    const int q = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char *src = box + (size_t)q * Lq * row_bytes;
    const int pieces = row_bytes >> 4;
    for (int r = wave; r < Lq; r += 4) {
        uint32_t acc = 0;
        for (int p = lane; p < pieces; p += 64) {
            const i32x4 v = *reinterpret_cast<const i32x4 *>(src + (size_t)r * row_bytes + (p << 4));
            acc |= (uint32_t)(v[0] | v[1] | v[2] | v[3]);
        }
        const bool nz = __ballot(acc != 0) != 0;
        if (lane == 0) pos[r + 1] = nz ? 1 : 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        pos[0] = 0;
        for (int r = 0; r < Lq; ++r) pos[r + 1] += pos[r];
        if (counts) counts[q] = pos[Lq];
    }
    __syncthreads();
    if (out == nullptr) return;
    char *dst = out + (size_t)q_off[q] * row_bytes;
    for (int r = wave; r < Lq; r += 4) {
        if (pos[r + 1] == pos[r]) continue;
        for (int p = lane; p < pieces; p += 64)
            *reinterpret_cast<i32x4 *>(dst + (size_t)pos[r] * row_bytes + (p << 4)) =
                *reinterpret_cast<const i32x4 *>(src + (size_t)r * row_bytes + (p << 4));
    }
}

}  // namespace msim
