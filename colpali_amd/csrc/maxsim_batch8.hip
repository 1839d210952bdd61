// K1b8 -- the MFMA-bound end of the MaxSim scorer on gfx950 (MI355X): EIGHT token tiles per operand fetch.
// Same arithmetic as K1s / K1b (colpali_engine/utils/processing_utils.py:179,
// colpali_engine/loss/late_interaction_losses.py:297-298), bit-identical scores, different register plan.
//
// Why: K1b (maxsim_batch.hip) runs two 256-register waves per SIMD, 4 token tiles (128 B-operand registers) each, and reads
// the 8 operand fragments of a 32-row slab once per 4 tiles = one ds_read_b128 per 8 MFMAs.  On real operand values the chip
// is power-bound, and the bare loop with that instruction mix tops out at 1.73 PFLOP/s (msim_probe_mfma variant 12); K1b sits
// on that ceiling.  The LDS -> VGPR operand path is the largest item on top of the MFMAs themselves.  Here ONE 512-register
// wave per SIMD holds up to 8 tiles -- 256 B-operand registers, most of them in AGPRs (MFMA srcA/srcB read either file on
// gfx950; hipcc is told so with "+a" constraints, left alone it shuffles the overflow through v_accvgpr copies) -- so a
// fragment read feeds 16 MFMAs, and everything a second wave used to cover is covered by hand instead:
//   * the 8 operand fragments of the NEXT slab are fetched underneath the MFMAs of the current one (two fragment sets,
//     the slab loop is unrolled by two so no register is ever copied);
//   * the LDS-DMA pieces of the ring refill are issued one or two per token tile BETWEEN the MFMAs, a constant 8 per chunk
//     and wave, so the `s_waitcnt vmcnt(N)` in front of the chunk barrier is a constant too and the slab body has no branch;
//   * one raw s_barrier per chunk of NW slabs, placed at the START of the chunk's last slab: it publishes the next chunk
//     (whose first fragments that slab prefetches) and frees the current one (whose last fragments are in registers already).
// msim_probe_mfma variants 12 / 13 / 18 price the plan: 1.73 (K1b's mix) -> 1.74 (8 tiles, no prefetch) -> 1.83-1.85 PFLOP/s
// (8 tiles + prefetch), profiles/r03_logs/ab_probe_mix8.log.
//
// NW waves share one document stream (each loads one whole slab of every chunk):
//   NW = 4: one workgroup per CU, up to 32 tiles per query block  (17+ tiles)
//   NW = 2: two workgroups per CU, up to 16 tiles per query block (9..16 tiles; one query block, `nt` stream)
// Grid, balanced query blocks, XCD-shared document ranges and the convoy are K1b's (maxsim_batch.hip).
#pragma once
#include <type_traits>

#include "maxsim_batch.hip"
#include "maxsim_common.hpp"
#include "maxsim_stream.hip"

namespace msim {

template <int N>
__device__ __forceinline__ void wait_lgkmcnt() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

// TPQ : token tiles (32 tokens) per query, 1..4; a wave holds whole queries, up to 8 / TPQ of them
// NW  : waves per workgroup = slabs per chunk
// RING: chunks in the shared LDS ring
// AUX : cache policy of the LDS-DMA loads (2 = nt when one query block streams the corpus, 0 when blocks share ranges via L2)
// VAR : measurement variants (bit 0: folds deferred by a whole tile on two accumulator sets with the interleave pinned;
//       bit 1 / bit 2: KNOCK-OUTS that give wrong scores -- no LDS-DMA issue inside the slab body / no chunk barrier)
template <int TPQ, bool F16, int NW, int RING, int AUX, int VAR = 0>
__global__ __launch_bounds__(NW * 64, 1) void maxsim_batch8_kernel(const uint16_t *__restrict__ Q,
                                                                const uint16_t *__restrict__ D,
                                                                const int32_t *__restrict__ d_off,
                                                                const uint8_t *__restrict__ clamp0,
                                                                float *__restrict__ scores, BatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int kChunkSlabs = NW;
    constexpr int kChunkRows = kChunkSlabs * kSlabRows;
    constexpr int kChunkBytes = kChunkSlabs * kSlabBytes;
    constexpr int PPS = 8 / NW;                            // LDS-DMA pieces a wave issues per slab it computes
    static_assert(NW == 1 || NW == 2 || NW == 4, "a wave issues 8 / NW pieces per slab");
    static_assert(RING >= 3, "one chunk being read, one landed or landing, one being refilled");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // ---- which (query block, document range) is this workgroup?  (as K1b)
    const int sub = a.n_ranges >> 3;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int qblock, range;
    if (sub > 1) { qblock = slot % a.n_qblocks; range = xcd * sub + slot / a.n_qblocks; }
    else         { qblock = slot;               range = xcd; }
    if (qblock >= a.n_qblocks || range >= a.n_ranges) return;
    const long long total_rows = d_off[a.n_d];
    const int d_lo = lower_bound_doc(d_off, a.n_d, (total_rows * range) / a.n_ranges);
    const int d_hi = (range + 1 == a.n_ranges) ? a.n_d
                                                : lower_bound_doc(d_off, a.n_d, (total_rows * (range + 1)) / a.n_ranges);
    int *const my_prog = a.convoy ? a.convoy + (size_t)range * a.n_qblocks : nullptr;
    if (d_lo >= d_hi) {
        if (my_prog && threadIdx.x == 0) __hip_atomic_store(my_prog + qblock, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    constexpr int kConvoyWindow = 768 * 1024 / kChunkBytes;   // chunks a workgroup may lead by: 768 KiB of the shared stream
    constexpr int kConvoyEvery8 = 256 * 1024 / kChunkBytes;   // publish / look every 256 KiB (a power of two)
    bool convoy_on = my_prog != nullptr;
    int g_chunk = 0;

    // ---- this wave's queries: block-local query j lives in wave j % NW
    static_assert(TPQ >= 1 && TPQ <= 4, "a wave holds whole queries of at most 4 token tiles");
    constexpr int QPW = 8 / TPQ;
    constexpr int NTMAX = QPW * TPQ;
    const int q_base = a.n_q / a.n_qblocks, q_extra = a.n_q % a.n_qblocks;
    const int qb0 = qblock * q_base + (qblock < q_extra ? qblock : q_extra);
    const int qb_n = q_base + (qblock < q_extra ? 1 : 0);                     // <= NW * QPW
    const int my_q = wave < qb_n ? (qb_n - 1 - wave) / NW + 1 : 0;
    QueryTile qt[NTMAX];
#pragma unroll
    for (int t = 0; t < NTMAX; ++t) {
        const bool live = t / TPQ < my_q;
        const int q = qb0 + wave + NW * (t / TPQ);
        load_query_tile(qt[t], Q + (size_t)(live ? q : 0) * a.Lq * kDim, (t % TPQ) * kTokTile, a.Lq, lane, live);
    }
    wait_vmcnt<0>();
    // tiles 0 and 1 stay in VGPRs, the others are pinned to AGPRs: 64 + 192 B-operand registers, which leaves the VGPR file to
    // the two fragment sets (64), the accumulators, the running maxima and the address constants
#pragma unroll
    for (int t = 0; t < NTMAX; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ks = 0; ks < kKSteps16; ++ks) {
                if (t >= 2) asm volatile("" : "+a"(qt[t].f[h][ks]));
                else asm volatile("" : "+v"(qt[t].f[h][ks]));
            }

    // ---- per-lane address constants (the slab image of K1s / K1b)
    const int l16 = lane & 15, l4 = lane >> 4;
    const int src_base = l4 * kRowBytes + ((l16 ^ l4) << 4);     // piece j of a slab reads at src_base ^ ((j & 3) << 6)
    int rd_off[2][kKSteps16];
    slab_rd_offsets16(lane, rd_off);

    // ---- producer: the flattened (document, chunk) sequence of [d_lo, d_hi); this wave loads slab `wave` of every chunk
    int p_idx = d_lo, p_row = 0, p_len = 0, p_slot = 0;
    const __amdgpu_buffer_rsrc_t null_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)D, 0, 0, 0x00020000);
    __amdgpu_buffer_rsrc_t p_rsrc = null_rsrc;
    auto p_open = [&]() {
        while (p_idx < d_hi) {
            const int r0 = d_off[p_idx], r1 = d_off[p_idx + 1];
            p_len = r1 - r0;
            if (p_len > 0) {
                p_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(D + (size_t)r0 * kDim), 0, p_len * kRowBytes, 0x00020000);
                p_row = 0;
                return;
            }
            ++p_idx;
        }
    };
    p_open();
    // the chunk whose pieces are being issued (8 per wave; requests past the end of the stream go through an empty descriptor:
    // they fetch nothing and count like any other load, so the number of outstanding loads is a constant)
    __amdgpu_buffer_rsrc_t pc_rsrc = null_rsrc;
    int pc_soff = 0, pc_dst = 0, pending = 0;
    auto p_begin = [&]() {
        const bool live = p_idx < d_hi;
        pc_rsrc = live ? p_rsrc : null_rsrc;
        pc_soff = live ? (p_row + wave * kSlabRows) * kRowBytes : 0;      // rows past the document end read as zeros (bounds check)
        pc_dst = p_slot * kChunkBytes + wave * kSlabBytes;
        pending = 8;
        p_slot = (p_slot + 1 == RING) ? 0 : p_slot + 1;
        if (live) {
            p_row += kChunkRows;
            if (p_row >= p_len) {
                ++p_idx;
                p_open();
            }
        }
    };
    auto issue_piece = [&]() {
        const int j = 8 - pending;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(pc_rsrc, MSIM_LDS(smem + pc_dst + j * 1024), 16, src_base ^ ((j & 3) << 6),
                                                 pc_soff + j * 1024, 0, AUX);
        --pending;
    };
#pragma unroll 1
    for (int i = 0; i < RING - 1; ++i) {                   // prologue: chunks 0 .. RING-2 in flight
        p_begin();
#pragma unroll
        for (int j = 0; j < 8; ++j) issue_piece();
    }

    const bool ref_bf16 = (a.flags & kFlagRefBf16) != 0;

    // Chunk hand-over.  Called at the start of the LAST slab of the chunk being read (and once before the first slab): all 8
    // pieces of the youngest chunk are out, the chunk after the current one must have landed (8 * (RING - 2) younger loads may
    // stay in flight), this wave's fragment reads of the current chunk have returned; after the barrier the next chunk is
    // visible to everyone and the current chunk's slot belongs to the producer again.
    auto hand_over = [&]() {
        while (pending > 0) issue_piece();                 // only after a document tail (a chunk with fewer than NW slabs)
        wait_vmcnt<8 * (RING - 2)>();
        wait_lgkmcnt<0>();
        if (convoy_on && wave == 0 && (g_chunk & (kConvoyEvery8 - 1)) == 0) {
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
        if constexpr ((VAR & 4) == 0) __builtin_amdgcn_s_barrier();
        p_begin();
        if constexpr ((VAR & 2) != 0) pending = 0;
        if constexpr ((VAR & 8) != 0) {                    // all 8 pieces of the refill in one burst behind the barrier
#pragma unroll
            for (int j = 0; j < 8; ++j) issue_piece();
        }
    };

    auto run = [&](auto nt_c) {
        constexpr int NT = decltype(nt_c)::value;
        constexpr int NTA = NT > 0 ? NT : 1;
        float m[NTA][2];
#pragma unroll
        for (int t = 0; t < NTA; ++t) m[t][0] = m[t][1] = -INFINITY;

        TileAcc carry;                                     // VAR bit 0: the last tile of a slab, folded underneath the next slab
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int g = 0; g < 2; ++g) carry.a[h][g] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        auto doc_epilogue = [&](int c) {                   // combine the lane groups, clamp, sum over tokens, store; reset the maxima
            if constexpr (NT > 0) {
                if constexpr ((VAR & 1) != 0) {
                    tile_fold(m[NTA - 1], carry);
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int g = 0; g < 2; ++g) carry.a[h][g] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                }
                bool clamp = false;
                if (clamp0 != nullptr) {
                    const uint64_t addr = reinterpret_cast<uint64_t>(clamp0) + (uint64_t)c;
                    clamp = ((scalar_load_u32(addr & ~3ull) >> ((addr & 3) * 8)) & 0xffu) != 0;
                }
                float tile_sum[NTA];
#pragma unroll
                for (int t = 0; t < NTA; ++t) tile_sum[t] = tile_finish<F16>(m[t], clamp, ref_bf16);
                if (lane == 0) {
#pragma unroll
                    for (int qq = 0; qq < NTA / TPQ; ++qq) {
                        float tot = 0.0f;
#pragma unroll
                        for (int tt = 0; tt < TPQ; ++tt) tot += tile_sum[qq * TPQ + tt];
                        if (ref_bf16) tot = round_to_input<F16>(tot);
                        scores[(size_t)(qb0 + wave + NW * qq) * a.ld + c] = tot;
                    }
                }
#pragma unroll
                for (int t = 0; t < NTA; ++t) m[t][0] = m[t][1] = -INFINITY;
            }
        };

        auto fetch = [&](bf16x8 (&af)[2][kKSteps16], int lds_addr) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int ks = 0; ks < kKSteps16; ++ks) af[g][ks] = *reinterpret_cast<const bf16x8 *>(smem + lds_addr + rd_off[g][ks]);
        };

        // one slab against this wave's NT tiles; PPS refill pieces go out between the tiles
        auto body = [&](auto tail_c, const bf16x8 (&af)[2][kKSteps16], int rows_left) {
            constexpr bool kTail = decltype(tail_c)::value;
            if constexpr (NT == 0) {
                if constexpr ((VAR & 10) == 0) {
#pragma unroll
                    for (int i = 0; i < PPS; ++i) issue_piece();
                }
            } else if constexpr ((VAR & 1) != 0) {
                // two accumulator sets: the 16 -> 1 fold of tile t-1 is spread over the 16 MFMAs of tile t (no wait states between a
                // tile's last MFMA and the reads of its accumulators); the fold of the slab's last tile is carried into the next slab
                TileAcc acc[2];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int g = 0; g < 2; ++g) acc[t & 1].a[h][g] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if constexpr ((VAR & 10) == 0) {
#pragma unroll
                        for (int i = 0; i < PPS; ++i)
                            if ((i * NT) / PPS == t) issue_piece();
                    }
#pragma unroll
                    for (int ks = 0; ks < kKSteps16; ++ks)
#pragma unroll
                        for (int hg = 0; hg < 4; ++hg)
                            acc[t & 1].a[hg >> 1][hg & 1] = mfma16<F16>(af[hg & 1][ks], qt[t].f[hg >> 1][ks], acc[t & 1].a[hg >> 1][hg & 1]);
                    if constexpr (kTail) tile_mask_tail(acc[t & 1], rows_left, lane);
                    if (t > 0) tile_fold(m[t - 1], acc[(t - 1) & 1]);
                    else tile_fold(m[NT - 1], carry);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x2, 1, 0);
                    }
                }
                carry = acc[(NT - 1) & 1];
            } else {
                TileAcc prev;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    TileAcc acc;
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int g = 0; g < 2; ++g) acc.a[h][g] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if constexpr ((VAR & 10) == 0) {
#pragma unroll
                        for (int i = 0; i < PPS; ++i)
                            if ((i * NT) / PPS == t) issue_piece();
                    }
#pragma unroll
                    for (int ks = 0; ks < kKSteps16; ++ks)
#pragma unroll
                        for (int hg = 0; hg < 4; ++hg)
                            acc.a[hg >> 1][hg & 1] = mfma16<F16>(af[hg & 1][ks], qt[t].f[hg >> 1][ks], acc.a[hg >> 1][hg & 1]);
                    if constexpr (kTail) tile_mask_tail(acc, rows_left, lane);
                    if (t > 0) tile_fold(m[t - 1], prev);      // the fold of tile t-1 runs underneath the MFMAs of tile t
                    prev = acc;
                }
                tile_fold(m[NT - 1], prev);
            }
        };

        // ---- consumer state (wave-uniform, identical in every wave of the workgroup)
        int c_idx = d_lo;
        auto skip_empty = [&]() {
            while (c_idx < d_hi && d_off[c_idx + 1] - d_off[c_idx] <= 0) {
                doc_epilogue(c_idx);
                ++c_idx;
            }
        };
        skip_empty();
        if (c_idx < d_hi) {
            int len = d_off[c_idx + 1] - d_off[c_idx];
            int row = 0, sl = 0, c_slot = 0;
            hand_over();                                    // chunk 0 is visible; the refill of slot RING-1 starts
            bf16x8 af0[2][kKSteps16], af1[2][kKSteps16];
            fetch(af0, 0);
            // one slab item; returns false after the last one
            auto item = [&](const bf16x8 (&cur)[2][kKSteps16], bf16x8 (&nxt)[2][kKSteps16]) -> bool {
                const int rows_left = len - row;
                const bool last_of_doc = rows_left <= kSlabRows;
                const bool last_of_chunk = last_of_doc || sl == kChunkSlabs - 1;
                int nxt_addr = c_slot * kChunkBytes + (sl + 1) * kSlabBytes;
                int n_sl = sl + 1, n_slot = c_slot;
                if (last_of_chunk) {
                    hand_over();
                    n_slot = (c_slot + 1 == RING) ? 0 : c_slot + 1;
                    n_sl = 0;
                    nxt_addr = n_slot * kChunkBytes;
                }
                // unconditional (after the last slab of the range it reads a slot nobody needs): a conditional fetch would make the
                // compiler's s_waitcnt placement merge "fetched" and "not fetched" and wait for the prefetch at the top of the body
                fetch(nxt, nxt_addr);
                if (rows_left >= kSlabRows) body(std::false_type{}, cur, kSlabRows);
                else body(std::true_type{}, cur, rows_left);
                sl = n_sl;
                c_slot = n_slot;
                if (last_of_doc) {
                    doc_epilogue(c_idx);
                    ++c_idx;
                    skip_empty();
                    if (c_idx >= d_hi) return false;
                    len = d_off[c_idx + 1] - d_off[c_idx];
                    row = 0;
                } else {
                    row += kSlabRows;
                }
                return true;
            };
            for (;;) {
                if (!item(af0, af1)) break;
                if (!item(af1, af0)) break;
            }
        }
        wait_vmcnt<0>();                                    // no LDS-DMA write may outlive the workgroup's LDS allocation
        if (my_prog && threadIdx.x == 0)
            __hip_atomic_store(my_prog + qblock, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };   // run

    switch (my_q) {                                        // wave-uniform; every body executes the same barriers
        case 0: run(std::integral_constant<int, 0>{}); break;
        case 1: run(std::integral_constant<int, TPQ>{}); break;
        case 2: if constexpr (QPW >= 2) run(std::integral_constant<int, 2 * TPQ>{}); break;
        case 3: if constexpr (QPW >= 3) run(std::integral_constant<int, 3 * TPQ>{}); break;
        case 4: if constexpr (QPW >= 4) run(std::integral_constant<int, 4 * TPQ>{}); break;
        case 5: if constexpr (QPW >= 5) run(std::integral_constant<int, 5 * TPQ>{}); break;
        case 6: if constexpr (QPW >= 6) run(std::integral_constant<int, 6 * TPQ>{}); break;
        case 7: if constexpr (QPW >= 7) run(std::integral_constant<int, 7 * TPQ>{}); break;
        default: if constexpr (QPW >= 8) run(std::integral_constant<int, 8 * TPQ>{}); break;
    }
}

}  // namespace msim
