// K1sP / K1bP -- the tuned MaxSim kernels for embeddings WIDER than 128 (ColQwen3: dim = 320,
// colpali_engine/models/qwen3/colqwen3/modeling_colqwen3.py:48), 16-bit.  Same arithmetic as K1s / K1b
// (colpali_engine/utils/processing_utils.py:179); same building blocks: LDS-DMA with a per-document buffer
// descriptor, the 32 x 256 B slab image XOR-swizzled on the source address, the swapped 16x16x32 MFMA tiling of
// maxsim_common.hpp with the running max in registers.
//
// A row of dim * 2 bytes is streamed as PANELS column panels of 256 bytes (128 elements): panel p of a 32-row slab is
// one 8 KiB "panel-slab" with exactly the LDS image of a dim-128 slab, so the swizzle, the operand fetches and the
// conflict-freeness carry over unchanged; the MFMA chain of a (slab, token tile) simply runs across the panels
// (4 k-steps of 32 per full panel, KS_LAST / 2 in the last one) before the max is folded.  dim = ((PANELS-1)*8 + KS_LAST) * 16 is a
// compile-time constant of each instantiation.  In the last panel the lanes whose 16-byte chunk lies beyond the row
// re-read chunk 0 of their row instead (their LDS slots are never consumed): no byte outside the row is fetched.
//
// K1sP: one wave = one pipeline, wave-private ring of 4 panel-slabs, no barrier (HBM-bound regime, <= 4 token tiles:
//       a wave runs alone on its SIMD, so the 80 VGPRs per resident token tile fit).
// K1bP: 8 waves x NT <= 2 token tiles, one stage = the PANELS panel-slabs of ONE 32-row slab shared by the workgroup,
//       4-deep ring, one raw s_barrier per stage, XCD-aware grid as in K1b (MFMA-bound regime).
#pragma once
#include "maxsim_common.hpp"
#include "maxsim_stream.hip"
#include "maxsim_batch.hip"

namespace msim {

// the panel kernels keep the [n_q, Lq, dim] query box of rounds 1-3 (whole queries of 32-token tiles per wave)
struct PanelStreamArgs {
    long long ld;   // leading dimension of scores
    int n_q, Lq, n_d;
    unsigned flags;
};
struct PanelBatchArgs {
    long long ld;
    int n_q, Lq, n_d;
    int n_qblocks;       // query blocks: block b holds n_q / n_qblocks (+1 for the first n_q % n_qblocks) queries
    int n_ranges;        // document ranges (multiple of 8: XCD x owns ranges x*sub .. x*sub+sub-1)
    unsigned flags;
};

constexpr int kPanelBytes = 256;             // one panel of a row
constexpr int kPanelRing = 4;                // K1sP: panel-slabs per wave-private ring
constexpr int kPanelStages = 4;              // K1bP: stages in the shared ring

// per-lane LDS-DMA source offsets of one 1 KiB wave-instruction (rows 4j..4j+3 of a panel-slab, j = instruction & 3):
// lane (l4 = lane >> 4, l16 = lane & 15) fills physical chunk l16 of LDS row 4i + l4, which holds logical chunk
// l16 ^ l4 ^ (j << 2); chunks at or beyond `valid_chunks` (last panel) are redirected to chunk 0 of the same row.
__device__ __forceinline__ int panel_src_off(int lane, int j, int row_bytes, int valid_chunks) {
    const int l16 = lane & 15, l4 = lane >> 4;
    int c = (l16 ^ l4) ^ (j << 2);
    c = c < valid_chunks ? c : 0;
    return l4 * row_bytes + (c << 4);
}

template <int QT, int TPQ, int PANELS, int KS_LAST, bool F16, int AUX>
__global__ __launch_bounds__(256) void maxsim_stream_panels_kernel(const uint16_t *__restrict__ Q, const uint16_t *__restrict__ D,
                                                                   const int32_t *__restrict__ d_off,
                                                                   const uint8_t *__restrict__ clamp0,
                                                                   float *__restrict__ scores, PanelStreamArgs a) {
    constexpr int KT = (PANELS - 1) * 8 + KS_LAST;      // k-steps of 16 elements
    constexpr int DIM = KT * 16;
    constexpr int ROW_BYTES = DIM * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *ring = smem + wave * (kPanelRing * kSlabBytes);
    const int gw = blockIdx.x * 4 + wave;
    const int GW = gridDim.x * 4;

    // ---- query fragments (B operands of the 16x16x32 tiling, maxsim_common.hpp): [tile][token half][k-step of 32]
    static_assert(KS_LAST % 2 == 0, "the last panel must hold whole k-steps of 32");
    constexpr int KT32 = KT / 2;
    bf16x8 qf[QT][2][KT32];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int q = t / TPQ;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = (t % TPQ) * kTokTile + 16 * h + (lane & 15);
            const bool valid = row < a.Lq;
            const uint16_t *p = Q + ((size_t)q * a.Lq + (valid ? row : 0)) * DIM + (lane >> 4) * 8;
#pragma unroll
            for (int ks = 0; ks < KT32; ++ks) {
                bf16x8 v = *reinterpret_cast<const bf16x8 *>(p + ks * 32);
                qf[t][h][ks] = valid ? v : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
    }
    wait_vmcnt<0>();
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ks = 0; ks < KT32; ++ks) asm volatile("" : "+v"(qf[t][h][ks]));

    int src_full[4], src_last[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        src_full[j] = panel_src_off(lane, j, ROW_BYTES, 16);
        src_last[j] = panel_src_off(lane, j, ROW_BYTES, 2 * KS_LAST);
    }
    int rd_off[2][kKSteps16];
    slab_rd_offsets16(lane, rd_off);

    // ---- producer cursor (wave-uniform): next panel-slab to request = (document, first row of the slab, panel)
    int p_idx = gw, p_row = 0, p_len = 0, p_pan = 0;
    __amdgpu_buffer_rsrc_t p_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)D, 0, 0, 0x00020000);
    auto p_open = [&]() {
        while (p_idx < a.n_d) {
            const int r0 = d_off[p_idx], r1 = d_off[p_idx + 1];
            p_len = r1 - r0;
            if (p_len > 0) {
                p_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(D + (size_t)r0 * DIM), 0, p_len * ROW_BYTES, 0x00020000);
                p_row = 0;
                p_pan = 0;
                return;
            }
            p_idx += GW;
        }
    };
    p_open();
    int p_slot = 0;
    auto produce = [&]() -> bool {
        if (p_idx >= a.n_d) return false;
        char *dst = ring + p_slot * kSlabBytes;
        const int soff = p_row * ROW_BYTES + p_pan * kPanelBytes;
        const bool last = p_pan == PANELS - 1;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(p_rsrc, MSIM_LDS(dst + i * 1024), 16, last ? src_last[i & 3] : src_full[i & 3],
                                                     soff + i * 4 * ROW_BYTES, 0, AUX);
        p_slot = (p_slot + 1 == kPanelRing) ? 0 : p_slot + 1;
        if (++p_pan == PANELS) {
            p_pan = 0;
            p_row += kSlabRows;
            if (p_row >= p_len) {
                p_idx += GW;
                p_open();
            }
        }
        return true;
    };
#pragma unroll
    for (int i = 0; i < kPanelRing - 1; ++i) produce();

    const bool ref_bf16 = (a.flags & kFlagRefBf16) != 0;
    int c_slot = 0;

    for (int c_idx = gw; c_idx < a.n_d; c_idx += GW) {
        const int len = d_off[c_idx + 1] - d_off[c_idx];
        const int nslab = (len + kSlabRows - 1) / kSlabRows;
        float m[QT][2];
#pragma unroll
        for (int t = 0; t < QT; ++t) m[t][0] = m[t][1] = -INFINITY;

        for (int s = 0; s < nslab; ++s) {
            TileAcc acc[QT];
#pragma unroll
            for (int t = 0; t < QT; ++t)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int g = 0; g < 2; ++g) acc[t].a[h][g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int p = 0; p < PANELS; ++p) {
                const int nks = p == PANELS - 1 ? KS_LAST / 2 : kKSteps16;     // k-steps of 32 in this panel
                if (produce()) wait_vmcnt<8 * (kPanelRing - 1)>(); else wait_vmcnt<0>();
                const char *src = ring + c_slot * kSlabBytes;
                c_slot = (c_slot + 1 == kPanelRing) ? 0 : c_slot + 1;
                bf16x8 af[2][kKSteps16];
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int ks = 0; ks < kKSteps16; ++ks)
                        if (ks < nks) af[g][ks] = *reinterpret_cast<const bf16x8 *>(src + rd_off[g][ks]);
#pragma unroll
                for (int t = 0; t < QT; ++t)
#pragma unroll
                    for (int ks = 0; ks < kKSteps16; ++ks)
                        if (ks < nks) {
#pragma unroll
                            for (int hg = 0; hg < 4; ++hg)
                                acc[t].a[hg >> 1][hg & 1] =
                                    mfma16<F16>(af[hg & 1][ks], qf[t][hg >> 1][p * kKSteps16 + ks], acc[t].a[hg >> 1][hg & 1]);
                        }
            }
            const int rows_left = len - s * kSlabRows;
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                if (rows_left < kSlabRows) tile_mask_tail(acc[t], rows_left, lane);
                tile_fold(m[t], acc[t]);
            }
        }

        bool clamp = false;
        if (clamp0 != nullptr) {
            const uint64_t addr = reinterpret_cast<uint64_t>(clamp0) + (uint64_t)c_idx;
            clamp = ((scalar_load_u32(addr & ~3ull) >> ((addr & 3) * 8)) & 0xffu) != 0;
        }
        float tile_sum[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) tile_sum[t] = tile_finish<F16>(m[t], clamp, ref_bf16);
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < QT / TPQ; ++q) {
                float tot = 0.0f;
#pragma unroll
                for (int tt = 0; tt < TPQ; ++tt) tot += tile_sum[q * TPQ + tt];
                if (ref_bf16) tot = round_to_input<F16>(tot);
                scores[(size_t)q * a.ld + c_idx] = tot;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
template <int NT, int TPQ, int PANELS, int KS_LAST, bool F16>
__global__ __launch_bounds__(512, 2) void maxsim_batch_panels_kernel(const uint16_t *__restrict__ Q, const uint16_t *__restrict__ D,
                                                                      const int32_t *__restrict__ d_off,
                                                                      const uint8_t *__restrict__ clamp0,
                                                                      float *__restrict__ scores, PanelBatchArgs a) {
    constexpr int KT = (PANELS - 1) * 8 + KS_LAST;
    constexpr int DIM = KT * 16;
    constexpr int ROW_BYTES = DIM * 2;
    constexpr int kStage = PANELS * kSlabBytes;          // the PANELS panel-slabs of one 32-row slab
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // ---- which (query block, document range) is this workgroup?  (same XCD-aware mapping as K1b)
    const int sub = a.n_ranges >> 3;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int qblock, range;
    if (sub > 1) { qblock = slot % a.n_qblocks; range = xcd * sub + slot / a.n_qblocks; }
    else         { qblock = slot;               range = xcd; }
    if (qblock >= a.n_qblocks || range >= a.n_ranges) return;
    const long long row0 = d_off[0], total_rows = (long long)d_off[a.n_d] - row0;     // d_off may be a slice of absolute offsets
    const int d_lo = lower_bound_doc(d_off, a.n_d, row0 + (total_rows * range) / a.n_ranges);
    const int d_hi = (range + 1 == a.n_ranges) ? a.n_d
                                                : lower_bound_doc(d_off, a.n_d, row0 + (total_rows * (range + 1)) / a.n_ranges);
    if (d_lo >= d_hi) return;

    static_assert(NT >= 1 && NT <= 2 && NT % TPQ == 0, "a wave holds whole queries");
    constexpr int q_per_wave = NT / TPQ;
    const int q_first = (qblock * kBatchWaves + wave) * q_per_wave;
    static_assert(KS_LAST % 2 == 0, "the last panel must hold whole k-steps of 32");
    constexpr int KT32 = KT / 2;
    bf16x8 qf[NT][2][KT32];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int q = q_first + t / TPQ;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = (t % TPQ) * kTokTile + 16 * h + (lane & 15);
            const bool valid = q < a.n_q && row < a.Lq;
            const uint16_t *p = Q + ((size_t)(valid ? q : 0) * a.Lq + (valid ? row : 0)) * DIM + (lane >> 4) * 8;
#pragma unroll
            for (int ks = 0; ks < KT32; ++ks) {
                bf16x8 v = *reinterpret_cast<const bf16x8 *>(p + ks * 32);
                qf[t][h][ks] = valid ? v : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
    }
    wait_vmcnt<0>();
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ks = 0; ks < KT32; ++ks) asm volatile("" : "+v"(qf[t][h][ks]));
    const bool wave_has_queries = q_first < a.n_q;

    // ---- this wave's share of a stage's 8 * PANELS LDS-DMA wave-instructions: instruction `wave` (rows 4*wave .. 4*wave+3)
    // of every panel-slab
    const int my_src_full = panel_src_off(lane, wave & 3, ROW_BYTES, 16) + wave * 4 * ROW_BYTES;
    const int my_src_last = panel_src_off(lane, wave & 3, ROW_BYTES, 2 * KS_LAST) + wave * 4 * ROW_BYTES;
    const int my_lds = wave * 1024;
    int rd_off[2][kKSteps16];
    slab_rd_offsets16(lane, rd_off);

    // ---- producer cursor over the flattened (document, slab) sequence of [d_lo, d_hi)
    int p_idx = d_lo, p_row = 0, p_len = 0;
    __amdgpu_buffer_rsrc_t p_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)D, 0, 0, 0x00020000);
    auto p_open = [&]() {
        while (p_idx < d_hi) {
            const int r0 = d_off[p_idx], r1 = d_off[p_idx + 1];
            p_len = r1 - r0;
            if (p_len > 0) {
                p_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(D + (size_t)r0 * DIM), 0, p_len * ROW_BYTES, 0x00020000);
                p_row = 0;
                return;
            }
            ++p_idx;
        }
    };
    p_open();
    int p_slot = 0;
    auto produce = [&]() -> bool {
        if (p_idx >= d_hi) return false;
        char *dst = smem + p_slot * kStage;
        const int soff = p_row * ROW_BYTES;               // rows past the document end read as zeros (bounds check)
#pragma unroll
        for (int u = 0; u < PANELS; ++u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(p_rsrc, MSIM_LDS(dst + my_lds + u * kSlabBytes), 16,
                                                     u == PANELS - 1 ? my_src_last : my_src_full, soff + u * kPanelBytes, 0, 0);
        p_slot = (p_slot + 1 == kPanelStages) ? 0 : p_slot + 1;
        p_row += kSlabRows;
        if (p_row >= p_len) {
            ++p_idx;
            p_open();
        }
        return true;
    };
#pragma unroll
    for (int i = 0; i < kPanelStages - 1; ++i) produce();

    const bool ref_bf16 = (a.flags & kFlagRefBf16) != 0;
    int c_slot = 0;

    for (int c_idx = d_lo; c_idx < d_hi; ++c_idx) {
        const int len = d_off[c_idx + 1] - d_off[c_idx];
        const int nslab = (len + kSlabRows - 1) / kSlabRows;
        float m[NT][2];
#pragma unroll
        for (int t = 0; t < NT; ++t) m[t][0] = m[t][1] = -INFINITY;

        for (int s = 0; s < nslab; ++s) {
            // my share of this stage has landed once at most (stages - 2) later stages of mine are still in flight
            if (p_idx < d_hi) wait_vmcnt<PANELS * (kPanelStages - 2)>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();   // everyone's share landed; everyone is done reading the previous stage
            produce();                      // refill the stage that was read in the previous iteration
            const char *st = smem + c_slot * kStage;
            c_slot = (c_slot + 1 == kPanelStages) ? 0 : c_slot + 1;
            if (wave_has_queries) {
                TileAcc acc[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int g = 0; g < 2; ++g) acc[t].a[h][g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int p = 0; p < PANELS; ++p)
#pragma unroll
                    for (int ks = 0; ks < kKSteps16; ++ks)
                        if (ks < (p == PANELS - 1 ? KS_LAST / 2 : kKSteps16)) {
                            bf16x8 af[2];
#pragma unroll
                            for (int g = 0; g < 2; ++g) af[g] = *reinterpret_cast<const bf16x8 *>(st + p * kSlabBytes + rd_off[g][ks]);
#pragma unroll
                            for (int t = 0; t < NT; ++t)
#pragma unroll
                                for (int hg = 0; hg < 4; ++hg)
                                    acc[t].a[hg >> 1][hg & 1] =
                                        mfma16<F16>(af[hg & 1], qf[t][hg >> 1][p * kKSteps16 + ks], acc[t].a[hg >> 1][hg & 1]);
                        }
                const int rows_left = len - s * kSlabRows;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if (rows_left < kSlabRows) tile_mask_tail(acc[t], rows_left, lane);
                    tile_fold(m[t], acc[t]);
                }
            }
        }

        if (wave_has_queries) {
            bool clamp = false;
            if (clamp0 != nullptr) {
                const uint64_t addr = reinterpret_cast<uint64_t>(clamp0) + (uint64_t)c_idx;
                clamp = ((scalar_load_u32(addr & ~3ull) >> ((addr & 3) * 8)) & 0xffu) != 0;
            }
            float tile_sum[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) tile_sum[t] = tile_finish<F16>(m[t], clamp, ref_bf16);
            if (lane == 0) {
#pragma unroll
                for (int qq = 0; qq < q_per_wave; ++qq) {
                    float tot = 0.0f;
#pragma unroll
                    for (int tt = 0; tt < TPQ; ++tt) tot += tile_sum[qq * TPQ + tt];
                    if (ref_bf16) tot = round_to_input<F16>(tot);
                    if (q_first + qq < a.n_q) scores[(size_t)(q_first + qq) * a.ld + c_idx] = tot;
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------
// K1bPF -- K1bP on the FLAT token layout of maxsim_common.hpp (round 5): the queries are one token matrix Qt [T, DIM] + token offsets,
// a query block holds WHOLE queries (at most 8 * MAXU units of 16 tokens and 64 queries: one 8-lane group of the 512 threads per query
// in the reduction), its units are dealt to the eight waves round-robin, and the per-token maxima of a document go through the LDS
// table where 8 lanes per query add their query's tokens (reduce_query_tokens: the same order as K1s / K1b, whatever the batch).
// What the [n_q, Lq, DIM] box of K1bP pays for and this form does not: the rows that pad a query to a multiple of 32 tokens (Lq = 40:
// 64 rows in the box, 48 here) and to the longest query of the call (ragged questions, processing_utils.py:86).  Queries of more than
// two 32-token tiles -- which K1bP does not take at all -- are just more units.
// Document stream, stage ring, swizzled panel-slab image and the MFMA chain across the panels are K1bP's; block -> (query block,
// document range) mapping, the one-table epilogue and its barrier rule (a one-slab document gets a barrier of its own between the
// previous document's sums and its writes) are K1b's, with the 32-row slab in the place of K1b's chunk.
// NW / STAGES (round 6): the plan ladder of width 320.  The 8-wave shape is a barrier per 32-row slab with 20 MFMAs per unit behind it:
// right for full blocks (32 units), barrier-bound for the small batches of the HBM-bound end (10 units over 8 waves = 20-40 MFMAs per
// barrier: 4 x 40 tokens reached 0.49 of the HBM bound where width 128 reaches 0.79, profiles/r06_logs/ab_wide_ladder.log).  NW = 4
// (<= 16 units) and NW = 2 (<= 8 units) give every wave up to four units, halve / quarter the barrier's width and -- with a 3-stage
// ring of 72 KiB -- put two workgroups on a CU; a wave issues 8 / NW of the stage's eight 4-row DMA groups.
template <bool F16, int PANELS, int KS_LAST, int MAXU, int NW = kBatchWaves, int STAGES = kPanelStages>
__global__ __launch_bounds__(NW * 64, 2) void maxsim_batch_panels_flat_kernel(const uint16_t *__restrict__ Qt, const uint16_t *__restrict__ D,
                                                                           const int32_t *__restrict__ d_off,
                                                                           const uint8_t *__restrict__ clamp0,
                                                                           float *__restrict__ scores, BatchArgs a) {
    constexpr int KT = (PANELS - 1) * 8 + KS_LAST;
    constexpr int DIM = KT * 16;
    constexpr int ROW_BYTES = DIM * 2;
    constexpr int kStage = PANELS * kSlabBytes;
    constexpr int kGroups = 8 / NW;                        // 4-row DMA groups of a stage per wave
    static_assert(NW == 2 || NW == 4 || NW == 8, "2, 4 or 8 waves share a stage");
    static_assert(KS_LAST % 2 == 0, "the last panel must hold whole k-steps of 32");
    static_assert(MAXU >= 1 && MAXU <= 4, "a wave holds up to four units: 4 x KT / 2 operand registers each");
    constexpr int KT32 = KT / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *const tokmax = smem + STAGES * kStage;                             // per-token max table: NW * MAXU units x 16 tokens x 16 B
    int *const rtab = reinterpret_cast<int *>(tokmax + NW * MAXU * kUnitTok * 16);   // the queries' token ranges: 64 x 2 ints

    // ---- which (query block, document range) is this workgroup?  (K1b's XCD-aware mapping)
    const int sub = a.n_ranges >> 3;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int qblock, range;
    if (sub > 1) { qblock = slot % a.n_qblocks; range = xcd * sub + slot / a.n_qblocks; }
    else         { qblock = slot;               range = xcd; }
    if (qblock >= a.n_qblocks || range >= a.n_ranges) return;
    const long long row0 = d_off[0], total_rows = (long long)d_off[a.n_d] - row0;     // d_off may be a slice of absolute offsets
    const int want_lo = (int)(row0 + (total_rows * range) / a.n_ranges), want_hi = (int)(row0 + (total_rows * (range + 1)) / a.n_ranges);
    const int d_lo = lower_bound_wave(a.n_d, want_lo, lane, [&](int k) { return d_off[k]; });
    const int d_hi = (range + 1 == a.n_ranges) ? a.n_d : lower_bound_wave(a.n_d, want_hi, lane, [&](int k) { return d_off[k]; });
    if (d_lo >= d_hi) return;

    // ---- this block's tokens; block-local unit u lives in wave u % NW (slot u / NW)
    const int qb0 = a.blk_q0[qblock];
    const int qb_n = a.blk_q0[qblock + 1] - qb0;                             // <= 64
    const int tok0 = flat_qoff(a.fq, qb0);
    const int n_tok = flat_qoff(a.fq, qb0 + qb_n) - tok0;                    // <= 16 * NW * MAXU
    const int n_units = (n_tok + kUnitTok - 1) / kUnitTok;
    const int my_nu = wave < n_units ? (n_units - 1 - wave) / NW + 1 : 0;    // wave-uniform

    // ---- this wave's share of a stage's 8 * PANELS LDS-DMA wave-instructions (K1bP): rows 4 g .. + 3 of every panel-slab for the
    // row groups g = wave, wave + NW, ...
    // (arrays of the fixed size 4, of which kGroups are used: with the dependent extent [kGroups] hipcc's HOST pass silently drops the
    // kernel's definition -- the handle stays an undefined symbol of the library; ROCm 7.2)
    int my_src_full[4], my_src_last[4], my_lds[4];
#pragma unroll
    for (int i = 0; i < kGroups; ++i) {
        const int grp = wave + NW * i;
        my_src_full[i] = panel_src_off(lane, grp & 3, ROW_BYTES, 16) + grp * 4 * ROW_BYTES;
        my_src_last[i] = panel_src_off(lane, grp & 3, ROW_BYTES, 2 * KS_LAST) + grp * 4 * ROW_BYTES;
        my_lds[i] = grp * 1024;
    }
    int rd_off[2][kKSteps16];
    slab_rd_offsets16(lane, rd_off);

    // ---- producer cursor over the flattened (document, slab) sequence of [d_lo, d_hi)
    int p_idx = d_lo, p_row = 0, p_len = 0;
    __amdgpu_buffer_rsrc_t p_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)D, 0, 0, 0x00020000);
    auto p_open = [&]() {
        while (p_idx < d_hi) {
            const int r0 = d_off[p_idx], r1 = d_off[p_idx + 1];
            p_len = r1 - r0;
            if (p_len > 0) {
                p_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(D + (size_t)r0 * DIM), 0, p_len * ROW_BYTES, 0x00020000);
                p_row = 0;
                return;
            }
            ++p_idx;
        }
    };
    p_open();
    int p_slot = 0;
    auto produce = [&]() -> bool {
        if (p_idx >= d_hi) return false;
        char *dst = smem + p_slot * kStage;
        const int soff = p_row * ROW_BYTES;               // rows past the document end read as zeros (bounds check)
#pragma unroll
        for (int i = 0; i < kGroups; ++i)
#pragma unroll
            for (int u = 0; u < PANELS; ++u)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(p_rsrc, MSIM_LDS(dst + my_lds[i] + u * kSlabBytes), 16,
                                                         u == PANELS - 1 ? my_src_last[i] : my_src_full[i], soff + u * kPanelBytes, 0, 0);
        p_slot = (p_slot + 1 == STAGES) ? 0 : p_slot + 1;
        p_row += kSlabRows;
        if (p_row >= p_len) {
            ++p_idx;
            p_open();
        }
        return true;
    };
#pragma unroll
    for (int i = 0; i < STAGES - 1; ++i) produce();

    // ---- the block's units (B operands, [unit][k-step of 32]), loaded behind the first stages' LDS-DMA requests
    bf16x8 qf[MAXU][KT32];
#pragma unroll
    for (int t = 0; t < MAXU; ++t) {
        const int row = (wave + NW * t) * kUnitTok + (lane & 15);
        const bool valid = t < my_nu && row < n_tok;
        const uint16_t *p = Qt + ((size_t)tok0 + (valid ? row : 0)) * DIM + (lane >> 4) * 8;
#pragma unroll
        for (int ks = 0; ks < KT32; ++ks) {
            const bf16x8 v = *reinterpret_cast<const bf16x8 *>(p + ks * 32);
            qf[t][ks] = valid ? v : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
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
        for (int ks = 0; ks < KT32; ++ks) asm volatile("" : "+v"(qf[t][ks]));

    const bool ref_bf16 = (a.flags & kFlagRefBf16) != 0;
    const bool round_total = ref_bf16 && !(a.flags & kFlagPartial);
    int c_slot = 0;

    auto reduce_doc = [&](int doc, bool clamp) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));        // opaque: nothing derived from it stays live across the slab loop
        const int rq = tid >> 3, ri = tid & 7;
        if (rq < qb_n) {
            float tot = reduce_query_tokens<F16>(tokmax, rtab[2 * rq], rtab[2 * rq + 1], ri, clamp, ref_bf16);
            if (round_total) tot = round_to_input<F16>(tot);
            if (ri == 0) scores[(size_t)(qb0 + rq) * a.ld + doc] = tot;
        }
    };
    auto lds_barrier = [&]() {               // a barrier that also orders this wave's table writes
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    auto run = [&](auto nu_c) {
    constexpr int NU = decltype(nu_c)::value;
    constexpr int NUA = NU > 0 ? NU : 1;
    int pend_doc = -1;                                      // document whose maxima wait in the table (the same in all waves)
    bool pend_clamp = false;
    for (int c_idx = d_lo; c_idx < d_hi; ++c_idx) {
        const int len = d_off[c_idx + 1] - d_off[c_idx];
        const int nslab = (len + kSlabRows - 1) / kSlabRows;
        bool clamp = false;
        if (clamp0 != nullptr) {
            const uint64_t addr = reinterpret_cast<uint64_t>(clamp0) + (uint64_t)c_idx;
            clamp = ((scalar_load_u32(addr & ~3ull) >> ((addr & 3) * 8)) & 0xffu) != 0;
        }
        if (nslab == 0) {           // a document without rows never enters the ring: every token's max is over nothing (-inf, or 0 under clamp0)
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            const int rq = tid >> 3;
            if (rq < qb_n && (tid & 7) == 0)
                scores[(size_t)(qb0 + rq) * a.ld + c_idx] = (rtab[2 * rq + 1] > rtab[2 * rq] && !clamp) ? -INFINITY : 0.0f;
            continue;
        }
        float m[NUA];
#pragma unroll
        for (int t = 0; t < NUA; ++t) m[t] = -INFINITY;
        for (int s = 0; s < nslab; ++s) {
            // my share of this stage has landed once at most (stages - 2) later stages of mine are still in flight
            if (p_idx < d_hi) wait_vmcnt<PANELS * kGroups * (STAGES - 2)>(); else wait_vmcnt<0>();
            lds_barrier();                  // everyone's share landed; everyone is done reading the previous stage (and has written its maxima)
            produce();                      // refill the stage that was read in the previous iteration
            if (s == 0 && pend_doc >= 0) reduce_doc(pend_doc, pend_clamp);   // the previous document's token sums, behind its barrier
            const char *st = smem + c_slot * kStage;
            c_slot = (c_slot + 1 == STAGES) ? 0 : c_slot + 1;
            if constexpr (NU > 0) {
                UnitAcc acc[NUA];
#pragma unroll
                for (int t = 0; t < NUA; ++t)
#pragma unroll
                    for (int g = 0; g < 2; ++g) acc[t].a[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int p = 0; p < PANELS; ++p)
#pragma unroll
                    for (int ks = 0; ks < kKSteps16; ++ks)
                        if (ks < (p == PANELS - 1 ? KS_LAST / 2 : kKSteps16)) {
                            bf16x8 af[2];
#pragma unroll
                            for (int g = 0; g < 2; ++g) af[g] = *reinterpret_cast<const bf16x8 *>(st + p * kSlabBytes + rd_off[g][ks]);
#pragma unroll
                            for (int t = 0; t < NUA; ++t)
#pragma unroll
                                for (int g = 0; g < 2; ++g)
                                    acc[t].a[g] = mfma16<F16>(af[g], qf[t][p * kKSteps16 + ks], acc[t].a[g]);
                        }
                const int rows_left = len - s * kSlabRows;
#pragma unroll
                for (int t = 0; t < NUA; ++t) {
                    if (rows_left < kSlabRows) unit_mask_tail(acc[t], rows_left, lane);
                    unit_fold(m[t], acc[t]);
                }
            }
        }
        // ---- document epilogue: this wave's maxima into the workgroup's ONE table; the sums are taken behind the next barrier.  A
        // one-slab document has no barrier between the previous document's sums and these writes, so it gets one
        if (nslab == 1) lds_barrier();
        if constexpr (NU > 0) {
#pragma unroll
            for (int t = 0; t < NUA; ++t) store_token_max(tokmax, wave + NW * t, m[t], lane);
        }
        pend_doc = c_idx;
        pend_clamp = clamp;
    }
    if (pend_doc >= 0) {
        lds_barrier();
        reduce_doc(pend_doc, pend_clamp);
    }
    };   // run

    switch (my_nu) {                                       // wave-uniform; every body executes the same barriers
        case 0: run(std::integral_constant<int, 0>{}); break;
        case 1: run(std::integral_constant<int, 1>{}); break;
        case 2: if constexpr (MAXU >= 2) run(std::integral_constant<int, 2>{}); break;
        case 3: if constexpr (MAXU >= 3) run(std::integral_constant<int, 3>{}); break;
        default: if constexpr (MAXU >= 4) run(std::integral_constant<int, 4>{}); break;
    }
}

}  // namespace msim
