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

}  // namespace msim
