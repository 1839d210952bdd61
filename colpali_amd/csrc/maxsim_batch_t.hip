// K1t -- the TRANSPOSED batch kernel for gfx950 (MI355X): long queries against many SHORT documents, all pairs.
//
// The reference trainer's symmetric direction (trainer/contrastive_trainer.py:202-206, compute_symetric_loss) calls the loss with the
// pages as `query_embeddings` [B, 780, 128] and the gathered queries as `doc_embeddings` [C, 32, 128]:
//     scores[p, c] = sum over the page's rows r of  max over the query's tokens t of  <P[p, r, :], Q[c, t, :]>
//     (colpali_engine/loss/late_interaction_losses.py:297-298: einsum("bnd,csd->bcns") -> amax(dim=3) -> sum(dim=2))
// -- the same 52 GFLOP as the forward direction at BASELINE config 5's shape, reduced the other way round.  K1b streams documents and
// keeps query tokens resident: here that is 256 "documents" of ONE slab each under a 780-token "query" cut into seven pieces, and every
// 32-row document pays a chunk barrier, a table write and a pass of token sums (104 us for 21 us of MFMA work, rocprofv3, round 5).
//
// K1t swaps what streams and what is resident, and the MFMA operand roles with it:
//   * the PAGE streams through the workgroup's LDS ring exactly like a K1b document (128-row chunks, LDS-DMA, swizzled slab image, one
//     raw s_barrier per chunk, counted vmcnt);
//   * the short documents' rows are the RESIDENT operand: a document of Ld <= 16 U rows is U units of 16 rows (16 VGPRs each), a wave
//     holds DPW = 8 / U whole documents, a block of 8 waves 8 DPW documents;
//   * v_mfma_f32_16x16x32 with A = the resident rows, B = the streamed page rows: D puts one PAGE ROW per lane column (l & 15) and
//     four resident rows in the lane's registers, so "max over the document's rows" is an in-lane fold over the document's units and
//     ONE exchange across the four lane groups (xor 16, xor 32) per (document, 16 page rows) -- and the sum over the page's rows is a
//     running per-lane sum, folded over the 16 lanes once per page.  No token table, no cross-wave reduction, nothing per document.
// Determinism: a (page, document) score is summed in an order fixed by the page length alone (row groups of 16 in order, then the
// xor-8/4/2/1 butterfly): independent of which other documents or pages share the launch.
#pragma once
#include <type_traits>

#include "maxsim_batch.hip"
#include "maxsim_common.hpp"
#include "maxsim_stream.hip"

namespace msim {

// max / min over the four 16-lane groups of a wave (lanes l, l ^ 16, l ^ 32, l ^ 48), the result in all four -- on the VALU:
// v_permlane16_swap / v_permlane32_swap exchange whole 16- / 32-lane rows between two copies of the register, so max(copy0, copy1)
// IS the pair maximum in every lane.  (__shfl_xor compiles to ds_bpermute_b32: an LDS-pipe round trip per step, 16 per slab here.)
__device__ __forceinline__ float group_max4(float v) {
    uint32_t u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = __builtin_fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    u = __float_as_uint(v);
    r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ uint32_t group_min4(uint32_t u) {
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    u = r[0] < r[1] ? r[0] : r[1];
    r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return r[0] < r[1] ? r[0] : r[1];
}

struct BatchTArgs {
    long long ld;       // leading dimension of scores [n_q, ld]
    int n_q, Lq;        // streamed side: n_q pages of Lq rows each (a dense box)
    int n_d, Ld;        // resident side: n_d documents of Ld rows each (a dense box), Ld <= 16 * U
    int n_blocks;       // document blocks of 8 * DPW documents
    int n_slots;        // page slots per XCD: the workgroup of (xcd, slot) walks pages xcd + 8 * (slot / n_blocks), then += 8 * slots_p
    int slots_p;        // page slots per XCD = n_slots / n_blocks
    int Lq_pad;         // ROUTE: bytes per (page, document) of the routing = Lq rounded up to 32
};

// U   : 16-row units per resident document (Ld <= 16 * U)
// DPW : documents per wave (U * DPW <= 8 units = 128 operand registers)
// ROUTE (round 6): also write the ROUTING of every (page, document, page row) -- the resident row that won the max, one byte,
//   route[(page * n_d + doc) * Lq_pad + row], Lq_pad = Lq rounded up to 64, bytes of rows in [Lq, Lq_pad) unspecified -- for the dense
//   hard-max backward on the matrix cores (maxsim_dense_t.hip: ColbertLoss / ColbertSigmoidLoss in the trainer's symmetric direction,
//   late_interaction_losses.py:140-164, contrastive_trainer.py:202-206).  The first maximal row wins a tie (maxsim_pairs.hip's rule);
//   the scores are bit-identical to the ROUTE = false form (the same MFMAs and the same max3 chain; the row is found by comparing the
//   lane's candidates with the exchanged maximum).
template <bool F16, int U, int DPW, bool ROUTE = false>
__global__ __launch_bounds__(512, 2) void maxsim_batch_t_kernel(const uint16_t *__restrict__ Q, const uint16_t *__restrict__ D,
                                                                float *__restrict__ scores, int32_t *__restrict__ q_lengths,
                                                                uint8_t *__restrict__ route, BatchTArgs a) {
    static_assert(U * DPW <= 8 && U >= 1 && DPW >= 1, "a wave holds at most 8 units");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = 8;
    constexpr int kRing = 3;
    constexpr int kCSlabs = NW / 2;                         // 4 slabs = 128 page rows per chunk
    constexpr int kCRows = kCSlabs * kSlabRows;
    constexpr int kCBytes = kCSlabs * kSlabBytes;
    constexpr int NU = U * DPW;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    if (slot >= a.n_slots) return;
    const int block = slot % a.n_blocks;
    const int pslot = slot / a.n_blocks;
    const int doc0 = (block * NW + wave) * DPW;              // this wave's first document

    QueryUnit qu[NU];           // the resident rows (loaded below, behind the first page's first LDS-DMA requests)
    const int l16 = lane & 15, l4 = lane >> 4;
    int src_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) src_off[j] = l4 * kRowBytes + (((l16 ^ l4) ^ (j << 2)) << 4);
    int rd_off[2][kKSteps16];
    slab_rd_offsets16(lane, rd_off);
    const int my_lds_off = (wave >> 1) * kSlabBytes + (wave & 1) * 4096;
    const int my_row_off = (wave >> 1) * kSlabRows + (wave & 1) * 16;
    // resident rows beyond the document's end (the last unit when Ld is not a multiple of 16) are zero rows in the registers: their
    // similarities are masked to -inf (a real similarity may be negative)
    // (round 6, measured and NOT kept at config 5's shape, 32 pages x 256 documents of 32 rows: a template form without this
    // wave-uniform branch behind every unit's 8 MFMAs -- the slab body as ONE basic block, 130 instead of 356 VALU per 64 MFMAs --
    // 50.9 -> 55.0 us; two documents per wave with a two-chunk ring, i.e. two workgroups per CU, 50.9 -> 56.7 us.  The kernel is
    // bound by its per-chunk latency chain, not by instruction issue or occupancy.)
    const bool need_mask = a.Ld != U * kUnitTok;            // wave-uniform
    const int row_lim = a.Ld - 4 * l4;                      // resident row 16 u + 4 l4 + r exists iff 16 u + r < row_lim

    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    // optional by-product: q_lengths[page] = rows of the page whose FIRST component is non-zero -- the `lengths` of
    // late_interaction_losses.py:296, which the loss epilogue would otherwise collect with one cache line per row on ONE CU (21 us for
    // 32 pages of 780 rows).  Every row passes through this workgroup's LDS anyway: wave 0 of document block 0 counts.
    const bool count_here = q_lengths != nullptr && block == 0 && wave == 0;
    const int first_off = (lane & 31) * kRowBytes + ((lane & 15) << 4);      // chunk 0 of row (lane & 31) in the swizzled slab image
    int p_slot = 0, c_slot = 0;
    const int nchunk = (a.Lq + kCRows - 1) / kCRows;
    int page = xcd + 8 * pslot;
    if (page >= a.n_q) return;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(Q + (size_t)page * a.Lq * kDim), 0, a.Lq * kRowBytes, 0x00020000);
    int p_ch = 0;
    auto produce = [&]() {
        if (p_ch >= nchunk) return;
        char *dst = smem + p_slot * kCBytes + my_lds_off;
        const int soff = (p_ch * kCRows + my_row_off) * kRowBytes;       // rows past the page end read as zeros (bounds check)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, MSIM_LDS(dst + j * 1024), 16, src_off[j], soff + j * 1024, 0, 0);
        p_slot = (p_slot + 1 == kRing) ? 0 : p_slot + 1;
        ++p_ch;
    };
    produce();                                                  // the first page's ring fill (two chunks ahead) ...
    produce();

    // ---- ... and BEHIND it (both in flight together) the resident rows: unit (d, u) = rows 16 u .. 16 u + 15 of document doc0 + d,
    // as MFMA A operands
#pragma unroll
    for (int d = 0; d < DPW; ++d)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = doc0 + d < a.n_d;
            load_query_unit(qu[d * U + u], D + (size_t)(live ? doc0 + d : 0) * a.Ld * kDim, u * kUnitTok, a.Ld, lane, live);
        }
    wait_vmcnt<0>();
#pragma unroll
    for (int t = 0; t < NU; ++t)
#pragma unroll
        for (int ks = 0; ks < kKSteps16; ++ks) asm volatile("" : "+v"(qu[t].f[ks]));

    for (;;) {
        float sum[DPW];
#pragma unroll
        for (int d = 0; d < DPW; ++d) sum[d] = 0.0f;
        int n_real = 0;
        uint8_t *rbase[DPW];                    // ROUTE: this page's routing bytes of the wave's documents
        if constexpr (ROUTE) {
#pragma unroll
            for (int d = 0; d < DPW; ++d) rbase[d] = route + ((size_t)page * a.n_d + (doc0 + d < a.n_d ? doc0 + d : 0)) * a.Lq_pad;
        }

        auto slab = [&](int src_lds, auto tail, int rows_left, int row0) {
            constexpr bool kTail = decltype(tail)::value;
            bf16x8 af[2][kKSteps16];
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int ks = 0; ks < kKSteps16; ++ks) af[g][ks] = *reinterpret_cast<const bf16x8 *>(smem + src_lds + rd_off[g][ks]);
            if (count_here) {      // rows beyond the page end arrived as zeros (bounds-checked descriptor): they count as padding
                const uint16_t first = *reinterpret_cast<const uint16_t *>(smem + src_lds + first_off);
                n_real += __popcll(__ballot(lane < 32 && (first & 0x7fffu) != 0));
            }
            float x[DPW][2];
#pragma unroll
            for (int d = 0; d < DPW; ++d) {
                float m0 = -INFINITY, m1 = -INFINITY;
                f32x4 keep0[U], keep1[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < kKSteps16; ++ks) {
                        acc0 = mfma16<F16>(qu[d * U + u].f[ks], af[0][ks], acc0);
                        acc1 = mfma16<F16>(qu[d * U + u].f[ks], af[1][ks], acc1);
                    }
                    if (need_mask && (u + 1) * kUnitTok > a.Ld) {        // a unit that reaches beyond the document (wave-uniform branch)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (u * kUnitTok + r >= row_lim) { acc0[r] = -INFINITY; acc1[r] = -INFINITY; }
                        }
                    }
                    m0 = max3(m0, acc0[0], acc0[1]);
                    m0 = max3(m0, acc0[2], acc0[3]);
                    m1 = max3(m1, acc1[0], acc1[1]);
                    m1 = max3(m1, acc1[2], acc1[3]);
                    if constexpr (ROUTE) {
                        // this lane's candidates of the unit, kept until the document's maximum is known (below): 8 registers per unit
                        keep0[u] = acc0;
                        keep1[u] = acc1;
                    }
                }
                if constexpr (ROUTE) {
                    // (max, lowest maximal row) over the four lane groups, at once (nothing is kept across documents: the routing costs
                    // registers the plain form spends on hiding the exchange behind the next document's MFMAs)
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const float M = group_max4(g == 0 ? m0 : m1);
                        // the lowest row of THIS lane whose similarity is the maximum (descending scan: the first one stays), else 255
                        uint32_t vi = 255u;
#pragma unroll
                        for (int u = U - 1; u >= 0; --u)
#pragma unroll
                            for (int r = 3; r >= 0; --r) vi = (g == 0 ? keep0[u][r] : keep1[u][r]) == M ? (uint32_t)(u * kUnitTok + r + 4 * l4) : vi;
                        vi = group_min4(vi);
                        // lane group d & 3 stores document d's byte of page row row0 + 16 g + l16 (16 consecutive bytes per group)
                        const int row = row0 + 16 * g + l16;
                        if (l4 == (d & 3) && doc0 + d < a.n_d && row < a.Lq_pad) rbase[d][row] = row < a.Lq ? (uint8_t)vi : (uint8_t)255;
                        float v = M;
                        if constexpr (kTail) v = (16 * g + l16 < rows_left) ? v : 0.0f;
                        sum[d] += v;
                    }
                } else {
                    x[d][0] = m0;
                    x[d][1] = m1;
                }
            }
            // one exchange across the four lane groups per (document, 16 page rows), behind all of the slab's MFMAs
            if constexpr (!ROUTE) {
#pragma unroll
                for (int d = 0; d < DPW; ++d)
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        float v = group_max4(x[d][g]);
                        if constexpr (kTail) v = (16 * g + l16 < rows_left) ? v : 0.0f;   // page rows that do not exist add nothing
                        sum[d] += v;
                    }
            }
        };

        for (int ch = 0; ch < nchunk; ++ch) {
            if (p_ch < nchunk) wait_vmcnt<4 * (kRing - 2)>(); else wait_vmcnt<0>();
            lds_barrier();                  // everyone's share landed; everyone is done reading the previous chunk
            produce();
            const int cbuf = c_slot * kCBytes;
            c_slot = (c_slot + 1 == kRing) ? 0 : c_slot + 1;
            const int rows_in_chunk = a.Lq - ch * kCRows;
            const int n_full = rows_in_chunk >= kCRows ? kCSlabs : rows_in_chunk / kSlabRows;
#pragma unroll 1
            for (int sl = 0; sl < n_full; ++sl) slab(cbuf + sl * kSlabBytes, std::false_type{}, kSlabRows, ch * kCRows + sl * kSlabRows);
            const int rem = rows_in_chunk - n_full * kSlabRows;
            if (n_full < kCSlabs && rem > 0) slab(cbuf + n_full * kSlabBytes, std::true_type{}, rem, ch * kCRows + n_full * kSlabRows);
        }
        // ---- page done: fold the 16 page-row lanes (every lane group holds the same sums), one store per document
#pragma unroll
        for (int d = 0; d < DPW; ++d) {
            float s = sum[d];
            s += __shfl_xor(s, 8);
            s += __shfl_xor(s, 4);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 1);
            if (lane == 0 && doc0 + d < a.n_d) scores[(size_t)page * a.ld + doc0 + d] = s;
        }
        if (count_here && lane == 0) q_lengths[page] = n_real;
        page += 8 * a.slots_p;
        if (page >= a.n_q) break;
        lds_barrier();                      // the ring is re-filled for the next page only after every wave has read its last chunk
        rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(Q + (size_t)page * a.Lq * kDim), 0, a.Lq * kRowBytes, 0x00020000);
        p_ch = 0;
        static_assert(kRing == 3, "two chunks are requested ahead");
        produce();
        produce();
    }
}

}  // namespace msim
