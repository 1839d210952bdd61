// K1s -- "stream" MaxSim kernel for gfx950 (MI355X), the HBM-bound regime:
// a handful of queries (<= 8 token tiles of 32 tokens) scored against a large
// resident corpus.  Every byte of the corpus is read from HBM exactly once.
//
// Reference arithmetic replaced (no [b,c,n,s] tensor is ever materialised):
//   colpali_engine/utils/processing_utils.py:179
//       einsum("bnd,csd->bcns", Q, D).max(dim=3)[0].sum(dim=2)
//
// Structure (one wave = one independent pipeline, no workgroup barriers at all):
//   * the queries' tokens -- ONE flat token matrix, every query's real tokens back to back
//     (maxsim_common.hpp: the flat layout) -- live in registers for the whole kernel as the MFMA
//     B operand (16 VGPRs per 16-token unit, at most 8 units);
//   * each wave walks its own documents; a document is streamed in 32-patch
//     slabs (8 KiB) into a wave-private LDS ring (RING slabs: 4, or 2 for 3-4 token tiles
//     so that two workgroups share a CU -- see launch_stream in maxsim_abi.hip) by LDS-DMA
//     (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction, fully
//     coalesced, bounds-checked by a per-document buffer descriptor);
//   * the slab image is XOR-swizzled on the SOURCE address so the ds_read_b128
//     operand fetches are bank-conflict free (cdna_hip_programming.md T2 /
//     rule 21: linear destination, swizzled source, same swizzle on the read);
//   * swapped product mfma_f32_16x16x32_bf16(D_slab rows, Q^T) (maxsim_common.hpp: the
//     16x16x32 tiling -- half the accumulator traffic of 32x32x16, which is what counts
//     on a chip that clocks to its power budget): the C layout puts one query token per
//     lane column, so the max over patches is 4 v_max3 per 16-token unit and slab in
//     registers; per document the maxima pass through a wave-private LDS table where 8 lanes
//     per query add that query's tokens (whatever units they sit in) and one lane stores.
#pragma once
#include <type_traits>

#include "maxsim_common.hpp"

namespace msim {

struct StreamArgs {
    long long ld;   // leading dimension of scores
    FlatQ fq;       // where the queries sit in the flat token matrix
    int n_q;        // queries (<= 8: one 8-lane group of every wave per query)
    int n_d;
    unsigned flags;
};

constexpr unsigned kFlagRefBf16 = 1u;
constexpr unsigned kFlagPartial = 2u;     // internal (long queries scored in 128-token pieces): the token sum is a PARTIAL sum, not rounded here

constexpr int kStreamMaxUnits = 8;      // (nine / ten units with eight of them in AGPRs, as in K1b's ten-unit form, were tried in round 4:
                                        // next to the interleaved LDS-DMA issue the slab body does not fit 128 VGPRs and spills into the counted-vmcnt stream)
constexpr int kStreamTokBytes = kStreamMaxUnits * kUnitTok * 16 + 64;   // wave-private: per-token max table (8 units x 16 tokens x 4 lane groups x 4 B) + 8 token ranges

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// NU  : 16-token units of the flat token matrix held by every wave (1..8) = ceil(total tokens / 16)
// RING: slabs in the wave-private LDS ring
// F16 : embeddings are IEEE half instead of bfloat16
// AUX : cache-policy bits of the LDS-DMA loads (0 = default, 2 = nt: streamed once, do not keep in L2 / MALL)
// IL  : issue the 8 LDS-DMA pieces of the next slab BETWEEN the MFMAs of the current one instead of in a block in
//       front of them (the matrix pipe idles while a block of DMA instructions issues; one piece per NU MFMAs hides)
// At most 8 units (4 token tiles of 32): launch bounds ask for two waves per SIMD (<= 256 registers), which the 2-slab ring needs to
// put two workgroups on a CU; without the bound hipcc gives every unit an accumulator of its own and lands at 290 registers.
template <int NU, int RING, bool F16, int AUX = 0, bool IL = false>
__global__ __launch_bounds__(256, 2) void maxsim_stream_kernel(const uint16_t *__restrict__ Qt,      // [T, 128] flat query tokens
                                                            const uint16_t *__restrict__ D,       // [rows, 128] bf16
                                                            const int32_t *__restrict__ d_off,    // [n_d + 1]
                                                            const uint8_t *__restrict__ clamp0,   // [n_d] or null
                                                            float *__restrict__ scores,           // [n_q, ld]
                                                            StreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *ring = smem + wave * (RING * kSlabBytes);
    char *tokmax = smem + 4 * (RING * kSlabBytes) + wave * kStreamTokBytes;
    const int gw = blockIdx.x * 4 + wave;  // global wave id: this wave owns documents gw, gw+GW, ...
    const int GW = gridDim.x * 4;

    // ---- query fragments: B operands, resident for the whole kernel (16 VGPRs per 16-token unit)
    const int n_tok = flat_qoff(a.fq, a.n_q);
    QueryUnit qu[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) load_query_unit(qu[u], Qt, u * kUnitTok, n_tok, lane, true);
    // 8 lanes per query in the reduction; the queries' token ranges wait in LDS behind the table, written here by the lanes that read
    // them back (nothing derived from the lane id needs to stay in registers across the slab loop)
    int *const rtab = reinterpret_cast<int *>(tokmax + kStreamMaxUnits * kUnitTok * 16);
    if ((lane >> 3) < a.n_q) {
        rtab[2 * (lane >> 3)] = flat_qoff(a.fq, lane >> 3);
        rtab[2 * (lane >> 3) + 1] = flat_qoff(a.fq, (lane >> 3) + 1);
    }

    // the query loads are ordinary VMEM loads: retire them before the LDS-DMA stream starts so that the
    // compiler's own vmcnt waits for them never drain the ring later on
    wait_vmcnt<0>();
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int ks = 0; ks < kKSteps16; ++ks) asm volatile("" : "+v"(qu[u].f[ks]));

    // ---- per-lane address constants
    // LDS-DMA source: wave-instruction i of a slab fills LDS rows 4i..4i+3 linearly; lane (l4 = lane>>4,
    // l16 = lane&15) lands on physical chunk l16 of row 4i+l4, which must hold logical chunk
    // l16 ^ (row & 15) = l16 ^ l4 ^ ((i&3)<<2).
    const int l16 = lane & 15, l4 = lane >> 4;
    int src_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) src_off[j] = l4 * kRowBytes + (((l16 ^ l4) ^ (j << 2)) << 4);
    // operand fetch: fragment (g, ks) = rows 16g + (lane & 15), logical chunk 4 * ks + (lane >> 4)
    int rd_off[2][kKSteps16];
    slab_rd_offsets16(lane, rd_off);

    // ---- producer cursor (wave-uniform): next slab to request
    int p_idx = gw, p_row = 0, p_len = 0;
    __amdgpu_buffer_rsrc_t p_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)D, 0, 0, 0x00020000);
    auto p_open = [&]() {  // position on the next non-empty document at or after p_idx
        while (p_idx < a.n_d) {
            const int r0 = d_off[p_idx], r1 = d_off[p_idx + 1];
            p_len = r1 - r0;
            if (p_len > 0) {
                p_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(D + (size_t)r0 * kDim), 0,
                                                           p_len * kRowBytes, 0x00020000);
                p_row = 0;
                return;
            }
            p_idx += GW;
        }
    };
    p_open();
    int p_slot = 0;
    auto produce = [&]() -> bool {  // returns false when the stream is exhausted (nothing issued)
        if (p_idx >= a.n_d) return false;
        char *dst = ring + p_slot * kSlabBytes;
        const int soff = p_row * kRowBytes;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(p_rsrc, MSIM_LDS(dst + i * 1024), 16, src_off[i & 3],
                                                     soff + i * 1024, 0, AUX);
        p_slot = (p_slot + 1 == RING) ? 0 : p_slot + 1;
        p_row += kSlabRows;
        if (p_row >= p_len) {
            p_idx += GW;
            p_open();
        }
        return true;
    };

    // IL: every request is 8 loads, real or through an empty descriptor (out-of-range lanes fetch nothing and count like any
    // other load), so the number of outstanding loads is a constant and the instruction stream has no branches
    const __amdgpu_buffer_rsrc_t null_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)D, 0, 0, 0x00020000);
    auto advance = [&](bool live) {
        p_slot = (p_slot + 1 == RING) ? 0 : p_slot + 1;
        if (live) {
            p_row += kSlabRows;
            if (p_row >= p_len) {
                p_idx += GW;
                p_open();
            }
        }
    };
    // prologue: RING-1 slabs in flight
    if constexpr (IL) {
#pragma unroll
        for (int k = 0; k < RING - 1; ++k) {
            const bool live = p_idx < a.n_d;
            const __amdgpu_buffer_rsrc_t rs = live ? p_rsrc : null_rsrc;
            char *dst = ring + p_slot * kSlabBytes;
            const int soff = live ? p_row * kRowBytes : 0;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, MSIM_LDS(dst + i * 1024), 16, src_off[i & 3], soff + i * 1024, 0, AUX);
            advance(live);
        }
    } else {
#pragma unroll
        for (int i = 0; i < RING - 1; ++i) produce();
    }

    const bool ref_bf16 = (a.flags & kFlagRefBf16) != 0;
    int c_slot = 0;
    for (int c_idx = gw; c_idx < a.n_d; c_idx += GW) {
        const int len = d_off[c_idx + 1] - d_off[c_idx];
        float m[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) m[u] = -INFINITY;

        // one slab: request the next one (IL: its 8 pieces go out between the MFMAs), fetch the 8 operand fragments once, then the
        // MFMAs of all units (maxsim_common.hpp: slab_units, folds one group late).  kTail (rows past the document end masked to
        // -inf) is a compile-time variant so that the full-slab body has no branch in it.
        auto slab = [&](auto tail_c, int rows_left) {
            constexpr bool kTail = decltype(tail_c)::value;
            // next slab to request (its slot is the one consumed in the previous iteration: free again)
            bool nx_live = false;
            char *nx_dst = ring;
            int nx_soff = 0;
            __amdgpu_buffer_rsrc_t nx_rsrc = p_rsrc;
            if constexpr (IL) {
                wait_vmcnt<8 * (RING - 2)>();     // RING - 1 requests are outstanding: all but the oldest may stay in flight
                nx_live = p_idx < a.n_d;
                nx_dst = ring + p_slot * kSlabBytes;
                nx_soff = nx_live ? p_row * kRowBytes : 0;
                nx_rsrc = nx_live ? p_rsrc : null_rsrc;
            } else {
                // the slot consumed in the previous iteration is free again: refill it, then wait for this slab
                const bool issued = produce();
                if (issued)
                    wait_vmcnt<8 * (RING - 1)>();
                else
                    wait_vmcnt<0>();
            }

            const char *src = ring + c_slot * kSlabBytes;
            c_slot = (c_slot + 1 == RING) ? 0 : c_slot + 1;
            bf16x8 af[2][kKSteps16];
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int ks = 0; ks < kKSteps16; ++ks) af[g][ks] = *reinterpret_cast<const bf16x8 *>(src + rd_off[g][ks]);
            slab_units<F16, NU, kTail, true>(m, af, qu, rows_left, lane, [&](int mf) {
                if constexpr (IL) {                 // one DMA piece per NU MFMAs: 8 per slab (a slab is 8 * NU MFMAs)
                    if (mf % NU == 0) {
                        const int i = mf / NU;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(nx_rsrc, MSIM_LDS(nx_dst + i * 1024), 16, src_off[i & 3],
                                                                 nx_soff + i * 1024, 0, AUX);
                    }
                }
            });
            if constexpr (IL) advance(nx_live);                                  // the 8 pieces are out: advance the cursor
        };
        const int n_full = len / kSlabRows, rem = len - n_full * kSlabRows;
        for (int s = 0; s < n_full; ++s) slab(std::false_type{}, kSlabRows);
        if (rem > 0) slab(std::true_type{}, rem);

        // ---- document epilogue: the per-token maxima go to the wave's table in LDS (no barrier: the table is wave-private and a
        // wave's LDS operations complete in order), then 8 lanes per query add their query's tokens and one of them stores.
        // clamp0 is a byte array; fetch the aligned dword around the byte with an explicit scalar load
        // (a vector byte load would make the compiler wait vmcnt(0), i.e. drain the whole LDS-DMA ring)
        bool clamp = false;
        if (clamp0 != nullptr) {
            const uint64_t addr = reinterpret_cast<uint64_t>(clamp0) + (uint64_t)c_idx;
            clamp = ((scalar_load_u32(addr & ~3ull) >> ((addr & 3) * 8)) & 0xffu) != 0;
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) store_token_max(tokmax, u, m[u], lane);
        int ln = lane;
        asm volatile("" : "+v"(ln));            // opaque: no row pointer or table address is hoisted out of the document loop
        const int rq = ln >> 3, ri = ln & 7;
        if (rq < a.n_q) {
            float tot = reduce_query_tokens<F16>(tokmax, rtab[2 * rq], rtab[2 * rq + 1], ri, clamp, ref_bf16);
            if (ref_bf16 && !(a.flags & kFlagPartial)) tot = round_to_input<F16>(tot);
            if (ri == 0) scores[(size_t)rq * a.ld + c_idx] = tot;
        }
    }
}

}  // namespace msim
