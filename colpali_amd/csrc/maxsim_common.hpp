// Shared device helpers for the gfx950 MaxSim kernels (CDNA4 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace msim {

// s_memtime phase traces inside K1b / K3 (tools/trace_batch.py, tools/trace_head.py) exist only in `make trace` builds
#ifdef MSIM_TRACE
constexpr bool kTraceBuild = true;
#else
constexpr bool kTraceBuild = false;
#endif
// A/B knobs (MSIM_* environment variables), kernel variants that were measured and not kept, and the kernels behind them exist
// only in measurement builds (`make ab`, `make trace` -> tools/_ab/): the shipped library never reads the environment
#if defined(MSIM_AB) || defined(MSIM_TRACE)
constexpr bool kAbBuild = true;
#else
constexpr bool kAbBuild = false;
#endif

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 bf16 = one MFMA A/B operand (4 VGPRs)
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 MFMA accumulator fragment
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// Embedding element types the kernels are instantiated for (16-bit, MFMA K=16).  The register/LDS images are
// identical (8 elements = 16 B per lane), only the MFMA opcode and the scalar conversions differ.
constexpr int kDtypeBf16 = 0, kDtypeF16 = 1;

constexpr int kDim = 128;                 // embedding width the kernels are built for
constexpr int kRowBytes = kDim * 2;       // one bf16 patch row = 256 B = one LDS bank row
constexpr int kSlabRows = 32;             // MFMA M: patches per slab
constexpr int kSlabBytes = kSlabRows * kRowBytes;  // 8 KiB
constexpr int kKSteps = kDim / 16;        // 8 x (32x32x16) MFMAs per 32x32 output tile
constexpr int kTokTile = 32;              // MFMA N: query tokens per tile

#define MSIM_LDS(p) ((__attribute__((address_space(3))) void *)(p))

// v_max3_f32.  Written with the builtin, NOT as inline asm: the compiler's hazard recognizer does not look inside an asm
// statement, so an asm v_max3 that reads an accumulator straight after the MFMA that writes it gets no wait states (an
// 8-pass MFMA needs 12 before a VALU read; the hardware does not interlock) and reads a partial sum.
__device__ __forceinline__ float max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

// max over the 16 accumulator registers of one lane, folded into a running max
__device__ __forceinline__ float fold_max16(float m, const f32x16 &c) {
    m = max3(m, c[0], c[1]);
    m = max3(m, c[2], c[3]);
    m = max3(m, c[4], c[5]);
    m = max3(m, c[6], c[7]);
    m = max3(m, c[8], c[9]);
    m = max3(m, c[10], c[11]);
    m = max3(m, c[12], c[13]);
    m = max3(m, c[14], c[15]);
    return m;
}

// round-to-nearest-even to bf16, result kept as fp32 (torch's float->bfloat16->float): gfx950 converts in hardware
// (v_cvt_pk_bf16_f32: round to nearest even, NaN stays a quiet NaN)
__device__ __forceinline__ float bf16_round(float x) {
    const __bf16 b = (__bf16)x;
    return __uint_as_float((uint32_t)__builtin_bit_cast(uint16_t, b) << 16);
}

// D = A * B + C on one 32x32x16 tile; operands travel as 8 x 16-bit lanes whatever the element type
template <bool F16>
__device__ __forceinline__ f32x16 mfma32(const bf16x8 &a, const bf16x8 &b, const f32x16 &c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// round an fp32 value to the embedding dtype and back (what torch does when it stores an intermediate in that dtype)
template <bool F16>
__device__ __forceinline__ float round_to_input(float x) {
    if constexpr (F16) return (float)(_Float16)x;
    else return bf16_round(x);
}

// one stored 16-bit element -> fp32
template <bool F16>
__device__ __forceinline__ float elem_to_float(uint16_t v) {
    if constexpr (F16) return (float)__builtin_bit_cast(_Float16, v);
    else return __uint_as_float((uint32_t)v << 16);
}

// sum over lanes 0..31 of each 32-lane half (result valid in every lane of the half)
__device__ __forceinline__ float half_wave_sum(float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}

// 4-byte load through the scalar cache from a wave-uniform, 4-byte aligned address (read-only data)
__device__ __forceinline__ uint32_t scalar_load_u32(uint64_t addr) {
    uint32_t v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(addr) : "memory");
    return v;
}

// first index k in [0, n) with key(k) >= v (key non-decreasing), searched by the whole wave: 64 probes per round,
// so 8192 pairs take 3 rounds of one load each instead of 13 dependent loads
template <class KeyFn>
__device__ __forceinline__ int lower_bound_wave(int n, int v, int lane, KeyFn key) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int step = (hi - lo + 63) >> 6;
        const int idx = lo + lane * step;
        const bool below = idx < hi && key(idx) < v;
        const int cnt = __popcll(__ballot(below));          // probes 0 .. cnt-1 are below v, probe cnt (if any) is not
        if (cnt == 0) break;
        const int nhi = lo + cnt * step;
        lo = lo + (cnt - 1) * step + 1;
        hi = nhi < hi ? nhi : hi;
    }
    return __builtin_amdgcn_readfirstlane(lo);
}

// C/D layout of v_mfma_f32_32x32x16_bf16: lane holds column (lane & 31) and the 16 rows
//   row(reg) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// LDS image of a slab: 32 rows x 256 B, the 16-byte chunk index XOR-ed with (row & 15) so that the
// ds_read_b128 of one MFMA operand (32 different rows, the same logical chunk) is bank-conflict free.
// Byte offset inside the slab of logical chunk `c` of row `r`:
__device__ __forceinline__ int slab_swizzled_off(int r, int c) { return r * kRowBytes + ((c ^ (r & 15)) << 4); }

// ---------------------------------------------------------------------------------------------------------------------------
// The 16x16x32 tiling of the scorers K1s / K1b.  A 32-token query tile x a 32-row document slab x 128 is computed as
// 2 (token halves) x 2 (row groups) x 4 (k-steps) v_mfma_f32_16x16x32 instead of 8 v_mfma_f32_32x32x16: the same FLOP, the same
// operand bytes from LDS and from the query registers, the same number of max folds -- but half the accumulator registers read and
// written per FLOP.  MI355X clocks to its power budget, and on real operand values the matrix pipe sustains 2.07 PFLOP/s with this
// shape against 1.81 with 32x32x16 (msim_probe_mfma, registers only; 1.99 vs 1.71 with the max folds; both reach 2.45 on zeros):
// in the power-bound regimes (everything from 4 queries up) the tile shape is worth more than any scheduling detail.
//   A (M = 16 document rows): lane l supplies row l & 15, k-slice 8 * (l >> 4) .. +7 of the 32-wide k-step   (16 bytes)
//   B (N = 16 query tokens) : lane l supplies token l & 15, the same k-slice                                  (16 bytes)
//   D: lane l holds token l & 15 and document rows 4 * (l >> 4) + {0, 1, 2, 3}                                 (4 registers)
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <bool F16>
__device__ __forceinline__ f32x4 mfma16(const bf16x8 &a, const bf16x8 &b, const f32x4 &c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

constexpr int kKSteps16 = kDim / 32;      // 4 k-steps of 32

struct QueryTile {                         // B operands of one 32-token tile: [token half][k-step], 32 VGPRs
    bf16x8 f[2][kKSteps16];
};
struct TileAcc {                           // D of one (tile, slab): [token half][row group], 16 VGPRs
    f32x4 a[2][2];
};

// byte offsets of this lane's A fragments inside the swizzled slab image: fragment (g, ks) = rows 16g .. 16g+15, k = 32ks .. 32ks+31.
// Conflict-free for ds_read_b128: within each 16-lane access group the physical 16-byte chunk (4ks + (l >> 4)) ^ (row & 15) is distinct.
__device__ __forceinline__ void slab_rd_offsets16(int lane, int (&rd)[2][kKSteps16]) {
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int ks = 0; ks < kKSteps16; ++ks) rd[g][ks] = slab_swizzled_off(16 * g + (lane & 15), 4 * ks + (lane >> 4));
}

// B fragments of tokens tok0 .. tok0+31 of one query (Qq -> its [Lq][128] rows); tokens >= Lq (and a dead tile) are zero rows
__device__ __forceinline__ void load_query_tile(QueryTile &q, const uint16_t *__restrict__ Qq, int tok0, int Lq, int lane, bool live) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = tok0 + 16 * h + (lane & 15);
        const bool valid = live && row < Lq;
        const uint16_t *p = Qq + (size_t)(valid ? row : 0) * kDim + (lane >> 4) * 8;
#pragma unroll
        for (int ks = 0; ks < kKSteps16; ++ks) {
            const bf16x8 v = *reinterpret_cast<const bf16x8 *>(p + ks * 32);
            q.f[h][ks] = valid ? v : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
}

// rows of the slab at or beyond rows_left do not exist: -inf (after the MFMA; zero padding is a different thing, see clamp0)
__device__ __forceinline__ void tile_mask_tail(TileAcc &acc, int rows_left, int lane) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (16 * g + 4 * (lane >> 4) + r >= rows_left) acc.a[h][g][r] = -INFINITY;
}

// running per-token max over document rows: m[h] covers token 16h + (lane & 15), this lane's rows only (8 v_max3 per tile and slab)
__device__ __forceinline__ void tile_fold(float (&m)[2], const TileAcc &acc) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            m[h] = max3(m[h], acc.a[h][g][0], acc.a[h][g][1]);
            m[h] = max3(m[h], acc.a[h][g][2], acc.a[h][g][3]);
        }
}

// end of a document: combine the four lane groups (rows), clamp / round as asked, sum over the tile's 32 tokens.
// Add order = the 5-step butterfly over tokens (t ^ 16 first): token t + token t+16, then xor 8, 4, 2, 1 inside 16 lanes.
template <bool F16>
__device__ __forceinline__ float tile_finish(const float (&m)[2], bool clamp, bool ref_round) {
    float v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float x = fmaxf(m[h], __shfl_xor(m[h], 16));
        x = fmaxf(x, __shfl_xor(x, 32));
        if (clamp) x = fmaxf(x, 0.0f);
        if (ref_round) x = round_to_input<F16>(x);
        v[h] = x;
    }
    float s = v[0] + v[1];
    s += __shfl_xor(s, 8);
    s += __shfl_xor(s, 4);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 1);
    return s;
}


// ---------------------------------------------------------------------------------------------------------------------------
// Round 4: the FLAT token layout of the scorers K1s / K1b.  Queries are ragged in real use (processing_utils.py:86 appends 10
// augmentation tokens to a question of any length; the model's zero padding rows add exactly 0) -- a [n_q, Lq_max, 128] box spends
// MFMA work on every padding row.  The kernels therefore see ONE token matrix Qt [T, 128] (every query's real tokens back to back,
// like the packed corpus) plus the token offsets q_off [n_q + 1], and the unit of MFMA work is 16 consecutive tokens of that matrix
// (the N of v_mfma_f32_16x16x32), whatever query they belong to.  The per-token maxima of a document go through LDS, where
// 8 lanes per query add their query's tokens in an order that depends on nothing but the query's length.
constexpr int kUnitTok = 16;               // query tokens per unit = MFMA N

struct QueryUnit {                         // B operands of one 16-token unit: [k-step], 16 VGPRs
    bf16x8 f[kKSteps16];
};
struct UnitAcc {                           // D of one (unit, slab): [row group], 8 VGPRs
    f32x4 a[2];
};

// how a kernel finds its queries in the flat token matrix
struct FlatQ {
    const int32_t *q_off;   // [n_q + 1] token offsets, or null: uniform queries of Lq tokens each
    int Lq;                 // uniform length (q_off == null)
    int seg, n_seg;         // n_seg > 1 (uniform only): "query" p is PIECE p % n_seg (seg tokens, the last one shorter) of query p / n_seg
};
__device__ __forceinline__ int flat_qoff(const FlatQ &f, int i) {
    if (f.q_off) return f.q_off[i];
    if (f.n_seg <= 1) return i * f.Lq;
    const int r = i / f.n_seg, s = i - r * f.n_seg;
    const int o = s * f.seg;
    return r * f.Lq + (o < f.Lq ? o : f.Lq);
}

// B fragments of tokens tok0 .. tok0+15 of the token matrix Qt; tokens >= n_tok (and a dead unit) are zero rows
__device__ __forceinline__ void load_query_unit(QueryUnit &q, const uint16_t *__restrict__ Qt, int tok0, int n_tok, int lane, bool live) {
    const int row = tok0 + (lane & 15);
    const bool valid = live && row < n_tok;
    const uint16_t *p = Qt + (size_t)(valid ? row : 0) * kDim + (lane >> 4) * 8;
#pragma unroll
    for (int ks = 0; ks < kKSteps16; ++ks) {
        const bf16x8 v = *reinterpret_cast<const bf16x8 *>(p + ks * 32);
        q.f[ks] = valid ? v : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
}

__device__ __forceinline__ void unit_mask_tail(UnitAcc &acc, int rows_left, int lane) {
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (16 * g + 4 * (lane >> 4) + r >= rows_left) acc.a[g][r] = -INFINITY;
}

// running max of token (lane & 15) of the unit over this lane's document rows: 4 v_max3 per unit and slab
__device__ __forceinline__ void unit_fold(float &m, const UnitAcc &acc) {
    m = max3(m, acc.a[0][0], acc.a[0][1]);
    m = max3(m, acc.a[0][2], acc.a[0][3]);
    m = max3(m, acc.a[1][0], acc.a[1][1]);
    m = max3(m, acc.a[1][2], acc.a[1][3]);
}

// One 32-row slab (its 8 operand fragments `af` already in registers) against NU resident units.  Units are taken two at a time
// (a "group" = the 32-token tile of rounds 1-3): 16 MFMAs on four independent accumulator chains, round-robin; an odd last unit runs
// 8 MFMAs on two chains.  LATE: the 16 -> 1 max fold of a group is written behind the MFMAs of the NEXT group -- a VALU read of an
// accumulator needs 12 wait states after the MFMA that writes it, and straight-line code lets hipcc spend them on MFMAs instead of
// s_nop (K1s); K1b folds at once (two waves per SIMD cover it, and fewer live registers matter more there).
// `hook(mf)` runs in front of MFMA number mf (K1s issues its LDS-DMA pieces there).
template <bool F16, int NU, bool kTail, bool LATE, class Hook>
__device__ __forceinline__ void slab_units(float (&m)[NU], const bf16x8 (&af)[2][kKSteps16], const QueryUnit *qu, int rows_left,
                                           int lane, Hook &&hook) {
    constexpr int NG = (NU + 1) / 2;
    UnitAcc prev[2];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int n = NU - 2 * g >= 2 ? 2 : 1;      // units in this group (compile-time after unrolling)
        UnitAcc acc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 2; ++r) acc[u].a[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < kKSteps16; ++ks)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < 2 * n) {
                    hook((2 * g * kKSteps16 + ks * n) * 2 + j);
                    acc[j >> 1].a[j & 1] = mfma16<F16>(af[j & 1][ks], qu[2 * g + (j >> 1)].f[ks], acc[j >> 1].a[j & 1]);
                }
            }
        if constexpr (kTail) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (u < n) unit_mask_tail(acc[u], rows_left, lane);
        }
        if constexpr (LATE) {
            if (g > 0) {
                unit_fold(m[2 * g - 2], prev[0]);
                unit_fold(m[2 * g - 1], prev[1]);
            }
            prev[0] = acc[0];
            prev[1] = acc[1];
        } else {
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (u < n) unit_fold(m[2 * g + u], acc[u]);
        }
    }
    if constexpr (LATE) {
        unit_fold(m[2 * NG - 2], prev[0]);
        if constexpr (NU % 2 == 0) unit_fold(m[2 * NG - 1], prev[1]);
    }
}

// End of a document: the running maxima of this lane's units go to the per-token table in LDS -- token t of the table owns 4 floats,
// one per lane group (the four row quarters of a slab); a full-wave ds_write_b32 hits 64 different banks.
__device__ __forceinline__ void store_token_max(char *tokmax, int unit_in_table, float m, int lane) {
    *reinterpret_cast<float *>(tokmax + ((unit_in_table * kUnitTok + (lane & 15)) << 4) + ((lane >> 4) << 2)) = m;
}

// sum over the tokens [s, e) of the table: lane i of the query's 8 lanes adds tokens s+i, s+i+8, ... in that order, then the 8 lanes
// are folded xor 4, 2, 1 -- a pure function of the token values and e - s, the same in every kernel and for every batch composition
template <bool F16>
__device__ __forceinline__ float reduce_query_tokens(const char *tokmax, int s, int e, int i, bool clamp, bool ref_round) {
    float acc = 0.0f;
    for (int t = s + i; t < e; t += 8) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(tokmax + (t << 4));
        float x = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
        if (clamp) x = fmaxf(x, 0.0f);
        if (ref_round) x = round_to_input<F16>(x);
        acc += x;
    }
    acc += __shfl_xor(acc, 4);
    acc += __shfl_xor(acc, 2);
    acc += __shfl_xor(acc, 1);
    return acc;
}

// The same sum -- the same bits in the lane that is read, lane i == 0 of the query's eight -- with shorter latency chains (round 6,
// K1b on short documents: the token sums of a document sit between two chunk barriers):
//   * four table reads in flight per step instead of one (a token past the end adds +0.0 to a sum that starts at +0.0: no bit changes);
//   * the fold over the 8 lanes on DPP row shifts instead of three ds_bpermute round trips: lane 0 adds (a0+a4) + (a2+a6), then
//     + ((a1+a5) + (a3+a7)) -- exactly what the xor butterfly leaves in lane 0 (the other lanes hold partial sums: read lane 0 only).
template <int N>
__device__ __forceinline__ float row_shl_add(float acc) {
    const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x100 | N, 0xf, 0xf, true);   // lane l reads lane l + N of its row of 16
    return acc + __builtin_bit_cast(float, moved);
}
template <bool F16>
__device__ __forceinline__ float reduce_query_tokens_lane0(const char *tokmax, int s, int e, int i, bool clamp, bool ref_round) {
    float acc = 0.0f;
    for (int t = s + i; t < e; t += 32) {
        f32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const f32x4 *>(tokmax + ((t + 8 * k < e ? t + 8 * k : s) << 4));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float x = fmaxf(fmaxf(v[k][0], v[k][1]), fmaxf(v[k][2], v[k][3]));
            if (clamp) x = fmaxf(x, 0.0f);
            if (ref_round) x = round_to_input<F16>(x);
            if (t + 8 * k < e) acc += x;
        }
    }
    acc = row_shl_add<4>(acc);
    acc = row_shl_add<2>(acc);
    acc = row_shl_add<1>(acc);
    return acc;
}

}  // namespace msim
