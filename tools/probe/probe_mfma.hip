// Measurement aid behind msim_probe_mfma (include/maxsim_probe.h): the matrix-core ceiling of THIS machine, under its own power
// budget, for the instruction mix of the MaxSim kernels on REAL operand values.  Every SIMD of the chip runs
// v_mfma_f32_32x32x16_bf16 back to back on operand fragments taken from the caller's matrix (unit-norm bf16 rows: the values the
// scorer multiplies), two waves per SIMD like K1b, with nothing else in the way:
//   bit 0 of `variant`: the A operand comes from LDS (one conflict-free ds_read_b128 per NT MFMAs, K1b's operand path)
//                       instead of staying in registers;
//   bit 1            : the 16 -> 1 max fold of every accumulator tile (8 v_max3 per 8 MFMAs) runs next to the MFMAs.
// There is no HBM traffic, no barrier and no LDS-DMA.  MI355X clocks to its power budget (MI355X_MICROARCH.md, DVFS):
// the 2.5 PFLOP/s dense bf16 figure is 1024 SIMDs x 1024 FLOP/clk x 2.4 GHz; what the chip sustains on random data is lower
// and is the ceiling an MFMA-bound kernel can be held against -- bench.py reports it next to the spec peak.
#pragma once
#include "maxsim_common.hpp"
#include "maxsim_stream.hip"
namespace msim {
template <int NT, bool LDSA, bool FOLD>
__global__ __launch_bounds__(512, 2) void probe_mfma_kernel(const uint16_t *__restrict__ X, int iters, float *__restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t base = ((size_t)blockIdx.x * 8 + wave) * (NT + 1) * kTokTile;     // rows of X used by this wave
    bf16x8 qf[NT][kKSteps];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int ks = 0; ks < kKSteps; ++ks)
            qf[t][ks] = *reinterpret_cast<const bf16x8 *>(X + (base + t * kTokTile + (lane & 31)) * kDim + (lane >> 5) * 8 + ks * 16);
    bf16x8 areg[kKSteps];
#pragma unroll
    for (int ks = 0; ks < kKSteps; ++ks)
        areg[ks] = *reinterpret_cast<const bf16x8 *>(X + (base + NT * kTokTile + (lane & 31)) * kDim + (lane >> 5) * 8 + ks * 16);
    char *slab = smem + wave * kSlabBytes;                                           // wave-private slab, K1s/K1b's swizzled image
    int rd_off[kKSteps];
#pragma unroll
    for (int ks = 0; ks < kKSteps; ++ks) {
        rd_off[ks] = slab_swizzled_off(lane & 31, 2 * ks + (lane >> 5));
        if constexpr (LDSA) *reinterpret_cast<bf16x8 *>(slab + rd_off[ks]) = areg[ks];
    }
    __syncthreads();
    f32x16 acc[NT];
    float m[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        m[t] = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < kKSteps; ++ks) {
            bf16x8 af;
            if constexpr (LDSA) {
                int o = rd_off[ks];
                asm volatile("" : "+v"(o));                                          // a fresh read every iteration
                af = *reinterpret_cast<const bf16x8 *>(slab + o);
            } else {
                asm volatile("" : "+v"(areg[ks]));
                af = areg[ks];
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = mfma32<false>(af, qf[t][ks], acc[t]);
        }
        if constexpr (FOLD) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                m[t] = fold_max16(m[t], acc[t]);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        s += m[t] == -INFINITY ? 0.f : m[t];
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    }
    if (s == 123.456f) sink[0] = s;
}
// The same experiment on v_mfma_f32_16x16x32_bf16 (half the accumulator registers per FLOP): 2 x NT x 8 MFMAs of 16384 FLOP per
// iteration = the FLOP of the kernel above.  This is the tile shape K1s / K1b use (maxsim_common.hpp); with LDSA + FOLD it is their
// instruction mix: one ds_read_b128 per 8 MFMAs, 8 v_max3 per 16.
typedef __attribute__((ext_vector_type(4))) float f32x4;
// WAVES: waves per workgroup (8 = two per SIMD like K1b; 12 / 16 = three / four per SIMD with NT = 3 / 2 tiles each: does a third
// instruction stream per SIMD fill the bubbles two streams leave?)
template <int NT, bool LDSA, bool FOLD, int WAVES = 8>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void probe_mfma16_kernel(const uint16_t *__restrict__ X, int iters, float *__restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t base = ((size_t)blockIdx.x * WAVES + wave) * (NT + 1) * kTokTile;
    bf16x8 qf[NT][kKSteps];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int ks = 0; ks < kKSteps; ++ks)
            qf[t][ks] = *reinterpret_cast<const bf16x8 *>(X + (base + t * kTokTile + (lane & 31)) * kDim + (lane >> 5) * 8 + ks * 16);
    bf16x8 areg[kKSteps];
#pragma unroll
    for (int ks = 0; ks < kKSteps; ++ks)
        areg[ks] = *reinterpret_cast<const bf16x8 *>(X + (base + NT * kTokTile + (lane & 31)) * kDim + (lane >> 5) * 8 + ks * 16);
    char *slab = smem + wave * kSlabBytes;
    int rd_off[kKSteps];
#pragma unroll
    for (int ks = 0; ks < kKSteps; ++ks) {
        rd_off[ks] = slab_swizzled_off(16 * (ks >> 2) + (lane & 15), 4 * (ks & 3) + (lane >> 4));   // K1s/K1b's fragment (g, ks)
        if constexpr (LDSA) *reinterpret_cast<bf16x8 *>(slab + rd_off[ks]) = areg[ks];
    }
    __syncthreads();
    f32x4 acc[NT][2];
    float m[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        m[t] = -INFINITY;
        acc[t][0] = acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < kKSteps; ++ks) {
            bf16x8 af;
            if constexpr (LDSA) {
                int o = rd_off[ks];
                asm volatile("" : "+v"(o));
                af = *reinterpret_cast<const bf16x8 *>(slab + o);
            } else {
                asm volatile("" : "+v"(areg[ks]));
                af = areg[ks];
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, qf[t][ks], acc[t][0], 0, 0, 0);
                acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, qf[t][(ks + 1) & 7], acc[t][1], 0, 0, 0);
            }
        }
        if constexpr (FOLD) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    m[t] = max3(m[t], acc[t][h][0], acc[t][h][1]);
                    m[t] = max3(m[t], acc[t][h][2], acc[t][h][3]);
                    acc[t][h] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        s += m[t] == -INFINITY ? 0.f : m[t];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += acc[t][h][r];
    }
    if (s == 123.456f) sink[0] = s;
}
}  // namespace msim
template <bool LDSA, bool FOLD>
int run_probe_mfma16(const uint16_t *x, int iters, float *sink, hipStream_t st) {
    auto kern = msim::probe_mfma16_kernel<4, LDSA, FOLD>;
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 8 * msim::kSlabBytes, st, x, iters, sink);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
template <int NT, int WAVES, bool LDSA, bool FOLD>
int run_probe_mfma16w(const uint16_t *x, int iters, float *sink, hipStream_t st) {
    auto kern = msim::probe_mfma16_kernel<NT, LDSA, FOLD, WAVES>;
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, WAVES * msim::kSlabBytes) != hipSuccess) return -3;
        configured = true;
    }
    hipLaunchKernelGGL(kern, dim3(256), dim3(WAVES * 64), WAVES * msim::kSlabBytes, st, x, iters, sink);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
template <bool LDSA, bool FOLD>
int run_probe_mfma(const uint16_t *x, int iters, float *sink, hipStream_t st) {
    auto kern = msim::probe_mfma_kernel<4, LDSA, FOLD>;
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 8 * msim::kSlabBytes, st, x, iters, sink);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

namespace msim {
// Round 3: the EXACT slab body of K1b (QueryTile / TileAcc, 8 conflict-free ds_read_b128 per 32-row slab, then per token tile 16
// v_mfma_f32_16x16x32 into four accumulator chains and 8 v_max3), looped over a wave-private LDS slab with no DMA and no barrier, for
//   NT = 4, WAVES = 8: the shipped register plan (two waves per SIMD, 4 tiles = 128 B-operand registers each);
//   NT = 8, WAVES = 4: the plan the round-2 review asks to price before building it -- ONE 512-register wave per SIMD holding 8
//                      tiles (256 B-operand registers, which only fit if they live in AGPRs), i.e. one operand fetch per 16 MFMAs;
//   NT = 6, WAVES = 4: the same with 6 tiles.
// LDSA = false keeps the A fragments in registers (what the operand path through LDS costs in that plan).
template <int NT, int WAVES, bool LDSA, bool PF = false, int DEFER = 0>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void probe_mix_kernel(const uint16_t *__restrict__ X, int iters, float *__restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t base = ((size_t)blockIdx.x * WAVES + wave) * (NT + 1) * kTokTile;
    QueryTile qt[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) load_query_tile(qt[t], X + (base + (size_t)t * kTokTile) * kDim, 0, kTokTile, lane, true);
    // B operands of tiles 4.. are pinned to AGPRs (MFMA srcA/srcB accept either file on gfx950): left to itself hipcc keeps 256 VGPRs
    // and shuffles the overflow through AGPR copies (80 v_accvgpr_read/write per 128 MFMAs)
    wait_vmcnt<0>();
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ks = 0; ks < kKSteps16; ++ks) {
                if (t >= ((WAVES == 8 && NT > 4) ? 1 : 4)) asm volatile("" : "+a"(qt[t].f[h][ks]));   // 5 tiles on two waves per SIMD: 4 of them in AGPRs
                else asm volatile("" : "+v"(qt[t].f[h][ks]));
            }
    char *slab = smem + wave * kSlabBytes;
    int rd_off[2][kKSteps16];
    slab_rd_offsets16(lane, rd_off);
    bf16x8 areg[2][kKSteps16];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int ks = 0; ks < kKSteps16; ++ks) {
            areg[g][ks] = *reinterpret_cast<const bf16x8 *>(X + (base + (size_t)NT * kTokTile + 16 * g + (lane & 15)) * kDim + ks * 32 + (lane >> 4) * 8);
            *reinterpret_cast<bf16x8 *>(slab + rd_off[g][ks]) = areg[g][ks];
        }
    __syncthreads();
    float m[NT][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) m[t][0] = m[t][1] = -INFINITY;
    auto fetch = [&](bf16x8 (&af)[2][kKSteps16]) {
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int ks = 0; ks < kKSteps16; ++ks) {
                if constexpr (LDSA) {
                    int o = rd_off[g][ks];
                    asm volatile("" : "+v"(o));
                    af[g][ks] = *reinterpret_cast<const bf16x8 *>(slab + o);
                } else {
                    asm volatile("" : "+v"(areg[g][ks]));
                    af[g][ks] = areg[g][ks];
                }
            }
    };
    auto tile_mfmas = [&](TileAcc &acc, const bf16x8 (&af)[2][kKSteps16], int t) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int g = 0; g < 2; ++g) acc.a[h][g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < kKSteps16; ++ks)
#pragma unroll
            for (int hg = 0; hg < 4; ++hg)
                acc.a[hg >> 1][hg & 1] = mfma16<false>(af[hg & 1][ks], qt[t].f[hg >> 1][ks], acc.a[hg >> 1][hg & 1]);
    };
    auto compute = [&](const bf16x8 (&af)[2][kKSteps16]) {
        if constexpr (DEFER) {
            // the 16 -> 1 fold of tile t - 1 runs underneath the MFMAs of tile t (two accumulator sets): a lone wave on its SIMD has
            // nobody to cover the wait states between a tile's last MFMA and the VALU reads of its accumulators
            TileAcc acc[2];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                tile_mfmas(acc[t & 1], af, t);
                if (t > 0) tile_fold(m[t - 1], acc[(t - 1) & 1]);
                if constexpr (DEFER == 2) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {        // emitted order: two MFMAs, one fold instruction, ...
                        __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x2, 1, 0);
                    }
                }
            }
            tile_fold(m[NT - 1], acc[(NT - 1) & 1]);
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                TileAcc acc;
                tile_mfmas(acc, af, t);
                tile_fold(m[t], acc);
            }
        }
    };
    if constexpr (PF) {            // operand fragments of the NEXT slab fetched underneath the MFMAs of the current one
        bf16x8 af0[2][kKSteps16], af1[2][kKSteps16];
        fetch(af0);
        for (int it = 0; it < iters; it += 2) {
            fetch(af1);
            compute(af0);
            fetch(af0);
            compute(af1);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            bf16x8 af[2][kKSteps16];
            fetch(af);
            compute(af);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) s += (m[t][0] == -INFINITY ? 0.f : m[t][0]) + (m[t][1] == -INFINITY ? 0.f : m[t][1]);
    if (s == 123.456f) sink[0] = s;
}
}  // namespace msim
template <int NT, int WAVES, bool LDSA, bool PF = false, int DEFER = 0>
int run_probe_mix(const uint16_t *x, int iters, float *sink, hipStream_t st) {
    auto kern = msim::probe_mix_kernel<NT, WAVES, LDSA, PF, DEFER>;
    hipLaunchKernelGGL(kern, dim3(256), dim3(WAVES * 64), WAVES * msim::kSlabBytes, st, x, iters, sink);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
