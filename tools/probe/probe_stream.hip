// Measurement aid behind msim_probe_stream (include/maxsim_probe.h): the streaming ceiling of THIS machine for the access
// patterns the kernels use.  A CU-filling persistent grid pulls a row-major [M, H] 16-bit matrix through LDS with the same
// LDS-DMA instruction, cache policy (nt) and ring discipline as K1s / K3 -- PIECE bytes of ROWS rows per wave and ring
// slot, DEPTH slots, WAVES waves per workgroup -- and does nothing else (one ds_read per slot keeps the data dependency).
// bench.py reports it next to the 8 TB/s spec figure: the gap between a kernel and this number is what the kernel's
// own work costs; the gap between this number and the spec is the machine's.
#pragma once
#include "maxsim_common.hpp"
#include "maxsim_stream.hip"
namespace msim {
// SKEW (round 3: what spreads the 8 rows of ONE LDS-DMA instruction over the memory channels when the row stride is a multiple of 4 KiB?):
//   0 = K3's mapping: instruction i = rows 8i .. 8i+7 (consecutive), all at the same K chunk;  1 = rows i, i+4, .., i+28 (16 KiB apart);
//   2 = consecutive rows, every ROW of the instruction at a different K chunk (chunk + row-in-instruction, wrapping);
//   3 = consecutive rows, every INSTRUCTION of the wave at a different K chunk.
template <int PIECE, int ROWS, int DEPTH, int WAVES, int SKEW = 0>
__global__ __launch_bounds__(WAVES * 64) void probe_stream_kernel(const char *__restrict__ X, long long M, int H, float *__restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SLAB = ROWS * PIECE;
    constexpr int NI = SLAB / 1024;
    constexpr int LPR = PIECE / 16;                  // lanes per row piece
    constexpr int RPI = 64 / LPR;                    // rows per instruction
    constexpr int TILE = WAVES * ROWS;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *ring = smem + wave * (DEPTH * SLAB);
    const int row_bytes = H * 2;
    const int n_chunks = row_bytes / PIECE;
    const int n_tiles = (int)(M / TILE);
    const int rin = lane / LPR;                      // row inside the instruction
    const int src = (wave * ROWS + (SKEW == 1 ? rin * NI : rin)) * row_bytes + (lane % LPR) * 16;
    int p_tile = blockIdx.x, p_chunk = 0, p_slot = 0;
    float acc = 0.f;
    int c_slot = 0;
    const int my_tiles = blockIdx.x < n_tiles ? (n_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int total = my_tiles * n_chunks;
    for (int it = 0; it < total + DEPTH - 1; ++it) {
        const bool issue = it < total;
        if (issue) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(X + (size_t)p_tile * TILE * row_bytes), 0, TILE * row_bytes, 0x00020000);
            char *dst = ring + p_slot * SLAB;
            const int soff = p_chunk * PIECE;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if constexpr (SKEW == 1)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, MSIM_LDS(dst + i * 1024), 16, src, soff + i * row_bytes, 0, 2);
                else if constexpr (SKEW == 2)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, MSIM_LDS(dst + i * 1024), 16, src + ((p_chunk + rin) % n_chunks) * PIECE,
                                                             i * RPI * row_bytes, 0, 2);
                else if constexpr (SKEW == 3)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, MSIM_LDS(dst + i * 1024), 16, src, ((p_chunk + i) % n_chunks) * PIECE + i * RPI * row_bytes, 0, 2);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, MSIM_LDS(dst + i * 1024), 16, src, soff + i * RPI * row_bytes, 0, 2);
            }
            p_slot = p_slot + 1 == DEPTH ? 0 : p_slot + 1;
            if (++p_chunk == n_chunks) { p_chunk = 0; p_tile += gridDim.x; }
        }
        if (it >= DEPTH - 1) {
            if (issue) wait_vmcnt<NI * (DEPTH - 1)>(); else wait_vmcnt<0>();
            acc += *reinterpret_cast<float *>(ring + c_slot * SLAB + lane * 16);
            c_slot = c_slot + 1 == DEPTH ? 0 : c_slot + 1;
        }
    }
    if (acc == 123.456f) sink[0] = acc;
}
}  // namespace msim
template <int PIECE, int ROWS, int DEPTH, int WAVES, int SKEW = 0>
int run_probe(const char *X, long long M, int H, float *sink, hipStream_t st) {
    auto kern = msim::probe_stream_kernel<PIECE, ROWS, DEPTH, WAVES, SKEW>;
    constexpr int lds = WAVES * DEPTH * ROWS * PIECE;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -3;
    hipLaunchKernelGGL(kern, dim3(256), dim3(WAVES * 64), lds, st, X, M, H, sink);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

