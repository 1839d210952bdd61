// K3 -- embedding head for gfx950: the step immediately BEFORE the MaxSim path, producing its corpus format.
//
// Reference lines replaced (identical in every Col* model family):
//   colpali_engine/models/paligemma/colpali/modeling_colpali.py:67-72
//   colpali_engine/models/qwen2/colqwen2/modeling_colqwen2.py:65-69
//       proj = self.custom_text_proj(hidden_states)              # nn.Linear(hidden, 128)
//       proj = proj / proj.norm(dim=-1, keepdim=True)            # L2 normalisation
//       proj = proj * attention_mask.unsqueeze(-1)               # padded positions -> exactly 0
//   (+ :74-77 the optional image-token mask: same multiply, folded into the row map by the host)
// and the host steps that follow it on the way to the scorer (README.md:121-126: torch.unbind, .to("cpu"),
// later pad_sequence + H2D per block): here the normalised rows are written straight into the scorer's
// packed corpus blob, optionally WITHOUT the masked rows (a document that had masked rows gets the
// clamp0 flag instead -- a zero row and the clamp are the same thing to a max).
//
// Rounding chain reproduced in the model dtype (bf16 / fp16), one rounding where torch has one:
//   y = round(acc_fp32 + bias)          nn.Linear output tensor
//   n = round(sqrt(sum_fp32(y^2)))      Tensor.norm (fp32 accumulation inside, result in the tensor dtype)
//   o = round(y / n) * mask             true division, then an exact multiply by 0 or 1
//
// Shape of the work: M = all token rows (B*S, ~10^6 for 1k pages), N = 128, K = hidden size H (1536 .. 3584).
// 128 FLOP per byte of hidden state < the 312 FLOP/B ridge  =>  HBM-bound: the hidden states are streamed
// exactly once (nt), the 128 x H weight is re-read per row tile from the XCD's L2.
//
// Structure (persistent, one workgroup per CU, 8 compute waves = 2 per SIMD + 1 loader wave):
//   * row tile = 256 rows, K chunk = 64: A chunk [256][64] = 32 KiB, W chunk [128][64] = 16 KiB, both filled by
//     LDS-DMA (buffer_load_dwordx4 ... lds); ONE raw s_barrier per chunk, counted vmcnt (never 0 in steady state),
//     the rings keep running across row tiles;
//   * compute wave w owns rows 32w..32w+31 of the tile: it DMA-loads exactly the A rows it consumes into a 4-deep
//     ring (128 KiB; 3 chunks = 96 KiB of hidden states in flight per CU -- at ~3.5 us of loaded HBM latency the
//     bytes in flight are what bounds the stream) and keeps a 32 x 128 fp32 accumulator (4 MFMA tiles);
//   * the weight chunks come from the XCD's L2 through a 2-deep ring (32 KiB) filled by a DEDICATED loader wave:
//     vmcnt retires in order per wave, so a compute wave that also fetched its W share would have to wait for
//     its own deepest A prefetch every chunk; a separate wave has a separate counter (160 KiB of LDS in total);
//   * LDS image: 128-byte rows, the 16-byte chunk index XOR-ed with (row >> 1) & 7 on the SOURCE address and
//     on the ds_read_b128 (linear LDS-DMA destination; cdna_hip_programming.md rule 21): conflict free;
//   * epilogue per tile in registers: the C layout puts one output column per lane and 16 rows per lane, so the
//     row norm is 4 squares + a 5-step butterfly over the 32 lanes of a half; per-row destinations come from a
//     row map read through the scalar cache (no vector load in the loop: nothing drains the DMA ring).
#pragma once
#include "maxsim_common.hpp"
#include "maxsim_stream.hip"

namespace msim {

constexpr int kHeadWaves = 8;
constexpr int kHeadBM = 256;                       // rows per tile
constexpr int kHeadBK = 64;                        // K elements per chunk (128 B per row)
constexpr int kHeadN = 128;                        // output width
constexpr int kHeadABytes = kHeadBM * kHeadBK * 2;   // 32 KiB
constexpr int kHeadWBytes = kHeadN * kHeadBK * 2;    // 16 KiB
constexpr int kHeadRingA = 4;                        // hidden-state chunks (3 in flight)
constexpr int kHeadRingW = 2;                        // weight chunks
constexpr int kHeadWBase = kHeadRingA * kHeadABytes; // 128 KiB
constexpr int kHeadLds = kHeadWBase + kHeadRingW * kHeadWBytes;   // 160 KiB: the whole LDS of a CU
constexpr int kHeadThreads = (kHeadWaves + 1) * 64;  // 8 compute waves + the weight loader
// flag-synchronised variant: hidden-state ring 3 deep (96 KiB, 64 KiB in flight), weight ring 3 deep (48 KiB, the loader runs up
// to two chunks ahead), one 64-byte line of counters behind them
constexpr int kHeadFRingA = 3;
constexpr int kHeadFRingW = 3;
constexpr int kHeadFWBase = kHeadFRingA * kHeadABytes;                   // 96 KiB
constexpr int kHeadFFlags = kHeadFWBase + kHeadFRingW * kHeadWBytes;      // 144 KiB
constexpr int kHeadFLds = kHeadFFlags + 64;

struct HeadArgs {
    long long M;          // token rows
    int H;                // hidden size (multiple of 64)
    long long ld_out;     // elements between consecutive output rows (>= 128)
};

// row_map[m] (int32, ceil(M / 256) * 256 entries):  v >= 0: write the normalised row to out row v;
//   v == -1: drop the row;  v <= -2: write a row of zeros to out row (-2 - v)   (a masked position kept in place).
// FLAGS = false: the weight ring is handed over with one raw s_barrier per K chunk (all nine waves in lock step).
// FLAGS = true : no workgroup barrier at all.  The loader wave publishes "weight chunks landed" in an LDS counter and reads eight
//                "chunks consumed" counters before it overwrites a slot; a compute wave polls the first (only when its cached copy
//                runs out) and bumps its own counter right behind its last operand read of the chunk.  LDS operations of one wave
//                execute in order and an LDS-DMA load has landed once vmcnt says so, so counter writes need no fence.  The eight
//                private hidden-state streams are no longer coupled: a wave whose rows arrive late delays nobody, and the waves
//                drift out of phase, so one wave's operand reads overlap another's MFMAs.
// PIPE: the operand fetch of a chunk is software-pipelined by hand two k-steps ahead of the MFMAs (see the loop).
template <bool F16, bool FLAGS = false, bool PIPE = false>
__global__ __launch_bounds__(kHeadThreads) void embed_head_kernel(const uint16_t *__restrict__ X,     // [M, H]
                                                                     const uint16_t *__restrict__ W,     // [128, H]
                                                                     const uint16_t *__restrict__ bias,  // [128] or null
                                                                     const int32_t *__restrict__ row_map,
                                                                     uint16_t *__restrict__ out, HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int kRingA = FLAGS ? kHeadFRingA : kHeadRingA;
    constexpr int kRingW = FLAGS ? kHeadFRingW : kHeadRingW;
    constexpr int kWBase = FLAGS ? kHeadFWBase : kHeadWBase;
    volatile int *const f_ready = reinterpret_cast<volatile int *>(smem + kHeadFFlags);          // weight chunks landed
    volatile int *const f_done = reinterpret_cast<volatile int *>(smem + kHeadFFlags) + 1;       // [8] chunks consumed per compute wave
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int n_tiles = (int)((a.M + kHeadBM - 1) / kHeadBM);
    const int n_chunks = a.H / kHeadBK;
    const int row_bytes = a.H * 2;

    // ---- bias: one column per lane and column tile
    float bias_f[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bias_f[j] = bias != nullptr ? elem_to_float<F16>(bias[j * 32 + l31]) : 0.0f;
    wait_vmcnt<0>();
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(bias_f[j]));

    // ---- LDS image of a chunk: 128-byte rows, logical 16-byte chunk c of row r stored at physical chunk c ^ ((r >> 1) & 7).
    // One LDS-DMA wave-instruction fills 1 KiB = 8 rows x 128 B linearly: lane -> (row = lane >> 3, physical chunk lane & 7).
    const int my_tiles = blockIdx.x < n_tiles ? (n_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int total = my_tiles * n_chunks;                                   // chunks this workgroup walks (= its barriers)

    if constexpr (FLAGS) {
        if (threadIdx.x < 16) f_ready[threadIdx.x] = 0;      // the only workgroup barrier of the kernel: counters zeroed
        __syncthreads();
    }

    if (wave == kHeadWaves) {
        // ================= weight loader: W chunk c -> ring slot c & 1, one chunk ahead of the consumers
        const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)W, 0, kHeadN * row_bytes, 0x00020000);
        int w_src[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int n = i * 8 + (lane >> 3);                               // weight row = output column
            w_src[i] = n * row_bytes + ((((lane & 7) ^ ((n >> 1) & 7))) << 4);
        }
        auto load_w = [&](int c) {
            char *dst = smem + kWBase + (c % kRingW) * kHeadWBytes;
            const int soff = (c % n_chunks) * (kHeadBK * 2);
#pragma unroll
            for (int i = 0; i < 16; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, MSIM_LDS(dst + i * 1024), 16, w_src[i], soff, 0, 0);
        };
        if constexpr (FLAGS) {
            for (int c = 0; c < total; ++c) {
                // slot c % 3 was last read for chunk c - 3: every compute wave must have consumed c - 2 chunks
                if (c >= kRingW) {
                    for (;;) {
                        int v = f_done[lane & 7];
                        v = min(v, __shfl_xor(v, 1));
                        v = min(v, __shfl_xor(v, 2));
                        v = min(v, __shfl_xor(v, 4));
                        if (__builtin_amdgcn_readfirstlane(v) >= c - (kRingW - 1)) break;
                        __builtin_amdgcn_s_sleep(2);
                    }
                    asm volatile("" ::: "memory");
                }
                load_w(c);
                if (c >= 1) {
                    wait_vmcnt<16>();            // chunk c - 1 has landed (chunk c stays in flight)
                    if (lane == 0) *f_ready = c;
                }
            }
            wait_vmcnt<0>();
            if (lane == 0) *f_ready = total;
            return;
        }
        if (total > 0) load_w(0);
        for (int c = 0; c < total; ++c) {
            wait_vmcnt<0>();                 // W chunk c has landed
            __builtin_amdgcn_s_barrier();    // consumers may read it; they are done with chunk c - 1, whose slot is free again
            if (c + 1 < total) load_w(c + 1);
        }
        return;
    }

    // ================= compute waves
    int a_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wave * 32 + i * 8 + (lane >> 3);                     // row inside the tile (this wave's own rows)
        a_src[i] = row * row_bytes + ((((lane & 7) ^ ((row >> 1) & 7))) << 4);
    }
    const int a_dst = wave * 4096;                                           // 32 rows x 128 B

    // ---- operand fetch offsets: A row = 32*wave + l31, B row (column tile j) = 32*j + l31; logical chunk 2*ks + half
    int a_rd[4], b_rd[4];
    {
        const int arow = wave * 32 + l31;
        const int ax = ((arow >> 1) & 7) << 4, bx = ((l31 >> 1) & 7) << 4;   // ((32j + l31) >> 1) & 7 does not depend on j
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            a_rd[ks] = arow * 128 + ((((2 * ks + half) << 4)) ^ ax);
            b_rd[ks] = kWBase + l31 * 128 + ((((2 * ks + half) << 4)) ^ bx);
        }
    }

    // ---- producer cursor over the flattened (tile, chunk) sequence of this workgroup
    int p_tile = blockIdx.x, p_chunk = 0, p_slot = 0;
    __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)X, 0, 0, 0x00020000);
    auto p_open = [&]() {
        if (p_tile < n_tiles) {
            const long long row0 = (long long)p_tile * kHeadBM;
            const long long rows = a.M - row0 < kHeadBM ? a.M - row0 : kHeadBM;   // rows past M read as zeros (bounds check)
            a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(X + (size_t)row0 * a.H), 0, (int)(rows * row_bytes), 0x00020000);
        }
    };
    p_open();
    auto produce = [&]() -> bool {
        if (p_tile >= n_tiles) return false;
        char *dst = smem + p_slot * kHeadABytes + a_dst;
        const int soff = p_chunk * (kHeadBK * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i)   // hidden states: streamed once -> nt
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, MSIM_LDS(dst + i * 1024), 16, a_src[i], soff, 0, 2);
        p_slot = (p_slot + 1 == kRingA) ? 0 : p_slot + 1;
        if (++p_chunk == n_chunks) {
            p_chunk = 0;
            p_tile += gridDim.x;
            p_open();
        }
        return true;
    };
#pragma unroll
    for (int i = 0; i < kRingA - 1; ++i) produce();

    int c_slot = 0, c_count = 0, w_slot = 0, seen_ready = 0;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

        for (int ch = 0; ch < n_chunks; ++ch, ++c_count) {
            // the slot consumed in the previous iteration is private to this wave and free again: refill it, then wait for
            // this chunk's 4 loads (the rows are this wave's own -- no barrier is involved in the A stream at all)
            if (produce()) wait_vmcnt<4 * (kRingA - 1)>(); else wait_vmcnt<0>();
            if constexpr (FLAGS) {
                while (seen_ready <= c_count) {                    // weight chunk c_count not known to have landed: poll
                    seen_ready = *f_ready;
                    if (seen_ready <= c_count) __builtin_amdgcn_s_sleep(1);
                }
                asm volatile("" ::: "memory");                      // no operand read may move above the poll
            } else {
                __builtin_amdgcn_s_barrier();   // W chunk landed (loader wave); everyone finished reading the previous W chunk
            }
            const char *sa = smem + c_slot * kHeadABytes;
            const char *sw = smem + w_slot * kHeadWBytes;
            w_slot = (w_slot + 1 == kRingW) ? 0 : w_slot + 1;
            c_slot = (c_slot + 1 == kRingA) ? 0 : c_slot + 1;
            if constexpr (PIPE) {
                // Operand fetch software-pipelined by hand, two k-steps ahead of the MFMAs (three fragment buffers of 1 + 4 operands =
                // 60 VGPRs; nine waves per CU leave 168 per wave).  Left to itself hipcc keeps three fragment registers and issues
                // ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma sixteen times per chunk: every MFMA then waits out a full LDS round trip
                // (the loop was LDS-LATENCY bound: ~2400 cycles per chunk and wave for 512 cycles of MFMA).  The scheduling barriers
                // keep the compiler from folding the stages back together; the waitcnt pass still derives the counted lgkmcnt waits.
                bf16x8 fa[3], fw[3][4];
                auto fetch = [&](int ks, int buf) {
                    fa[buf] = *reinterpret_cast<const bf16x8 *>(sa + a_rd[ks]);
    #pragma unroll
                    for (int j = 0; j < 4; ++j) fw[buf][j] = *reinterpret_cast<const bf16x8 *>(sw + b_rd[ks] + j * 4096);
                };
                fetch(0, 0);
                fetch(1, 1);
                __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    if (ks + 2 < 4) fetch(ks + 2, (ks + 2) % 3);
                    __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                    for (int j = 0; j < 4; ++j)   // A = hidden rows (-> accumulator rows), B = weight rows = output columns (-> lane column)
                        acc[j] = mfma32<F16>(fa[ks % 3], fw[ks % 3][j], acc[j]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf16x8 af = *reinterpret_cast<const bf16x8 *>(sa + a_rd[ks]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bf16x8 bf = *reinterpret_cast<const bf16x8 *>(sw + b_rd[ks] + j * 4096);
                        // A = hidden rows (-> accumulator rows), B = weight rows = output columns (-> lane column)
                        acc[j] = mfma32<F16>(af, bf, acc[j]);
                    }
                }
            }
            if constexpr (FLAGS) {
                // every operand read of this chunk has been ISSUED (LDS executes a wave's operations in order): release the slot
                asm volatile("" ::: "memory");
                if (lane == 0) f_done[wave] = c_count + 1;
            }
        }

        // ---- epilogue: acc[j][r] of lane (l31, half) = (row 32*wave + row(r, half), column 32*j + l31)
        const long long row0 = (long long)tile * kHeadBM + wave * 32;
        const int32_t *rm = row_map + row0;          // wave-uniform address: read through the scalar cache
        float ss[16];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float y = round_to_input<F16>(acc[j][r] + bias_f[j]);
                acc[j][r] = y;
                ss[r] = j == 0 ? y * y : ss[r] + y * y;
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float nrm = round_to_input<F16>(sqrtf(half_wave_sum(ss[r])));
            const int v_lo = rm[acc_row(r, 0)], v_hi = rm[acc_row(r, 32)];   // compile-time offsets, scalar loads
            const int v = half ? v_hi : v_lo;
            if (v == -1) continue;
            const bool zero = v < 0;
            uint16_t *dst = out + (size_t)(zero ? -2 - v : v) * a.ld_out + l31;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float o = round_to_input<F16>(acc[j][r] / nrm);
                if (zero) o *= 0.0f;   // `proj * attention_mask`: +-0 (sign kept), NaN stays NaN -- what torch's multiply yields
                uint16_t bits;
                if constexpr (F16) bits = __builtin_bit_cast(uint16_t, (_Float16)o);
                else bits = (uint16_t)(__float_as_uint(o) >> 16);
                dst[j * 32] = bits;
            }
        }
    }
}

}  // namespace msim
