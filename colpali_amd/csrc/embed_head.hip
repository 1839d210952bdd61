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
    unsigned long long *trace;   // debug (MSIM_HEAD_TRACE_PTR): per wave of workgroup 0, cycles spent per phase; null in production
    int stagger;                 // > 0: workgroup b delays its start by (b % stagger) * stagger_sleep x ~64 cycles (see the kernel's prologue)
    int stagger_sleep;
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
// EPI2: MFMA operand roles swapped (A = weight rows = output columns, B = hidden rows), so that a lane ends a tile holding 64 of
//       the 128 output columns of ONE hidden row instead of one column of 16 rows.  The s_memtime trace of the first form
//       (tools/trace_head.py) showed the per-tile epilogue taking a quarter of the kernel: 16 five-step butterflies for the row
//       norms, 32 dependent scalar row-map loads and 64 two-byte stores per lane.  Here the row norm is an in-lane sum plus one
//       lane exchange, the row map is read once per lane, and a lane stores 16 x 8 bytes.
// DEEPW: hidden-state ring 3 deep, weight ring 3 deep with the loader TWO chunks ahead (still one s_barrier per chunk): the s_memtime
//       trace showed the loader on the critical path -- it issued chunk c+1 at barrier c and had to see it land (an L2 round trip of
//       16 KiB, ~1000-1500 ticks) before barrier c+1 -- while the compute waves never waited for their hidden states (vmcnt ~80 ticks).
// HALF: two workgroups per CU, each with 4 compute waves (128-row tiles) + its own loader and half the LDS (hidden-state ring 3, weight
//       ring 2): the lock-step phases of one workgroup overlap the other's, at the price of every weight chunk being fetched twice per CU.
// IL:   a compute wave no longer issues the four LDS-DMA pieces of its next hidden-state chunk in front of the barrier (where all
//       eight waves queue on the CU's one vector-memory issue path at the same moment, ~140 ticks per piece, with the matrix pipe
//       idle) but one piece behind every k-step's MFMAs of the chunk it is multiplying.
// EPI3: the output rows leave through LDS, as whole rows, with the `nt` (streaming) policy.  Knock-outs (profiles/r02_logs/
//       ab_head_epilogue.log): without its output stores the kernel ran 17-25 % faster although the output is 6-8 % of the bytes; the
//       arithmetic of the epilogue (64 divisions, 16 butterflies) is worth 5-7 %.  The first form stores one 2-byte element per lane and
//       instruction (64 store instructions per lane and tile, two 64-byte pieces each).  Here a wave writes 16 finished rows (bf16) into
//       the hidden-state slot it has just consumed (4 KiB, private, free until the next refill), reads them back as 16 bytes per lane and
//       stores whole 256-byte rows: 8 store instructions per lane and tile, full 128-byte lines.  That alone changed nothing; what the
//       stores cost is their way through L2 / MALL next to the streaming reads: with `nt` on them +4 % (H = 2048), +8-12 % (1536),
//       +13-14 % (3584) -- and the whole-row form gains 2-4 % more than the 2-byte form under the same policy.
// PAIR (round 3, measurement builds): the hidden-state chunks are requested TWO AT A TIME, every other iteration, piece by piece
//       (rows 8i..8i+7 of chunk c, then the same rows of chunk c + 1): the two 128-byte pieces of a row reach the memory system back
//       to back, 256 contiguous bytes of one DRAM page instead of two visits a chunk period apart.  Hidden-state ring 4, weight ring 2.
template <bool F16, bool FLAGS = false, bool PIPE = false, bool EPI2 = false, bool DEEPW = false, bool HALF = false, bool IL = false,
          bool EPI3 = false, bool PAIR = false>
__global__ __launch_bounds__(HALF ? 320 : kHeadThreads) void embed_head_kernel(const uint16_t *__restrict__ X,     // [M, H]
                                                                     const uint16_t *__restrict__ W,     // [128, H]
                                                                     const uint16_t *__restrict__ bias,  // [128] or null
                                                                     const int32_t *__restrict__ row_map,
                                                                     uint16_t *__restrict__ out, HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(!HALF || (!FLAGS && !DEEPW), "HALF has its own ring plan");
    static_assert(!IL || (!FLAGS && !PIPE && !HALF), "IL is built on the barrier form with the compiler's operand schedule");
    static_assert(!EPI3 || (!FLAGS && !EPI2 && !HALF && !IL), "EPI3 stages through the slot the barrier form refills at the top of the next chunk");
    static_assert(!PAIR || (!FLAGS && !DEEPW && !HALF && !IL), "PAIR is built on the 4 + 2 ring plan of the barrier form");
    constexpr int kHeadWaves = HALF ? 4 : 8;                 // compute waves (shadows the namespace constants below)
    constexpr int kHeadBM = kHeadWaves * 32;
    constexpr int kHeadABytes = kHeadBM * kHeadBK * 2;
    constexpr int kRingA = (FLAGS || DEEPW || HALF) ? kHeadFRingA : kHeadRingA;
    constexpr int kRingW = (FLAGS || DEEPW) ? kHeadFRingW : kHeadRingW;
    constexpr int kWBase = kRingA * kHeadABytes;
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
    for (int j = 0; j < 4; ++j) bias_f[j] = (!EPI2 && bias != nullptr) ? elem_to_float<F16>(bias[j * 32 + l31]) : 0.0f;
    wait_vmcnt<0>();
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(bias_f[j]));
    const uint32_t *bias32 = reinterpret_cast<const uint32_t *>(bias);     // EPI2: pairs of adjacent columns, read through the scalar cache

    // ---- LDS image of a chunk: 128-byte rows, logical 16-byte chunk c of row r stored at physical chunk c ^ ((r >> 1) & 7).
    // One LDS-DMA wave-instruction fills 1 KiB = 8 rows x 128 B linearly: lane -> (row = lane >> 3, physical chunk lane & 7).
    const int my_tiles = blockIdx.x < n_tiles ? (n_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int total = my_tiles * n_chunks;                                   // chunks this workgroup walks (= its barriers)

    // Start-up stagger.  Every workgroup walks the K chunks of its rows in the same order and at the same pace, so at any instant the
    // WHOLE CHIP requests bytes at one offset (mod the row stride) of 65 536 different rows.  When the row stride is a multiple of
    // 4 KiB (hidden 2048, 4096) those addresses differ only above bit 12 and camp on a fraction of the memory channels: the bare access
    // pattern loses 14 % there (tools/probe_strides.py: 6.04 TB/s at a 4096-byte stride, 7.06 at 4224).  Delaying workgroup b by
    // (b % stagger) chunk periods puts the workgroups at different K offsets for the rest of the launch -- without touching the order
    // in which any row accumulates its chunks (results are bit-identical by construction).
    if (a.stagger > 1) {
        const int phase = (int)(blockIdx.x % (unsigned)a.stagger);
        for (int i = 0; i < phase * a.stagger_sleep; ++i) __builtin_amdgcn_s_sleep(64);
    }

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
        if constexpr (DEEPW) {
            if (total > 0) load_w(0);
            if (total > 1) load_w(1);
            for (int c = 0; c < total; ++c) {
                if (c + 1 < total) wait_vmcnt<16>(); else wait_vmcnt<0>();   // W chunk c has landed (chunk c + 1 may still be in flight)
                __builtin_amdgcn_s_barrier();    // consumers may read chunk c; they are done with chunk c - 1, whose slot is free again
                if (c + 2 < total) load_w(c + 2);
            }
            return;
        }
        if (total > 0) load_w(0);
        unsigned long long tr_vm = 0, tr_bar = 0, tr_issue = 0;
        const bool tracing = kTraceBuild && a.trace != nullptr && blockIdx.x == 0;
        for (int c = 0; c < total; ++c) {
            const unsigned long long t0 = tracing ? __builtin_amdgcn_s_memtime() : 0;
            wait_vmcnt<0>();                 // W chunk c has landed
            const unsigned long long t1 = tracing ? __builtin_amdgcn_s_memtime() : 0;
            __builtin_amdgcn_s_barrier();    // consumers may read it; they are done with chunk c - 1, whose slot is free again
            const unsigned long long t2 = tracing ? __builtin_amdgcn_s_memtime() : 0;
            if (c + 1 < total) load_w(c + 1);
            if (tracing) {
                const unsigned long long t3 = __builtin_amdgcn_s_memtime();
                tr_vm += t1 - t0; tr_bar += t2 - t1; tr_issue += t3 - t2;
            }
        }
        if (tracing && lane == 0) {
            a.trace[wave * 8 + 0] = tr_issue; a.trace[wave * 8 + 1] = tr_vm; a.trace[wave * 8 + 2] = tr_bar; a.trace[wave * 8 + 3] = 0;
            a.trace[wave * 8 + 4] = (unsigned long long)total;
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
    int p_issued = 0;                                                        // chunks this wave has issued so far
    auto produce_advance = [&]() {
        ++p_issued;
        p_slot = (p_slot + 1 == kRingA) ? 0 : p_slot + 1;
        if (++p_chunk == n_chunks) {
            p_chunk = 0;
            p_tile += gridDim.x;
            p_open();
        }
    };
    auto produce = [&]() -> bool {
        if (p_tile >= n_tiles) return false;
        char *dst = smem + p_slot * kHeadABytes + a_dst;
        const int soff = p_chunk * (kHeadBK * 2);
#pragma unroll
        for (int i = 0; i < 4; ++i)   // hidden states: streamed once -> nt
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, MSIM_LDS(dst + i * 1024), 16, a_src[i], soff, 0, 2);
        produce_advance();
        return true;
    };
    // PAIR: two chunks at once, their pieces interleaved row group by row group
    auto produce2 = [&]() {
        char *dst[2];
        int soff[2];
        bool live[2];
        __amdgpu_buffer_rsrc_t rs[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            live[k] = p_tile < n_tiles;
            dst[k] = smem + p_slot * kHeadABytes + a_dst;
            soff[k] = p_chunk * (kHeadBK * 2);
            rs[k] = a_rsrc;
            if (live[k]) produce_advance();
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (live[k]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs[k], MSIM_LDS(dst[k] + i * 1024), 16, a_src[i], soff[k], 0, 2);
    };
#pragma unroll
    for (int i = 0; i < (PAIR ? 2 : kRingA - 1); ++i) produce();

    int c_slot = 0, c_count = 0, w_slot = 0, seen_ready = 0;
    unsigned long long tr_issue = 0, tr_vm = 0, tr_bar = 0, tr_comp = 0, tr_epi = 0;
    const bool tracing = kTraceBuild && !FLAGS && a.trace != nullptr && blockIdx.x == 0;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if constexpr (IL) {
            if (tile != (int)blockIdx.x) produce();      // the chunk the previous tile's last iteration left out (see `feed`)
        }

        for (int ch = 0; ch < n_chunks; ++ch, ++c_count) {
            // the slot consumed in the previous iteration is private to this wave and free again: refill it, then wait for
            // this chunk's 4 loads (the rows are this wave's own -- no barrier is involved in the A stream at all)
            const unsigned long long t0 = tracing ? __builtin_amdgcn_s_memtime() : 0;
            bool issued = false;
            if constexpr (IL) {
                // chunks c_count + 1 .. c_count + kRingA - 2 may stay in flight (fewer at the very end of this wave's sequence)
                const int ahead = p_issued - c_count - 1;
                if (ahead >= kRingA - 2) wait_vmcnt<4 * (kRingA - 2)>();
                else if (kRingA > 3 && ahead == 1) wait_vmcnt<4>();
                else wait_vmcnt<0>();
            } else if constexpr (PAIR) {
                if ((c_count & 1) == 0) produce2();
            } else {
                issued = produce();
            }
            const unsigned long long t1 = tracing ? __builtin_amdgcn_s_memtime() : 0;
            if constexpr (PAIR) {
                // loads retire in order: the youngest 4 * ahead are the chunks behind the one about to be read (an epilogue's stores
                // among them only make the wait longer, never shorter)
                const int ahead = p_issued - c_count - 1;
                if (ahead >= 3) wait_vmcnt<12>();
                else if (ahead == 2) wait_vmcnt<8>();
                else if (ahead == 1) wait_vmcnt<4>();
                else wait_vmcnt<0>();
            } else if constexpr (!IL) {
                if (issued) wait_vmcnt<4 * (kRingA - 1)>(); else wait_vmcnt<0>();
            }
            const unsigned long long t2 = tracing ? __builtin_amdgcn_s_memtime() : 0;
            if constexpr (FLAGS) {
                while (seen_ready <= c_count) {                    // weight chunk c_count not known to have landed: poll
                    seen_ready = *f_ready;
                    if (seen_ready <= c_count) __builtin_amdgcn_s_sleep(1);
                }
                asm volatile("" ::: "memory");                      // no operand read may move above the poll
            } else {
                __builtin_amdgcn_s_barrier();   // W chunk landed (loader wave); everyone finished reading the previous W chunk
            }
            const unsigned long long t3 = tracing ? __builtin_amdgcn_s_memtime() : 0;
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
                        acc[j] = EPI2 ? mfma32<F16>(fw[ks % 3][j], fa[ks % 3], acc[j]) : mfma32<F16>(fa[ks % 3], fw[ks % 3][j], acc[j]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                // wave-uniform: this wave still has chunks to fetch.  Not behind the last chunk of a tile: vmcnt counts the epilogue's
                // stores too and retires in order, so loads issued in front of them could only be waited for together with them --
                // that chunk is issued in one piece behind the epilogue instead (top of the tile loop)
                const bool feed = IL && p_tile < n_tiles && ch != n_chunks - 1;
                char *const pdst = smem + p_slot * kHeadABytes + a_dst;   // the slot read in the previous iteration
                const int psoff = p_chunk * (kHeadBK * 2);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf16x8 af = *reinterpret_cast<const bf16x8 *>(sa + a_rd[ks]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bf16x8 bf = *reinterpret_cast<const bf16x8 *>(sw + b_rd[ks] + j * 4096);
                        // A = hidden rows (-> accumulator rows), B = weight rows = output columns (-> lane column)
                        acc[j] = EPI2 ? mfma32<F16>(bf, af, acc[j]) : mfma32<F16>(af, bf, acc[j]);
                    }
                    if constexpr (IL) {
                        if (feed) __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, MSIM_LDS(pdst + ks * 1024), 16, a_src[ks], psoff, 0, 2);
                    }
                }
                if constexpr (IL) {
                    if (feed) produce_advance();
                }
            }
            if (tracing) {
                // the MFMAs have been issued, not retired: make the accumulators' values needed before stamping
                float sink = acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0];
                asm volatile("" ::"v"(sink));
                const unsigned long long t4 = __builtin_amdgcn_s_memtime();
                tr_issue += t1 - t0; tr_vm += t2 - t1; tr_bar += t3 - t2; tr_comp += t4 - t3;
            }
            if constexpr (FLAGS) {
                // every operand read of this chunk has been ISSUED (LDS executes a wave's operations in order): release the slot
                asm volatile("" ::: "memory");
                if (lane == 0) f_done[wave] = c_count + 1;
            }
        }

        const unsigned long long te0 = tracing ? __builtin_amdgcn_s_memtime() : 0;
        // ---- epilogue: acc[j][r] of lane (l31, half) = (row 32*wave + row(r, half), column 32*j + l31)
        const long long row0 = (long long)tile * kHeadBM + wave * 32;
        const int32_t *rm = row_map + row0;          // wave-uniform address: read through the scalar cache
        if constexpr (EPI2) {
            // acc[j][r] of lane (l31, half) = (hidden row 32*wave + l31, output column 32*j + acc_row(r, lane))
            int v = -1;
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int sv = rm[i];                      // wave-uniform address: scalar loads (merged by the compiler)
                v = l31 == i ? sv : v;
            }
            float ssq = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // the lane's column is 32j + acc_row(r, 0) + 4 * half: both candidates come from wave-uniform (scalar) loads
                    float bv = 0.0f;
                    if (bias != nullptr) {
                        const uint32_t lo = bias32[(j * 32 + acc_row(r & ~1, 0)) >> 1], hi = bias32[(j * 32 + acc_row(r & ~1, 0) + 4) >> 1];
                        const uint32_t pk = half ? hi : lo;
                        bv = elem_to_float<F16>((uint16_t)((r & 1) ? (pk >> 16) : (pk & 0xffffu)));
                    }
                    const float y = round_to_input<F16>(acc[j][r] + bv);
                    acc[j][r] = y;
                    ssq += y * y;
                }
            ssq += __shfl_xor(ssq, 32);                    // the other 64 columns of the same row
            const float nrm = round_to_input<F16>(sqrtf(ssq));
            // 64 quotients by one divisor: reciprocal, product, one Newton correction (2 FMAs) = the correctly rounded quotient in
            // all but ~1e-7 of the cases, which the rounding to 16 bits that follows absorbs; a subnormal norm takes the real division
            const float rcp = 1.0f / nrm;
            const bool tiny = nrm < 1e-30f;
            auto quot = [&](float y) {
                float qv = y * rcp;
                qv = fmaf(fmaf(-qv, nrm, y), rcp, qv);
                return tiny ? y / nrm : qv;
            };
            if (v != -1) {
                const bool zero = v < 0;
                uint16_t *dst = out + (size_t)(zero ? -2 - v : v) * a.ld_out + 4 * half;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {       // registers 4rg .. 4rg+3 = four consecutive columns: one 8-byte store
                        uint16_t b4[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float o = round_to_input<F16>(quot(acc[j][4 * rg + e]));
                            if (zero) o *= 0.0f;           // `proj * attention_mask`: +-0 (sign kept), NaN stays NaN
                            if constexpr (F16) b4[e] = __builtin_bit_cast(uint16_t, (_Float16)o);
                            else b4[e] = (uint16_t)(__float_as_uint(o) >> 16);
                        }
                        uint2 w2;
                        w2.x = (uint32_t)b4[0] | ((uint32_t)b4[1] << 16);
                        w2.y = (uint32_t)b4[2] | ((uint32_t)b4[3] << 16);
                        *reinterpret_cast<uint2 *>(dst + j * 32 + 8 * rg) = w2;
                    }
            }
            if (tracing) tr_epi += __builtin_amdgcn_s_memtime() - te0;
            continue;
        }
        float ss[16];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float y = round_to_input<F16>(acc[j][r] + bias_f[j]);
                acc[j][r] = y;
                ss[r] = j == 0 ? y * y : ss[r] + y * y;
            }
        if constexpr (EPI3) {
            // the slot consumed last: private to this wave (its own 32 rows x 128 B) and not refilled before the next produce()
            char *const stage = smem + (c_slot == 0 ? kRingA - 1 : c_slot - 1) * kHeadABytes + a_dst;
            typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int p = 0; p < 2; ++p) {                         // rows 16p .. 16p+15 of this wave's 32 = accumulator registers 8p .. 8p+7
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int r = 8 * p + rr;
                    const float nrm = round_to_input<F16>(sqrtf(half_wave_sum(ss[r])));
                    const int v_lo = rm[acc_row(r, 0)], v_hi = rm[acc_row(r, 32)];   // compile-time offsets, scalar loads
                    const bool zero = (half ? v_hi : v_lo) < -1;
                    // staged row = acc_row(r, lane) - 16p = (r & 3) + 8 * ((r >> 2) & 1) + 4 * half; bit 2 of it is `half`: XOR-ing the
                    // byte offset inside the row with half << 7 keeps the two halves of a write on different banks
                    char *const row = stage + ((rr & 3) + 8 * ((rr >> 2) & 1) + 4 * half) * 256;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float o = round_to_input<F16>(acc[j][r] / nrm);
                        if (zero) o *= 0.0f;   // `proj * attention_mask`: +-0 (sign kept), NaN stays NaN -- what torch's multiply yields
                        uint16_t bits;
                        if constexpr (F16) bits = __builtin_bit_cast(uint16_t, (_Float16)o);
                        else bits = (uint16_t)(__float_as_uint(o) >> 16);
                        *reinterpret_cast<uint16_t *>(row + ((64 * j + 2 * l31) ^ (half << 7))) = bits;
                    }
                }
                // read back: instruction i covers staged rows 4i .. 4i+3, lane -> (row 4i + (lane >> 4), 16-byte piece lane & 15)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int rl = 4 * i + (lane >> 4);
                    const u32x4_t val = *reinterpret_cast<const u32x4_t *>(stage + rl * 256 + ((((lane & 15) << 4)) ^ ((i & 1) << 7)));
                    const int v0 = rm[16 * p + 4 * i], v1 = rm[16 * p + 4 * i + 1], v2 = rm[16 * p + 4 * i + 2], v3 = rm[16 * p + 4 * i + 3];
                    const int sel = lane >> 4;
                    const int v = sel == 0 ? v0 : sel == 1 ? v1 : sel == 2 ? v2 : v3;
                    if (v != -1)
                        __builtin_nontemporal_store(val, reinterpret_cast<u32x4_t *>(out + (size_t)(v < 0 ? -2 - v : v) * a.ld_out + ((lane & 15) << 3)));
                }
            }
            if (tracing) tr_epi += __builtin_amdgcn_s_memtime() - te0;
            continue;
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
                __builtin_nontemporal_store(bits, dst + j * 32);   // streaming policy: see EPI3
            }
        }
        if (tracing) tr_epi += __builtin_amdgcn_s_memtime() - te0;
    }
    if (tracing && lane == 0) {
        a.trace[wave * 8 + 0] = tr_issue; a.trace[wave * 8 + 1] = tr_vm; a.trace[wave * 8 + 2] = tr_bar; a.trace[wave * 8 + 3] = tr_comp;
        a.trace[wave * 8 + 4] = (unsigned long long)c_count; a.trace[wave * 8 + 5] = tr_epi;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Backward of the norm / mask tail (training: modeling_colpali.py:65-78 sits inside the training graph).  What autograd derives for
//     proj = proj / proj.norm(dim=-1, keepdim=True);  proj = proj * mask
// with respect to the Linear output, written out:   dproj = mask * (g - y <g, y>) / n,   y = proj / n,   n = round(||proj||)
// (n rounded to the model dtype as Tensor.norm returns it; everything else in fp32, ONE rounding of the result -- the reference's
// autograd rounds five intermediates to the model dtype, so this is closer to the exact gradient than the reference is).
// One row per 16 lanes (16-byte pieces), HBM-bound: 2 x 256 B read + 256 B written per row.  The two GEMMs on either side of it
// (recomputing proj, and dX = dproj W, dW = dproj^T X) are plain library GEMMs on the host side (colpali_amd/embed.py).
template <bool F16>
__global__ __launch_bounds__(256) void embed_head_bwd_rows_kernel(const uint16_t *__restrict__ proj, const uint16_t *__restrict__ g,
                                                               const int32_t *__restrict__ row_map, long long M,
                                                               uint16_t *__restrict__ dproj) {
    const int l16 = threadIdx.x & 15;
    const long long rows_per_pass = (long long)gridDim.x * 16;
    for (long long m = (long long)blockIdx.x * 16 + (threadIdx.x >> 4); m < M; m += rows_per_pass) {
        const bf16x8 pv = *reinterpret_cast<const bf16x8 *>(proj + m * kHeadN + l16 * 8);
        const bf16x8 gv = *reinterpret_cast<const bf16x8 *>(g + m * kHeadN + l16 * 8);
        const bool keep = row_map[m] >= 0;
        float p[8], gg[8], ss = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            p[i] = elem_to_float<F16>((uint16_t)pv[i]);
            gg[i] = elem_to_float<F16>((uint16_t)gv[i]);
            ss += p[i] * p[i];
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        const float n = round_to_input<F16>(sqrtf(ss));
        const float inv = 1.0f / n;
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            p[i] *= inv;
            dot += gg[i] * p[i];
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
        bf16x8 ov;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float d = keep ? (gg[i] - p[i] * dot) * inv : 0.0f;
            uint16_t bits;
            if constexpr (F16) bits = __builtin_bit_cast(uint16_t, (_Float16)d);
            else bits = __builtin_bit_cast(uint16_t, (__bf16)d);
            ov[i] = (short)bits;
        }
        *reinterpret_cast<bf16x8 *>(dproj + m * kHeadN + l16 * 8) = ov;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Row map of the DENSE drop-in (colpali_amd.embedding_head): row m keeps its place, masked positions become zero rows
//   row_map[m] = (attention_mask[m] != 0 && (extra == null || extra[m] != 0)) ? m : -2 - m        (-1 beyond M: tile padding)
// -- `proj * attention_mask.unsqueeze(-1)` and the optional `proj * image_mask` of modeling_colpali.py:72-77 as ONE launch in front
// of the head instead of two or three torch element-wise kernels (a 0.5-1 ms head call is short enough for their launch gaps to show).
// Masks come in whatever dtype the model hands over: `kind` 0 = 1-byte (bool / uint8 / int8), 1 = int16, 2 = int32, 3 = int64,
// 4 = fp32, 5 = bf16, 6 = fp16 (floating zeros of either sign are "masked").
__device__ __forceinline__ bool mask_nonzero(const void *p, long long m, int kind) {
    switch (kind) {
        case 0: return static_cast<const uint8_t *>(p)[m] != 0;
        case 1: return static_cast<const uint16_t *>(p)[m] != 0;
        case 2: return static_cast<const uint32_t *>(p)[m] != 0;
        case 3: return static_cast<const unsigned long long *>(p)[m] != 0;
        case 4: return static_cast<const float *>(p)[m] != 0.0f;
        default: return (static_cast<const uint16_t *>(p)[m] & 0x7fffu) != 0;      // bf16 / fp16: anything but +-0
    }
}

__global__ __launch_bounds__(256) void head_row_map_kernel(const void *__restrict__ mask, int mask_kind, const void *__restrict__ extra,
                                                        int extra_kind, long long M, long long M_padded, int32_t *__restrict__ row_map) {
    const long long m = (long long)blockIdx.x * 256 + threadIdx.x;
    if (m >= M_padded) return;
    if (m >= M) { row_map[m] = -1; return; }
    const bool keep = mask_nonzero(mask, m, mask_kind) && (extra == nullptr || mask_nonzero(extra, m, extra_kind));
    row_map[m] = keep ? (int32_t)m : (int32_t)(-2 - m);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Row map of the PACKING form (colpali_amd.CorpusWriter.append): the kept positions of page b go to consecutive corpus rows
//   row_map[b * S + s] = keep(b, s) ? rows_before + (kept positions of pages < b) + (kept positions of page b before s) : -1
// -- what the host used to assemble from ten torch launches per batch (two cumsums, sums, a where, casts).  Two launches: the kept
// positions per page, then one workgroup per page that adds up the counts of the pages in front of it and ranks its own positions
// (wave ballots + a 4-entry LDS carry).  counts[b] (int64) goes back to the host at finish(); rows_after = rows_before + all counts.
__global__ __launch_bounds__(256) void head_page_count_kernel(const void *__restrict__ mask, int mask_kind, const void *__restrict__ extra,
                                                           int extra_kind, int S, long long *__restrict__ counts) {
    __shared__ int wave_sum[4];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int n = 0;
    for (int s0 = 0; s0 < S; s0 += 256) {
        const int sidx = s0 + threadIdx.x;
        const long long m = (long long)b * S + sidx;
        const bool keep = sidx < S && mask_nonzero(mask, m, mask_kind) && (extra == nullptr || mask_nonzero(extra, m, extra_kind));
        n += __popcll(__ballot(keep));
    }
    if (lane == 0) wave_sum[wave] = n;
    __syncthreads();
    if (threadIdx.x == 0) counts[b] = (long long)wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
}

__global__ __launch_bounds__(256) void head_writer_map_kernel(const void *__restrict__ mask, int mask_kind, const void *__restrict__ extra,
                                                           int extra_kind, int B, int S, const long long *__restrict__ rows_before,
                                                           const long long *__restrict__ counts, long long M_padded,
                                                           int32_t *__restrict__ row_map, long long *__restrict__ rows_after) {
    __shared__ long long part[256];
    __shared__ int wave_cnt[4];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (b == B) {                                         // the extra workgroup: tile padding of the map, and the new row count
        for (long long m = (long long)B * S + threadIdx.x; m < M_padded; m += 256) row_map[m] = -1;
        long long t = 0;
        for (int p = threadIdx.x; p < B; p += 256) t += counts[p];
        part[threadIdx.x] = t;
        __syncthreads();
        if (threadIdx.x == 0) {
            long long tot = *rows_before;
            for (int i = 0; i < 256; ++i) tot += part[i];
            *rows_after = tot;
        }
        return;
    }
    long long t = 0;
    for (int p = threadIdx.x; p < b; p += 256) t += counts[p];
    part[threadIdx.x] = t;
    __syncthreads();
    long long base = *rows_before;
    for (int i = 0; i < 256; ++i) base += part[i];         // every thread adds the same 256 numbers in the same order
    for (int s0 = 0; s0 < S; s0 += 256) {
        const int sidx = s0 + threadIdx.x;
        const long long m = (long long)b * S + sidx;
        const bool keep = sidx < S && mask_nonzero(mask, m, mask_kind) && (extra == nullptr || mask_nonzero(extra, m, extra_kind));
        const unsigned long long bal = __ballot(keep);
        __syncthreads();                                    // wave_cnt of the previous round has been read
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int before = __popcll(bal & ((1ull << lane) - 1ull)), total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            before += w < wave ? wave_cnt[w] : 0;
            total += wave_cnt[w];
        }
        if (sidx < S) row_map[m] = keep ? (int32_t)(base + before) : -1;
        base += total;
    }
}

}  // namespace msim
