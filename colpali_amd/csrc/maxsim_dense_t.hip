// The DENSE hard-max backward on the matrix cores, for long "queries" against many short "documents" -- the reference trainer's
// symmetric direction (trainer/contrastive_trainer.py:202-206) under a loss whose upstream gradient is dense: ColbertLoss, the
// trainer's DEFAULT (trainer/colmodel_training.py:33; late_interaction_losses.py:140-164), and ColbertSigmoidLoss (:440-465).
//     P [n_q, Lq, 128]  the pages (as `query_embeddings`),   R [n_d, Ld, 128]  the gathered queries (as `doc_embeddings`),  Ld <= 64
//     scores[p, c] = sum_i max_s <P[p, i], R[c, s]>          a(p, c, i) = the winning row s        G = dLoss / dscores [n_q, n_d]
//     dP[p, i, :] = sum_c G[p, c] * R[c, a(p, c, i), :]                      256 terms per row at BASELINE config 5's shape
//     dR[c, s, :] = sum_p sum_{i : a(p, c, i) = s} G[p, c] * P[p, i, :]      ~780 terms per row
// Both are genuinely dense contractions (52 GFLOP each as GEMMs at config 5's shape: W = G-scaled one-hot routing, [Lq x n_d Ld] per
// page), and until round 6 they ran as gathers: maxsim_bwd_dd_dense_kernel 576 us + maxsim_bwd_dq_kernel 162 us, ~1.6 GB of L2
// gathers for 6 MB of embeddings, 0.04 of the loss step's roof (rocprofv3, round 5).  Here:
//   * the forward (K1t, ROUTE = true) leaves the routing as ONE BYTE per (page, document, page row): route[p][c][Lq_pad];
//   * dense_t_image_kernel re-lays both boxes as MFMA operand fragments whose contraction index is the ROW index (what a backward
//     GEMM needs and a row-major embedding does not offer): img[n][k-step][16-column block][lane][8];
//   * dense_t_bwd_long_kernel  (dP): v_mfma_f32_16x16x32 with A = R's image (LDS ring, LDS-DMA), B = W built IN REGISTERS from the
//     routing bytes and bf16(G) -- one compare chain per (document, 16 page rows), no operand traffic at all;
//   * dense_t_bwd_short_kernel (dR): A = P's image (LDS ring), B = W^T built in registers with packed 16-bit arithmetic; the pages
//     are split over workgroups (fp32 partials, summed in split order by dense_t_bwd_short_sum_kernel: no float atomics).
// W carries G rounded to the embeddings' 16-bit dtype (one rounding per term, relative 2^-9 for bf16: the size of the rounding of
// the output itself); everything is summed in fp32 in a fixed order.
#pragma once
#include "maxsim_bwd.hip"
#include "maxsim_common.hpp"
#include "maxsim_stream.hip"

namespace msim {

constexpr int kDenseTMaxLd = 64;          // resident documents of at most this many rows
// Ring depths of the two backward kernels (stages of 16 KiB).  An LDS-DMA piece lands 1.1 - 1.8 us after its issue on this chip
// (MI355X_MICROARCH.md: "issued -> landed ~1.1 us"; measured here: with 2 stages of prefetch every stage took 0.9 us whatever the
// chip's load -- 56 or 224 workgroups, tools/ab_dense_t_sizes.sh), and a stage is ~0.25 us of MFMA work: 7 resp. 5 stages ahead.
constexpr int kDenseTLongSteps = 4, kDenseTLongRing = 2, kDenseTShortRing = 4;
constexpr int kFragBytes = 1024;          // one operand fragment: 64 lanes x 16 bytes
constexpr int kKStepBytes = 8 * kFragBytes;   // one 32-row k-step of an image: 8 column blocks

__host__ __device__ inline int dense_t_lq_pad(int Lq) { return (Lq + 63) / 64 * 64; }

// ---- operand images.  img[(n * KS + ks) * 8 + mb][lane][e] = X[n][32 ks + 8 (lane >> 4) + e][16 mb + (lane & 15)], zero for rows
// >= L.  One workgroup per (n, ks); wave w writes column blocks 2 w and 2 w + 1.  2-byte gathers straight from L2: the boxes are
// a few MiB, the kernel is a few microseconds, and nothing here is worth an LDS transpose.
__global__ __launch_bounds__(256) void dense_t_image_kernel(const uint16_t *__restrict__ X, uint16_t *__restrict__ img, int n, int L, int KS) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int box = blockIdx.x / KS, ks = blockIdx.x - box * KS;
    if (box >= n) return;
    const int l16 = lane & 15, l4 = lane >> 4;
    const uint16_t *src = X + (size_t)box * L * kDim;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int mb = 2 * wave + j;
        uint16_t v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int row = 32 * ks + 8 * l4 + e;
            v[e] = row < L ? src[(size_t)row * kDim + 16 * mb + l16] : (uint16_t)0;
        }
        uint4 o;
        o.x = v[0] | ((uint32_t)v[1] << 16);
        o.y = v[2] | ((uint32_t)v[3] << 16);
        o.z = v[4] | ((uint32_t)v[5] << 16);
        o.w = v[6] | ((uint32_t)v[7] << 16);
        *reinterpret_cast<uint4 *>(img + ((size_t)(blockIdx.x * 8 + mb) * 64 + lane) * 8) = o;
    }
}

struct DenseTArgs {
    long long ldg;      // leading dimension of G [n_q, ldg]
    int n_q, Lq;        // the long side: n_q pages of Lq rows
    int n_d, Ld;        // the short side: n_d documents of Ld rows, Ld <= kDenseTMaxLd
    int Lq_pad;         // bytes per (page, document) of the routing
    int ksp;            // k-steps per page in the page image = Lq_pad / 32
    int n_split;        // short-side kernel: page splits
    int pages_per;      // pages per split
    int dbg;            // measurement builds only (kAbBuild, MSIM_DENSE_T_DBG): 1 no LDS-DMA in the loop, 2 no MFMAs, 4 no W build, 8 no fragment reads
    unsigned long long *dbg_out;   // measurement builds only (MSIM_DENSE_T_DBG_OUT = a device address): per-wave s_memtime sums of dP's phases
};

// G[p, c] * upstream -> the embeddings' 16-bit dtype, in both halves of a word
template <bool F16>
__device__ __forceinline__ uint32_t weight_pair(float w) {
    uint32_t b;
    if constexpr (F16) b = __builtin_bit_cast(uint16_t, (_Float16)w);
    else b = __builtin_bit_cast(uint16_t, (__bf16)w);
    return b | (b << 16);
}

// Interleave hint for the instruction scheduler: N groups of (1 MFMA, an LDS read behind every DS_EVERY-th, VALU operations).  The hardware
// issues a wave's instructions in order: an MFMA occupies the matrix pipe for 16 cycles during which the SAME wave can issue the
// next step's operand reads and one-hot arithmetic -- but only if they stand between the MFMAs in the instruction stream.
template <int N, int VALU, int DS_EVERY>
__device__ __forceinline__ void interleave_hint() {
    if constexpr (N > 0) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   // one MFMA
        if constexpr (N % DS_EVERY == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // an LDS read behind every DS_EVERY-th
        if constexpr (VALU > 0) __builtin_amdgcn_sched_group_barrier(0x002, VALU, 0);        // VALU operations behind each
        interleave_hint<N - 1, VALU, DS_EVERY>();
    }
}

// s_waitcnt vmcnt(fl * PER): `fl` (wave-uniform, 0 .. MAXFL) later stages of PER LDS-DMA pieces each may stay in flight
template <int MAXFL, int PER>
__device__ __forceinline__ void wait_inflight(int fl) {
    if constexpr (MAXFL == 0) {
        wait_vmcnt<0>();
    } else {
        if (fl >= MAXFL) wait_vmcnt<MAXFL * PER>();
        else wait_inflight<MAXFL - 1, PER>(fl);
    }
}

// ---- dP.  Workgroup = 8 waves (two per SIMD) on 128 page rows of ONE page: wave w works on the 32 rows (two 16-row groups)
// 32 (w & 3) .. and the column blocks 4 (w >> 2) .. + 3.  The workgroup walks ALL documents in stages of NCS documents (32 KiB of
// image per stage, 3-stage LDS ring filled by LDS-DMA together with the stage's routing bytes for the 128 page rows).  A STEP =
// one (document, k-step): 4 A fragments from the ring (lane-linear ds_read_b128: conflict-free) x 2 row groups = 8 MFMAs per wave.
//   * software-pipelined over steps: while the MFMAs of step j run, the A fragments of step j + 1 are read and its W operands are
//     built (compare chain on the routing byte, ~13 VALU per row group);
//   * the barrier that publishes stage s + 1 stands in front of the LAST step of stage s, so the pipeline runs across stages (every
//     LDS read of stage s has been issued and waited for by then: its slot takes stage s + 3);
//   * an LDS-DMA piece costs the issuing wave 60-185 cycles (MI355X_MICROARCH.md): the 5 pieces a wave owes per stage are spread
//     over the stage's steps instead of standing in one block behind the barrier, and two waves per SIMD cover each other's stalls
//     (first build: 4 waves, 9 pieces in a block per stage -- 122 us for 27 us of MFMA work, rocprofv3).
// D layout: lane holds page row (lane & 15) and columns 16 mb + 4 (lane >> 4) + r: four consecutive columns = one 8-byte store per
// (row group, column block).  KS: k-steps per document (Ld <= 32 KS).
template <bool F16, int KS>
__global__ __launch_bounds__(512, 1) void dense_t_bwd_long_kernel(const uint16_t *__restrict__ Rimg, const uint8_t *__restrict__ route,
                                                                   const float *__restrict__ G, GScale gs, uint16_t *__restrict__ dP,
                                                                   DenseTArgs a) {
    static_assert(KS == 1 || KS == 2, "documents of at most 64 rows");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NCS = kDenseTLongSteps / KS;             // documents per stage
    constexpr int NS = NCS * KS;                           // steps per stage
    constexpr int kStageBytes = NS * kKStepBytes;          // 8 KiB per step
    constexpr int kRouteBytes = NCS * 128;                 // NCS slots x 128 page rows
    constexpr int kRing = kDenseTLongRing;                 // stages in the ring: kRing - 1 are in flight or landed ahead of the consumer
    constexpr int kPieces = kStageBytes / kFragBytes / 8;  // image pieces per wave and stage (2)
    constexpr int kDma = kPieces + 1;                      // + the routing bytes
    constexpr int kLutBytes = 9 * 16;                      // per document: the 8 one-hot W fragments (w at k-slot r) and the zero fragment
    static_assert(kPieces == NS, "one image piece per step");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tp = wave & 3, ch = wave >> 2;               // row pair, column half
    const int l16 = lane & 15, l4 = lane >> 4;
    const int page = blockIdx.y;
    const int tile0 = blockIdx.x * 128;
    // the arguments the loop needs, pinned in scalar registers: left to itself hipcc re-loads kernel arguments inside the loop
    // (s_load + s_waitcnt lgkmcnt(0)), and lgkmcnt is the counter of the LDS reads too -- every such wait drains the operand
    // prefetch of the software pipeline (first builds: 0.7 us per stage with NOTHING else in the loop, tools/ab_dense_t.sh)
    int n_d = a.n_d, Lq = a.Lq, Lq_pad = a.Lq_pad, dbg = kAbBuild ? a.dbg : 0;
    asm volatile("" : "+s"(n_d), "+s"(Lq), "+s"(Lq_pad), "+s"(dbg));
    char *route_lds = smem + kRing * kStageBytes;
    char *lut_lds = route_lds + kRing * kRouteBytes;                                      // [stage slot][document][9 W fragments of 16 bytes]
    uint32_t *w_lds = reinterpret_cast<uint32_t *>(lut_lds + kRing * NCS * kLutBytes);    // weight pairs of this page's n_d documents
    const int n_stages = (n_d + NCS - 1) / NCS;
    const __amdgpu_buffer_rsrc_t img_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)Rimg, 0, (int)((size_t)n_d * KS * kKStepBytes), 0x00020000);
    const size_t route_all = (size_t)a.n_q * n_d * Lq_pad;
    const __amdgpu_buffer_rsrc_t rt_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)route, 0, (int)(route_all < 0x7fffffffull ? route_all : 0x7fffffffull), 0x00020000);
    // G[page, :] * upstream, rounded to the embeddings' dtype, in both halves of a word, once per workgroup (documents past n_d, which
    // the last stage may touch, weigh 0); published by the first barrier below
    {
        const float up = load_gscale(gs);
        const float *grow = G + (size_t)page * a.ldg;
        const int n_pad = n_stages * NCS;
        for (int c = threadIdx.x; c < n_pad; c += 512) w_lds[c] = weight_pair<F16>(c < n_d ? grow[c] * up : 0.0f);
        wait_vmcnt<0>();
        __syncthreads();
    }
    // The W operand of a (document, 16 page rows, k-step) -- 8 k-slots per lane, at most one of them non-zero: the document's weight at
    // the slot of the row that won -- is one of NINE 16-byte patterns per document.  Wave 0 writes them into the ring when it starts
    // a stage's DMA; every wave then fetches its operand with ONE ds_read_b128 at pattern min(winning row - first row of the lane's
    // slots, 8) instead of building it with 15 VALU operations per (row group, step): the kernel is bound by instruction issue
    // (rocprofv3 --pmc, round 6: MFMA pipe busy 26 % with 4.8 VALU + 1.9 SALU per MFMA in the register-built form).
    auto build_lut = [&](int t) {
        if (t >= n_stages || wave != 0 || lane >= NCS * 9) return;
        const int cc = lane / 9, r = lane - 9 * cc;
        const uint32_t wpair = w_lds[t * NCS + cc];
        const uint32_t word = (r & 1) ? (wpair & 0xffff0000u) : (wpair & 0xffffu);
        i32x4 v = {0, 0, 0, 0};
        if (r < 8) v[r >> 1] = (int)word;
        *reinterpret_cast<i32x4 *>(lut_lds + ((t % kRing) * NCS + cc) * kLutBytes + r * 16) = v;
    };

    // piece j < kPieces: 1 KiB of stage t's image (documents past n_d read as zeros: bounds check); piece kPieces: the stage's routing
    // bytes of this workgroup's 128 page rows, 128-byte slot q of the stage's area = document q % NCS (every wave issues one piece of
    // two slots: the same count for everyone keeps the vmcnt arithmetic uniform; waves beyond the first NCS / 2 repeat them)
    auto piece = [&](int t, int j) {
        if (t >= n_stages) return;
        if (kAbBuild && (dbg & 1) && t >= 2) return;
        const int slot = t % kRing;
        if (j < kPieces) {
            const int off = (wave * kPieces + j) * kFragBytes;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(img_rsrc, MSIM_LDS(smem + slot * kStageBytes + off), 16, lane * 16, t * kStageBytes + off, 0, 0);
        } else {
            constexpr int kRoutePieces = NCS > 2 ? NCS / 2 : 1;                     // 256 bytes = two documents' 128 page rows per piece
            const int q = 2 * (wave % kRoutePieces) + (lane >> 5);
            int c = t * NCS + q % NCS;
            c = c < n_d ? c : n_d - 1;
            const int voff = (int)(((size_t)page * n_d + c) * Lq_pad) + tile0 + 4 * (lane & 31);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rt_rsrc, MSIM_LDS(route_lds + slot * kRouteBytes + (wave % kRoutePieces) * 256), 4, voff, 0, 0, 0);
        }
    };
    // this wave's share of stage t has landed (the stages t + 1 .. t + kRing - 2, all issued by now, may stay in flight), then the
    // workgroup-wide publication
    unsigned long long t_vm = 0, t_lgkm = 0, t_bar = 0, t_switch = 0, t_loop = 0;      // measurement builds: s_memtime sums
    auto publish = [&](int t) {
        unsigned long long c0 = 0, c1 = 0, c2 = 0;
        if (kAbBuild && a.dbg_out) c0 = __builtin_amdgcn_s_memtime();
        wait_inflight<kRing - 2, kDma>((t + kRing - 1 < n_stages ? t + kRing - 1 : n_stages) - (t + 1));
        if (kAbBuild && a.dbg_out) c1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (kAbBuild && a.dbg_out) c2 = __builtin_amdgcn_s_memtime();
        if (kAbBuild && (dbg & 16)) return;
        __builtin_amdgcn_s_barrier();
        if (kAbBuild && a.dbg_out) {
            t_vm += c1 - c0;
            t_lgkm += c2 - c1;
            t_bar += __builtin_amdgcn_s_memtime() - c2;
        }
    };
    if (kAbBuild && (dbg & 64)) return;
    for (int t = 0; t < kRing - 1; ++t) {
#pragma unroll
        for (int j = 0; j < kDma; ++j) piece(t, j);
        build_lut(t);
    }

    const bool live = tile0 + tp * 32 < Lq;
    if (!live) {                                   // a wave without page rows only feeds the ring (wave-uniform; the same barriers)
        publish(0);
        for (int s = 0; s < n_stages; ++s) {
#pragma unroll
            for (int j = 0; j < kDma; ++j) piece(s + kRing - 1, j);
            if (s + 1 < n_stages) publish(s + 1);
        }
        return;
    }

    f32x4 acc[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[t][mb] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto load_win = [&](int (&win)[NCS][2], const char *rt) {
#pragma unroll
        for (int cc = 0; cc < NCS; ++cc)
#pragma unroll
            for (int t = 0; t < 2; ++t) win[cc][t] = *reinterpret_cast<const uint8_t *>(rt + cc * 128 + tp * 32 + 16 * t + l16);
    };
    auto load_af = [&](bf16x8 (&af)[4], const char *st, int step) {
        if (kAbBuild && (dbg & 8)) return;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) af[mb] = *reinterpret_cast<const bf16x8 *>(st + (step * 8 + 4 * ch + mb) * kFragBytes + lane * 16);
    };
    // this lane's 8 k-slots of k-step ks are the resident rows 32 ks + 8 l4 + e: pattern (winning row - 32 ks - 8 l4) if that is in
    // 0..7, else the zero pattern
    auto load_w = [&](bf16x8 (&wf)[2], const int (&win)[2], const char *lut, int ks) {
        if (kAbBuild && (dbg & 4)) return;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint32_t rel = (uint32_t)(win[t] - (32 * ks + 8 * l4));
            wf[t] = *reinterpret_cast<const bf16x8 *>(lut + (rel < 8u ? rel : 8u) * 16);
        }
    };

    // operand registers in two sets that alternate by step parity (NS is even: step 0 of every stage uses set 0): no register copies in
    // a loop that is bound by instruction issue (an `af = afn` rotation costs 3 v_mov per MFMA here)
    static_assert(NS % 2 == 0, "operand sets alternate by step parity");
    int win[NCS][2];
    bf16x8 af[2][4], wf[2][2];
    publish(0);
    piece(kRing - 1, 0);
    piece(kRing - 1, kPieces);
    build_lut(kRing - 1);
    load_win(win, route_lds);
    load_af(af[0], smem, 0);
    load_w(wf[0], win[0], lut_lds, 0);
    unsigned long long loop0 = 0;
    if (kAbBuild && a.dbg_out) loop0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < (kAbBuild && (dbg & 32) ? 1 : n_stages); ++s) {
        const int slot = s % kRing;
        const char *st = smem + slot * kStageBytes;
        const char *lut = lut_lds + slot * NCS * kLutBytes;
        const bool more = s + 1 < n_stages;
#pragma unroll
        for (int step = 0; step < NS; ++step) {
            const int cur = step & 1, nxt = cur ^ 1;
            if (step + 1 < NS) {
                piece(s + kRing - 1, step + 1);    // the rest of stage s + kRing - 1, one piece per step
                load_af(af[nxt], st, step + 1);
                load_w(wf[nxt], win[(step + 1) / KS], lut + ((step + 1) / KS) * kLutBytes, (step + 1) % KS);
            } else if (more) {                     // the switch to stage s + 1, in front of the last step's MFMAs
                publish(s + 1);
                unsigned long long c0 = 0;
                if (kAbBuild && a.dbg_out) c0 = __builtin_amdgcn_s_memtime();
                piece(s + kRing, 0);               // stage s + kRing goes where stage s was: every read of stage s has been issued and waited for
                piece(s + kRing, kPieces);
                build_lut(s + kRing);
                const int nslot = slot + 1 == kRing ? 0 : slot + 1;
                load_win(win, route_lds + nslot * kRouteBytes);
                load_af(af[nxt], smem + nslot * kStageBytes, 0);
                load_w(wf[nxt], win[0], lut_lds + nslot * NCS * kLutBytes, 0);
                if (kAbBuild && a.dbg_out) t_switch += __builtin_amdgcn_s_memtime() - c0;
            }
            if (!(kAbBuild && (dbg & 2))) {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    acc[0][mb] = mfma16<F16>(af[cur][mb], wf[cur][0], acc[0][mb]);
                    acc[1][mb] = mfma16<F16>(af[cur][mb], wf[cur][1], acc[1][mb]);
                }
            }
            interleave_hint<8, 1, 1>();
        }
    }
    if (kAbBuild && a.dbg_out) {
        t_loop = __builtin_amdgcn_s_memtime() - loop0;
        if (lane == 0) {
            unsigned long long *o = a.dbg_out + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 8;
            o[0] = t_loop; o[1] = t_vm; o[2] = t_lgkm; o[3] = t_bar; o[4] = t_switch; o[5] = n_stages;
        }
    }
    // ---- write-out: one rounding of the fp32 sums to the embeddings' dtype
    constexpr int DT = F16 ? kDtypeF16 : kDtypeBf16;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int row = tile0 + tp * 32 + 16 * t + l16;
        if (row >= Lq) continue;
        uint16_t *o = dP + ((size_t)page * Lq + row) * kDim + 64 * ch + 4 * l4;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            uint2 v;
            v.x = (uint32_t)float_to_elem16<DT>(acc[t][mb][0]) | ((uint32_t)float_to_elem16<DT>(acc[t][mb][1]) << 16);
            v.y = (uint32_t)float_to_elem16<DT>(acc[t][mb][2]) | ((uint32_t)float_to_elem16<DT>(acc[t][mb][3]) << 16);
            *reinterpret_cast<uint2 *>(o + 16 * mb) = v;
        }
    }
}

// ---- dR.  Workgroup = 8 waves (two per SIMD) on 4 NC documents and one page split: wave w works on the NC documents of group
// w & 3 and on HALF of their NC NSB (document, 16-row block) combinations (w >> 2).  The workgroup walks its pages' image in stages
// of two k-steps (16 KiB, 3-stage LDS ring, LDS-DMA, plus each document group's routing bytes: 64 page rows x NC documents).  A
// STEP = one k-step: 8 A fragments, and per combination a W^T operand built for this lane's row s = 16 sb + (lane & 15) from 8
// routing bytes with packed 16-bit arithmetic (expand, xor with s, saturating 1 - x, multiply by the weight pair: 3 operations per
// register) and 8 MFMAs.  Software-pipelined like the kernel above (next step's fragments, routing bytes and first W^T under this
// step's MFMAs; the stage switch in front of a stage's last step; the wave's 3 LDS-DMA pieces per stage spread over the steps).
// D layout: lane holds document row (lane & 15) and columns 16 mb + 4 (lane >> 4) + r.
// NSB: 16-row blocks per document (Ld <= 16 NSB), NC: documents per wave pair; NC * NSB = 4 combinations, two per wave.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

template <bool F16, int NSB, int NC>
__global__ __launch_bounds__(512, 4) void dense_t_bwd_short_kernel(const uint16_t *__restrict__ Pimg, const uint8_t *__restrict__ route,
                                                                    const float *__restrict__ G, GScale gs, float *__restrict__ partial,
                                                                    DenseTArgs a) {
    static_assert(NSB * NC == 4 && NC <= 4, "four (document, row block) combinations per wave pair");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NS = 2;                                      // steps (k-steps) per stage
    constexpr int kStageBytes = NS * kKStepBytes;              // 16 KiB
    constexpr int kRouteBytes = 4 * 256;                       // per stage: [document group][4 slots][64 page rows]
    constexpr int kRing = kDenseTShortRing;
    constexpr int kPieces = kStageBytes / kFragBytes / 8;      // image pieces per wave and stage (2)
    constexpr int kDma = kPieces + 1;
    static_assert(kPieces == NS, "one image piece per step");
    constexpr int NJ = 2;                                      // combinations per wave
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int dw = wave & 3, half = wave >> 2;
    const int l16 = lane & 15, l4 = lane >> 4;
    const int doc0 = (blockIdx.x * 4 + dw) * NC;               // this wave pair's first document
    const int split = blockIdx.y;
    // the arguments the loop needs, pinned in scalar registers (see dense_t_bwd_long_kernel)
    int n_d = a.n_d, Ld = a.Ld, Lq_pad = a.Lq_pad, ksp = a.ksp, dbg = kAbBuild ? a.dbg : 0;
    asm volatile("" : "+s"(n_d), "+s"(Ld), "+s"(Lq_pad), "+s"(ksp), "+s"(dbg));
    const int page_lo = split * a.pages_per;
    const int page_hi = page_lo + a.pages_per < a.n_q ? page_lo + a.pages_per : a.n_q;
    char *route_lds = smem + kRing * kStageBytes;
    uint32_t *w_lds = reinterpret_cast<uint32_t *>(route_lds + kRing * kRouteBytes);      // [page of the split][4 NC documents of the workgroup]
    const int st_per_page = ksp / NS;
    const int n_stages = (page_hi - page_lo) * st_per_page;
    const __amdgpu_buffer_rsrc_t img_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)Pimg, 0, (int)((size_t)a.n_q * ksp * kKStepBytes), 0x00020000);
    const size_t route_all = (size_t)a.n_q * n_d * Lq_pad;
    const __amdgpu_buffer_rsrc_t rt_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)route, 0, (int)(route_all < 0x7fffffffull ? route_all : 0x7fffffffull), 0x00020000);
    // G[page, document] * upstream as weight pairs for this split's pages and this workgroup's documents, once (published by the first barrier)
    {
        const float up = load_gscale(gs);
        const int n_w = (page_hi - page_lo) * 4 * NC;
        for (int i = threadIdx.x; i < n_w; i += 512) {
            const int pg = page_lo + i / (4 * NC), c = blockIdx.x * 4 * NC + i % (4 * NC);
            w_lds[i] = weight_pair<F16>(c < n_d ? G[(size_t)pg * a.ldg + c] * up : 0.0f);
        }
        wait_vmcnt<0>();
    }

    // The stage being issued: stage i_t = k-steps NS i_sp .. of page i_pg (cursors instead of i_t / st_per_page: a division by a
    // run-time value is ~25 scalar instructions, and this loop is bound by instruction issue).
    int i_t = 0, i_pg = page_lo, i_sp = 0;
    auto next_stage = [&]() {
        ++i_t;
        if (++i_sp == st_per_page) { i_sp = 0; ++i_pg; }
    };
    // piece j < kPieces: 1 KiB of the image of the stage being issued; piece kPieces: this document group's routing bytes of the stage's
    // 64 page rows, slot (lane >> 4) = document doc0 + (lane >> 4) % NC (both waves of a pair issue it: uniform vmcnt arithmetic)
    auto piece = [&](int j) {
        if (i_t >= n_stages) return;
        if (kAbBuild && (dbg & 1) && i_t >= 2) return;
        const int slot = i_t % kRing;
        if (j < kPieces) {
            const int off = (wave * kPieces + j) * kFragBytes;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(img_rsrc, MSIM_LDS(smem + slot * kStageBytes + off), 16, lane * 16,
                                                     (i_pg * ksp + i_sp * NS) * kKStepBytes + off, 0, 0);
        } else {
            int c = doc0 + (lane >> 4) % NC;
            c = c < n_d ? c : n_d - 1;
            const int voff = (int)(((size_t)i_pg * n_d + c) * Lq_pad) + i_sp * 32 * NS + 4 * l16;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rt_rsrc, MSIM_LDS(route_lds + slot * kRouteBytes + dw * 256), 4, voff, 0, 0, 0);
        }
    };
    auto publish = [&](int t) {
        wait_inflight<kRing - 2, kDma>((t + kRing - 1 < n_stages ? t + kRing - 1 : n_stages) - (t + 1));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (kAbBuild && (dbg & 16)) return;
        __builtin_amdgcn_s_barrier();
    };
    if (kAbBuild && (dbg & 64)) return;
    for (int k = 0; k < kRing - 1; ++k) {
#pragma unroll
        for (int j = 0; j < kDma; ++j) piece(j);
        next_stage();
    }

    f32x4 acc[NJ][8];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) acc[j][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    // this wave's combinations j = 0, 1: document jd[j] (of the pair's NC), row block jsb[j]
    int jd[NJ], jsb[NJ];
    uint32_t s2[NJ];                           // this lane's document row, in both halves of a word
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int comb = half * NJ + j;
        jd[j] = comb / NSB;
        jsb[j] = comb % NSB;
        s2[j] = (uint32_t)(16 * jsb[j] + l16) * 0x00010001u;
    }
    constexpr bool kOneDoc = NSB >= NJ;        // both combinations of a wave are row blocks of ONE document: one set of routing bytes

    // the weight pairs of page pg for this wave's combinations: LDS broadcast reads
    auto load_weights = [&](uint32_t (&wp)[NJ], int pg) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) wp[j] = w_lds[(pg - page_lo) * 4 * NC + dw * NC + jd[j]];
    };
    // the routing of this lane's 8 k-slots = page rows 32 kk + 8 l4 + e of the stage, per combination's document
    auto load_rb = [&](uint2 (&rb)[NJ], const char *rt, int kk) {
        rb[0] = *reinterpret_cast<const uint2 *>(rt + jd[0] * 64 + kk * 32 + 8 * l4);
        rb[1] = kOneDoc ? rb[0] : *reinterpret_cast<const uint2 *>(rt + jd[1] * 64 + kk * 32 + 8 * l4);
    };
    auto build_w = [&](uint2 rbd, uint32_t wpd, uint32_t s2j) -> bf16x8 {
        if (kAbBuild && (dbg & 4)) return __builtin_bit_cast(bf16x8, i32x4{(int)rbd.x, (int)rbd.y, (int)wpd, (int)s2j});
        uint32_t h[4];                         // bytes -> 16-bit halves (shared by the two row blocks of one document: hipcc folds them)
        h[0] = __builtin_amdgcn_perm(0u, rbd.x, 0x0c010c00u);
        h[1] = __builtin_amdgcn_perm(0u, rbd.x, 0x0c030c02u);
        h[2] = __builtin_amdgcn_perm(0u, rbd.y, 0x0c010c00u);
        h[3] = __builtin_amdgcn_perm(0u, rbd.y, 0x0c030c02u);
        const u16x2 w2 = __builtin_bit_cast(u16x2, wpd);
        const u16x2 one = {1, 1};
        i32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u16x2 x = __builtin_bit_cast(u16x2, h[j] ^ s2j);                 // 0 where the winning row is this lane's
            const u16x2 hit = __builtin_elementwise_sub_sat(one, x);               // 1 there, 0 elsewhere
            v[j] = (int)__builtin_bit_cast(uint32_t, (u16x2)(hit * w2));
        }
        return __builtin_bit_cast(bf16x8, v);
    };

    // The pipeline runs in HALF steps (one k-step's column blocks 0-3, then 4-7: 2 combinations x 4 MFMAs each): the A fragments of
    // the next half step are read under this one's MFMAs, in two alternating sets of 4 -- 32 registers instead of the 64 a whole-step
    // double buffer takes, which keeps the kernel under 128 VGPRs: four waves per SIMD, i.e. it shares a CU with the dP kernel that
    // msim_dense_t_bwd runs beside it on a second stream.  The routing bytes and the first W^T of the next k-step alternate by k-step.
    auto load_af_half = [&](bf16x8 (&af)[4], const char *st, int kk, int hf) {
        if (kAbBuild && (dbg & 8)) return;
#pragma unroll
        for (int m = 0; m < 4; ++m) af[m] = *reinterpret_cast<const bf16x8 *>(st + (kk * 8 + 4 * hf + m) * kFragBytes + lane * 16);
    };
    static_assert(NS % 2 == 0, "operand sets alternate by parity");
    uint32_t wp[NJ], wpn[NJ];
    uint2 rb[2][NJ];
    bf16x8 af[2][4], wf0[2], wf1;
    if (n_stages > 0) {
        publish(0);
        load_weights(wp, page_lo);
        piece(0);
        piece(kPieces);
        load_rb(rb[0], route_lds + dw * 256, 0);
        load_af_half(af[0], smem, 0, 0);
        wf0[0] = build_w(rb[0][0], wp[0], s2[0]);
    }
    int c_pg = page_lo, c_sp = 0, slot = 0;    // the stage being consumed
    for (int s = 0; s < (kAbBuild && (dbg & 32) ? 1 : n_stages); ++s) {
        const char *st = smem + slot * kStageBytes;
        const char *rt = route_lds + slot * kRouteBytes + dw * 256;
        const bool more = s + 1 < n_stages;
        const bool new_page = more && c_sp + 1 == st_per_page;            // wave-uniform
        if (new_page) load_weights(wpn, c_pg + 1);
#pragma unroll
        for (int hs = 0; hs < 2 * NS; ++hs) {
            const int kk = hs >> 1, hf = hs & 1;
            const int cur = hs & 1, nxt = cur ^ 1;         // A fragment sets
            const int rcur = kk & 1, rnxt = rcur ^ 1;      // routing bytes / first W^T of a k-step
            if (hf == 0) {
                if (kk + 1 < NS) {
                    piece(kk + 1);                         // the rest of stage s + kRing - 1
                    load_rb(rb[rnxt], rt, kk + 1);
                }
                load_af_half(af[nxt], st, kk, 1);
                wf1 = build_w(rb[rcur][1], wp[1], s2[1]);
            } else if (kk + 1 < NS) {
                load_af_half(af[nxt], st, kk + 1, 0);
            } else if (more) {                             // the switch to stage s + 1, in front of the stage's last MFMAs
                publish(s + 1);
                next_stage();                              // stage s + kRing goes where stage s was
                piece(0);
                piece(kPieces);
                const int nslot = slot + 1 == kRing ? 0 : slot + 1;
                load_af_half(af[nxt], smem + nslot * kStageBytes, 0, 0);
                load_rb(rb[rnxt], route_lds + nslot * kRouteBytes + dw * 256, 0);
            }
            if (!(kAbBuild && (dbg & 2))) {
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[0][4 * hf + m] = mfma16<F16>(af[cur][m], wf0[rcur], acc[0][4 * hf + m]);
            }
            if (hf == 1) {
                // the next k-step's first W^T (its routing bytes were requested above or a half step ago; a new page brings new weights)
                if (kk + 1 == NS && new_page) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) wp[j] = wpn[j];
                }
                wf0[rnxt] = build_w(rb[rnxt][0], wp[0], s2[0]);
            }
            if (!(kAbBuild && (dbg & 2))) {
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[1][4 * hf + m] = mfma16<F16>(af[cur][m], wf1, acc[1][4 * hf + m]);
            }
            interleave_hint<8, 2, 2>();
        }
        slot = slot + 1 == kRing ? 0 : slot + 1;
        if (++c_sp == st_per_page) { c_sp = 0; ++c_pg; }
    }
    // ---- this split's partial sums: partial[((split * n_d + c) * Ld + s) * 128 + column]
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int c = doc0 + jd[j];
        const int srow = 16 * jsb[j] + l16;
        if (c >= n_d || srow >= Ld) continue;
        float *o = partial + (((size_t)split * n_d + c) * Ld + srow) * kDim + 4 * l4;
#pragma unroll
        for (int mb = 0; mb < 8; ++mb)
            *reinterpret_cast<float4 *>(o + 16 * mb) = make_float4(acc[j][mb][0], acc[j][mb][1], acc[j][mb][2], acc[j][mb][3]);
    }
}

// dR[c, s, :] = the splits' partials added in split order, rounded once to the embeddings' dtype; one thread per 4 columns
template <bool F16>
__global__ __launch_bounds__(256) void dense_t_bwd_short_sum_kernel(const float *__restrict__ partial, uint16_t *__restrict__ dR,
                                                                    long long n_elems, int n_split) {
    const long long idx = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (idx >= n_elems) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < n_split; ++z) {
        const float4 v = *reinterpret_cast<const float4 *>(partial + (size_t)z * n_elems + idx);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    constexpr int DT = F16 ? kDtypeF16 : kDtypeBf16;
    uint2 o;
    o.x = (uint32_t)float_to_elem16<DT>(s.x) | ((uint32_t)float_to_elem16<DT>(s.y) << 16);
    o.y = (uint32_t)float_to_elem16<DT>(s.z) | ((uint32_t)float_to_elem16<DT>(s.w) << 16);
    *reinterpret_cast<uint2 *>(dR + idx) = o;
}

}  // namespace msim
