// Pair-list kernels: MaxSim with arg-max for an explicit list of (query, document) pairs, and the
// backward of the contraction for such a list.  They serve the training losses:
//
//   colpali_engine/loss/late_interaction_losses.py:297-298 (+ :91)
//       raw = einsum("bnd,csd->bcns", Q, D); scores = raw.amax(dim=3).sum(dim=2)
//   autograd of that expression: d scores[b,c] / d raw[b,c,n,s] = [s == argmax_s raw[b,c,n,:]]
//       dQ[b,n,:] = sum_c G[b,c] * D[c, a(b,c,n), :]
//       dD[c,s,:] = sum_b sum_{n : a(b,c,n) = s} G[b,c] * Q[b,n,:]
//   ColbertPairwiseCELoss (:309-313) has exactly two non-zero G entries per query (the positive and
//   the hardest negative), so the backward recomputes the arg-max for those 2B pairs instead of
//   saving the [B,C,Lq,Ld] similarity tensor the reference keeps alive for autograd.
//   The paired variants (:235-238, :381-384: "bnd,bsd->bns", "bnd,blsd->blns") are pair lists too.
//
// Determinism: no floating-point atomics anywhere; every output element has one owner thread that
// accumulates in a fixed order.
// Ties: the first maximal patch wins (the reference's amax backward splits the gradient evenly
// among exact ties; exact ties only occur at all-zero padding rows, whose gradient the model's
// `proj * attention_mask` multiplies by zero -- modeling_colpali.py:72, modeling_colqwen2.py:69).
#pragma once
#include "maxsim_common.hpp"
#include "maxsim_stream.hip"

namespace msim {

constexpr int kPairsRing = 2;

struct PairsArgs {
    int n_q, Lq, n_d, n_pairs;
};

// WPP = 1: one wave per pair (4 pairs per workgroup, wave-private LDS ring as in K1s) -- the throughput form for long pair lists.
// WPP = 4: one WORKGROUP per pair, wave w takes slabs w, w + 4, ... of the document and the four (max, arg-max) sets are combined
// through LDS (lower row wins a tie: the first maximum) -- the latency form for short lists: the pairwise loss recomputes the routing
// of 2B = 64 pairs, and one wave walking 25 slabs one LDS-DMA round trip at a time took 20 us of a 150 us step (rocprofv3, round 5).
// TPQ = ceil(Lq / 32) token tiles of the pair's query live in registers.
// RING: slabs in each wave's private ring.  2 keeps two workgroups on a CU (the throughput form); the latency forms (few workgroups, each
// wave waiting for its own LDS-DMA round trips) take 4: three slabs in flight instead of one.
template <int TPQ, bool F16, int WPP = 1, int RING = kPairsRing>
__global__ __launch_bounds__(256, (RING > 2 ? 1 : 2)) void maxsim_pairs_argmax_kernel(const uint16_t *__restrict__ Q,
                                                                  const uint16_t *__restrict__ D,
                                                                  const int32_t *__restrict__ d_off,
                                                                  const uint8_t *__restrict__ clamp0,
                                                                  const int32_t *__restrict__ pairs,   // [n_pairs, 2]
                                                                  float *__restrict__ out_scores,      // [n_pairs] or null
                                                                  int32_t *__restrict__ out_argmax,    // [n_pairs, Lq] or null
                                                                  PairsArgs a) {
    static_assert(WPP == 1 || WPP == 4, "one wave or one workgroup per pair");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *ring = smem + wave * (RING * kSlabBytes);
    float *comb_m = reinterpret_cast<float *>(smem + 4 * RING * kSlabBytes);        // WPP = 4: [4][TPQ * 32]
    int *comb_a = reinterpret_cast<int *>(comb_m + 4 * TPQ * kTokTile);
    const int gw = WPP == 1 ? blockIdx.x * 4 + wave : blockIdx.x;
    const int GW = WPP == 1 ? gridDim.x * 4 : gridDim.x;
    const int s_first = WPP == 1 ? 0 : wave;

    const int l16 = lane & 15, l4 = lane >> 4;
    int src_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) src_off[j] = l4 * kRowBytes + (((l16 ^ l4) ^ (j << 2)) << 4);
    int rd_off[kKSteps];
#pragma unroll
    for (int ks = 0; ks < kKSteps; ++ks) rd_off[ks] = slab_swizzled_off(lane & 31, 2 * ks + (lane >> 5));

    for (int p = gw; p < a.n_pairs; p += GW) {
        const int q = pairs[2 * p], c = pairs[2 * p + 1];
        if (q < 0 || q >= a.n_q || c < 0 || c >= a.n_d) continue;   // caller error: leave the outputs untouched (uniform per workgroup for WPP = 4)
        // ---- query fragments of this pair
        bf16x8 qf[TPQ][kKSteps];
#pragma unroll
        for (int t = 0; t < TPQ; ++t) {
            const int row = t * kTokTile + (lane & 31);
            const bool valid = row < a.Lq;
            const uint16_t *qp = Q + ((size_t)q * a.Lq + (valid ? row : 0)) * kDim + (lane >> 5) * 8;
#pragma unroll
            for (int ks = 0; ks < kKSteps; ++ks) {
                bf16x8 v = *reinterpret_cast<const bf16x8 *>(qp + ks * 16);
                qf[t][ks] = valid ? v : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
        wait_vmcnt<0>();   // also retires every LDS-DMA / store of the previous pair
#pragma unroll
        for (int t = 0; t < TPQ; ++t)
#pragma unroll
            for (int ks = 0; ks < kKSteps; ++ks) asm volatile("" : "+v"(qf[t][ks]));

        const int r0 = d_off[c];
        const int len = d_off[c + 1] - r0;
        const int nslab = (len + kSlabRows - 1) / kSlabRows;
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void *)(D + (size_t)r0 * kDim), 0, len * kRowBytes, 0x00020000);
        int p_s = s_first, p_slot = 0, c_slot = 0;
        auto produce = [&]() -> bool {
            if (p_s >= nslab) return false;
            char *dst = ring + p_slot * kSlabBytes;
            const int soff = p_s * kSlabBytes;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, MSIM_LDS(dst + i * 1024), 16, src_off[i & 3], soff + i * 1024, 0, 0);
            p_slot = (p_slot + 1 == RING) ? 0 : p_slot + 1;
            p_s += WPP;
            return true;
        };
#pragma unroll
        for (int i = 0; i < RING - 1; ++i) produce();

        // nothing left to request: slab s has landed once at most the (requested - consumed - 1) later slabs are still in flight
        auto wait_tail = [&](int next_req, int cur) {
            const int behind = (next_req - cur) / WPP - 1;       // slabs requested after `cur`
            if (RING > 2 && behind >= 2) wait_vmcnt<16>();
            else if (RING > 2 && behind == 1) wait_vmcnt<8>();
            else wait_vmcnt<0>();
        };
        float m[TPQ];
        int am[TPQ];
#pragma unroll
        for (int t = 0; t < TPQ; ++t) { m[t] = -INFINITY; am[t] = -1; }

        for (int s = s_first; s < nslab; s += WPP) {
            if (produce()) wait_vmcnt<8 * (RING - 1)>(); else wait_tail(p_s, s);
            const char *src = ring + c_slot * kSlabBytes;
            c_slot = (c_slot + 1 == RING) ? 0 : c_slot + 1;
            bf16x8 af[kKSteps];
#pragma unroll
            for (int ks = 0; ks < kKSteps; ++ks) af[ks] = *reinterpret_cast<const bf16x8 *>(src + rd_off[ks]);
            const int row0 = s * kSlabRows;
#pragma unroll
            for (int t = 0; t < TPQ; ++t) {
                f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < kKSteps; ++ks)
                    acc = mfma32<F16>(af[ks], qf[t][ks], acc);
                // rows are visited in increasing order inside a lane, strict '>' keeps the first maximum
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row0 + acc_row(r, lane);
                    const float v = (row < len) ? acc[r] : -INFINITY;
                    if (v > m[t]) { m[t] = v; am[t] = row; }
                }
            }
        }

        bool clamp = false;
        if (clamp0 != nullptr) {
            const uint64_t addr = reinterpret_cast<uint64_t>(clamp0) + (uint64_t)c;
            clamp = ((scalar_load_u32(addr & ~3ull) >> ((addr & 3) * 8)) & 0xffu) != 0;
        }
        float v_t[TPQ];
        int a_t[TPQ];
#pragma unroll
        for (int t = 0; t < TPQ; ++t) {
            const float om = __shfl_xor(m[t], 32);
            const int oam = __shfl_xor(am[t], 32);
            v_t[t] = m[t];
            a_t[t] = am[t];
            if (om > v_t[t] || (om == v_t[t] && (unsigned)oam < (unsigned)a_t[t])) { v_t[t] = om; a_t[t] = oam; }
        }
        if constexpr (WPP > 1) {
            // the four waves' (max, row) per token through LDS; wave 0 keeps the largest value, the lowest row among equals
            if (lane < 32) {
#pragma unroll
                for (int t = 0; t < TPQ; ++t) {
                    comb_m[wave * (TPQ * kTokTile) + t * kTokTile + lane] = v_t[t];
                    comb_a[wave * (TPQ * kTokTile) + t * kTokTile + lane] = a_t[t];
                }
            }
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int t = 0; t < TPQ; ++t) {
                    float v = -INFINITY;
                    int arg = -1;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const float ov = comb_m[w * (TPQ * kTokTile) + t * kTokTile + (lane & 31)];
                        const int oa = comb_a[w * (TPQ * kTokTile) + t * kTokTile + (lane & 31)];
                        if (ov > v || (ov == v && (unsigned)oa < (unsigned)arg)) { v = ov; arg = oa; }
                    }
                    v_t[t] = v;
                    a_t[t] = arg;
                }
            }
            __syncthreads();                                  // the table is free for the workgroup's next pair
            if (wave != 0) continue;
        }
        float total = 0.0f;
#pragma unroll
        for (int t = 0; t < TPQ; ++t) {
            float v = v_t[t];
            int arg = a_t[t];
            if (clamp && !(v >= 0.0f)) { v = 0.0f; arg = -1; }   // the reference's zero padding row wins
            const int tok = t * kTokTile + (lane & 31);
            if (out_argmax != nullptr && lane < 32 && tok < a.Lq) out_argmax[(size_t)p * a.Lq + tok] = arg;
            total += half_wave_sum(v);
        }
        if (out_scores != nullptr && lane == 0) out_scores[p] = total;
    }
}

// ---- ALL (query, document) pairs with the routing: the forward of the in-batch losses whose upstream gradient is dense (ColbertLoss,
// ColbertSigmoidLoss: late_interaction_losses.py:152-164, :444-465), which keeps the per-token arg-max for the backward.  Through the
// pair-list kernel above every one of the B x C pairs streamed ITS document for ITS query: 32 queries x 256 pages = 8192 streams of
// 200 KB, 1.6 GB out of L2 for 51 MB of pages (98 us, rocprofv3 round 5).  Here a wave takes GQ queries at once (GQ x TPQ1 <= 4 token
// tiles in registers) and walks ONE document for all of them: a quarter of the traffic at Lq = 32, the same MFMAs; scores and routing
// leave in the layout of the row-major all-pairs list, scores[q, c] and argmax[(q * n_d + c), token].
template <int TPQ1, int GQ, bool F16>
__global__ __launch_bounds__(256, 2) void maxsim_allpairs_argmax_kernel(const uint16_t *__restrict__ Q,
                                                                     const uint16_t *__restrict__ D,
                                                                     const int32_t *__restrict__ d_off,
                                                                     const uint8_t *__restrict__ clamp0,
                                                                     float *__restrict__ out_scores,      // [n_q, ld] or null
                                                                     long long ld,
                                                                     int32_t *__restrict__ out_argmax,    // [n_q * n_d, Lq] or null
                                                                     PairsArgs a) {
    constexpr int TPQ = TPQ1 * GQ;                         // token tiles a wave holds
    static_assert(TPQ <= 4, "at most four 32-token tiles per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *ring = smem + wave * (kPairsRing * kSlabBytes);
    const int n_groups = (a.n_q + GQ - 1) / GQ;
    const int n_work = n_groups * a.n_d;                   // (query group, document), document-major inside a group
    const int gw = blockIdx.x * 4 + wave, GW = gridDim.x * 4;
    const int l16 = lane & 15, l4 = lane >> 4;
    int src_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) src_off[j] = l4 * kRowBytes + (((l16 ^ l4) ^ (j << 2)) << 4);
    int rd_off[kKSteps];
#pragma unroll
    for (int ks = 0; ks < kKSteps; ++ks) rd_off[ks] = slab_swizzled_off(lane & 31, 2 * ks + (lane >> 5));

    int cur_group = -1;
    bf16x8 qf[TPQ][kKSteps];
    for (int w = gw; w < n_work; w += GW) {
        const int g = w / a.n_d, c = w - g * a.n_d;
        if (g != cur_group) {                               // (at BASELINE config 5's size every wave has exactly one work item)
#pragma unroll
            for (int t = 0; t < TPQ; ++t) {
                const int q = g * GQ + t / TPQ1, row = (t % TPQ1) * kTokTile + (lane & 31);
                const bool valid = q < a.n_q && row < a.Lq;
                const uint16_t *qp = Q + ((size_t)(valid ? q : 0) * a.Lq + (valid ? row : 0)) * kDim + (lane >> 5) * 8;
#pragma unroll
                for (int ks = 0; ks < kKSteps; ++ks) {
                    bf16x8 v = *reinterpret_cast<const bf16x8 *>(qp + ks * 16);
                    qf[t][ks] = valid ? v : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                }
            }
            cur_group = g;
        }
        wait_vmcnt<0>();   // the query fragments; also retires every LDS-DMA / store of the previous work item
#pragma unroll
        for (int t = 0; t < TPQ; ++t)
#pragma unroll
            for (int ks = 0; ks < kKSteps; ++ks) asm volatile("" : "+v"(qf[t][ks]));

        const int r0 = d_off[c];
        const int len = d_off[c + 1] - r0;
        const int nslab = (len + kSlabRows - 1) / kSlabRows;
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void *)(D + (size_t)r0 * kDim), 0, len * kRowBytes, 0x00020000);
        int p_s = 0, p_slot = 0, c_slot = 0;
        auto produce = [&]() -> bool {
            if (p_s >= nslab) return false;
            char *dst = ring + p_slot * kSlabBytes;
            const int soff = p_s * kSlabBytes;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, MSIM_LDS(dst + i * 1024), 16, src_off[i & 3], soff + i * 1024, 0, 0);
            p_slot = (p_slot + 1 == kPairsRing) ? 0 : p_slot + 1;
            ++p_s;
            return true;
        };
        produce();
        float m[TPQ];
        int am[TPQ];
#pragma unroll
        for (int t = 0; t < TPQ; ++t) { m[t] = -INFINITY; am[t] = -1; }
        // one slab; kTail: rows past the document's end are masked -- only the LAST slab can have any, and the two extra VALU
        // operations per element (this kernel is bound by its VALU work: 3 per element for the running arg-max) are kept out of the others
        auto slab_body = [&](int s, auto tail_c) {
            constexpr bool kTail = decltype(tail_c)::value;
            if (produce()) wait_vmcnt<8 * (kPairsRing - 1)>(); else wait_vmcnt<0>();
            const char *src = ring + c_slot * kSlabBytes;
            c_slot = (c_slot + 1 == kPairsRing) ? 0 : c_slot + 1;
            bf16x8 af[kKSteps];
#pragma unroll
            for (int ks = 0; ks < kKSteps; ++ks) af[ks] = *reinterpret_cast<const bf16x8 *>(src + rd_off[ks]);
            const int row0 = s * kSlabRows;
#pragma unroll
            for (int t = 0; t < TPQ; ++t) {
                f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < kKSteps; ++ks) acc = mfma32<F16>(af[ks], qf[t][ks], acc);
#pragma unroll
                for (int r = 0; r < 16; ++r) {                  // rows in increasing order inside a lane: strict '>' keeps the first maximum
                    const int row = row0 + acc_row(r, lane);
                    const float v = (!kTail || row < len) ? acc[r] : -INFINITY;
                    if (v > m[t]) { m[t] = v; am[t] = row; }
                }
            }
        };
        for (int s = 0; s + 1 < nslab; ++s) slab_body(s, std::false_type{});
        if (nslab > 0) slab_body(nslab - 1, std::true_type{});
        bool clamp = false;
        if (clamp0 != nullptr) {
            const uint64_t addr = reinterpret_cast<uint64_t>(clamp0) + (uint64_t)c;
            clamp = ((scalar_load_u32(addr & ~3ull) >> ((addr & 3) * 8)) & 0xffu) != 0;
        }
#pragma unroll
        for (int j = 0; j < GQ; ++j) {
            const int q = g * GQ + j;
            float total = 0.0f;
#pragma unroll
            for (int tt = 0; tt < TPQ1; ++tt) {
                const int t = j * TPQ1 + tt;
                const float om = __shfl_xor(m[t], 32);
                const int oam = __shfl_xor(am[t], 32);
                float v = m[t];
                int arg = am[t];
                if (om > v || (om == v && (unsigned)oam < (unsigned)arg)) { v = om; arg = oam; }
                if (clamp && !(v >= 0.0f)) { v = 0.0f; arg = -1; }   // the reference's zero padding row wins
                const int tok = tt * kTokTile + (lane & 31);
                if (out_argmax != nullptr && q < a.n_q && lane < 32 && tok < a.Lq)
                    out_argmax[((size_t)q * a.n_d + c) * a.Lq + tok] = arg;
                total += half_wave_sum(v);
            }
            if (out_scores != nullptr && q < a.n_q && lane == 0) out_scores[(size_t)q * ld + c] = total;
        }
    }
}

// ---- the TRANSPOSED pair kernel: long queries against short documents -- the trainer's symmetric direction
// (trainer/contrastive_trainer.py:202-206: a 780-token page as `query_embeddings`, a 32-token query as `doc_embeddings`).  The long
// side streams (the query's tokens, 32 per slab, through the same wave-private LDS ring and swizzle), the short side is resident
// (the document's rows, TPD tiles of 32, as the A operand): D[doc row][token] puts one TOKEN per lane column and 16 document rows
// in its registers, so "max over the document's rows" is the same in-lane fold as everywhere else, finished per slab: every slab
// yields 32 finished (max, arg-max) results, nothing is carried across slabs.  One workgroup per pair, wave w takes slabs w, w + 4, ...
// (the results are per token: no cross-wave combine except the score sum, in wave order).  Until round 5 these pairs went to the
// generic kernel (fragment-shaped global loads, one wave per pair): 53 us for the 64 pairs of the pairwise loss.
template <int TPD, bool F16, int RING = 4>
__global__ __launch_bounds__(256, (RING > 2 ? 1 : 2)) void maxsim_pairs_argmax_t_kernel(const uint16_t *__restrict__ Q,
                                                                    const uint16_t *__restrict__ D,
                                                                    const int32_t *__restrict__ d_off,
                                                                    const uint8_t *__restrict__ clamp0,
                                                                    const int32_t *__restrict__ pairs,   // [n_pairs, 2]
                                                                    float *__restrict__ out_scores,      // [n_pairs] or null
                                                                    int32_t *__restrict__ out_argmax,    // [n_pairs, Lq] or null
                                                                    PairsArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *ring = smem + wave * (RING * kSlabBytes);
    float *wave_sum = reinterpret_cast<float *>(smem + 4 * RING * kSlabBytes);      // [4]
    const int l16 = lane & 15, l4 = lane >> 4;
    int src_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) src_off[j] = l4 * kRowBytes + (((l16 ^ l4) ^ (j << 2)) << 4);
    int rd_off[kKSteps];
#pragma unroll
    for (int ks = 0; ks < kKSteps; ++ks) rd_off[ks] = slab_swizzled_off(lane & 31, 2 * ks + (lane >> 5));
    const int nslab = (a.Lq + kSlabRows - 1) / kSlabRows;

    for (int p = blockIdx.x; p < a.n_pairs; p += gridDim.x) {
        const int q = pairs[2 * p], c = pairs[2 * p + 1];
        if (q < 0 || q >= a.n_q || c < 0 || c >= a.n_d) continue;   // caller error: leave the outputs untouched
        const int r0 = d_off[c];
        const int len = d_off[c + 1] - r0;                          // <= 32 * TPD (the host checked max_doc_rows)
        // ---- the document's rows: resident A operands
        bf16x8 df[TPD][kKSteps];
#pragma unroll
        for (int t = 0; t < TPD; ++t) {
            const int row = t * kTokTile + (lane & 31);
            const bool valid = row < len;
            const uint16_t *dp = D + ((size_t)r0 + (valid ? row : 0)) * kDim + (lane >> 5) * 8;
#pragma unroll
            for (int ks = 0; ks < kKSteps; ++ks) {
                bf16x8 v = *reinterpret_cast<const bf16x8 *>(dp + ks * 16);
                df[t][ks] = valid ? v : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
        wait_vmcnt<0>();
#pragma unroll
        for (int t = 0; t < TPD; ++t)
#pragma unroll
            for (int ks = 0; ks < kKSteps; ++ks) asm volatile("" : "+v"(df[t][ks]));
        bool clamp = false;
        if (clamp0 != nullptr) {
            const uint64_t addr = reinterpret_cast<uint64_t>(clamp0) + (uint64_t)c;
            clamp = ((scalar_load_u32(addr & ~3ull) >> ((addr & 3) * 8)) & 0xffu) != 0;
        }
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void *)(Q + (size_t)q * a.Lq * kDim), 0, a.Lq * kRowBytes, 0x00020000);
        int p_s = wave, p_slot = 0, c_slot = 0;
        auto produce = [&]() -> bool {
            if (p_s >= nslab) return false;
            char *dst = ring + p_slot * kSlabBytes;
            const int soff = p_s * kSlabBytes;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, MSIM_LDS(dst + i * 1024), 16, src_off[i & 3], soff + i * 1024, 0, 0);
            p_slot = (p_slot + 1 == RING) ? 0 : p_slot + 1;
            p_s += 4;
            return true;
        };
#pragma unroll
        for (int i = 0; i < RING - 1; ++i) produce();
        float total = 0.0f;
        for (int s = wave; s < nslab; s += 4) {
            if (produce()) {
                wait_vmcnt<8 * (RING - 1)>();
            } else {
                const int behind = (p_s - s) / 4 - 1;             // slabs requested after this one (nothing more can be requested)
                if (RING > 2 && behind >= 2) wait_vmcnt<16>();
                else if (RING > 2 && behind == 1) wait_vmcnt<8>();
                else wait_vmcnt<0>();
            }
            const char *src = ring + c_slot * kSlabBytes;
            c_slot = (c_slot + 1 == RING) ? 0 : c_slot + 1;
            bf16x8 sf[kKSteps];
#pragma unroll
            for (int ks = 0; ks < kKSteps; ++ks) sf[ks] = *reinterpret_cast<const bf16x8 *>(src + rd_off[ks]);
            float m = -INFINITY;
            int am = -1;
#pragma unroll
            for (int t = 0; t < TPD; ++t) {
                f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < kKSteps; ++ks) acc = mfma32<F16>(df[t][ks], sf[ks], acc);
                // document rows are visited in increasing order inside a lane, strict '>' keeps the first maximum
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = t * kTokTile + acc_row(r, lane);
                    const float v = (row < len) ? acc[r] : -INFINITY;
                    if (v > m) { m = v; am = row; }
                }
            }
            const float om = __shfl_xor(m, 32);
            const int oam = __shfl_xor(am, 32);
            if (om > m || (om == m && (unsigned)oam < (unsigned)am)) { m = om; am = oam; }
            if (clamp && !(m >= 0.0f)) { m = 0.0f; am = -1; }       // the reference's zero padding row wins
            const int tok = s * kSlabRows + (lane & 31);
            const bool live = tok < a.Lq;
            if (out_argmax != nullptr && lane < 32 && live) out_argmax[(size_t)p * a.Lq + tok] = am;
            total += live ? m : 0.0f;                               // both halves hold the same value: summed over 32 lanes below
        }
        if (out_scores != nullptr) {
            const float wsum = half_wave_sum(total);                 // lanes 0..31: the sum over this wave's tokens
            __syncthreads();                                         // the previous pair's table has been read
            if (lane == 0) wave_sum[wave] = wsum;
            __syncthreads();
            if (threadIdx.x == 0) out_scores[p] = ((wave_sum[0] + wave_sum[1]) + wave_sum[2]) + wave_sum[3];
        }
    }
}

// first index k in [0, n) with key(k) >= v, key non-decreasing
template <class KeyFn>
__device__ __forceinline__ int lower_bound_idx(int n, int v, KeyFn key) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (key(mid) < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

constexpr int kBwdRows = 64;   // document rows per backward tile (maxsim_bwd.hip)

}  // namespace msim
