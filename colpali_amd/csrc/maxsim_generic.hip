// K1g -- generic MaxSim kernels for gfx950: any embedding width and fp32 embeddings, any query length.
// Same arithmetic as K1s/K1b (colpali_engine/utils/processing_utils.py:179,
// colpali_engine/loss/late_interaction_losses.py:297-298), used whenever the tuned dim=128 16-bit
// kernels do not apply: ColQwen3's dim=320, the reference's own unit-test shape (fp32, dim=32,
// tests/utils/test_processing_utils.py:15-35), fp32 embeddings of a model loaded in fp32, queries
// longer than 128 tokens.
//
// fp32 uses v_mfma_f32_32x32x2_f32: exact fp32 products and accumulation (bitwise an fmaf chain,
// cdna_hip_programming.md section 3) at the fp32 vector rate -- what the reference computes on fp32 tensors
// up to summation order, not a reduced-precision shortcut.
//
// Row layout contract: one embedding row is `row_bytes` = dim * sizeof(elem) bytes, a multiple of 32
// (the host pads the width with zero columns, which changes no dot product) and at most 4 KiB.
//
// Forward structure: a workgroup of 8 waves shares T query token tiles (32 tokens each) staged in LDS,
// rows padded by 16 B so the ds_read_b128 operand fetches are bank-conflict free; each wave walks its own
// documents and loads the A fragments (32 document rows x 32 B per MFMA step) straight from global
// memory, reusing each A fragment for the T resident tiles.  blockIdx.y selects the query group
// (whole queries when a query fits T tiles, otherwise one query processed in ceil(tpq/T) sub-passes whose
// partial sums are accumulated by the same thread in a fixed order: no atomics, deterministic).
#pragma once
#include "maxsim_common.hpp"
#include "maxsim_stream.hip"
#include "maxsim_pairs.hip"

namespace msim {

constexpr int kDtypeF32 = 2;
constexpr int kGenericMaxRowBytes = 4096;
constexpr int kGenericWaves = 8;

struct GenericArgs {
    long long ld;
    int n_q, Lq, n_d;
    int row_bytes;   // dim * element size, multiple of 32
    unsigned flags;
};

typedef __attribute__((ext_vector_type(4))) float f32x4;

// one MFMA "step" = 32 bytes of every row: lanes 0-31 hold bytes [0,16), lanes 32-63 bytes [16,32) of the step
template <int DT>
__device__ __forceinline__ f32x16 mfma_step(const bf16x8 &a, const bf16x8 &b, f32x16 c) {
    if constexpr (DT == kDtypeF32) {
        // the 16 bytes are 4 consecutive fp32 of the lane's row: k = 8*step + 4*(lane>>5) + e; A and B use the same map
        const f32x4 af = __builtin_bit_cast(f32x4, a);
        const f32x4 bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
        for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[e], bf[e], c, 0, 0, 0);
        return c;
    } else {
        return mfma32<DT == kDtypeF16>(a, b, c);
    }
}

template <int DT>
__device__ __forceinline__ float round_generic(float x) {
    if constexpr (DT == kDtypeF32) return x;
    else return round_to_input<DT == kDtypeF16>(x);
}

template <int DT>
__device__ __forceinline__ float load_elem(const char *p) {
    if constexpr (DT == kDtypeF32) return *reinterpret_cast<const float *>(p);
    else return elem_to_float<DT == kDtypeF16>(*reinterpret_cast<const uint16_t *>(p));
}
template <int DT>
constexpr int elem_size() { return DT == kDtypeF32 ? 4 : 2; }

template <int DT, int T>
__global__ __launch_bounds__(kGenericWaves * 64) void maxsim_generic_kernel(const char *__restrict__ Q,
                                                                            const char *__restrict__ D,
                                                                            const int32_t *__restrict__ d_off,
                                                                            const uint8_t *__restrict__ clamp0,
                                                                            float *__restrict__ scores, GenericArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row_bytes = a.row_bytes;
    const int q_stride = row_bytes + 16;          // padded LDS row: (row_bytes/16 + 1) is odd -> conflict free
    const int tile_bytes = kTokTile * q_stride;
    const int n_steps = row_bytes >> 5;
    const int n16 = row_bytes >> 4;               // 16-byte pieces per row
    const int tpq = (a.Lq + kTokTile - 1) / kTokTile;
    const bool whole = tpq <= T;                  // a group = floor(T / tpq) whole queries, one pass
    const int qpg = whole ? T / tpq : 1;
    const int n_pass = whole ? 1 : (tpq + T - 1) / T;
    const int g = blockIdx.y;
    const int gw = blockIdx.x * kGenericWaves + wave, GW = gridDim.x * kGenericWaves;
    const bool ref_round = (a.flags & kFlagRefBf16) != 0;
    const int half_off = (lane >> 5) * 16;

    for (int pass = 0; pass < n_pass; ++pass) {
        // ---- stage the T token tiles of this pass (zero rows beyond Lq / beyond the last query)
        if (pass > 0) __syncthreads();
        for (int idx = threadIdx.x; idx < T * kTokTile * n16; idx += kGenericWaves * 64) {
            const int t = idx / (kTokTile * n16);
            const int rem = idx - t * (kTokTile * n16);
            const int r = rem / n16, p = rem - r * n16;
            const int q = whole ? g * qpg + t / tpq : g;
            const int tt = whole ? t % tpq : pass * T + t;
            const int tok = tt * kTokTile + r;
            const bool valid = (whole ? t < qpg * tpq : tt < tpq) && q < a.n_q && tok < a.Lq;
            i32x4 v = {0, 0, 0, 0};
            if (valid) v = *reinterpret_cast<const i32x4 *>(Q + ((size_t)q * a.Lq + tok) * row_bytes + p * 16);
            *reinterpret_cast<i32x4 *>(smem + t * tile_bytes + r * q_stride + p * 16) = v;
        }
        __syncthreads();
        const char *q_lds = smem + (lane & 31) * q_stride + half_off;

        for (int c = gw; c < a.n_d; c += GW) {
            const int r0 = d_off[c];
            const int len = d_off[c + 1] - r0;
            const char *doc = D + (size_t)r0 * row_bytes;
            float m[T];
#pragma unroll
            for (int t = 0; t < T; ++t) m[t] = -INFINITY;
            for (int s0 = 0; s0 < len; s0 += kSlabRows) {
                int row = s0 + (lane & 31);
                row = row < len ? row : len - 1;   // stay inside the document; masked below
                const char *arow = doc + (size_t)row * row_bytes + half_off;
                f32x16 acc[T];
#pragma unroll
                for (int t = 0; t < T; ++t) acc[t] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                int j = 0;
#pragma unroll 1
                for (; j + 4 <= n_steps; j += 4) {   // 4 A fragments in flight per trip (explicit: the trip count is a run-time value)
                    bf16x8 av[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) av[u] = *reinterpret_cast<const bf16x8 *>(arow + (j + u) * 32);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int t = 0; t < T; ++t) {
                            const bf16x8 bv = *reinterpret_cast<const bf16x8 *>(q_lds + t * tile_bytes + (j + u) * 32);
                            acc[t] = mfma_step<DT>(av[u], bv, acc[t]);
                        }
                }
#pragma unroll 1
                for (; j < n_steps; ++j) {
                    const bf16x8 av = *reinterpret_cast<const bf16x8 *>(arow + j * 32);
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        const bf16x8 bv = *reinterpret_cast<const bf16x8 *>(q_lds + t * tile_bytes + j * 32);
                        acc[t] = mfma_step<DT>(av, bv, acc[t]);
                    }
                }
                const int rows_left = len - s0;
                if (rows_left < kSlabRows) {
#pragma unroll
                    for (int t = 0; t < T; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (acc_row(r, lane) >= rows_left) acc[t][r] = -INFINITY;
                }
#pragma unroll
                for (int t = 0; t < T; ++t) m[t] = fold_max16(m[t], acc[t]);
            }
            const bool clamp = clamp0 != nullptr && clamp0[c] != 0;
            float tile_sum[T];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                float v = fmaxf(m[t], __shfl_xor(m[t], 32));
                if (clamp) v = fmaxf(v, 0.0f);
                if (ref_round) v = round_generic<DT>(v);
                tile_sum[t] = half_wave_sum(v);
            }
            if (lane == 0) {
                if (whole) {
                    for (int qq = 0; qq < qpg; ++qq) {
                        const int q = g * qpg + qq;
                        if (q >= a.n_q) break;
                        float tot = 0.0f;
#pragma unroll
                        for (int t = 0; t < T; ++t)
                            if (t >= qq * tpq && t < (qq + 1) * tpq) tot += tile_sum[t];
                        if (ref_round) tot = round_generic<DT>(tot);
                        scores[(size_t)q * a.ld + c] = tot;
                    }
                } else {
                    float *dst = scores + (size_t)g * a.ld + c;
                    float tot = pass > 0 ? *dst : 0.0f;   // this same thread wrote the earlier passes of (g, c)
#pragma unroll
                    for (int t = 0; t < T; ++t)
                        if (pass * T + t < tpq) tot += tile_sum[t];
                    if (ref_round && pass == n_pass - 1) tot = round_generic<DT>(tot);
                    *dst = tot;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Pair list: MaxSim (+ arg-max routing) of explicit (query, document) pairs, generic width / dtype.
// One wave per pair; both operands are fragment-shaped global loads (the 32 query rows of a tile stay in L1).
// Semantics identical to maxsim_pairs_argmax_kernel (first maximum wins, -1 = the zero padding row).
template <int DT>
__global__ __launch_bounds__(256) void maxsim_generic_pairs_argmax_kernel(const char *__restrict__ Q,
                                                                          const char *__restrict__ D,
                                                                          const int32_t *__restrict__ d_off,
                                                                          const uint8_t *__restrict__ clamp0,
                                                                          const int32_t *__restrict__ pairs,
                                                                          float *__restrict__ out_scores,
                                                                          int32_t *__restrict__ out_argmax,
                                                                          PairsArgs a, int row_bytes) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gw = blockIdx.x * 4 + wave, GW = gridDim.x * 4;
    const int n_steps = row_bytes >> 5;
    const int tpq = (a.Lq + kTokTile - 1) / kTokTile;
    const int half_off = (lane >> 5) * 16;

    for (int p = gw; p < a.n_pairs; p += GW) {
        const int q = pairs[2 * p], c = pairs[2 * p + 1];
        if (q < 0 || q >= a.n_q || c < 0 || c >= a.n_d) continue;   // caller error: leave the outputs untouched
        const int r0 = d_off[c];
        const int len = d_off[c + 1] - r0;
        const char *doc = D + (size_t)r0 * row_bytes;
        const bool clamp = clamp0 != nullptr && clamp0[c] != 0;
        float total = 0.0f;
        for (int tt = 0; tt < tpq; ++tt) {
            const int tok = tt * kTokTile + (lane & 31);
            const bool tok_valid = tok < a.Lq;
            const char *qrow = Q + ((size_t)q * a.Lq + (tok_valid ? tok : 0)) * row_bytes + half_off;
            float m = -INFINITY;
            int am = -1;
            for (int s0 = 0; s0 < len; s0 += kSlabRows) {
                int row = s0 + (lane & 31);
                row = row < len ? row : len - 1;
                const char *arow = doc + (size_t)row * row_bytes + half_off;
                f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                int j = 0;
#pragma unroll 1
                for (; j + 4 <= n_steps; j += 4) {
                    bf16x8 av[4], bv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        av[u] = *reinterpret_cast<const bf16x8 *>(arow + (j + u) * 32);
                        bv[u] = *reinterpret_cast<const bf16x8 *>(qrow + (j + u) * 32);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (!tok_valid) bv[u] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                        acc = mfma_step<DT>(av[u], bv[u], acc);
                    }
                }
#pragma unroll 1
                for (; j < n_steps; ++j) {
                    const bf16x8 av = *reinterpret_cast<const bf16x8 *>(arow + j * 32);
                    bf16x8 bv = *reinterpret_cast<const bf16x8 *>(qrow + j * 32);
                    if (!tok_valid) bv = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    acc = mfma_step<DT>(av, bv, acc);
                }
                // rows are visited in increasing order inside a lane, strict '>' keeps the first maximum
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = s0 + acc_row(r, lane);
                    const float v = (rr < len) ? acc[r] : -INFINITY;
                    if (v > m) { m = v; am = rr; }
                }
            }
            const float om = __shfl_xor(m, 32);
            const int oam = __shfl_xor(am, 32);
            float v = m;
            int arg = am;
            if (om > v || (om == v && (unsigned)oam < (unsigned)arg)) { v = om; arg = oam; }
            if (clamp && !(v >= 0.0f)) { v = 0.0f; arg = -1; }   // the reference's zero padding row wins
            if (out_argmax != nullptr && lane < 32 && tok_valid) out_argmax[(size_t)p * a.Lq + tok] = arg;
            total += half_wave_sum(v);
        }
        if (out_scores != nullptr && lane == 0) out_scores[p] = total;
    }
}

}  // namespace msim

namespace msim {

// ---------------------------------------------------------------------------------------------------------
// Plain similarity matrix  out[i, j] = <A[i, :], B[j, :]>  (fp32 accumulate, no reduction), any width / dtype:
//   colpali_engine/utils/processing_utils.py:126       torch.einsum("bd,cd->bc", qs, ps)          (score_single_vector)
//   colpali_engine/interpretability/similarity_map_utils.py:50   torch.einsum("nk,ijk->nij", query, image_grid)
// Workgroup = 8 waves sharing T tiles of 32 A rows in LDS (K1g staging); every wave walks 32-row slabs of B with
// fragment-shaped global loads.  MFMA roles are chosen so that one lane holds one B row (= output column): a store
// instruction writes 32 consecutive floats of an output row.
struct SimArgs {
    long long ld;          // leading dimension of out
    int n_a, n_b, row_bytes;
    unsigned flags;        // kFlagRefBf16: round every dot product to the input dtype (what torch stores)
};

template <int DT, int T>
__global__ __launch_bounds__(kGenericWaves * 64) void sim_matrix_kernel(const char *__restrict__ A, const char *__restrict__ B,
                                                                        float *__restrict__ out, SimArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row_bytes = a.row_bytes;
    const int q_stride = row_bytes + 16;
    const int tile_bytes = kTokTile * q_stride;
    const int n_steps = row_bytes >> 5;
    const int n16 = row_bytes >> 4;
    const int a0 = blockIdx.y * (T * kTokTile);          // first A row of this workgroup's group of tiles
    const int half_off = (lane >> 5) * 16;
    const bool ref_round = (a.flags & kFlagRefBf16) != 0;

    for (int idx = threadIdx.x; idx < T * kTokTile * n16; idx += kGenericWaves * 64) {
        const int r = idx / n16, p = idx - r * n16;       // r = row inside the group (0 .. 32T-1)
        i32x4 v = {0, 0, 0, 0};
        if (a0 + r < a.n_a) v = *reinterpret_cast<const i32x4 *>(A + (size_t)(a0 + r) * row_bytes + p * 16);
        *reinterpret_cast<i32x4 *>(smem + (r >> 5) * tile_bytes + (r & 31) * q_stride + p * 16) = v;
    }
    __syncthreads();
    const char *a_lds = smem + (lane & 31) * q_stride + half_off;

    const int n_slabs = (a.n_b + kSlabRows - 1) / kSlabRows;
    for (int s = blockIdx.x * kGenericWaves + wave; s < n_slabs; s += gridDim.x * kGenericWaves) {
        const int b0 = s * kSlabRows;
        int brow = b0 + (lane & 31);
        brow = brow < a.n_b ? brow : a.n_b - 1;
        const char *bptr = B + (size_t)brow * row_bytes + half_off;
        f32x16 acc[T];
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        int j = 0;
#pragma unroll 1
        for (; j + 4 <= n_steps; j += 4) {
            bf16x8 bv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) bv[u] = *reinterpret_cast<const bf16x8 *>(bptr + (j + u) * 32);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const bf16x8 av = *reinterpret_cast<const bf16x8 *>(a_lds + t * tile_bytes + (j + u) * 32);
                    acc[t] = mfma_step<DT>(av, bv[u], acc[t]);   // A rows -> accumulator rows, B rows -> lane column
                }
        }
#pragma unroll 1
        for (; j < n_steps; ++j) {
            const bf16x8 bv = *reinterpret_cast<const bf16x8 *>(bptr + j * 32);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const bf16x8 av = *reinterpret_cast<const bf16x8 *>(a_lds + t * tile_bytes + j * 32);
                acc[t] = mfma_step<DT>(av, bv, acc[t]);
            }
        }
        const int col = b0 + (lane & 31);
        if (col < a.n_b) {
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int arow = a0 + t * kTokTile + acc_row(r, lane);
                    if (arow < a.n_a) out[(size_t)arow * a.ld + col] = ref_round ? round_generic<DT>(acc[t][r]) : acc[t][r];
                }
        }
    }
}

}  // namespace msim
