// Hierarchical token pooling on the MI355X: the reference pools every page on the CPU with SciPy,
//   colpali_engine/compression/token_pooling/hierarchical_token_pooling.py:83-146
//       similarities = torch.mm(embedding, embedding.t());  distances = 1 - similarities            (:117-118)
//       Z = linkage(distances, metric="euclidean", method="ward")                                   (:120)
//       labels = fcluster(Z, t=max(n // pool_factor, 1), criterion="maxclust") - 1                  (:121-122)
//       pooled[c] = normalize(mean(embedding[labels == c]))                                         (:127-140)
// SciPy's `linkage` on a 2-D array treats the ROWS of the [n, n] matrix as observations: pdist (double) over those rows,
// then the nearest-neighbour-chain Ward algorithm, a stable sort of the merges, union-find relabelling, and for
// `maxclust` the smallest threshold leaving at most t flat clusters, numbered by a depth-first walk of the dendrogram.
// Everything that decides an index is done in IEEE double WITHOUT fused multiply-add, in SciPy's operation order, so
// that equal inputs give equal merges (the tests compare the labels with SciPy's own, ties included).
//
//   pool_gram_kernel     X = 1 - E E^T per page, fp32 (MFMA: bf16/f16 products are exact in fp32; fp32 inputs use the
//                        exact-fp32 MFMA) -- the one step whose rounding cannot be made bit-equal to the CPU GEMM's
//   pool_pdist_kernel    D[i][j] = sqrt(sum_k (X[i][k] - X[j][k])^2), double, sequential in k, square symmetric output
//   pool_cluster_kernel  one workgroup per page: NN-chain Ward (parallel arg-min scans and Lance-Williams updates,
//                        the chain logic itself is sequential by nature), bitonic sort, union-find, k-th smallest
//                        criterion, depth-first numbering -> labels
//   pool_reduce_kernel   mean + L2 normalisation of every cluster's rows, written in cluster-id order
#pragma once
#include "maxsim_common.hpp"
#include "maxsim_generic.hip"

namespace msim {

constexpr int kPoolMaxN = 2048;           // rows of one page the cluster kernel holds in LDS
constexpr int kPoolMaxRows = 32768;       // rows of one page at all (n^2 int32 grid / workspace arithmetic; 12.9 GB of workspace at the cap)
constexpr int kPoolThreads = 256;

struct PoolArgs {
    int n_pages, dim, row_bytes, pool_factor;
};

// ---------------------------------------------------------------------------------------------------------
// X[c][i][j] = 1 - <E[c][i], E[c][j]>   grid = (j slabs, i tiles of 32, page); one wave per (i tile, j slab) pair
template <int DT>
__global__ __launch_bounds__(256) void pool_gram_kernel(const char *__restrict__ E, const int32_t *__restrict__ d_off,
                                                        const int64_t *__restrict__ ws_off, float *__restrict__ X, PoolArgs a) {
    const int c = blockIdx.z;
    const int r0 = d_off[c];
    const int n = d_off[c + 1] - r0;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i0 = blockIdx.y * 32;
    const int j0 = (blockIdx.x * 4 + wave) * 32;
    if (i0 >= n || j0 >= n) return;
    const int half_off = (lane >> 5) * 16;
    const char *base = E + (size_t)r0 * a.row_bytes;
    int ri = i0 + (lane & 31), rj = j0 + (lane & 31);
    ri = ri < n ? ri : n - 1;
    rj = rj < n ? rj : n - 1;
    const char *pa = base + (size_t)ri * a.row_bytes + half_off;
    const char *pb = base + (size_t)rj * a.row_bytes + half_off;
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int n_steps = a.row_bytes >> 5;
#pragma unroll 1
    for (int s = 0; s < n_steps; ++s) {
        const bf16x8 av = *reinterpret_cast<const bf16x8 *>(pa + s * 32);
        const bf16x8 bv = *reinterpret_cast<const bf16x8 *>(pb + s * 32);
        acc = mfma_step<DT>(av, bv, acc);              // rows i -> accumulator rows, rows j -> lane column
    }
    float *out = X + ws_off[c];
    const int j = j0 + (lane & 31);
    if (j < n) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + acc_row(r, lane);
            if (i < n) out[(size_t)i * n + j] = 1.0f - acc[r];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// D[c][i][j] = || X[c][i][:] - X[c][j][:] ||_2 in double, the sum taken in column order with one rounding per product
// and per addition (scipy.spatial.distance.pdist on the float64 copy of X).  16 x 16 pairs per workgroup, X tiles staged
// through LDS in 64-column chunks; only tiles with tj >= ti are computed, the result is mirrored.
__global__ __launch_bounds__(256) void pool_pdist_kernel(const int32_t *__restrict__ d_off, const int64_t *__restrict__ ws_off,
                                                         const float *__restrict__ X, double *__restrict__ Dm) {
#pragma clang fp contract(off)
    __shared__ float xi[16][65], xj[16][65];
    const int c = blockIdx.z;
    const int n = d_off[c + 1] - d_off[c];
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (tj < ti || ti * 16 >= n || tj * 16 >= n) return;
    const float *x = X + ws_off[c];
    double *d = Dm + ws_off[c];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i = ti * 16 + ty, j = tj * 16 + tx;
    double s = 0.0;
    for (int k0 = 0; k0 < n; k0 += 64) {
        __syncthreads();
        for (int e = threadIdx.x; e < 16 * 64; e += 256) {
            const int r = e >> 6, k = e & 63;
            const int gi = ti * 16 + r, gj = tj * 16 + r;
            xi[r][k] = (gi < n && k0 + k < n) ? x[(size_t)gi * n + k0 + k] : 0.0f;
            xj[r][k] = (gj < n && k0 + k < n) ? x[(size_t)gj * n + k0 + k] : 0.0f;
        }
        __syncthreads();
        const int kmax = n - k0 < 64 ? n - k0 : 64;
        for (int k = 0; k < kmax; ++k) {
            const double df = (double)xi[ty][k] - (double)xj[tx][k];
            const double sq = df * df;
            s = s + sq;
        }
    }
    if (i < n && j < n) {
        const double r = sqrt(s);
        d[(size_t)i * n + j] = r;
        d[(size_t)j * n + i] = r;
    }
}

// ---------------------------------------------------------------------------------------------------------
struct PoolLds {
    double zdist[kPoolMaxN];      // merge distance (becomes the sort key, then MC)
    double mc[kPoolMaxN];         // max distance below each node / scratch keys of the selection
    int zx[kPoolMaxN], zy[kPoolMaxN];
    int zorder[kPoolMaxN];        // original merge index (stable sort tie-break), later DFS stack
    int size[kPoolMaxN];
    int chain[kPoolMaxN];
    int parent[2 * kPoolMaxN];
    double red_val[kPoolThreads];
    int red_idx[kPoolThreads];
    int bcast[4];
    double bcast_d;
};

// A page above kPoolMaxN rows keeps the same state in global memory: the page's own region of the fp32 workspace X, which nothing reads
// after pool_pdist_kernel (4 n^2 bytes; the state is at most 72 n).  The reduction scratch stays in LDS.
struct PoolBig {
    double *zdist, *mc;
    int *zx, *zy, *zorder, *size, *chain, *parent;
    double *red_val;
    int *red_idx;
};

// Ward's Lance-Williams update exactly as scipy's _ward evaluates it (left to right, no contraction)
__device__ __forceinline__ double ward_update(double d_xi, double d_yi, double d_xy, int nx, int ny, int ni) {
#pragma clang fp contract(off)
    const double fi = (double)ni;
    const double t = 1.0 / (double)(nx + ny + ni);
    double a = (fi + (double)nx) * t;
    a = a * d_xi;
    a = a * d_xi;
    double b = (fi + (double)ny) * t;
    b = b * d_yi;
    b = b * d_yi;
    double cc = fi * t;
    cc = cc * d_xy;
    cc = cc * d_xy;
    double r = a + b;
    r = r - cc;
    return sqrt(r);
}

template <class State>
__device__ __forceinline__ void pool_cluster_page(State &L, const int n, const int t_max, double *__restrict__ D, int32_t *__restrict__ lab,
                                                  int32_t *__restrict__ n_cluster_out) {
    const int tid = threadIdx.x;

    for (int i = tid; i < n; i += kPoolThreads) L.size[i] = 1;
    __syncthreads();

    // ================= nearest-neighbour chain (scipy _hierarchy.nn_chain, ward)
    int chain_len = 0;                                      // kept identical in every thread
    for (int k = 0; k < n - 1; ++k) {
        if (chain_len == 0) {
            // first active cluster
            int first = n;
            for (int i = tid; i < n; i += kPoolThreads)
                if (L.size[i] > 0) { first = i; break; }
            L.red_idx[tid] = first;
            __syncthreads();
            for (int s = kPoolThreads / 2; s > 0; s >>= 1) {
                if (tid < s && L.red_idx[tid + s] < L.red_idx[tid]) L.red_idx[tid] = L.red_idx[tid + s];
                __syncthreads();
            }
            if (tid == 0) L.chain[0] = L.red_idx[0];
            chain_len = 1;
            __syncthreads();
        }
        int x, y;
        double current_min;
        while (true) {
            x = L.chain[chain_len - 1];
            int y_prev = -1;
            current_min = INFINITY;
            if (chain_len > 1) {
                y_prev = L.chain[chain_len - 2];
                current_min = D[(size_t)x * n + y_prev];
            }
            // arg-min over the active clusters, first index among equal values
            double bv = INFINITY;
            int bi = n;
            const double *row = D + (size_t)x * n;
            for (int i = tid; i < n; i += kPoolThreads) {
                if (L.size[i] == 0 || i == x) continue;
                const double v = row[i];
                if (v < bv) { bv = v; bi = i; }
            }
            L.red_val[tid] = bv;
            L.red_idx[tid] = bi;
            __syncthreads();
            for (int s = kPoolThreads / 2; s > 0; s >>= 1) {
                if (tid < s) {
                    const double ov = L.red_val[tid + s];
                    const int oi = L.red_idx[tid + s];
                    if (ov < L.red_val[tid] || (ov == L.red_val[tid] && oi < L.red_idx[tid])) {
                        L.red_val[tid] = ov;
                        L.red_idx[tid] = oi;
                    }
                }
                __syncthreads();
            }
            const double best_v = L.red_val[0];
            const int best_i = L.red_idx[0];
            __syncthreads();                                 // everyone has read the result before the arrays are reused
            y = y_prev;
            if (best_v < current_min) {                      // strictly smaller than the previous chain element's distance
                current_min = best_v;
                y = best_i;
            }
            if (chain_len > 1 && y == y_prev) break;
            if (tid == 0) L.chain[chain_len] = y;
            chain_len += 1;
            __syncthreads();
        }
        chain_len -= 2;
        if (x > y) { const int tmp = x; x = y; y = tmp; }
        const int nx = L.size[x], ny = L.size[y];
        __syncthreads();                                     // sizes read by everyone before they change
        if (tid == 0) {
            L.zx[k] = x;
            L.zy[k] = y;
            L.zdist[k] = current_min;
            L.zorder[k] = k;
            L.size[x] = 0;
            L.size[y] = nx + ny;
        }
        __syncthreads();
        // Lance-Williams update of the distances to the merged cluster (kept in slot y), rows x and y are contiguous
        const double *rx = D + (size_t)x * n;
        double *ry = D + (size_t)y * n;
        for (int i = tid; i < n; i += kPoolThreads) {
            const int ni = L.size[i];
            if (ni == 0 || i == y) continue;
            const double nd = ward_update(rx[i], ry[i], current_min, nx, ny, ni);
            ry[i] = nd;
            D[(size_t)i * n + y] = nd;
        }
        __threadfence_block();
        __syncthreads();
    }

    // ================= stable sort of the n-1 merges by distance (bitonic on (dist, original index))
    const int m = n - 1;
    int p2 = 1;
    while (p2 < m) p2 <<= 1;
    for (int i = m + tid; i < p2; i += kPoolThreads) { L.zdist[i] = INFINITY; L.zorder[i] = 0x7fffffff; L.zx[i] = 0; L.zy[i] = 0; }
    __syncthreads();
    for (int kk = 2; kk <= p2; kk <<= 1)
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < p2; i += kPoolThreads) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & kk) == 0;
                    const double di = L.zdist[i], dl = L.zdist[l];
                    const int oi = L.zorder[i], ol = L.zorder[l];
                    const bool gt = di > dl || (di == dl && oi > ol);
                    if (gt == up) {
                        L.zdist[i] = dl; L.zdist[l] = di;
                        L.zorder[i] = ol; L.zorder[l] = oi;
                        const int xi_ = L.zx[i], yi_ = L.zy[i];
                        L.zx[i] = L.zx[l]; L.zy[i] = L.zy[l];
                        L.zx[l] = xi_; L.zy[l] = yi_;
                    }
                }
            }
            __syncthreads();
        }

    // ================= union-find relabelling (scipy `label`) and the subtree maxima, sequential by nature
    for (int i = tid; i < 2 * n - 1; i += kPoolThreads) L.parent[i] = i;
    __syncthreads();
    if (tid == 0) {
        int next = n;
        for (int i = 0; i < m; ++i) {
            int xr = L.zx[i], yr = L.zy[i], a_;
            a_ = xr; while (L.parent[a_] != a_) a_ = L.parent[a_];
            { int w = xr; while (L.parent[w] != a_ && w != a_) { const int nw = L.parent[w]; L.parent[w] = a_; w = nw; } }
            xr = a_;
            a_ = yr; while (L.parent[a_] != a_) a_ = L.parent[a_];
            { int w = yr; while (L.parent[w] != a_ && w != a_) { const int nw = L.parent[w]; L.parent[w] = a_; w = nw; } }
            yr = a_;
            L.zx[i] = xr < yr ? xr : yr;
            L.zy[i] = xr < yr ? yr : xr;
            L.parent[xr] = next;
            L.parent[yr] = next;
            ++next;
            double mm = L.zdist[i];
            if (L.zx[i] >= n && L.mc[L.zx[i] - n] > mm) mm = L.mc[L.zx[i] - n];
            if (L.zy[i] >= n && L.mc[L.zy[i] - n] > mm) mm = L.mc[L.zy[i] - n];
            L.mc[i] = mm;
        }
    }
    __syncthreads();

    // ================= maxclust: cutoff = the (n - t)-th smallest criterion value.  A node is a flat cluster root iff its
    // criterion (a subtree maximum, so parents never have a smaller one) is <= cutoff, hence #clusters = n - #{MC <= cutoff}.
    const int kth = n - t_max;                              // >= 1 here
    // rank selection without another sort: count, for every candidate, how many values are <= it
    for (int i = tid; i < m; i += kPoolThreads) {
        const double v = L.mc[i];
        int le = 0;
        for (int j = 0; j < m; ++j) le += L.mc[j] <= v ? 1 : 0;
        L.zorder[i] = le;                                   // #values <= mc[i]
    }
    __syncthreads();
    {
        double bv = INFINITY;                               // smallest v with count(<= v) >= kth
        for (int i = tid; i < m; i += kPoolThreads)
            if (L.zorder[i] >= kth && L.mc[i] < bv) bv = L.mc[i];
        L.red_val[tid] = bv;
        __syncthreads();
        for (int s = kPoolThreads / 2; s > 0; s >>= 1) {
            if (tid < s && L.red_val[tid + s] < L.red_val[tid]) L.red_val[tid] = L.red_val[tid + s];
            __syncthreads();
        }
    }
    const double cutoff = L.red_val[0];
    __syncthreads();

    // ================= depth-first numbering of the flat clusters (scipy cluster_monocrit), sequential
    if (tid == 0) {
        int *stack = L.chain;                                // reuse
        int *visited = L.size;                               // reuse as flags for internal nodes (index node - n)
        for (int i = 0; i < m; ++i) visited[i] = 0;
        int sp = 0, n_cluster = 0, leader = -1;
        stack[0] = 2 * n - 2;
        while (sp >= 0) {
            const int root = stack[sp] - n;
            const int lc = L.zx[root], rc = L.zy[root];
            if (leader == -1 && L.mc[root] <= cutoff) { leader = root; ++n_cluster; }
            if (lc >= n && !visited[lc - n]) { visited[lc - n] = 1; stack[++sp] = lc; continue; }
            if (rc >= n && !visited[rc - n]) { visited[rc - n] = 1; stack[++sp] = rc; continue; }
            if (lc < n) { if (leader == -1) ++n_cluster; lab[lc] = n_cluster - 1; }
            if (rc < n) { if (leader == -1) ++n_cluster; lab[rc] = n_cluster - 1; }
            if (leader == root) leader = -1;
            --sp;
        }
        *n_cluster_out = n_cluster;
    }
}

// One workgroup per page.  BIG = false: every page has at most kPoolMaxN rows and its state lives in LDS (dynamic, sizeof(PoolLds)).
// BIG = true: longer pages carve the same arrays out of their X region (see PoolBig); shorter pages of the same call still use LDS.
template <bool BIG>
__global__ __launch_bounds__(kPoolThreads) void pool_cluster_kernel(const int32_t *__restrict__ d_off,
                                                                    const int64_t *__restrict__ ws_off, float *__restrict__ X,
                                                                    double *__restrict__ Dm,
                                                                    int32_t *__restrict__ labels,      // [total rows], 0-based
                                                                    int32_t *__restrict__ n_clusters,  // [n_pages]
                                                                    int pool_factor) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    PoolLds &L = *reinterpret_cast<PoolLds *>(smem_raw);
    const int c = blockIdx.x;
    const int r0 = d_off[c];
    const int n = d_off[c + 1] - r0;
    const int tid = threadIdx.x;
    int32_t *lab = labels + r0;
    if (n <= 0) { if (tid == 0) n_clusters[c] = 0; return; }
    int t_max = n / pool_factor;
    t_max = t_max < 1 ? 1 : t_max;
    if (n == 1 || pool_factor == 1 || t_max >= n) {        // nothing to merge: every row is its own cluster, in index order
        for (int i = tid; i < n; i += kPoolThreads) lab[i] = i;
        if (tid == 0) n_clusters[c] = n;
        return;
    }
    double *D = Dm + ws_off[c];
    if constexpr (BIG) {
        if (n > kPoolMaxN) {
            int p2 = 1;
            while (p2 < n - 1) p2 <<= 1;                     // the bitonic sort pads the merge list to a power of two
            char *p = reinterpret_cast<char *>(X + ws_off[c]);
            p += (16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15;
            PoolBig B;
            B.zdist = reinterpret_cast<double *>(p);  p += (size_t)p2 * 8;
            B.mc = reinterpret_cast<double *>(p);     p += (size_t)p2 * 8;
            B.zx = reinterpret_cast<int *>(p);        p += (size_t)p2 * 4;
            B.zy = reinterpret_cast<int *>(p);        p += (size_t)p2 * 4;
            B.zorder = reinterpret_cast<int *>(p);    p += (size_t)p2 * 4;
            B.size = reinterpret_cast<int *>(p);      p += (size_t)n * 4;
            B.chain = reinterpret_cast<int *>(p);     p += (size_t)n * 4;
            B.parent = reinterpret_cast<int *>(p);
            B.red_val = L.red_val;
            B.red_idx = L.red_idx;
            pool_cluster_page(B, n, t_max, D, lab, n_clusters + c);
            return;
        }
    }
    pool_cluster_page(L, n, t_max, D, lab, n_clusters + c);
}

// ---------------------------------------------------------------------------------------------------------
// pooled row of every cluster: mean of its member rows (fp32, members added in index order), L2-normalised
// (torch.nn.functional.normalize: v / max(||v||, 1e-12)), cast to the embedding dtype.  One workgroup per page;
// wave w handles clusters w, w + 4, ...
template <int DT>
__global__ __launch_bounds__(256) void pool_reduce_kernel(const char *__restrict__ E, const int32_t *__restrict__ d_off,
                                                          const int32_t *__restrict__ labels, const int32_t *__restrict__ out_off,
                                                          char *__restrict__ out, int dim_logical, int row_bytes_in, int row_bytes_out) {
    constexpr int ES = elem_size<DT>();
    const int c = blockIdx.x;
    const int r0 = d_off[c];
    const int n = d_off[c + 1] - r0;
    const int k = out_off[c + 1] - out_off[c];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char *base = E + (size_t)r0 * row_bytes_in;
    const int32_t *lab = labels + r0;
    for (int cl = wave; cl < k; cl += 4) {
        float ssq = 0.0f;
        // first pass: mean per column (kept in registers for up to 32 columns per lane = dim <= 2048)
        float v[32];
        int count = 0;
#pragma unroll
        for (int u = 0; u < 32; ++u) v[u] = 0.0f;
        for (int i = 0; i < n; ++i) {
            if (lab[i] != cl) continue;
            ++count;
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const int col = lane + u * 64;
                if (col < dim_logical) v[u] += load_elem<DT>(base + (size_t)i * row_bytes_in + (size_t)col * ES);
            }
        }
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            v[u] = v[u] / (float)count;
            const int col = lane + u * 64;
            if (col < dim_logical) ssq += v[u] * v[u];
        }
        for (int s = 32; s > 0; s >>= 1) ssq += __shfl_xor(ssq, s);
        float nrm = sqrtf(ssq);
        nrm = nrm > 1e-12f ? nrm : 1e-12f;
        char *dst = out + (size_t)(out_off[c] + cl) * row_bytes_out;
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int col = lane + u * 64;
            if (col >= dim_logical) continue;
            const float o = v[u] / nrm;
            if constexpr (DT == kDtypeF32) *reinterpret_cast<float *>(dst + (size_t)col * 4) = o;
            else if constexpr (DT == kDtypeF16) *reinterpret_cast<_Float16 *>(dst + (size_t)col * 2) = (_Float16)o;
            else *reinterpret_cast<uint16_t *>(dst + (size_t)col * 2) = (uint16_t)(__float_as_uint(bf16_round(o)) >> 16);
        }
    }
}

}  // namespace msim
