// C ABI of tools/probe/libmaxsim_probe.so (include/maxsim_probe.h): measurement aids only -- the product library
// (colpali_amd/csrc/) contains none of this.  Built by `make -C tools/probe` (and by __graft_entry__.build()).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/maxsim.h"
#include "../../include/maxsim_probe.h"
#include "probe_stream.hip"
#include "probe_mfma.hip"

namespace {
thread_local char g_err[512] = "";
int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace

extern "C" {

const char *msim_probe_last_error(void) { return g_err; }

int msim_probe_stream(int variant, const void *X, int64_t rows, int row_elems, float *sink, void *stream) {
    if (!X || !sink) return fail(MSIM_EINVAL, "null pointer argument");
    if (rows <= 0 || row_elems <= 0) return fail(MSIM_EINVAL, "bad size (rows=%lld row_elems=%d)", (long long)rows, row_elems);
    if (reinterpret_cast<uintptr_t>(X) & 15) return fail(MSIM_EINVAL, "X must be 16-byte aligned");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const char *x = static_cast<const char *>(X);
    const int row_bytes = row_elems * 2;
    int piece = 0, tile_rows = 256;
    switch (variant) {
        case MSIM_PROBE_ROWS256B: piece = 256; tile_rows = 128; break;
        case MSIM_PROBE_PIECES128B: piece = 128; break;
        case 11: case 12: case 13: piece = 128; break;          // round 3: the 128-byte pattern with other lane -> (row, chunk) mappings
        case MSIM_PROBE_PIECES512B: piece = 512; tile_rows = 128; break;
        case 21: piece = 512; tile_rows = 64; break;             // round 5: K3's split-K candidates -- what still fits the LDS
        case 22: piece = 512; tile_rows = 128; break;
        case 23: piece = 256; tile_rows = 128; break;
        default: return fail(MSIM_EINVAL, "unknown probe variant %d", variant);
    }
    if (row_bytes % piece != 0 || rows % tile_rows != 0)
        return fail(MSIM_EUNSUPPORTED, "probe variant %d needs rows %% %d == 0 and a row of a multiple of %d bytes", variant, tile_rows, piece);
    if (rows * (int64_t)row_bytes / tile_rows > 0x7fffffff) return fail(MSIM_EUNSUPPORTED, "matrix too large for the probe");
    int rc;
    switch (variant) {
        case MSIM_PROBE_ROWS256B: rc = run_probe<256, 32, 4, 4>(x, rows, row_elems, sink, st); break;     // K1s: 4 waves, 4 slabs of 8 KiB each
        case MSIM_PROBE_PIECES128B: rc = run_probe<128, 32, 4, 8>(x, rows, row_elems, sink, st); break;   // K3's hidden-state stream
        case 11: rc = run_probe<128, 32, 4, 8, 1>(x, rows, row_elems, sink, st); break;   // 128-byte pieces, rows of an instruction 16 KiB apart
        case 12: rc = run_probe<128, 32, 4, 8, 2>(x, rows, row_elems, sink, st); break;   // ... every row of an instruction at another K chunk
        case 13: rc = run_probe<128, 32, 4, 8, 3>(x, rows, row_elems, sink, st); break;   // ... every instruction of a wave at another K chunk
        // round 5 (the review's "measured freeze" of K3): the widened visits in the shapes the CU can hold next to the weight ring
        case 21: rc = run_probe<512, 8, 3, 8>(x, rows, row_elems, sink, st); break;    // 512-byte pieces x 8 rows x 8 waves x 3 slots = 96 KiB
        case 22: rc = run_probe<512, 16, 2, 8>(x, rows, row_elems, sink, st); break;   // 512-byte pieces x 16 rows x 8 waves x 2 slots = 128 KiB
        case 23: rc = run_probe<256, 16, 3, 8>(x, rows, row_elems, sink, st); break;   // 256-byte pieces x 16 rows x 8 waves x 3 slots = 96 KiB
        default: rc = run_probe<512, 16, 2, 8>(x, rows, row_elems, sink, st); break;
    }
    if (rc) return fail(MSIM_ELAUNCH, "probe_stream_kernel launch failed (variant %d)", variant);
    return MSIM_OK;
}

int msim_probe_mfma(int variant, const void *X, int64_t rows, int iters, float *sink, void *stream) {
    if (!X || !sink) return fail(MSIM_EINVAL, "null pointer argument");
    if (reinterpret_cast<uintptr_t>(X) & 15) return fail(MSIM_EINVAL, "X must be 16-byte aligned");
    if (rows < 256LL * 8 * 5 * 32) return fail(MSIM_EINVAL, "the MFMA probe needs at least %d rows of operands", 256 * 8 * 5 * 32);
    if (variant >= 8 && rows < 256LL * 16 * 3 * 32) return fail(MSIM_EINVAL, "MFMA probe variants 8..11 need at least %d rows of operands", 256 * 16 * 3 * 32);
    if (variant >= 25 && rows < 256LL * 4 * 11 * 32) return fail(MSIM_EINVAL, "MFMA probe variants 25..28 need at least %d rows of operands", 256 * 4 * 11 * 32);
    if (iters <= 0 || variant < 0 || variant > 28) return fail(MSIM_EINVAL, "bad probe arguments (variant=%d iters=%d)", variant, iters);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const uint16_t *x = static_cast<const uint16_t *>(X);
    int rc;
    switch (variant) {
        case 0: rc = run_probe_mfma<false, false>(x, iters, sink, st); break;
        case 1: rc = run_probe_mfma<true, false>(x, iters, sink, st); break;
        case 2: rc = run_probe_mfma<false, true>(x, iters, sink, st); break;
        case 4: rc = run_probe_mfma16<false, false>(x, iters, sink, st); break;   // 16x16x32 tiles, registers only
        case 5: rc = run_probe_mfma16<true, false>(x, iters, sink, st); break;    // + A fragments from LDS
        case 6: rc = run_probe_mfma16<false, true>(x, iters, sink, st); break;    // + max folds
        case 7: rc = run_probe_mfma16<true, true>(x, iters, sink, st); break;     // K1s / K1b's instruction mix
        // the same mix with three / four waves per SIMD (FLOP = 256 x 12 x iters x 48 x 16384 resp. 256 x 16 x iters x 32 x 16384)
        case 8: rc = run_probe_mfma16w<3, 12, true, true>(x, iters, sink, st); break;
        case 9: rc = run_probe_mfma16w<2, 16, true, true>(x, iters, sink, st); break;
        case 10: rc = run_probe_mfma16w<3, 12, false, false>(x, iters, sink, st); break;
        case 11: rc = run_probe_mfma16w<2, 16, false, false>(x, iters, sink, st); break;
        // round 3: K1b's exact slab body (FLOP = 256 x WAVES x iters x NT x 16 x 16384); 12 = the shipped plan, 13..17 = one
        // 512-register wave per SIMD with 8 / 6 token tiles per operand fetch
        case 12: rc = run_probe_mix<4, 8, true>(x, iters, sink, st); break;
        case 13: rc = run_probe_mix<8, 4, true>(x, iters, sink, st); break;
        case 14: rc = run_probe_mix<8, 4, false>(x, iters, sink, st); break;
        case 15: rc = run_probe_mix<6, 4, true>(x, iters, sink, st); break;
        case 16: rc = run_probe_mix<4, 8, false>(x, iters, sink, st); break;
        case 17: rc = run_probe_mix<4, 4, true>(x, iters, sink, st); break;      // the shipped tile count, ONE wave per SIMD
        case 18: rc = run_probe_mix<8, 4, true, true>(x, iters, sink, st); break;  // 13 with the next slab's fragments prefetched
        case 19: rc = run_probe_mix<6, 4, true, true>(x, iters, sink, st); break;
        case 20: rc = run_probe_mix<8, 4, true, true, 1>(x, iters, sink, st); break;   // 18 + the fold of tile t-1 under the MFMAs of tile t
        case 21: rc = run_probe_mix<8, 4, true, true, 2>(x, iters, sink, st); break;   // ... with the interleave pinned (sched_group_barrier)
        case 22: rc = run_probe_mix<6, 4, true, true, 1>(x, iters, sink, st); break;
        case 23: rc = run_probe_mix<8, 4, false, false, 1>(x, iters, sink, st); break;  // A in registers, deferred folds
        case 24: rc = run_probe_mix<5, 8, true>(x, iters, sink, st); break;              // two waves per SIMD x FIVE tiles (4 of them in AGPRs)
        // round 6, the ridge (10 queries x 32 tokens = 20 units = 10 tiles): ONE 512-register wave per SIMD holding all of them
        // (320 B-operand registers, 192 of them AGPRs), the body a barrier-free band would run -- no DMA, no barrier: an upper bound
        case 25: rc = run_probe_mix<10, 4, true>(x, iters, sink, st); break;
        case 26: rc = run_probe_mix<10, 4, true, true>(x, iters, sink, st); break;
        case 27: rc = run_probe_mix<10, 4, true, true, 1>(x, iters, sink, st); break;
        case 28: rc = run_probe_mix<10, 4, false, false, 1>(x, iters, sink, st); break;
        default: rc = run_probe_mfma<true, true>(x, iters, sink, st); break;
    }
    if (rc) return fail(MSIM_ELAUNCH, "probe_mfma_kernel launch failed (variant %d)", variant);
    return MSIM_OK;
}

}  // extern "C"
