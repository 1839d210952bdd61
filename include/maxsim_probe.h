/*
 * include/maxsim_probe.h -- C ABI of tools/probe/libmaxsim_probe.so: MEASUREMENT AIDS, not part of the product.
 *
 * The shipped library (colpali_amd/csrc/libmaxsim_gfx950.so, include/maxsim.h) contains none of this.  bench.py and the
 * tools/ scripts load the probe library to measure what THIS machine can deliver for the kernels' access pattern and
 * instruction mix (a streaming ceiling, a matrix-core ceiling under the socket's power budget), so that a kernel's
 * roofline fraction can be read against the machine and not only against the spec sheet.  No reference counterpart.
 * Same conventions as include/maxsim.h (plain C types, caller-owned buffers, asynchronous on `stream`, 0 / negative codes).
 */
#ifndef COLPALI_AMD_MAXSIM_PROBE_H
#define COLPALI_AMD_MAXSIM_PROBE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char *msim_probe_last_error(void);

/*
 * Measurement aid (no reference counterpart): the streaming ceiling of this machine for the access patterns
 * of the kernels above.  Pulls the row-major 16-bit matrix X [rows, row_elems] through LDS once with the kernels'
 * own LDS-DMA instruction, cache policy and ring discipline and does nothing else; the caller times the launch
 * (bench.py reports bytes / time next to the 8 TB/s spec figure).
 *   MSIM_PROBE_ROWS256B    256-byte pieces of 32 rows per wave and ring slot; with row_elems = 128 these are whole
 *                          rows, i.e. msim_fwd's document stream
 *   MSIM_PROBE_PIECES128B  128-byte pieces of 32 rows per wave                   (msim_embed_head's hidden states)
 *   MSIM_PROBE_PIECES512B  512-byte pieces of 16 rows per wave                   (the best pattern found for wide rows)
 * rows must be a multiple of 256 and the row a multiple of the piece; sink = 4 bytes of device memory (never
 * written in practice: it only keeps the loads alive).
 */
#define MSIM_PROBE_ROWS256B 0
#define MSIM_PROBE_PIECES128B 1
#define MSIM_PROBE_PIECES512B 2
int msim_probe_stream(int variant, const void *X, int64_t rows, int row_elems, float *sink, void *stream);

/*
 * Measurement aid (no reference counterpart): the matrix-core ceiling of this machine under its own power budget for the
 * MaxSim kernels' MFMA (v_mfma_f32_32x32x16_bf16, two waves per SIMD, four 32-token tiles per wave) on the operand values
 * the scorer multiplies.  X = row-major [rows, 128] bf16 (unit-norm rows), rows >= 256 * 8 * 5 * 32 = 327 680; every wave
 * keeps 5 tiles of X in registers / LDS and issues `iters` x 32 MFMAs; no HBM traffic, no barrier.  The caller times the
 * launch: FLOP = 256 workgroups x 8 waves x iters x 32 x 32768.
 *   variant bit 0: A operand re-read from LDS per k-step (msim_fwd's operand path) instead of held in registers
 *   variant bit 1: the 16 -> 1 max fold of every accumulator tile runs next to the MFMAs
 *   variants 4..7: the same four mixes on v_mfma_f32_16x16x32_bf16, the tile shape msim_fwd's kernels use (variant - 4 = the bits
 *   above): 7 = their instruction mix (A fragments from LDS + max folds), 4 = MFMAs alone
 *   variants 8..11 (rows >= 256 * 16 * 3 * 32 = 393 216): the 16x16x32 mix with MORE waves per SIMD -- 8: 12 waves x 3 tiles, 9: 16 waves x
 *   2 tiles (both A from LDS + folds), 10 / 11: the same two shapes with everything in registers.  FLOP = 256 x 12 x iters x 48 x 16384
 *   (8, 10) resp. 256 x 16 x iters x 32 x 16384 (9, 11).
 *   variants 12..23 (round 3, iters even): msim_fwd's EXACT slab body (8 fragment reads per 32-row slab, then per token tile 16
 *   v_mfma_f32_16x16x32 + 8 v_max3) under other register plans -- 12: the shipped plan (8 waves x 4 tiles), 16: the same with the
 *   fragments in registers, 17: 4 waves x 4 tiles; 13 / 15: ONE 512-register wave per SIMD x 8 / 6 tiles (B operands in AGPRs),
 *   14: 13 with the fragments in registers, 18 / 19: 13 / 15 with the next slab's fragments prefetched, 20 / 21 / 22: + the fold
 *   of tile t-1 under tile t (21: interleave pinned), 23: 14 with deferred folds.  FLOP = 256 x waves x iters x tiles x 16 x 16384.
 */
int msim_probe_mfma(int variant, const void *X, int64_t rows, int iters, float *sink, void *stream);

#ifdef __cplusplus
}
#endif

#endif /* COLPALI_AMD_MAXSIM_PROBE_H */
