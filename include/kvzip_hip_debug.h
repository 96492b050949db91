/* kvzip_hip_debug.h - TEST HOOKS of libkvzip_hip.so (not part of the drop-in boundary; include/kvzip_hip.h is).
 * They expose pieces of the kernels' arithmetic to the test-suite: the rounding chain on raw 16-bit patterns, the static partition
 * of the row-statistics pass, the invariant-divisor arithmetic.  tests/test_abi.py checks that they are exported and bound. */
#ifndef KVZIP_HIP_DEBUG_H
#define KVZIP_HIP_DEBUG_H
#include "kvzip_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Test hook for the rounding chain of a1: out[i] = half( float(in[i]) / float(sqrt(D)) ) computed exactly as the
 * scoring kernels do (exact-reciprocal multiply when the host's exhaustive search found one, IEEE division
 * otherwise or when force_division != 0).  rcp_used (host pointer, optional) receives the constant (0 = division). */
int kvz_debug_round_chain(const void* in_bits, int n, int D, int dtype, int force_division, void* out_bits,
                          float* rcp_used, kvz_stream_t stream);
/* test hook, host only: the static partition of the row-statistics pass (kvz_score.hip, PaPlan) for a geometry.
 * unit / tile: 257 entries each; block b owns the key tiles from (unit[b], tile[b]) up to (unit[b+1], tile[b+1]). */
int kvz_debug_score_plan(int sink, int m, int q_len, int G, int Hkv, uint16_t* unit, uint16_t* tile,
                         int* n_blocks, int* max_seg, int* rows_per_unit);
/* test hook, host only: n / d and n % d as the kernels compute them (multiply-shift by a launch-invariant divisor). */
int kvz_debug_fastdiv(int d, int n, int* quotient, int* remainder);

/* test / tuning hook: set a tuning knob of the library ("attn_items", "flash_min_rows", "flash2_min_blocks", "flash2_xcd", "flash2_split",
 * "sel_blocks", "emit_blocks", "score_prune"; value <= 0 restores the default - the on / off knobs flash2_xcd, flash2_split and score_prune
 * take 0 as "off" and negative values as "default") and return its previous value (< 0: unknown name).
 * "score_prune" (both dtypes, deferred-log entry points, chunks of >= 32 query positions): 3 (default) = key-per-lane row statistics +
 * candidate keys per row group + gathered column maxima, 5 = candidate (group, key block) pairs (round 5), 0 = two full passes over Q.K^T,
 * 1 / 4 = check variants, 16 + mask = launches left out (time measurements only, results unusable) (kvz_score.hip); KVZIP_SCORE_PRUNE in the
 * environment presets it.
 * Process-wide, not thread-safe: for tests and probes.  kvz_debug_get_tunable: the current value. */
int kvz_debug_set_tunable(const char* name, int value);
int kvz_debug_get_tunable(const char* name);

/* measurement hook (round 6): a plain 16-bytes-per-lane copy KERNEL, dst[0 .. nbytes) = src[0 .. nbytes) (both 16-byte aligned, nbytes a
 * multiple of 16) - the yardstick the micro-architecture guide quotes for achievable HBM bandwidth (read + write), next to the runtime's
 * own device-to-device copy.  variant 0: ordinary loads / stores, 1: non-temporal (the compaction kernel's kind).  bench.py times both on
 * the box it runs on and reports the gather against the better one (roofline_stages.compact_gather.frac_of_box_copy_kernel). */
int kvz_debug_copy_kernel(void* dst, const void* src, size_t nbytes, int variant, kvz_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
