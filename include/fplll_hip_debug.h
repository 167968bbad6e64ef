/*
 * fplll_hip_debug.h — entry points of libfplll_hip.so that are NOT part of the drop-in boundary
 * (include/fplll_hip.h): the host half of two device protocols exposed on its own so that the CPU
 * test-suite can check it without a GPU, and the calibration stream of the profiling recipe.
 * Nothing here is needed to use the library.
 */
#ifndef FPLLL_HIP_DEBUG_H
#define FPLLL_HIP_DEBUG_H

#include "fplll_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* What the mailbox service of fphip_gso_bkz_strategies answers for a block of `bs` rows with the
 * stored r_ii `r[i]` and doubled row exponents `e2[i]`: enumeration radius (scaled like r[0]), the
 * chosen pruning set (index into coeff_off, -1 = none) and its expectation — BKZReduction::
 * svp_reduction's radius / get_pruning arithmetic (bkz.cpp:82-98, 311-323) with the host libm.
 * tests/test_bkzs_host_cpu.py compares it with the oracle bit for bit. */
int fphip_debug_bkz_radius(const fphip_strategies *S, double gh_factor, int bs, int flags, double delta,
                           const double *r, const int *e2, double *max_dist, int *prune,
                           double *expectation);
/* The plan the service draws for rerandomize_block(lo, hi, density) (bkz.cpp:43-80) from `rnd`:
 * plan[0..n_moves) = move_row(b, a) as b | a << 8, then n_ops row additions a | b << 8 | add << 16. */
int fphip_debug_bkz_plan(fphip_rand_fn rnd, void *rnd_user, int lattice, int lo, int hi, int density,
                         unsigned *plan, int *n_moves, int *n_ops);
/* FETCH_SIZE calibration (profiles/r01_fetch_size_calibration.csv): streams `rows` rows of `row_bytes`
 * bytes, `stride` bytes apart, with the sweep kernels' load instruction; milliseconds in *ms_out. */
int fphip_debug_stream(fphip_ctx *ctx, long long rows, int row_bytes, long long stride, double *ms_out);

/* The device's double-double arithmetic (csrc/ftx.h) element-wise on host arrays, for its unit test
 * against multiprecision.  op: 0 add, 1 sub, 2 mul, 3 div, 4 sqrt(a), 5 nint(a). */
int fphip_debug_dd_op(fphip_ctx *ctx, int op, int count, const double *ahi, const double *alo,
                      const double *bhi, const double *blo, double *ohi, double *olo);

#ifdef __cplusplus
}
#endif
#endif
