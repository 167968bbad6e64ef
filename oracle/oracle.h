/*
 * oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement ("port") of the reference algorithms on the hot path, used by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the *checker*.  The product
 * path (fplll_amd/, include/) never links, loads or calls anything declared here.
 *
 * Parity pin: every function here is checked against the real reference build
 * (oracle/_ref/libfplll.so, built from /root/reference by oracle/Makefile) by
 * oracle/ref_driver.cpp → tests/golden/*.json and tests/test_oracle_vs_ref.py.
 */
#ifndef FPLLL_AMD_ORACLE_H
#define FPLLL_AMD_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_MAX_DIM 260

/* Solution sink.  Mirrors extenum_cb_process_sol (fplll/enum/enumerate_ext_api.h:62-63):
 * receives the squared norm and the coefficient vector, returns the NEW enumeration bound. */
typedef double (*oracle_sol_cb)(void *user, double dist, const double *sol);
/* Mirrors extenum_cb_process_subsol (enumerate_ext_api.h:70-71). */
typedef void (*oracle_subsol_cb)(void *user, double dist, const double *subsol, int offset);

/*
 * SVP enumeration (no target, no subtree, primal), restating
 *   EnumerationDyn::prepare_enumeration   fplll/enum/enumerate.cpp:161-216
 *   EnumerationBase::enumerate_loop       fplll/enum/enumerate_base.cpp:152-195 (prologue)
 *   EnumerationBase::enumerate_recursive  fplll/enum/enumerate_base.cpp:24-118
 *   EnumerationDyn::set_bounds/process_solution/process_subsolution  enumerate.cpp:218-249
 *
 * mut     : d×d row-major, mut[i*d+j] = mu(j,i) for j>i (the layout callback_set_config fills with
 *           mutranspose=true, enumerate_ext.cpp:108-121); other entries ignored.
 * rdiag   : r_ii (normalised), pruning : d coefficients or NULL (= all 1.0).
 * cb      : NULL → behave like FastEvaluator(max_sols=1, BEST_N): new bound = dist, best kept in
 *           best_sol/best_dist.
 * nodes   : [d+1] per-level counts, "fplll rule" (count after the bound test, :31-33), including the
 *           nodes[i]-- compensation of enumerate_base.cpp:181-184.
 * returns : number of solutions passed to cb (or accepted by the built-in evaluator).
 */
int64_t oracle_enumerate(int d, const double *mut, const double *rdiag, const double *pruning,
                         double maxdist, int findsubsols, oracle_sol_cb cb, oracle_subsol_cb subcb,
                         void *user, uint64_t *nodes, double *best_sol, double *best_dist);

/* The dual-enumeration instantiation of the same walk (dualenum = true, enumerate_base.cpp:57-61,
 * 103-105; enumerate.cpp:185-189) on inputs already transformed as EnumerationDyn::enumerate does
 * for a dual call (mut negated and index-reversed, rdiag inverted and reversed, enumerate.cpp:107-123);
 * built-in FastEvaluator(1).  The caller reverses best_sol (enumerate.cpp:154-158). */
int64_t oracle_enumerate_dual(int d, const double *mut, const double *rdiag, const double *pruning,
                              double maxdist, uint64_t *nodes, double *best_sol, double *best_dist);
int64_t oracle_enumerate_dual_cb(int d, const double *mut, const double *rdiag, const double *pruning,
                                 double maxdist, oracle_sol_cb cb, void *user, uint64_t *nodes);

/* ------------------------------------------------------------------------------------------
 * GSO / size reduction for ZT=long, FT=double, GSO_ROW_EXPO on (the BKZ fast path,
 * fplll/bkz.cpp:816-829).  State layout = one contiguous lattice; see gso_oracle.c.
 * ------------------------------------------------------------------------------------------ */
typedef struct oracle_gso oracle_gso;

oracle_gso *oracle_gso_create(int d, int n, const int64_t *b /* d×n row-major */, int row_expo);
void oracle_gso_destroy(oracle_gso *g);
/* MatGSOInterface::update_gso_row(i, last_j)   gso_interface.cpp:131-164.  returns 0 on non-finite mu */
int oracle_gso_update_row(oracle_gso *g, int i, int last_j);
/* MatGSOInterface::update_gso()                gso_interface.h:767-775 */
int oracle_gso_update_all(oracle_gso *g);
/* LLLReduction::babai(kappa, sr_end=kappa, sr_start=0)   lll.cpp:166-224.
 * returns 1 ok, 0 = RED_GSO_FAILURE, -1 = RED_BABAI_FAILURE */
int oracle_gso_babai(oracle_gso *g, int kappa, int sr_end, int sr_start, double eta);
/* LLLReduction::size_reduction(kappa_min,kappa_end)      lll.h:107-122 */
int oracle_gso_size_reduction(oracle_gso *g, int kappa_min, int kappa_end, double eta);
/* LLLReduction::lll(kappa_min, kappa_start, kappa_end, 0)  lll.cpp:44-164 + MatGSO::move_row
 * gso.cpp:289-366.  1 success, 0 GSO failure, -1 babai failure, -2 multiplier, -3 LLL failure.
 * Rows below kappa_start must have been updated by the caller (as in the reference).
 * info[4] (nullable): final_kappa, n_swaps, zeros, iterations. */
int oracle_gso_lll(oracle_gso *g, int kappa_min, int kappa_start, int kappa_end, double delta,
                   double eta, int *info);
int oracle_gso_lll_flags(oracle_gso *g, int kappa_min, int kappa_start, int kappa_end, double delta,
                         double eta, int flags, int *info);
/* BKZReduction<Z_NR<long>,FP_NR<double>>::bkz() (bkz.cpp:522-668) with empty strategies (no pruning,
 * no preprocessing: BASELINE config 2), primal, flags BKZ_DEFAULT or BKZ_MAX_LOOPS.  1 RED_SUCCESS,
 * 8 RED_BKZ_LOOPS_LIMIT, <= 0 failure.  info[3] (nullable): tours, nodes lo, nodes hi. */
int oracle_gso_bkz(oracle_gso *g, int block_size, double delta, double eta, int use_max_loops,
                   int max_loops, int *info);
/* Strategies (bkz_param.h:22-66; the content of a strategies JSON, bkz_param.cpp:82-157), flattened:
 * block size b has preprocessing block sizes pre[pre_off[b] .. pre_off[b+1]) and pruning sets
 * prune_off[b] .. prune_off[b+1]; set p has gh_factor prune_gh[p], expectation prune_exp[p] and the
 * coefficients coeff[coeff_off[p] .. coeff_off[p+1]) (none = no pruning).  Every block size
 * 0..max_block_size has at least one pruning set (bkz_param.cpp:150-154). */
typedef struct
{
  int max_block_size;
  const int *pre_off;      /* [max_block_size + 2] */
  const int *pre;
  const int *prune_off;    /* [max_block_size + 2] */
  const double *prune_gh;
  const double *prune_exp;
  const int *coeff_off;    /* [number of pruning sets + 1] */
  const double *coeff;
} oracle_strategies;
/* gmp_urandomm_ui(RandGen::get_gmp_state(), n) of the reference (nr/nr_rand.inl); the tests supply
 * it from the same libgmp (oracle/gmp_rng.c) so that rerandomize_block (bkz.cpp:43-80) draws the
 * same numbers. */
typedef unsigned long (*oracle_rand_fn)(void *user, unsigned long n);
/* BKZReduction::bkz() with a BKZParam(block_size, strategies, delta, flags, max_loops, ...,
 * gh_factor): preprocessing tours, pruning selection, GH bound, rerandomisation (bkz.cpp:43-124,
 * 274-441, 522-668).  flags are fplll's (MAX_LOOPS 0x4, BOUNDED_LLL 0x10, AUTO_ABORT 0x20, GH_BND
 * 0x80).  info[5]: tours, nodes lo, nodes hi, enumeration calls, rerandomisations. */
int oracle_gso_bkz_param(oracle_gso *g, int block_size, double delta, double eta, int flags,
                         int max_loops, double gh_factor, const oracle_strategies *strat,
                         oracle_rand_fn rnd, void *rnd_user, int *info);
/* in-loop pruning hook of oracle_gso_bkz_param (NULL: off): fn(user, bs, gso_r[bs], radius, coefficients[bs] out,
 * expectation out) -> 1 when it pruned the block (else the strategies' set stays); top-level primal blocks
 * of at least min_block rows */
typedef int (*oracle_inloop_fn)(void *user, int bs, const double *gso_r, double radius, double *coefficients,
                                double *expectation);
void oracle_gso_bkz_set_inloop(oracle_inloop_fn fn, void *user, int min_block);
/* one svp_reduction(kappa, block_size, empty strategies, dual), bkz.cpp:274-358 */
int oracle_gso_svp_reduction(oracle_gso *g, int kappa, int block_size, int dual, double delta,
                             double eta, int *clean, uint64_t *nodes);
/* radius and pruning choice of the block [kappa, kappa+bs) in the current state (bkz.cpp:309-325) */
void oracle_gso_bkz_radius(oracle_gso *g, int kappa, int bs, int flags, double delta, double gh_factor,
                           const oracle_strategies *strat, double *max_dist_out, int *prune_out);
/* raw (scaled) state access; true values need the row exponents (gso_interface.h:694-732) */
const double *oracle_gso_mu(const oracle_gso *g);      /* d×d */
const double *oracle_gso_r(const oracle_gso *g);       /* d×d */
const double *oracle_gso_bf(const oracle_gso *g);      /* d×n */
const int64_t *oracle_gso_b(const oracle_gso *g);      /* d×n */
const int64_t *oracle_gso_row_expo(const oracle_gso *g); /* d */
/* get_mu / get_r with exponents applied (gso_interface.h:706-732) */
double oracle_gso_get_mu(const oracle_gso *g, int i, int j);
double oracle_gso_get_r(const oracle_gso *g, int i, int j);

/* ------------------------------------------------------------------------------------------
 * Householder R-factor for MatHouseholder<Z_NR<long>, FP_NR<double>>: refresh_R_bf() + update_R()
 * over all rows (householder.h:532-536, householder.cpp:27-245).  R, V: d×n; sigma, row_expo: d.
 * Only R(i, j<=i) is meaningful to callers (the tail of a row is scratch in the reference too).
 * ------------------------------------------------------------------------------------------ */
int oracle_hh_update_all(int d, int n, const int64_t *b, int row_expo_on, double *R, double *V,
                         double *sigma, int64_t *row_expo);
int oracle_hh_size_reduce(int d, int n, int64_t *b, int row_expo_on, int k, int end, int start, double *R,
                          int64_t *row_expo);
/* HLLLReduction<Z_NR<long>,FP_NR<double>>::hlll() (hlll.cpp:26-169) over MatHouseholder with
 * HOUSEHOLDER_ROW_EXPO: b (d x n, row-major) is reduced in place.  1 RED_SUCCESS, -2 multiplier
 * beyond 63 bits, -4 RED_HLLL_SR_FAILURE, -5 RED_HLLL_NORM_FAILURE.  info[2] (nullable): swaps,
 * loop iterations. */
int oracle_hlll(int d, int n, int64_t *b, int row_expo_on, double delta, double eta, double theta,
                double c, int *info);

#ifdef __cplusplus
}
#endif
#endif
