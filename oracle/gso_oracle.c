/*
 * gso_oracle.c — TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Plain-C restatement of fplll's floating-point Gram-Schmidt + size reduction for
 * MatGSO<Z_NR<long>, FP_NR<double>> with GSO_ROW_EXPO (the BKZ fast path, fplll/bkz.cpp:816-829),
 * all rows discovered (the state after update_gso()), no transform matrices, no integer Gram.
 * Pinned bit-exact against the real reference by tests/test_gso_oracle_vs_ref.py
 * (fixtures from oracle/ref_driver.cpp `gsofix`).
 *
 * Every sum keeps the reference's order and its separate multiply / add roundings
 * (nr/nr_FP_d.inl:178; compile with -ffp-contract=off).
 */
#define _DEFAULT_SOURCE /* M_E, M_PI, lgamma under -std=c11 */
#include "oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

struct oracle_gso
{
  int d, n;           /* rows, columns */
  int row_expo_on;
  int64_t *b;         /* d×n */
  double *bf;         /* d×n   (gso_interface.h:548) */
  double *gf;         /* d×d   lazy float Gram, NaN = invalid (gso_interface.h:575) */
  double *mu, *r;     /* d×d   (gso_interface.h:589,604) */
  int64_t *row_expo;  /* d     (gso_interface.h:167) */
  int *valid_cols;    /* gso_valid_cols (gso_interface.h:608) */
  int n_known_cols;
  double *babai_mu;
  int64_t *babai_expo;
  int64_t *tmp_col_expo;
};

#define B(g, i, j) ((g)->b[(size_t)(i) * (g)->n + (j)])
#define BF(g, i, j) ((g)->bf[(size_t)(i) * (g)->n + (j)])
#define GF(g, i, j) ((g)->gf[(size_t)(i) * (g)->d + (j)])
#define MU(g, i, j) ((g)->mu[(size_t)(i) * (g)->d + (j)])
#define R(g, i, j) ((g)->r[(size_t)(i) * (g)->d + (j)])

static long fexponent(double x) { return (long)ilogb(x) + 1; } /* nr_FP_d.inl:44 */
static long fexponent_l(double x) { return (x == 0.0) ? (long)INT_MIN + 1 : (long)ilogb(x) + 1; }

/* MatGSO::update_bf, gso.cpp:24-48 */
static void update_bf(oracle_gso *g, int i)
{
  int n = g->n_known_cols;
  if (g->row_expo_on)
  {
    long max_expo = LONG_MIN;
    for (int j = 0; j < n; j++)
    {
      int e;
      BF(g, i, j)        = frexp((double)B(g, i, j), &e); /* nr_Z_misc.inl:17-22 */
      g->tmp_col_expo[j] = e;
      if (e > max_expo)
        max_expo = e;
    }
    for (int j = 0; j < n; j++)
      BF(g, i, j) = ldexp(BF(g, i, j), (int)(g->tmp_col_expo[j] - max_expo));
    g->row_expo[i] = max_expo;
  }
  else
  {
    for (int j = 0; j < n; j++)
      BF(g, i, j) = (double)B(g, i, j);
    g->row_expo[i] = 0;
  }
}

oracle_gso *oracle_gso_create(int d, int n, const int64_t *b, int row_expo)
{
  oracle_gso *g   = (oracle_gso *)calloc(1, sizeof *g);
  g->d            = d;
  g->n            = n;
  g->row_expo_on  = row_expo;
  g->b            = (int64_t *)malloc(sizeof(int64_t) * d * n);
  g->bf           = (double *)calloc((size_t)d * n, sizeof(double));
  g->gf           = (double *)malloc(sizeof(double) * d * d);
  g->mu           = (double *)calloc((size_t)d * d, sizeof(double));
  g->r            = (double *)calloc((size_t)d * d, sizeof(double));
  g->row_expo     = (int64_t *)calloc(d, sizeof(int64_t));
  g->valid_cols   = (int *)calloc(d, sizeof(int));
  g->babai_mu     = (double *)calloc(d, sizeof(double));
  g->babai_expo   = (int64_t *)calloc(d, sizeof(int64_t));
  g->tmp_col_expo = (int64_t *)calloc(n, sizeof(int64_t));
  memcpy(g->b, b, sizeof(int64_t) * d * n);
  /* all rows discovered: n_known_cols = max over rows of size_nz (gso.cpp:56-82) */
  int nk = 1;
  for (int i = 0; i < d; ++i)
    for (int j = n - 1; j >= 0; --j)
      if (B(g, i, j) != 0)
      {
        if (j + 1 > nk)
          nk = j + 1;
        break;
      }
  g->n_known_cols = nk;
  for (int i = 0; i < d; ++i)
  {
    update_bf(g, i);
    for (int j = 0; j < d; ++j)
      GF(g, i, j) = NAN;
  }
  return g;
}

void oracle_gso_destroy(oracle_gso *g)
{
  if (!g)
    return;
  free(g->b);
  free(g->bf);
  free(g->gf);
  free(g->mu);
  free(g->r);
  free(g->row_expo);
  free(g->valid_cols);
  free(g->babai_mu);
  free(g->babai_expo);
  free(g->tmp_col_expo);
  free(g);
}

/* MatGSO::get_gram, gso.h:314-331 with NumVect::dot_product, nr/numvect.h:386-396 */
static double get_gram(oracle_gso *g, int i, int j)
{
  if (isnan(GF(g, i, j)))
  {
    double res = BF(g, i, 0) * BF(g, j, 0);
    for (int c = 1; c < g->n_known_cols; c++)
      res = res + BF(g, i, c) * BF(g, j, c);
    GF(g, i, j) = res;
  }
  return GF(g, i, j);
}

/* MatGSOInterface::update_gso_row, gso_interface.cpp:131-164 */
int oracle_gso_update_row(oracle_gso *g, int i, int last_j)
{
  int j = g->valid_cols[i] > 0 ? g->valid_cols[i] : 0;
  for (; j <= last_j; j++)
  {
    double ftmp1 = get_gram(g, i, j);
    for (int k = 0; k < j; k++)
    {
      double ftmp2 = MU(g, j, k) * R(g, i, k);
      ftmp1        = ftmp1 - ftmp2;
    }
    R(g, i, j) = ftmp1;
    if (i > j)
    {
      MU(g, i, j) = ftmp1 / R(g, j, j);
      if (!isfinite(MU(g, i, j)))
        return 0;
    }
  }
  g->valid_cols[i] = j;
  return 1;
}

int oracle_gso_update_all(oracle_gso *g)
{
  for (int i = 0; i < g->d; i++)
    if (!oracle_gso_update_row(g, i, i))
      return 0;
  return 1;
}

/* MatGSOInterface::row_op_end(first=i,last=i+1), gso_interface.cpp:32-53 */
static void row_op_end(oracle_gso *g, int i)
{
  update_bf(g, i);
  for (int j = 0; j <= i; j++) /* invalidate_gram_row, gso.cpp:50-54 */
    GF(g, i, j) = NAN;
  for (int j = i + 1; j < g->d; j++)
    GF(g, j, i) = NAN;
  g->valid_cols[i] = 0;
  for (int j = i + 1; j < g->d; j++)
    if (g->valid_cols[j] > i)
      g->valid_cols[j] = i;
}

/* MatGSO::row_addmul_we for ZT=long without OP_FORCE_LONG, gso.cpp:236-262;
 * returns 0 when the multiplier needs the 2^expo path (not restated: never taken when
 * |X·2^expo_add| < 2^63) */
static int row_addmul_we(oracle_gso *g, int i, int j, double x, long expo_add)
{
  long expo;
  if (x == 0)
    expo = 0;
  else
  {
    expo = fexponent(x) + expo_add - 63; /* numeric_limits<long>::digits */
    if (expo < 0)
      expo = 0;
  }
  long lx = (long)ldexp(x, (int)(expo_add - expo)); /* nr_FP_d.inl:46-53 */
  if (expo != 0)
    return 0;
  int n = g->n_known_cols;
  if (lx == 1)
    for (int c = n - 1; c >= 0; c--) /* row_add → NumVect::add, numvect.h:268-272 */
      B(g, i, c) = (int64_t)((uint64_t)B(g, i, c) + (uint64_t)B(g, j, c));
  else if (lx == -1)
    for (int c = n - 1; c >= 0; c--)
      B(g, i, c) = (int64_t)((uint64_t)B(g, i, c) - (uint64_t)B(g, j, c));
  else if (lx != 0)
    for (int c = n - 1; c >= 0; c--) /* addmul_si, numvect.h:324-329 */
      B(g, i, c) = (int64_t)((uint64_t)B(g, i, c) + (uint64_t)B(g, j, c) * (uint64_t)lx);
  return 1;
}

/* MatGSOInterface::get_max_mu_exp, gso_interface.cpp:88-98 */
static long get_max_mu_exp(oracle_gso *g, int i, int ncols)
{
  long max_expo = LONG_MIN;
  for (int j = 0; j < ncols; j++)
  {
    long expo  = g->row_expo[i] - g->row_expo[j];
    long expo2 = fexponent(MU(g, i, j));
    if (expo + expo2 > max_expo)
      max_expo = expo + expo2;
  }
  return max_expo;
}

/* LLLReduction::babai, lll.cpp:166-224 */
int oracle_gso_babai(oracle_gso *g, int kappa, int sr_end, int sr_start, double eta)
{
  long max_expo = LONG_MAX;
  for (int iter = 0;; iter++)
  {
    if (!oracle_gso_update_row(g, kappa, sr_end - 1))
      return 0; /* RED_GSO_FAILURE */
    int loop_needed = 0;
    for (int j = sr_end - 1; j >= sr_start && !loop_needed; j--)
    {
      double f = ldexp(MU(g, kappa, j), (int)(g->row_expo[kappa] - g->row_expo[j])); /* get_mu */
      f        = fabs(f);
      loop_needed |= (f > eta);
    }
    if (!loop_needed)
      break;
    if (iter >= 2)
    {
      long new_max_expo = get_max_mu_exp(g, kappa, sr_end);
      if (new_max_expo > max_expo - 5) /* SIZE_RED_FAILURE_THRESH, defs.h:146 */
        return -1;                     /* RED_BABAI_FAILURE */
      max_expo = new_max_expo;
    }
    for (int j = sr_start; j < sr_end; j++)
    {
      g->babai_mu[j]   = MU(g, kappa, j);
      g->babai_expo[j] = g->row_expo[kappa] - g->row_expo[j];
    }
    for (int j = sr_end - 1; j >= sr_start; j--)
    {
      /* rnd_we, nr_FP_d.inl:226-233 */
      double bm = g->babai_mu[j], X;
      long e    = g->babai_expo[j];
      if (fexponent(bm) + e >= 53)
        X = bm;
      else
        X = ldexp(rint(ldexp(bm, (int)e)), (int)-e);
      if (X == 0.0)
        continue;
      for (int k = sr_start; k < j; k++)
      {
        double t       = X * MU(g, j, k);
        g->babai_mu[k] = g->babai_mu[k] - t;
      }
      if (!row_addmul_we(g, kappa, j, -X, e))
        return -2; /* multiplier beyond 63 bits: path not restated */
    }
    row_op_end(g, kappa);
  }
  return 1;
}

/* LLLReduction::size_reduction, lll.h:107-122 */
int oracle_gso_size_reduction(oracle_gso *g, int kappa_min, int kappa_end, double eta)
{
  for (int k = kappa_min; k < kappa_end; k++)
  {
    if (k > 0)
    {
      int rc = oracle_gso_babai(g, k, k, 0, eta);
      if (rc != 1)
        return rc;
    }
    if (!oracle_gso_update_row(g, k, k))
      return 0;
  }
  return 1;
}

/* ------------------------------------------------------------------------------------------
 * LLL driver (pinned by tests/test_lll_oracle_vs_ref.py against `lllfix` fixtures)
 * ------------------------------------------------------------------------------------------ */

/* MatGSO::move_row, gso.cpp:289-366 (no transforms, float Gram).  Rows of b, bf, mu, r, row_expo and
 * gso_valid_cols rotate with the row; the lower-triangular Gram cache is permuted symmetrically
 * (what rotate_gram_left/right, nr/matrix.cpp:65-92, do by element swaps). */
static void move_row(oracle_gso *g, int old_r, int new_r)
{
  const int d = g->d, n = g->n;
  if (old_r == new_r)
    return;
  const int lo = old_r < new_r ? old_r : new_r, hi = old_r < new_r ? new_r : old_r;
  for (int i = lo; i < d; i++) /* invalidate_gso_row(i, lo) */
    if (g->valid_cols[i] > lo)
      g->valid_cols[i] = lo;
  int *src = (int *)malloc(sizeof(int) * d); /* new position p holds the old row src[p] */
  for (int p = 0; p < d; p++)
    src[p] = p;
  if (new_r < old_r)
  {
    src[new_r] = old_r;
    for (int p = new_r + 1; p <= old_r; p++)
      src[p] = p - 1;
  }
  else
  {
    src[new_r] = old_r;
    for (int p = old_r; p < new_r; p++)
      src[p] = p + 1;
  }
#define PERMUTE_ROWS(type, arr, width)                                           \
  do                                                                             \
  {                                                                              \
    type *tmp_ = (type *)malloc(sizeof(type) * (size_t)(hi - lo + 1) * (width)); \
    for (int p = lo; p <= hi; p++)                                               \
      memcpy(tmp_ + (size_t)(p - lo) * (width), (arr) + (size_t)src[p] * (width), \
             sizeof(type) * (width));                                            \
    memcpy((arr) + (size_t)lo * (width), tmp_, sizeof(type) * (size_t)(hi - lo + 1) * (width)); \
    free(tmp_);                                                                  \
  } while (0)
  PERMUTE_ROWS(int64_t, g->b, n);
  PERMUTE_ROWS(double, g->bf, n);
  PERMUTE_ROWS(double, g->mu, d);
  PERMUTE_ROWS(double, g->r, d);
  PERMUTE_ROWS(int64_t, g->row_expo, 1);
  PERMUTE_ROWS(int, g->valid_cols, 1);
#undef PERMUTE_ROWS
  double *ng = (double *)malloc(sizeof(double) * d * d);
  for (int i = 0; i < d; i++)
    for (int j = 0; j <= i; j++)
    {
      int a = src[i], b = src[j];
      ng[(size_t)i * d + j] = a >= b ? GF(g, a, b) : GF(g, b, a);
    }
  for (int i = 0; i < d; i++)
    for (int j = 0; j <= i; j++)
      GF(g, i, j) = ng[(size_t)i * d + j];
  free(ng);
  free(src);
}

/* Z_NR<long>::exponent, nr/nr_Z_l.inl:30-48 */
static long zexponent(int64_t v)
{
  int e;
  double f = frexp((double)v, &e);
  if ((double)v > 0x1p53 && fabs(f) == 0.5)
  {
    uint64_t y = (uint64_t)(v < 0 ? -v : v);
    long k     = 0;
    for (; y; k++, y >>= 1)
      ;
    return k;
  }
  return e;
}

/* LLLReduction::lll(kappa_min, kappa_start, kappa_end, 0), lll.cpp:44-164 (no early reduction, no
 * Siegel).  Returns 1 RED_SUCCESS, 0 RED_GSO_FAILURE, -1 RED_BABAI_FAILURE, -2 multiplier beyond 63
 * bits, -3 RED_LLL_FAILURE.  info[0..3] = final_kappa, n_swaps, zeros, iterations. */
static int oracle_gso_lll_impl(oracle_gso *g, int kappa_min, int kappa_start, int kappa_end, double delta_in,
                               double eta, int siegel, int early_red, int *info);
int oracle_gso_lll(oracle_gso *g, int kappa_min, int kappa_start, int kappa_end, double delta,
                   double eta, int *info)
{
  return oracle_gso_lll_impl(g, kappa_min, kappa_start, kappa_end, delta, eta, 0, 0, info);
}
/* ... with LLL_SIEGEL (flags & 4): swap_threshold = delta - eta^2 (lll.cpp:40) and the tests compare with
 * lovasz_tests[kappa] instead of [kappa - 1] (lll.cpp:122,134); with LLL_EARLY_RED (flags & 2, lll.cpp:35,
 * 84-99, lll.h:125-140): whenever kappa reaches a new maximum that is a power of two, every row from kappa on
 * is size-reduced against the rows below kappa.  (The reference locks n_known_cols meanwhile and forgets the
 * rows it discovered for this: their Gram and GSO entries are recomputed later from the same vectors — the
 * same numbers, since the columns beyond n_known_cols of a discovered row are zero — so the cache here
 * simply keeps them.)  One call = one LLLReduction object: last_early_red starts at 0. */
int oracle_gso_lll_flags(oracle_gso *g, int kappa_min, int kappa_start, int kappa_end, double delta,
                         double eta, int flags, int *info)
{
  return oracle_gso_lll_impl(g, kappa_min, kappa_start, kappa_end, delta, eta, (flags & 4) != 0, (flags & 2) != 0,
                             info);
}
static int oracle_gso_lll_impl(oracle_gso *g, int kappa_min, int kappa_start, int kappa_end, double delta_in,
                               double eta, int siegel, int early_red, int *info)
{
  const int n = g->n;
  int kappa_max = 0, last_early_red = 0;
  const double delta = siegel ? delta_in - eta * eta : delta_in; /* swap_threshold */
  if (kappa_end == -1)
    kappa_end = g->d;
  int kappa = kappa_start + 1;
  int d     = kappa_end - kappa_min;
  int zeros = 0, n_swaps = 0, final_kappa = 0;
  double *lovasz = (double *)calloc(g->d + 1, sizeof(double));
  int status     = 1;
  long long iter = 0;
  for (; zeros < d; zeros++)
  { /* b_row_is_zero(0) */
    int z = 1;
    for (int c = 0; c < n; c++)
      if (B(g, 0, c) != 0)
        z = 0;
    if (!z)
      break;
    move_row(g, kappa_min, kappa_end - 1 - zeros);
  }
  if (zeros < d)
  {
    int rc = 1;
    if (kappa_start > 0)
      rc = oracle_gso_babai(g, kappa_start, kappa_start, 0, eta);
    if (rc == 1 && !oracle_gso_update_row(g, kappa_start, kappa_start))
      rc = 0;
    if (rc != 1)
    {
      status      = rc;
      final_kappa = kappa_start;
      goto done;
    }
  }
  {
    long max_exp = 0;
    for (int i = 0; i < g->d; i++)
      for (int c = 0; c < n; c++)
      {
        long e = zexponent(B(g, i, c));
        if (e > max_exp)
          max_exp = e;
      }
    long long max_iter = (long long)(d - 2 * d * (d + 1) * ((max_exp + 3) / log(delta_in)));
    for (iter = 0; iter < max_iter && kappa < kappa_end - zeros; iter++)
    {
      int rc = 1;
      if (kappa > kappa_max)
      { /* lll.cpp:84-99 */
        kappa_max = kappa;
        if (early_red && (kappa & (kappa - 1)) == 0 && kappa > last_early_red)
        { /* early_reduction(kappa, 0), lll.h:125-140 */
          for (int i = kappa; i < g->d && rc == 1; i++)
            rc = oracle_gso_babai(g, i, kappa, 0, eta);
          if (rc == 1)
            last_early_red = kappa;
        }
      }
      if (rc == 1)
        rc = oracle_gso_babai(g, kappa, kappa, 0, eta);
      if (rc != 1)
      {
        status      = rc;
        final_kappa = kappa;
        goto done;
      }
      lovasz[0] = get_gram(g, kappa, kappa);
      for (int i = 1; i <= kappa; i++)
      {
        double t  = MU(g, kappa, i - 1) * R(g, kappa, i - 1);
        lovasz[i] = lovasz[i - 1] - t;
      }
      double f = R(g, kappa - 1, kappa - 1) * delta;
      if (g->row_expo_on)
        f = ldexp(f, (int)(2 * (g->row_expo[kappa - 1] - g->row_expo[kappa])));
      if (f > lovasz[siegel ? kappa : kappa - 1])
      {
        n_swaps++;
        int old_k = kappa;
        for (kappa--; kappa > kappa_min; kappa--)
        {
          f = R(g, kappa - 1, kappa - 1) * delta;
          if (g->row_expo_on)
            f = ldexp(f, (int)(2 * (g->row_expo[kappa - 1] - g->row_expo[old_k])));
          if (f < lovasz[siegel ? kappa : kappa - 1])
            break;
        }
        if (lovasz[kappa] > 0)
          move_row(g, old_k, kappa);
        else
        {
          zeros++;
          move_row(g, old_k, kappa_end - zeros);
          kappa = old_k;
          continue;
        }
      }
      /* set_r(kappa, kappa, lovasz[kappa]), gso_interface.h:742-749 */
      R(g, kappa, kappa) = lovasz[kappa];
      if (g->valid_cols[kappa] == kappa)
        g->valid_cols[kappa]++;
      kappa++;
    }
    status = (kappa < kappa_end - zeros) ? -3 : 1;
  }
done:
  free(lovasz);
  if (info)
  {
    info[0] = final_kappa;
    info[1] = n_swaps;
    info[2] = zeros;
    info[3] = (int)(iter & 0x7fffffff);
  }
  return status;
}

/* ------------------------------------------------------------------------------------------
 * BKZ driver (pinned by tests/test_bkz_oracle_vs_ref.py against `bkzfix` fixtures):
 * BKZReduction<Z_NR<long>, FP_NR<double>>::bkz() with BKZ_DEFAULT or BKZ_MAX_LOOPS flags and no
 * strategies (BKZParam fills Strategy::EmptyStrategy for every block size, bkz_param.h:124-132: no
 * pruning, no preprocessing, expectation 1 — BASELINE config 2), primal only.  Restates
 *   bkz()              bkz.cpp:522-668     tour / trunc_tour / hkz   bkz.cpp:360-441
 *   svp_reduction      bkz.cpp:274-358     svp_preprocessing (its lll call)  bkz.cpp:100-124
 *   svp_postprocessing / _generic          bkz.cpp:126-272
 *   EnumerationDyn::enumerate (normalisation)  enum/enumerate.cpp:58-159
 *   MatGSOInterface::row_op_end(first,last)    gso_interface.cpp:32-53
 * ------------------------------------------------------------------------------------------ */
static void row_op_end_range(oracle_gso *g, int first, int last)
{
  for (int i = first; i < last; i++)
  {
    update_bf(g, i);
    for (int j = 0; j <= i; j++)
      GF(g, i, j) = NAN;
    for (int j = i + 1; j < g->d; j++)
      GF(g, j, i) = NAN;
    g->valid_cols[i] = 0;
  }
  for (int i = last; i < g->d; i++)
    if (g->valid_cols[i] > first)
      g->valid_cols[i] = first;
}

static void swap_b_rows(oracle_gso *g, int i, int j)
{
  for (int c = 0; c < g->n; c++)
  {
    int64_t t  = B(g, i, c);
    B(g, i, c) = B(g, j, c);
    B(g, j, c) = t;
  }
}

/* 1 ok, else the failing status of babai / update_gso_row */
static int lll_size_reduction(oracle_gso *g, int kappa_min, int kappa_end, int sr_start, double eta)
{
  for (int k = kappa_min; k < kappa_end; k++)
  {
    if (k > 0)
    {
      int rc = oracle_gso_babai(g, k, k, sr_start, eta);
      if (rc != 1)
        return rc;
    }
    if (!oracle_gso_update_row(g, k, k))
      return 0;
  }
  return 1;
}

static int svp_postprocessing(oracle_gso *g, int kappa, int bs, const double *sol, int dual)
{
  int nz_vectors = 0, i_vector = -1;
  for (int i = bs - 1; i >= 0; i--)
    if (sol[i] != 0.0)
    {
      nz_vectors++;
      if (i_vector == -1 && fabs(sol[i]) == 1)
        i_vector = i;
    }
  const int pos = dual ? kappa + bs - 1 : kappa;
  if (nz_vectors == 1)
  {
    move_row(g, kappa + i_vector, pos);
  }
  else if (i_vector != -1)
  {
    int sol_i = (int)sol[i_vector];
    if (dual)
      sol_i *= -1;
    for (int i = 0; i < bs; ++i)
      if (sol[i] != 0.0 && i != i_vector)
      {
        int ok = dual ? row_addmul_we(g, kappa + i, kappa + i_vector, sol_i * sol[i], 0)
                      : row_addmul_we(g, kappa + i_vector, kappa + i, sol_i * sol[i], 0);
        if (!ok)
          return -2;
      }
    if (dual)
      row_op_end_range(g, kappa, kappa + bs);
    else
      row_op_end_range(g, kappa + i_vector, kappa + i_vector + 1);
    move_row(g, kappa + i_vector, pos);
  }
  else
  { /* svp_postprocessing_generic, bkz.cpp:205-272 */
    double *x = (double *)malloc(sizeof(double) * bs);
    for (int i = 0; i < bs; i++)
    {
      x[i] = sol[i];
      if (x[i] < 0)
      {
        x[i] = -x[i];
        for (int c = 0; c < g->n; c++) /* negate_row_of_b */
          B(g, i + kappa, c) = -B(g, i + kappa, c);
      }
    }
    int off = 1;
    while (off < bs)
    {
      int k = bs - 1;
      while (k - off >= 0)
      {
        if (!(x[k] == 0.0 && x[k - off] == 0.0))
        {
          if (x[k] < x[k - off])
          {
            double t = x[k]; x[k] = x[k - off]; x[k - off] = t;
            swap_b_rows(g, kappa + k - off, kappa + k);
          }
          while (x[k - off] != 0.0)
          {
            while (x[k - off] <= x[k])
            {
              x[k] = x[k] - x[k - off];
              int ok = dual ? row_addmul_we(g, kappa + k, kappa + k - off, -1.0, 0) /* row_sub */
                            : row_addmul_we(g, kappa + k - off, kappa + k, 1.0, 0); /* row_add */
              if (!ok)
              {
                free(x);
                return -2;
              }
            }
            double t = x[k]; x[k] = x[k - off]; x[k - off] = t;
            swap_b_rows(g, kappa + k - off, kappa + k);
          }
        }
        k -= 2 * off;
      }
      off *= 2;
    }
    free(x);
    row_op_end_range(g, kappa, kappa + bs);
    if (!dual)
      move_row(g, kappa + bs - 1, kappa);
  }
  return 1;
}

/* ---- BKZParam / Strategy (bkz_param.h:22-66,68-170) as the oracle sees them ---------------- */
typedef struct
{
  int block_size;
  int flags;                      /* BKZ_GH_BND 0x80, BKZ_BOUNDED_LLL 0x10 (defs.h:262-275) */
  double gh_factor;               /* BKZ_DEF_GH_FACTOR 1.1 */
  double min_success_probability; /* BKZ_DEF_MIN_SUCCESS_PROBABILITY 0.5 */
  int rerandomization_density;    /* BKZ_DEF_RERANDOMIZATION_DENSITY 3 */
} bkz_par;

typedef struct
{
  double delta;                      /* BKZReduction::delta = param.delta (bkz.cpp:38) */
  double lll_delta, eta;             /* of the LLLReduction object */
  const oracle_strategies *strat;    /* NULL: empty strategies */
  oracle_rand_fn rnd;                /* gmp_urandomm_ui(RandGen::get_gmp_state(), n) */
  void *rnd_user;
  uint64_t nodes;
  uint64_t enum_calls, rerandomizations;
  double sld_potential; /* BKZReduction::sld_potential */
  const void *top_par;  /* the BKZParam of the top-level tour (in-loop pruning applies to its blocks only) */
} bkz_ctx;

/* In-loop pruning (the product's FPHIP_BKZ_PRUNE_IN_LOOP; oracle/ref_driver.cpp: InloopBKZ drives the real
 * reference the same way): where svp_reduction picks a pruning set of the strategies (bkz.cpp:325) a
 * top-level primal block of at least min_block rows asks this hook for coefficients computed on its own
 * r-profile and radius.  Test infrastructure: the hook is installed by tests/conftest.py (which hands the
 * request to the PRODUCT's pruner: the CPU suite then pins "this schedule + that pruner = the driven
 * reference" without a GPU). */
static oracle_inloop_fn g_inloop_fn = NULL;
static void *g_inloop_user          = NULL;
static int g_inloop_min_block       = 0;
void oracle_gso_bkz_set_inloop(oracle_inloop_fn fn, void *user, int min_block)
{
  g_inloop_fn        = fn;
  g_inloop_user      = user;
  g_inloop_min_block = min_block;
}

/* MatGSOInterface::get_root_det / get_log_det, gso_interface.cpp:220-242 */
static double get_root_det(oracle_gso *g, int start_row, int end_row)
{
  if (start_row < 0)
    start_row = 0;
  if (end_row > g->d)
    end_row = g->d;
  double h       = (double)(end_row - start_row);
  double log_det = 0.0;
  for (int i = start_row; i < end_row; ++i)
  {
    double r = ldexp(R(g, i, i), (int)(2 * g->row_expo[i])); /* get_r, gso_interface.h:720-732 */
    log_det += log(r);
  }
  double root_det = log_det / h;
  return exp(root_det);
}

/* adjust_radius_to_gh_bound, gso_interface.cpp:260-276 */
static void adjust_radius_to_gh_bound(double *max_dist, long max_dist_expo, int block_size,
                                      double root_det, double gh_factor)
{
  double t = (double)block_size / 2.0 + 1;
  t        = lgamma(t);
  t        = pow(M_E, t * 2.0 / (double)block_size);
  t        = t / M_PI;
  double f = t;
  f        = f * root_det;
  f        = ldexp(f, (int)-max_dist_expo);
  f        = f * gh_factor;
  if (f < *max_dist)
    *max_dist = f;
}

/* BKZReduction::get_pruning (bkz.cpp:82-98) + Strategy::get_pruning (bkz_param.cpp:64-80):
 * index into strat->prune_* of the pruning set whose gh_factor is closest, or -1 (no strategies) */
static int get_pruning(oracle_gso *g, const bkz_ctx *cx, int kappa, int bs)
{
  const oracle_strategies *S = cx->strat;
  if (!S)
    return -1;
  long max_dist_expo = (long)(2 * g->row_expo[kappa]);
  double max_dist    = R(g, kappa, kappa);
  double gh_max_dist = max_dist;
  double root_det    = get_root_det(g, kappa, kappa + bs);
  adjust_radius_to_gh_bound(&gh_max_dist, max_dist_expo, bs, root_det, 1.0);
  double radius    = max_dist * pow(2, max_dist_expo);
  double gh        = gh_max_dist * pow(2, max_dist_expo);
  double gh_factor = radius / gh;
  double closest   = pow(2, 80);
  int best         = S->prune_off[bs];
  for (int p = S->prune_off[bs]; p < S->prune_off[bs + 1]; ++p)
    if (fabs(S->prune_gh[p] - gh_factor) < closest)
    {
      closest = fabs(S->prune_gh[p] - gh_factor);
      best    = p;
    }
  return best;
}

/* BKZReduction::rerandomize_block, bkz.cpp:43-80 */
static int rerandomize_block(oracle_gso *g, bkz_ctx *cx, int min_row, int max_row, int density)
{
  if (max_row - min_row < 2)
    return 1;
  if (max_row - min_row == 2)
    return 1; /* the reference never returns from this case (gmp_urandomm_ui(state, 1) == 0 for ever,
                 bkz.cpp:53-58); it does not occur with sane strategies (3-blocks are not pruned) */
  cx->rerandomizations++;
  size_t niter = 4 * (size_t)(max_row - min_row);
  for (size_t i = 0; i < niter; ++i)
  {
    size_t a = cx->rnd(cx->rnd_user, (unsigned long)(max_row - min_row - 1)) + min_row;
    size_t b = a;
    while (b == a)
      b = cx->rnd(cx->rnd_user, (unsigned long)(max_row - min_row - 1)) + min_row;
    move_row(g, (int)b, (int)a);
  }
  for (long a = min_row; a < max_row - 2; ++a)
    for (long i = 0; i < density; i++)
    {
      size_t b = cx->rnd(cx->rnd_user, (unsigned long)(max_row - (a + 1) - 1)) + a + 1;
      if (cx->rnd(cx->rnd_user, 2))
        row_addmul_we(g, (int)a, (int)b, 1.0, 0); /* row_add */
      else
        row_addmul_we(g, (int)a, (int)b, -1.0, 0); /* row_sub */
    }
  row_op_end_range(g, min_row, max_row);
  return 1;
}

static int bkz_tour(oracle_gso *g, bkz_ctx *cx, const bkz_par *par, int min_row, int max_row,
                    int *clean);

/* BKZReduction::svp_preprocessing, bkz.cpp:100-124 */
static int svp_preprocessing(oracle_gso *g, bkz_ctx *cx, const bkz_par *par, int kappa, int bs)
{
  int lll_start = (par->flags & 0x10) ? kappa : 0;
  int rc        = oracle_gso_lll(g, lll_start, lll_start, kappa + bs, cx->lll_delta, cx->eta, NULL);
  if (rc != 1)
    return rc;
  const oracle_strategies *S = cx->strat;
  if (!S)
    return 1;
  for (int p = S->pre_off[bs]; p < S->pre_off[bs + 1]; ++p)
  { /* BKZParam(*it, strategies, LLL_DEF_DELTA, BKZ_GH_BND): every other field at its default */
    bkz_par prepar = {S->pre[p], 0x80, 1.1, 0.5, 3};
    int dummy      = 1;
    rc             = bkz_tour(g, cx, &prepar, kappa, kappa + bs, &dummy);
    if (rc != 1)
      return rc;
  }
  return 1;
}

/* BKZReduction::svp_reduction (primal), bkz.cpp:274-358.
 * returns 1 ok (clean flag in *clean), else a failure status */
static int svp_reduction(oracle_gso *g, bkz_ctx *cx, const bkz_par *par, int kappa, int bs, int *clean,
                         int dual)
{
  const double eta = cx->eta;
  const int first  = dual ? kappa + bs - 1 : kappa;
  int rc           = lll_size_reduction(g, 0, first + 1, 0, eta);
  if (rc != 1)
    return rc;
  double old_first    = R(g, first, first);
  long old_first_expo = (long)(2 * g->row_expo[first]);
  int rerandomize     = 0;
  double remaining_probability = 1.0;
  double *rdiag   = (double *)calloc(bs, sizeof(double));
  double *mut     = (double *)calloc((size_t)bs * bs, sizeof(double));
  double *sol     = (double *)calloc(bs, sizeof(double));
  uint64_t *nodes = (uint64_t *)calloc(bs + 1, sizeof(uint64_t));
  rc              = 1;
  while (remaining_probability > 1. - par->min_success_probability)
  {
    if (rerandomize)
      rerandomize_block(g, cx, kappa + 1, kappa + bs, par->rerandomization_density);
    rc = svp_preprocessing(g, cx, par, kappa, bs);
    if (rc != 1)
      break;
    if (getenv("ORACLE_BKZ_DEBUG"))
      for (int i = kappa; i < kappa + bs; ++i)
        if (g->valid_cols[i] < i + 1) /* the enumeration would read a stale GSO row */
          fprintf(stderr, "STALE row %d (valid %d) in block kappa %d bs %d\n", i, g->valid_cols[i], kappa, bs);
    /* radius, bkz.cpp:309-323 */
    double max_dist    = R(g, first, first);
    long max_dist_expo = (long)(2 * g->row_expo[first]);
    if (dual)
    { /* max_dist.pow_si(max_dist, -1): nr_FP_d.inl:189-192 */
      max_dist      = pow(max_dist, (double)-1);
      max_dist_expo = -max_dist_expo;
    }
    max_dist = max_dist * cx->delta;
    if ((par->flags & 0x80) && bs > 30)
    {
      double root_det = get_root_det(g, kappa, kappa + bs);
      adjust_radius_to_gh_bound(&max_dist, max_dist_expo, bs, root_det, par->gh_factor);
    }
    const int pr           = get_pruning(g, cx, kappa, bs);
    const double *pruning  = NULL;
    double expectation     = 1.0; /* PruningParams(): no pruning, expectation 1 (pruner.h:47) */
    double inloop_co[256], inloop_r[256];
    if (pr >= 0)
    {
      expectation = cx->strat->prune_exp[pr];
      if (cx->strat->coeff_off[pr + 1] > cx->strat->coeff_off[pr])
        pruning = cx->strat->coeff + cx->strat->coeff_off[pr];
    }
    if (g_inloop_fn && (const void *)par == cx->top_par && !dual && bs >= g_inloop_min_block && bs >= 4 && bs <= 256)
    { /* prune THIS block: r_ii with the row exponents applied (get_r), radius as the driver passes it */
      for (int i = 0; i < bs; ++i)
        inloop_r[i] = ldexp(R(g, kappa + i, kappa + i), (int)(2 * g->row_expo[kappa + i]));
      const double radius = max_dist * pow(2, (double)max_dist_expo);
      double ex           = 1.0;
      if (isfinite(radius) && radius > 0 && g_inloop_fn(g_inloop_user, bs, inloop_r, radius, inloop_co, &ex) == 1)
      {
        pruning     = inloop_co;
        expectation = ex;
      }
    }
    /* EnumerationDyn::enumerate, enumerate.cpp:88-141 */
    long normexp = -1;
    for (int i = 0; i < bs; ++i)
    {
      long rexpo = (long)(2 * g->row_expo[i + kappa]);
      long e     = rexpo + fexponent_l(R(g, i + kappa, i + kappa));
      if (e > normexp)
        normexp = e;
    }
    if (dual)
      normexp = -normexp; /* the inverse of r is normalised, enumerate.cpp:96-101 */
    double maxdist = ldexp(max_dist, (int)(max_dist_expo - normexp));
    memset(mut, 0, sizeof(double) * (size_t)bs * bs);
    for (int i = 0; i < bs; ++i)
    {
      long rexpo = (long)(2 * g->row_expo[i + kappa]);
      if (dual)
        rdiag[bs - i - 1] = 1.0 / ldexp(R(g, i + kappa, i + kappa), (int)(rexpo + normexp));
      else
        rdiag[i] = ldexp(R(g, i + kappa, i + kappa), (int)(rexpo - normexp));
      for (int j = i + 1; j < bs; ++j)
      {
        double m = ldexp(MU(g, j + kappa, i + kappa), (int)(g->row_expo[j + kappa] - g->row_expo[i + kappa]));
        if (dual)
          mut[(size_t)(bs - j - 1) * bs + (bs - i - 1)] = -m;
        else
          mut[(size_t)i * bs + j] = m;
      }
    }
    double best_dist = 0.0;
    if (getenv("ORACLE_BKZ_DEBUG"))
      fprintf(stderr, "call kappa %d dim %d maxdist %a r0 %a pruning %d dual %d\n", kappa, bs, maxdist, rdiag[0], pr, dual);
    int64_t nsol;
    if (dual)
    {
      nsol = oracle_enumerate_dual(bs, mut, rdiag, pruning, maxdist, nodes, sol, &best_dist);
      if (nsol > 0)
        for (int a = 0, b2 = bs - 1; a < b2; ++a, --b2)
        { /* reverse_by_swap, enumerate.cpp:154-158 */
          double t = sol[a]; sol[a] = sol[b2]; sol[b2] = t;
        }
    }
    else
      nsol = oracle_enumerate(bs, mut, rdiag, pruning, maxdist, 0, NULL, NULL, NULL, nodes, sol,
                              &best_dist);
    cx->enum_calls++;
    uint64_t t = 0;
    for (int i = 0; i <= bs; ++i)
      t += nodes[i];
    cx->nodes += t;
    if (getenv("ORACLE_BKZ_DEBUG"))
      fprintf(stderr, "   nodes %llu nsol %lld\n", (unsigned long long)t, (long long)nsol);
    if (nsol > 0)
    {
      rc = svp_postprocessing(g, kappa, bs, sol, dual);
      if (rc != 1)
        break;
      rerandomize = 0;
    }
    else
      rerandomize = 1;
    remaining_probability *= (1 - expectation);
  }
  free(rdiag); free(mut); free(sol); free(nodes);
  if (rc != 1)
    return rc;
  rc = lll_size_reduction(g, 0, first + 1, 0, eta);
  if (rc != 1)
    return rc;
  double new_first    = R(g, first, first);
  long new_first_expo = (long)(2 * g->row_expo[first]);
  new_first           = ldexp(new_first, (int)(new_first_expo - old_first_expo));
  *clean              = dual ? (old_first >= new_first) : (old_first <= new_first);
  return 1;
}

/* BKZReduction::trunc_tour, bkz.cpp:382-399 */
static int bkz_trunc_tour(oracle_gso *g, bkz_ctx *cx, const bkz_par *par, int min_row, int max_row, int *clean)
{
  int c1 = 1, rc;
  for (int kappa = min_row; kappa < max_row - par->block_size; ++kappa)
  {
    rc = svp_reduction(g, cx, par, kappa, par->block_size, &c1, 0);
    if (rc != 1)
      return rc;
    *clean &= c1;
  }
  return 1;
}

/* BKZReduction::trunc_dtour, bkz.cpp:401-413 */
static int bkz_trunc_dtour(oracle_gso *g, bkz_ctx *cx, const bkz_par *par, int min_row, int max_row, int *clean)
{
  int c1 = 1, rc;
  for (int kappa = max_row - par->block_size; kappa > min_row; --kappa)
  {
    rc = svp_reduction(g, cx, par, kappa, par->block_size, &c1, 1);
    if (rc != 1)
      return rc;
    *clean &= c1;
  }
  return 1;
}

/* BKZReduction::hkz, bkz.cpp:415-441 */
static int bkz_hkz(oracle_gso *g, bkz_ctx *cx, const bkz_par *par, int min_row, int max_row, int *clean)
{
  int c1 = 1, rc;
  for (int kappa = min_row; kappa < max_row - 1; ++kappa)
  {
    rc = svp_reduction(g, cx, par, kappa, max_row - kappa, &c1, 0);
    if (rc != 1)
      return rc;
    *clean &= c1;
  }
  lll_size_reduction(g, max_row - 1, max_row, max_row - 2, cx->eta); /* bkz.cpp:437 */
  return 1;
}

/* MatGSOInterface::get_log_det / get_slide_potential, gso_interface.cpp:230-258 (the stored r_ii are
 * read as they are, like the reference's get_r) */
static double get_log_det(oracle_gso *g, int start_row, int end_row)
{
  if (start_row < 0)
    start_row = 0;
  if (end_row > g->d)
    end_row = g->d;
  double log_det = 0.0;
  for (int i = start_row; i < end_row; ++i)
    log_det += log(ldexp(R(g, i, i), (int)(2 * g->row_expo[i])));
  return log_det;
}

static double get_slide_potential(oracle_gso *g, int start_row, int end_row, int block_size)
{
  double potential = 0.0;
  int p            = (end_row - start_row) / block_size;
  if ((end_row - start_row) % block_size == 0)
    --p;
  for (int i = 0; i < p; ++i)
    potential += (p - i) * get_log_det(g, i * block_size, (i + 1) * block_size);
  return potential;
}

/* BKZReduction::sd_tour, bkz.cpp:443-463 */
static int bkz_sd_tour(oracle_gso *g, bkz_ctx *cx, const bkz_par *par, int min_row, int max_row, int *clean_out)
{
  int clean = 1;
  int rc    = bkz_trunc_dtour(g, cx, par, min_row, max_row, &clean);
  if (rc != 1)
    return rc;
  rc = bkz_trunc_tour(g, cx, par, min_row, max_row, &clean);
  *clean_out = clean;
  return rc;
}

/* BKZReduction::slide_tour, bkz.cpp:465-520 */
static int bkz_slide_tour(oracle_gso *g, bkz_ctx *cx, const bkz_par *par, int min_row, int max_row, int *clean_out)
{
  int p = (max_row - min_row) / par->block_size;
  if ((max_row - min_row) % par->block_size)
    ++p;
  int clean, rc, c1 = 1;
  do
  {
    clean = 1;
    for (int i = 0; i < p; ++i)
    {
      int kappa      = min_row + i * par->block_size;
      int block_size = max_row - kappa < par->block_size ? max_row - kappa : par->block_size;
      rc             = svp_reduction(g, cx, par, kappa, block_size, &c1, 0);
      if (rc != 1)
        return rc;
      clean &= c1;
    }
    if (par->flags & 0x10)
    {
      int info[4];
      rc = oracle_gso_lll(g, min_row, min_row, max_row, cx->lll_delta, cx->eta, info);
      if (rc != 1)
        return rc;
      if (info[1] > 0)
        clean = 0;
    }
  } while (!clean);
  for (int i = 0; i < p - 1; ++i)
  {
    int kappa = min_row + i * par->block_size + 1;
    rc        = svp_reduction(g, cx, par, kappa, par->block_size, &c1, 1);
    if (rc != 1)
      return rc;
  }
  double new_potential = get_slide_potential(g, min_row, max_row, par->block_size);
  if (new_potential >= cx->sld_potential)
  {
    *clean_out = 1;
    return 1;
  }
  cx->sld_potential = new_potential;
  *clean_out        = 0;
  return 1;
}

/* BKZReduction::tour = trunc_tour + hkz, bkz.cpp:360-441 */
static int bkz_tour(oracle_gso *g, bkz_ctx *cx, const bkz_par *par, int min_row, int max_row,
                    int *clean_out)
{
  int clean = 1, c1 = 1, rc;
  const int block_size = par->block_size;
  for (int kappa = min_row; kappa < max_row - block_size; ++kappa)
  { /* trunc_tour */
    rc = svp_reduction(g, cx, par, kappa, block_size, &c1, 0);
    if (rc != 1)
      return rc;
    clean &= c1;
  }
  int hkz_min = max_row - block_size > 0 ? max_row - block_size : 0;
  for (int kappa = hkz_min; kappa < max_row - 1; ++kappa)
  { /* hkz */
    rc = svp_reduction(g, cx, par, kappa, max_row - kappa, &c1, 0);
    if (rc != 1)
      return rc;
    clean &= c1;
  }
  lll_size_reduction(g, max_row - 1, max_row, max_row - 2, cx->eta); /* bkz.cpp:437 */
  *clean_out = clean;
  return 1;
}

/* MatGSOInterface::get_current_slope(start_row, stop_row), gso_interface.cpp:198-218 */
static double current_slope(oracle_gso *g, int start_row, int stop_row)
{
  int n     = stop_row - start_row;
  double v1 = 0, v2 = (double)(n + 1) * n * (n - 1) / 12.0, weight = (1.0 - n) / 2.0;
  for (int i = start_row; i < stop_row; i++)
  {
    oracle_gso_update_row(g, i, i);
    double f    = R(g, i, i);
    long expo   = (long)(2 * g->row_expo[i]);
    double logf = log(f);
    v1 += weight * (logf + expo * log(2.0));
    weight++;
  }
  return v1 / v2;
}

/* BKZReduction::bkz() (bkz.cpp:522-668), primal BKZ.  flags: fplll's BKZ_MAX_LOOPS 0x4,
 * BKZ_BOUNDED_LLL 0x10, BKZ_AUTO_ABORT 0x20 (BKZAutoAbort::test_abort with scale 1.0 and 5 tours,
 * bkz.cpp:800-809), BKZ_GH_BND 0x80, BKZ_SD_VARIANT 0x100 (self-dual BKZ: dual + primal truncated
 * tours), BKZ_SLD_RED 0x200 (slide reduction).  strat NULL = empty strategies.
 * returns 1 RED_SUCCESS, 8 RED_BKZ_LOOPS_LIMIT, else a failure status (<= 0).
 * info[0] = tours executed, info[1..2] = enumeration nodes (lo, hi 32 bits), info[3] = enumeration
 * calls, info[4] = rerandomisations. */
int oracle_gso_bkz_param(oracle_gso *g, int block_size, double delta, double eta, int flags,
                         int max_loops, double gh_factor, const oracle_strategies *strat,
                         oracle_rand_fn rnd, void *rnd_user, int *info)
{
  int num_rows = g->d;
  for (; num_rows > 0; num_rows--)
  { /* trailing zero rows are not part of the lattice, bkz.cpp:35-37 */
    int z = 1;
    for (int c = 0; c < g->n; c++)
      if (B(g, num_rows - 1, c) != 0)
        z = 0;
    if (!z)
      break;
  }
  bkz_ctx cx;
  memset(&cx, 0, sizeof cx);
  cx.delta     = delta;
  cx.lll_delta = delta;
  cx.eta       = eta;
  cx.strat     = strat;
  cx.rnd       = rnd;
  cx.rnd_user  = rnd_user;
  bkz_par par  = {block_size, flags, gh_factor, 0.5, 3};
  cx.top_par   = &par;
  int status = 1, tours = 0;
  const int sd = (flags & 0x100) != 0, sld = (flags & 0x200) != 0; /* BKZ_SD_VARIANT, BKZ_SLD_RED */
  if (sd && sld)
    return -3;
  if (sd && !(flags & (0x4 | 0x8 | 0x20)))
    flags |= 0x20; /* "SD Variant of BKZ requires explicit termination condition", bkz.cpp:548-554 */
  const int auto_abort    = (flags & 0x20) != 0;
  const int use_max_loops = (flags & 0x4) != 0;
  int no_dec       = -1;
  double old_slope = DBL_MAX; /* numeric_limits<double>::max(), bkz.h BKZAutoAbort ctor */
  if (block_size < 2)
    goto done;
  if (sld)
  { /* bkz.cpp:567-571 */
    oracle_gso_update_all(g);
    cx.sld_potential = get_slide_potential(g, 0, num_rows, block_size);
  }
  if (sd)
  { /* bkz.cpp:576-577 */
    int rc0 = oracle_gso_lll(g, 0, 0, num_rows, cx.lll_delta, cx.eta, NULL);
    (void)rc0;
  }
  for (int i = 0;; ++i)
  {
    if (use_max_loops && i >= max_loops)
    {
      status = 8;
      break;
    }
    if (auto_abort)
    { /* BKZAutoAbort::test_abort(1.0, 5) */
      double new_slope = -current_slope(g, 0, num_rows);
      if (no_dec == -1 || new_slope < 1.0 * old_slope)
        no_dec = 0;
      else
        no_dec++;
      if (new_slope < old_slope)
        old_slope = new_slope;
      if (no_dec >= 5)
        break;
    }
    int clean = 1;
    int rc    = sd ? bkz_sd_tour(g, &cx, &par, 0, num_rows, &clean)
                : sld ? bkz_slide_tour(g, &cx, &par, 0, num_rows, &clean)
                      : bkz_tour(g, &cx, &par, 0, num_rows, &clean);
    if (rc != 1)
    {
      status = rc;
      goto done;
    }
    ++tours;
    if (clean || block_size >= num_rows)
      break;
  }
  /* post-processing, bkz.cpp:627-668 */
  if (sd)
  { /* hkz reduce the last window, which sd leaves unreduced */
    int dummy = 1;
    int rc    = bkz_hkz(g, &cx, &par, num_rows - block_size, num_rows, &dummy);
    if (rc != 1)
      status = rc;
  }
  if (sld)
  { /* hkz reduce the blocks (which are otherwise only svp and dual svp reduced) */
    int p = num_rows / block_size;
    if (num_rows % block_size)
      ++p;
    for (int j = 0; j < p; ++j)
    {
      int kappa = j * block_size + 1;
      int end   = num_rows < kappa + block_size - 1 ? num_rows : kappa + block_size - 1;
      int dummy = 1;
      int rc    = bkz_hkz(g, &cx, &par, kappa, end, &dummy);
      if (rc != 1)
      {
        status = rc;
        break;
      }
    }
  }
done:
  if (info)
  {
    info[0] = tours;
    info[1] = (int)(cx.nodes & 0xffffffffu);
    info[2] = (int)(cx.nodes >> 32);
    info[3] = (int)cx.enum_calls;
    info[4] = (int)cx.rerandomizations;
  }
  return status;
}

/* The radius and the pruning set svp_reduction (primal) would use for the block [kappa, kappa+bs) of
 * the CURRENT state (rows must be valid): bkz.cpp:309-325.  max_dist is scaled by 2^-(2 row_expo[kappa])
 * like r(kappa,kappa); *prune = index into strat->prune_* or -1.  For checking the product's host-side
 * mailbox service on the CPU. */
void oracle_gso_bkz_radius(oracle_gso *g, int kappa, int bs, int flags, double delta, double gh_factor,
                           const oracle_strategies *strat, double *max_dist_out, int *prune_out)
{
  bkz_ctx cx;
  memset(&cx, 0, sizeof cx);
  cx.delta           = delta;
  cx.strat           = strat;
  const int first    = (flags & 0x20000) ? kappa + bs - 1 : kappa; /* 0x20000: a dual block */
  double max_dist    = R(g, first, first);
  long max_dist_expo = (long)(2 * g->row_expo[first]);
  if (flags & 0x20000)
  {
    max_dist      = pow(max_dist, (double)-1);
    max_dist_expo = -max_dist_expo;
  }
  max_dist = max_dist * delta;
  if ((flags & 0x80) && bs > 30)
  {
    double root_det = get_root_det(g, kappa, kappa + bs);
    adjust_radius_to_gh_bound(&max_dist, max_dist_expo, bs, root_det, gh_factor);
  }
  *max_dist_out = max_dist;
  *prune_out    = get_pruning(g, &cx, kappa, bs);
}

/* One BKZReduction::svp_reduction(kappa, block_size, BKZParam(block_size, {}), dual) (bkz.cpp:274-358)
 * with empty strategies — the call the reference's own dual-SVP known-answer test makes
 * (tests/test_svp.cpp:214-262).  returns 1 or a failure status; *clean as the reference's return. */
int oracle_gso_svp_reduction(oracle_gso *g, int kappa, int block_size, int dual, double delta,
                             double eta, int *clean, uint64_t *nodes)
{
  bkz_ctx cx;
  memset(&cx, 0, sizeof cx);
  cx.delta     = delta;
  cx.lll_delta = delta;
  cx.eta       = eta;
  bkz_par par  = {block_size, 0, 1.1, 0.5, 3};
  int c        = 1;
  int rc       = svp_reduction(g, &cx, &par, kappa, block_size, &c, dual);
  if (clean)
    *clean = c;
  if (nodes)
    *nodes = cx.nodes;
  return rc;
}

/* the strategy-less form used by the device parity tests: use_max_loops bit 0 = BKZ_MAX_LOOPS,
 * bit 1 = BKZ_AUTO_ABORT.  info[3]: tours, nodes lo, nodes hi. */
int oracle_gso_bkz(oracle_gso *g, int block_size, double delta, double eta, int use_max_loops,
                   int max_loops, int *info)
{
  int inf5[5];
  int flags = ((use_max_loops & 1) ? 0x4 : 0) | ((use_max_loops & 2) ? 0x20 : 0);
  int st    = oracle_gso_bkz_param(g, block_size, delta, eta, flags, max_loops, 1.1, NULL, NULL, NULL, inf5);
  if (info)
  {
    info[0] = inf5[0];
    info[1] = inf5[1];
    info[2] = inf5[2];
  }
  return st;
}

const double *oracle_gso_mu(const oracle_gso *g) { return g->mu; }
const double *oracle_gso_r(const oracle_gso *g) { return g->r; }
const double *oracle_gso_bf(const oracle_gso *g) { return g->bf; }
const int64_t *oracle_gso_b(const oracle_gso *g) { return g->b; }
const int64_t *oracle_gso_row_expo(const oracle_gso *g) { return g->row_expo; }

/* get_mu / get_r, gso_interface.h:694-732 */
double oracle_gso_get_mu(const oracle_gso *g, int i, int j)
{
  return ldexp(MU(g, i, j), (int)(g->row_expo[i] - g->row_expo[j]));
}
double oracle_gso_get_r(const oracle_gso *g, int i, int j)
{
  return ldexp(R(g, i, j), (int)(g->row_expo[i] + g->row_expo[j]));
}
