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
int oracle_gso_lll(oracle_gso *g, int kappa_min, int kappa_start, int kappa_end, double delta,
                   double eta, int *info)
{
  const int n = g->n;
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
    long long max_iter = (long long)(d - 2 * d * (d + 1) * ((max_exp + 3) / log(delta)));
    for (iter = 0; iter < max_iter && kappa < kappa_end - zeros; iter++)
    {
      int rc = oracle_gso_babai(g, kappa, kappa, 0, eta);
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
      if (f > lovasz[kappa - 1])
      {
        n_swaps++;
        int old_k = kappa;
        for (kappa--; kappa > kappa_min; kappa--)
        {
          f = R(g, kappa - 1, kappa - 1) * delta;
          if (g->row_expo_on)
            f = ldexp(f, (int)(2 * (g->row_expo[kappa - 1] - g->row_expo[old_k])));
          if (f < lovasz[kappa - 1])
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

static int svp_postprocessing(oracle_gso *g, int kappa, int bs, const double *sol)
{
  int nz_vectors = 0, i_vector = -1;
  for (int i = bs - 1; i >= 0; i--)
    if (sol[i] != 0.0)
    {
      nz_vectors++;
      if (i_vector == -1 && fabs(sol[i]) == 1)
        i_vector = i;
    }
  if (nz_vectors == 1)
  {
    move_row(g, kappa + i_vector, kappa);
  }
  else if (i_vector != -1)
  {
    int sol_i = (int)sol[i_vector];
    for (int i = 0; i < bs; ++i)
      if (sol[i] != 0.0 && i != i_vector)
        if (!row_addmul_we(g, kappa + i_vector, kappa + i, sol_i * sol[i], 0))
          return -2;
    row_op_end_range(g, kappa + i_vector, kappa + i_vector + 1);
    move_row(g, kappa + i_vector, kappa);
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
              if (!row_addmul_we(g, kappa + k - off, kappa + k, 1.0, 0)) /* row_add */
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
    move_row(g, kappa + bs - 1, kappa);
  }
  return 1;
}

/* returns 1 ok (clean flag in *clean), else a failure status */
static int svp_reduction(oracle_gso *g, int kappa, int bs, double delta, double eta, int *clean,
                         uint64_t *total_nodes)
{
  int rc = lll_size_reduction(g, 0, kappa + 1, 0, eta);
  if (rc != 1)
    return rc;
  double old_first   = R(g, kappa, kappa);
  long old_first_expo = (long)(2 * g->row_expo[kappa]);
  /* svp_preprocessing: lll(0, 0, kappa + bs) (no recursive preprocessing in the empty strategy) */
  rc = oracle_gso_lll(g, 0, 0, kappa + bs, delta, eta, NULL);
  if (rc != 1)
    return rc;
  /* radius, bkz.cpp:311-317 */
  double max_dist    = R(g, kappa, kappa);
  long max_dist_expo = (long)(2 * g->row_expo[kappa]);
  max_dist           = max_dist * delta;
  /* EnumerationDyn::enumerate, enumerate.cpp:88-141 */
  long normexp = -1;
  for (int i = 0; i < bs; ++i)
  {
    long rexpo = (long)(2 * g->row_expo[i + kappa]);
    long e     = rexpo + fexponent_l(R(g, i + kappa, i + kappa));
    if (e > normexp)
      normexp = e;
  }
  double maxdist = ldexp(max_dist, (int)(max_dist_expo - normexp));
  double *rdiag  = (double *)calloc(bs, sizeof(double));
  double *mut    = (double *)calloc((size_t)bs * bs, sizeof(double));
  double *sol    = (double *)calloc(bs, sizeof(double));
  uint64_t *nodes = (uint64_t *)calloc(bs + 1, sizeof(uint64_t));
  for (int i = 0; i < bs; ++i)
  {
    long rexpo = (long)(2 * g->row_expo[i + kappa]);
    rdiag[i]   = ldexp(R(g, i + kappa, i + kappa), (int)(rexpo - normexp));
    for (int j = i + 1; j < bs; ++j)
      mut[(size_t)i * bs + j] =
          ldexp(MU(g, j + kappa, i + kappa), (int)(g->row_expo[j + kappa] - g->row_expo[i + kappa]));
  }
  double best_dist = 0.0;
  if (getenv("ORACLE_BKZ_DEBUG"))
    fprintf(stderr, "call dim %d maxdist %a r0 %a\n", bs, maxdist, rdiag[0]);
  int64_t nsol = oracle_enumerate(bs, mut, rdiag, NULL, maxdist, 0, NULL, NULL, NULL, nodes, sol,
                                  &best_dist);
  if (total_nodes)
    for (int i = 0; i <= bs; ++i)
      *total_nodes += nodes[i];
  if (getenv("ORACLE_BKZ_DEBUG"))
  {
    uint64_t t = 0;
    for (int i = 0; i <= bs; ++i)
      t += nodes[i];
    fprintf(stderr, "   nodes %llu nsol %lld\n", (unsigned long long)t, (long long)nsol);
  }
  rc = 1;
  if (nsol > 0)
    rc = svp_postprocessing(g, kappa, bs, sol);
  free(rdiag); free(mut); free(sol); free(nodes);
  if (rc != 1)
    return rc;
  rc = lll_size_reduction(g, 0, kappa + 1, 0, eta);
  if (rc != 1)
    return rc;
  double new_first    = R(g, kappa, kappa);
  long new_first_expo = (long)(2 * g->row_expo[kappa]);
  new_first           = ldexp(new_first, (int)(new_first_expo - old_first_expo));
  *clean              = (old_first <= new_first);
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

/* BKZReduction::bkz().  use_max_loops: bit 0 = BKZ_MAX_LOOPS (with max_loops), bit 1 = BKZ_AUTO_ABORT
 * (BKZAutoAbort::test_abort with scale 1.0 and 5 tours, bkz.cpp:800-809, bkz_param.h defaults).
 * returns 1 RED_SUCCESS, 8 RED_BKZ_LOOPS_LIMIT, else a failure status (<= 0).
 * info[0] = tours executed, info[1..2] = enumeration nodes (lo, hi 32 bits). */
int oracle_gso_bkz(oracle_gso *g, int block_size, double delta, double eta, int use_max_loops,
                   int max_loops, int *info)
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
  uint64_t nodes = 0;
  int status = 1, tours = 0;
  const int auto_abort = (use_max_loops & 2) != 0;
  use_max_loops &= 1;
  int no_dec       = -1;
  double old_slope = DBL_MAX; /* numeric_limits<double>::max(), bkz.h BKZAutoAbort ctor */
  if (block_size < 2)
    goto done;
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
    int clean = 1, c1 = 1, rc;
    /* trunc_tour */
    for (int kappa = 0; kappa < num_rows - block_size; ++kappa)
    {
      rc = svp_reduction(g, kappa, block_size, delta, eta, &c1, &nodes);
      if (rc != 1)
      {
        status = rc;
        goto done;
      }
      clean &= c1;
    }
    /* hkz */
    int min_row = num_rows - block_size > 0 ? num_rows - block_size : 0;
    for (int kappa = min_row; kappa < num_rows - 1; ++kappa)
    {
      rc = svp_reduction(g, kappa, num_rows - kappa, delta, eta, &c1, &nodes);
      if (rc != 1)
      {
        status = rc;
        goto done;
      }
      clean &= c1;
    }
    lll_size_reduction(g, num_rows - 1, num_rows, num_rows - 2, eta); /* bkz.cpp:437 */
    ++tours;
    if (clean || block_size >= num_rows)
      break;
  }
done:
  if (info)
  {
    info[0] = tours;
    info[1] = (int)(nodes & 0xffffffffu);
    info[2] = (int)(nodes >> 32);
  }
  return status;
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
