/*
 * hh_oracle.c — TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Plain-C restatement of fplll's Householder R-factor computation for
 * MatHouseholder<Z_NR<long>, FP_NR<double>> (fplll/householder.{h,cpp}): refresh_R_bf (:186-245),
 * update_R(i, true) (:151-184) and update_R_last (:27-146, the default build: no
 * HOUSEHOLDER_PRECOMPUTE_INVERSE, no DEBUG), i.e. MatHouseholder::update_R() over all rows
 * (householder.h:532-536).  Pinned against the real reference by tests/test_hh_oracle_vs_ref.py.
 * Sums keep the reference's order (dot_product ascending, nr/numvect.h:386-396; addmul element-wise
 * with two roundings, numvect.h:300-305).  Compile with -ffp-contract=off.
 */
#include "oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* R, V: d×n row-major; sigma, row_expo: d.  returns 0. */
int oracle_hh_update_all(int d, int n, const int64_t *b, int row_expo_on, double *R, double *V,
                         double *sigma, int64_t *row_expo)
{
  long *tmp_expo = (long *)malloc(sizeof(long) * n);
  int n_known_cols = 0;
  memset(V, 0, sizeof(double) * d * n);
  /* refresh_R_bf() for every row, householder.cpp:186-245 */
  for (int i = 0; i < d; ++i)
  {
    int nz = 1;
    for (int j = n - 1; j >= 0; --j)
      if (b[(size_t)i * n + j] != 0)
      {
        nz = j + 1;
        break;
      }
    if (nz > n_known_cols)
      n_known_cols = nz;
    double *Ri = R + (size_t)i * n;
    if (row_expo_on)
    {
      long max_expo = LONG_MIN;
      for (int j = 0; j < n_known_cols; ++j)
      {
        int e;
        Ri[j]       = frexp((double)b[(size_t)i * n + j], &e);
        tmp_expo[j] = e;
        if (e > max_expo)
          max_expo = e;
      }
      for (int j = 0; j < n_known_cols; ++j)
        Ri[j] = ldexp(Ri[j], (int)(tmp_expo[j] - max_expo));
      row_expo[i] = max_expo;
    }
    else
    {
      for (int j = 0; j < n_known_cols; ++j)
        Ri[j] = (double)b[(size_t)i * n + j];
      row_expo[i] = 0;
    }
    for (int j = n_known_cols; j < n; ++j)
      Ri[j] = 0.0;
  }
  /* update_R(i, true) for every row */
  for (int i = 0; i < d; ++i)
  {
    double *Ri = R + (size_t)i * n;
    for (int j = 0; j < i; ++j)
    {
      const double *Vj = V + (size_t)j * n;
      double s         = Vj[j] * Ri[j]; /* dot_product(beg=j, n) */
      for (int c = j + 1; c < n; ++c)
        s = s + Vj[c] * Ri[c];
      s = -s;
      for (int c = n - 1; c >= j; --c) /* addmul, numvect.h:300-305 */
        Ri[c] = Ri[c] + Vj[c] * s;
      Ri[j] = sigma[j] * Ri[j];
    }
    /* update_R_last(i), householder.cpp:27-146 */
    double *Vi = V + (size_t)i * n;
    sigma[i]   = (Ri[i] < 0.0) ? -1.0 : 1.0;
    double f3;
    if (i + 1 == n)
      f3 = 0.0;
    else
    {
      f3 = Ri[i + 1] * Ri[i + 1];
      for (int c = i + 2; c < n; ++c)
        f3 = f3 + Ri[c] * Ri[c];
    }
    double f1 = Ri[i] * Ri[i];
    f1        = f1 + f3;
    if (f1 != 0.0)
    {
      double f2 = sqrt(f1);
      double f0 = sigma[i] * f2;
      f1        = Ri[i] + f0;
      f3        = -f3;
      f3        = f3 / f1;
      if (f3 != 0.0)
      {
        f0    = -f0;
        f0    = f0 * f3;
        f0    = sqrt(f0);
        Vi[i] = f3 / f0;
        Ri[i] = f2;
        for (int c = n - 1; c >= i + 1; --c)
          Vi[c] = Ri[c] / f0;
      }
      else
      {
        Vi[i] = 0.0;
        if (Ri[i] < 0.0)
          Ri[i] = -Ri[i];
        for (int c = i + 1; c < n; ++c)
          Vi[c] = 0.0;
      }
    }
    else
    {
      Ri[i] = 0.0;
      Vi[i] = 0.0;
      for (int c = i + 1; c < n; ++c)
        Vi[c] = 0.0;
    }
  }
  free(tmp_expo);
  return 0;
}

/* MatHouseholder::size_reduce(k, size_reduction_end, size_reduction_start), householder.cpp:402-451, with
 * row_addmul_we (:522-559), on the state refresh_R_bf() + update_R() leave (oracle_hh_update_all): b row k and
 * R row k are updated in place (R(k, c) for every c < k: line 5 of Algorithm 3 plus what :551-556 add beyond it —
 * the row is invalidated by the reference afterwards).  Returns the reference's flag (1 reduced / 0 not), -1 for a
 * multiplier beyond 63 bits (path not restated).  Pinned by the `hhsr` fixtures (tests/test_hh_oracle_vs_ref.py). */
int oracle_hh_size_reduce(int d, int n, int64_t *b, int row_expo_on, int k, int end, int start, double *R,
                          int64_t *row_expo)
{
  double *V     = (double *)malloc(sizeof(double) * (size_t)d * n);
  double *sigma = (double *)malloc(sizeof(double) * (size_t)d);
  oracle_hh_update_all(d, n, b, row_expo_on, R, V, sigma, row_expo);
  free(V);
  free(sigma);
  int n_known_cols = 0;
  for (int i = 0; i < d; ++i)
    for (int j = n - 1; j >= 0; --j)
      if (b[(size_t)i * n + j] != 0)
      {
        if (j + 1 > n_known_cols)
          n_known_cols = j + 1;
        break;
      }
  int reduced = 0;
  double *Rk  = R + (size_t)k * n;
  for (int i = end - 1; i >= start; --i)
  {
    const double *Ri = R + (size_t)i * n;
    double x         = Rk[i] / Ri[i];
    long ea          = (long)(row_expo[k] - row_expo[i]);
    long fx          = (x == 0.0) ? (long)INT_MIN + 1 : (long)ilogb(x) + 1;
    if (!(fx + ea >= 53)) /* rnd_we, nr_FP_d.inl:226-233 */
      x = ldexp(rint(ldexp(x, (int)ea)), (int)-ea);
    x = -x;
    if (x != 0.0)
    {
      fx        = (long)ilogb(x) + 1;
      long expo = fx + ea - 63;
      if (expo > 0)
        return -1;
      long lx = (long)ldexp(x, (int)ea);
      for (int c = n_known_cols - 1; c >= 0; --c)
        b[(size_t)k * n + c] =
            (int64_t)((uint64_t)b[(size_t)k * n + c] + (uint64_t)b[(size_t)i * n + c] * (uint64_t)lx);
      /* R[k].addmul(R[i], x, k), householder.cpp:551-556: ALL k leading entries — R(k, i) becomes the remainder and
       * the entries between i and k pick up x times the tail update_R_last left in row i (never zeroed in a
       * non-DEBUG build, :104-111); x = +-1: add / sub, the same floats */
      for (int c = k - 1; c >= 0; --c)
        Rk[c] = Rk[c] + Ri[c] * x;
      reduced = 1;
    }
  }
  return reduced;
}

/* ------------------------------------------------------------------------------------------
 * HLLL: HLLLReduction<Z_NR<long>, FP_NR<double>>::hlll() over MatHouseholder with
 * HOUSEHOLDER_ROW_EXPO (| HOUSEHOLDER_OP_FORCE_LONG), the LM_FAST configuration of
 * hlll_reduction_zf (wrapper.cpp:790-806).  Restates
 *   HLLLReduction::hlll / lovasz_test / size_reduction / verify_size_reduction  hlll.cpp:26-499
 *   compute_dR / compute_eR                                                     hlll.h:148-159
 *   MatHouseholder::update_R(i,false) / update_R_last / refresh_R_bf / refresh_R / swap /
 *   size_reduce / row_addmul_we                             householder.cpp:27-261,372-451,522-559
 * recover_R (householder.h:597-608) restores, from R_history, exactly the values that
 * refresh_R(i) + update_R(i,false) recompute (same operands, same operation order), so it is
 * restated as that recomputation and the d×d×n R_history is not kept.
 * Pinned against the real reference by tests/test_hlll_oracle_vs_ref.py (`hlllfix` fixtures).
 * Returns 1 RED_SUCCESS, -2 multiplier beyond 63 bits (path not restated), -4 RED_HLLL_SR_FAILURE,
 * -5 RED_HLLL_NORM_FAILURE.  info[0] = number of swaps, info[1] = loop iterations.
 * ------------------------------------------------------------------------------------------ */
typedef struct
{
  int d, n, row_expo_on, n_known_cols;
  int64_t *b;
  double *bf, *R, *V, *sigma, *nsb, *dR, *eR;
  int64_t *rexp;
  int *init_row_size;
  long *tmp_expo;
} hh_state;

static long hh_fexponent(double x) { return (x == 0.0) ? (long)INT_MIN + 1 : (long)ilogb(x) + 1; }

static void hh_refresh_R(hh_state *s, int i)
{
  for (int j = 0; j < s->n_known_cols; ++j)
    s->R[(size_t)i * s->n + j] = s->bf[(size_t)i * s->n + j];
  for (int j = s->n_known_cols; j < s->n; ++j)
    s->R[(size_t)i * s->n + j] = 0.0;
}

static void hh_refresh_R_bf(hh_state *s, int i)
{
  const int n = s->n;
  if (s->init_row_size[i] > s->n_known_cols)
    s->n_known_cols = s->init_row_size[i];
  double *bf = s->bf + (size_t)i * n;
  if (s->row_expo_on)
  {
    long max_expo = LONG_MIN;
    for (int j = 0; j < s->n_known_cols; ++j)
    {
      int e;
      bf[j]          = frexp((double)s->b[(size_t)i * n + j], &e);
      s->tmp_expo[j] = e;
      if (e > max_expo)
        max_expo = e;
    }
    for (int j = 0; j < s->n_known_cols; ++j)
      bf[j] = ldexp(bf[j], (int)(s->tmp_expo[j] - max_expo));
    s->rexp[i] = max_expo;
  }
  else
  {
    for (int j = 0; j < s->n_known_cols; ++j)
      bf[j] = (double)s->b[(size_t)i * n + j];
    s->rexp[i] = 0;
  }
  for (int j = s->n_known_cols; j < n; ++j)
    bf[j] = 0.0;
  hh_refresh_R(s, i);
  /* norm_square_b_row, householder.h:538-551 */
  double f = bf[0] * bf[0];
  for (int j = 1; j < s->n_known_cols; ++j)
    f = f + bf[j] * bf[j];
  s->nsb[i] = f;
}

/* update_R(i, false), householder.cpp:151-184 */
static void hh_apply_reflectors(hh_state *s, int i)
{
  const int n = s->n;
  double *Ri  = s->R + (size_t)i * n;
  for (int j = 0; j < i; ++j)
  {
    const double *Vj = s->V + (size_t)j * n;
    double t         = Vj[j] * Ri[j];
    for (int c = j + 1; c < n; ++c)
      t = t + Vj[c] * Ri[c];
    t = -t;
    for (int c = n - 1; c >= j; --c)
      Ri[c] = Ri[c] + Vj[c] * t;
    Ri[j] = s->sigma[j] * Ri[j];
  }
}

/* update_R_last(i), householder.cpp:27-146 */
static void hh_update_R_last(hh_state *s, int i)
{
  const int n = s->n;
  double *Ri = s->R + (size_t)i * n, *Vi = s->V + (size_t)i * n;
  s->sigma[i] = (Ri[i] < 0.0) ? -1.0 : 1.0;
  double f3;
  if (i + 1 == n)
    f3 = 0.0;
  else
  {
    f3 = Ri[i + 1] * Ri[i + 1];
    for (int c = i + 2; c < n; ++c)
      f3 = f3 + Ri[c] * Ri[c];
  }
  double f1 = Ri[i] * Ri[i];
  f1        = f1 + f3;
  if (f1 != 0.0)
  {
    double f2 = sqrt(f1);
    double f0 = s->sigma[i] * f2;
    f1        = Ri[i] + f0;
    f3        = -f3;
    f3        = f3 / f1;
    if (f3 != 0.0)
    {
      f0    = -f0;
      f0    = f0 * f3;
      f0    = sqrt(f0);
      Vi[i] = f3 / f0;
      Ri[i] = f2;
      for (int c = n - 1; c >= i + 1; --c)
        Vi[c] = Ri[c] / f0;
    }
    else
    {
      Vi[i] = 0.0;
      if (Ri[i] < 0.0)
        Ri[i] = -Ri[i];
      for (int c = i + 1; c < n; ++c)
        Vi[c] = 0.0;
    }
  }
  else
  {
    Ri[i] = 0.0;
    Vi[i] = 0.0;
    for (int c = i + 1; c < n; ++c)
      Vi[c] = 0.0;
  }
}

/* MatHouseholder::size_reduce(k, k, 0), householder.cpp:402-451; -1 = multiplier overflow */
static int hh_size_reduce(hh_state *s, int k)
{
  const int n = s->n;
  int reduced = 0;
  double *Rk  = s->R + (size_t)k * n;
  for (int i = k - 1; i >= 0; --i)
  {
    const double *Ri = s->R + (size_t)i * n;
    double x         = Rk[i] / Ri[i];
    long ea          = (long)(s->rexp[k] - s->rexp[i]);
    if (!(hh_fexponent(x) + ea >= 53)) /* rnd_we, nr_FP_d.inl:226-233 */
      x = ldexp(rint(ldexp(x, (int)ea)), (int)-ea);
    x = -x;
    if (x != 0.0)
    {
      /* row_addmul_we(k, i, x, ea), householder.cpp:522-559 */
      long expo = hh_fexponent(x) + ea - 63;
      if (expo < 0)
        expo = 0;
      if (expo != 0)
        return -1;
      long lx = (long)ldexp(x, (int)ea);
      for (int c = s->n_known_cols - 1; c >= 0; --c)
        s->b[(size_t)k * n + c] =
            (int64_t)((uint64_t)s->b[(size_t)k * n + c] + (uint64_t)s->b[(size_t)i * n + c] * (uint64_t)lx);
      if (x == 1.0)
        for (int c = i - 1; c >= 0; --c)
          Rk[c] = Rk[c] + Ri[c];
      else if (x == -1.0)
        for (int c = i - 1; c >= 0; --c)
          Rk[c] = Rk[c] - Ri[c];
      else
        for (int c = i - 1; c >= 0; --c)
          Rk[c] = Rk[c] + Ri[c] * x;
      reduced = 1;
    }
  }
  return reduced;
}

int oracle_hlll(int d, int n, int64_t *b, int row_expo_on, double delta, double eta, double theta,
                double c, int *info)
{
  (void)eta;
  (void)c;
  hh_state S, *s = &S;
  memset(s, 0, sizeof S);
  s->d = d;
  s->n = n;
  s->row_expo_on = row_expo_on;
  s->b           = b;
  s->bf          = (double *)calloc((size_t)d * n, sizeof(double));
  s->R           = (double *)calloc((size_t)d * n, sizeof(double));
  s->V           = (double *)calloc((size_t)d * n, sizeof(double));
  s->sigma       = (double *)calloc(d, sizeof(double));
  s->nsb         = (double *)calloc(d, sizeof(double));
  s->dR          = (double *)calloc(d, sizeof(double));
  s->eR          = (double *)calloc(d, sizeof(double));
  s->rexp        = (int64_t *)calloc(d, sizeof(int64_t));
  s->init_row_size = (int *)calloc(d, sizeof(int));
  s->tmp_expo      = (long *)calloc(n, sizeof(long));
  double *prev_R   = (double *)calloc(d, sizeof(double));
  long *prev_expo  = (long *)calloc(d, sizeof(long));
  int64_t *tmprow  = (int64_t *)malloc(sizeof(int64_t) * n);
  double *tmpf     = (double *)malloc(sizeof(double) * n);
  for (int i = 0; i < d; ++i)
  {
    int nz = 1;
    for (int j = n - 1; j >= 0; --j)
      if (b[(size_t)i * n + j] != 0)
      {
        nz = j + 1;
        break;
      }
    s->init_row_size[i] = nz;
  }
  int status = 1, n_swaps = 0;
  long long iters = 0;
#define COMPUTE_DR(k)                                  \
  do                                                   \
  {                                                    \
    double t_ = s->R[(size_t)(k) * n + (k)];           \
    t_        = t_ * t_;                               \
    s->dR[k]  = delta * t_;                            \
  } while (0)
#define COMPUTE_ER(k) (s->eR[k] = delta * s->R[(size_t)(k) * n + (k)]) /* sic: hlll.h:155-159 uses delta */
  hh_refresh_R_bf(s, 0);
  hh_update_R_last(s, 0);
  COMPUTE_DR(0);
  COMPUTE_ER(0);
  int k = 1, k_max = 1, prev_k = -1;
  if (d < 2)
    goto done; /* the reference reads b[1] unconditionally; a 1-row basis is already reduced */
  hh_refresh_R_bf(s, 1);
  for (;;)
  {
    ++iters;
    /* ---- size_reduction(k, k, 0), hlll.cpp:262-351 */
    {
      int not_stop = 1, prev_not_stop = 1;
      const double approx = 0.1;
      hh_apply_reflectors(s, k);
      for (;;)
      {
        int reduced = hh_size_reduce(s, k);
        if (reduced < 0)
        {
          status = -2;
          goto done;
        }
        if (!reduced)
          break;
        double f0   = s->nsb[k];
        long expo0  = row_expo_on ? 2 * (long)s->rexp[k] : 0;
        hh_refresh_R_bf(s, k);
        double f1   = s->nsb[k];
        long expo1  = row_expo_on ? 2 * (long)s->rexp[k] : 0;
        f0          = approx * f0;
        f0          = ldexp(f0, (int)(expo0 - expo1));
        not_stop    = (f1 <= f0);
        hh_apply_reflectors(s, k);
        if (prev_not_stop || not_stop)
          prev_not_stop = not_stop;
        else
          break;
      }
    }
    /* ---- verify_size_reduction(k), hlll.cpp:455-496 */
    {
      const double *Rk = s->R + (size_t)k * n;
      double f1;
      if (n == k)
        f1 = 0.0;
      else
      {
        f1 = Rk[k] * Rk[k];
        for (int cc = k + 1; cc < n; ++cc)
          f1 = f1 + Rk[cc] * Rk[cc];
        f1 = sqrt(f1);
      }
      f1 = f1 * theta;
      for (int i = 0; i < k; ++i)
      {
        double f0 = fabs(Rk[i]);
        double f2 = ldexp(s->eR[i], (int)(s->rexp[i] - s->rexp[k]));
        f2        = f1 + f2;
        if (f0 > f2)
        {
          status = -4;
          goto done;
        }
      }
    }
    /* ---- lovasz_test(k), hlll.cpp:171-224 */
    int lov;
    {
      const double *Rk = s->R + (size_t)k * n;
      double f0 = s->nsb[k], f1;
      if (k - 1 == 0)
        f1 = 0.0;
      else
      {
        f1 = Rk[0] * Rk[0];
        for (int cc = 1; cc < k - 1; ++cc)
          f1 = f1 + Rk[cc] * Rk[cc];
      }
      f1         = f0 - f1;
      long expo1 = row_expo_on ? 2 * (long)s->rexp[k] : 0;
      long expo0 = (long)s->rexp[k - 1];
      f1         = ldexp(f1, (int)(expo1 - 2 * expo0));
      lov        = (s->dR[k - 1] <= f1);
    }
    if (lov)
    {
      hh_update_R_last(s, k);
      COMPUTE_DR(k);
      COMPUTE_ER(k);
      if (prev_k == k + 1)
      {
        double f0 = s->R[(size_t)k * n + k];
        double f1 = ldexp(prev_R[k], (int)(prev_expo[k] - (long)s->rexp[k]));
        if (f0 > f1)
        {
          status = -5;
          goto done;
        }
      }
      prev_k       = k;
      prev_R[k]    = s->R[(size_t)k * n + k];
      prev_expo[k] = (long)s->rexp[k];
      k++;
      if (k < d)
      {
        if (k > k_max)
        {
          k_max = k;
          hh_refresh_R_bf(s, k);
        }
        else
          hh_refresh_R(s, k);
      }
      else
        break; /* RED_SUCCESS */
    }
    else
    {
      /* swap(k-1, k), householder.cpp:372-398 */
      ++n_swaps;
      memcpy(tmprow, s->b + (size_t)(k - 1) * n, sizeof(int64_t) * n);
      memcpy(s->b + (size_t)(k - 1) * n, s->b + (size_t)k * n, sizeof(int64_t) * n);
      memcpy(s->b + (size_t)k * n, tmprow, sizeof(int64_t) * n);
      memcpy(tmpf, s->bf + (size_t)(k - 1) * n, sizeof(double) * n);
      memcpy(s->bf + (size_t)(k - 1) * n, s->bf + (size_t)k * n, sizeof(double) * n);
      memcpy(s->bf + (size_t)k * n, tmpf, sizeof(double) * n);
      {
        double t = s->sigma[k - 1]; s->sigma[k - 1] = s->sigma[k]; s->sigma[k] = t;
        int64_t e = s->rexp[k - 1]; s->rexp[k - 1] = s->rexp[k]; s->rexp[k] = e;
        int z = s->init_row_size[k - 1]; s->init_row_size[k - 1] = s->init_row_size[k]; s->init_row_size[k] = z;
        t = s->nsb[k - 1]; s->nsb[k - 1] = s->nsb[k]; s->nsb[k] = t;
      }
      prev_k = k;
      if (k - 1 == 0)
      {
        hh_refresh_R(s, 0);
        hh_update_R_last(s, 0);
        COMPUTE_DR(0);
        COMPUTE_ER(0);
        hh_refresh_R(s, 1);
        k = 1;
      }
      else
      {
        k--;
        hh_refresh_R(s, k); /* recover_R(k): see the header comment */
      }
    }
  }
done:
#undef COMPUTE_DR
#undef COMPUTE_ER
  if (info)
  {
    info[0] = n_swaps;
    info[1] = (int)(iters & 0x7fffffff);
  }
  free(s->bf); free(s->R); free(s->V); free(s->sigma); free(s->nsb); free(s->dR); free(s->eR);
  free(s->rexp); free(s->init_row_size); free(s->tmp_expo); free(prev_R); free(prev_expo);
  free(tmprow); free(tmpf);
  return status;
}
