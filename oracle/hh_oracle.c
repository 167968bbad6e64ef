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
