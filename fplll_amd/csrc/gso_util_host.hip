// gso_util_host.hip — the host-side members of MatGSOInterface that BKZ callers use between reductions
// (fplll/gso_interface.cpp:197-276): get_current_slope, get_log_det, get_root_det, get_slide_potential
// and the free function adjust_radius_to_gh_bound.  Stateless restatements over the values a caller has
// downloaded (fphip_gso_get_r's diagonal as STORED, i.e. without the row exponents, and
// fphip_gso_get_row_expo): every expression in the reference's operation order, the host's libm — the
// numbers are the reference's bit for bit (tests/test_gso_util_cpu.py, `ref_driver gsoutil`).  Host code
// only: the slope test of BKZ_AUTO_ABORT and the slide potential need log() of the host, which is why
// the device BKZ evaluates them on the host between tours as well (gso_host.hip).
#include <algorithm>
#include <cmath>
#include <cstdint>

#include "../../include/fplll_hip.h"

namespace
{
// get_r(h, i, i) (gso_interface.h:707-717): the stored value times 2^(2 row_expo[i])
inline double true_r(const double *r_diag, const int64_t *row_expo, int i)
{
  return row_expo ? std::ldexp(r_diag[i], (int)(2 * row_expo[i])) : r_diag[i];
}
double log_det(const double *r_diag, const int64_t *row_expo, int d, int start_row, int end_row)
{
  double acc = 0.0;
  start_row  = std::max(0, start_row);
  end_row    = std::min(d, end_row);
  for (int i = start_row; i < end_row; ++i)
    acc += std::log(true_r(r_diag, row_expo, i));
  return acc;
}
}  // namespace

// get_current_slope (gso_interface.cpp:197-218): least-squares slope of log r_ii over [start, stop);
// each term is log(stored r_ii) + (2 row_expo[i]) log 2, as get_r_exp hands the value out
extern "C" double fphip_gso_util_current_slope(const double *r_diag, const int64_t *row_expo, int start_row,
                                               int stop_row)
{
  const int n = stop_row - start_row;
  double v1 = 0, v2 = (double)(n + 1) * n * (n - 1) / 12.0, weight = (1.0 - n) / 2.0;
  for (int i = start_row; i < stop_row; i++)
  {
    const long expo = row_expo ? (long)(2 * row_expo[i]) : 0;
    const double lf = std::log(r_diag[i]);
    v1 += weight * (lf + expo * std::log(2.0));
    weight++;
  }
  return v1 / v2;
}
extern "C" double fphip_gso_util_log_det(const double *r_diag, const int64_t *row_expo, int d, int start_row,
                                         int end_row)
{
  return log_det(r_diag, row_expo, d, start_row, end_row);
}
// get_root_det (:220-228): exp(log_det / number of rows)
extern "C" double fphip_gso_util_root_det(const double *r_diag, const int64_t *row_expo, int d, int start_row,
                                          int end_row)
{
  start_row      = std::max(0, start_row);
  end_row        = std::min(d, end_row);
  const double h = (double)(end_row - start_row);
  return std::exp(log_det(r_diag, row_expo, d, start_row, end_row) / h);
}
// get_slide_potential (:244-258) — as the reference computes it: the blocks are counted from row 0
// whatever start_row is
extern "C" double fphip_gso_util_slide_potential(const double *r_diag, const int64_t *row_expo, int d,
                                                 int start_row, int end_row, int block_size)
{
  double potential = 0.0;
  int p            = (end_row - start_row) / block_size;
  if ((end_row - start_row) % block_size == 0)
    --p;
  for (int i = 0; i < p; ++i)
    potential += (p - i) * log_det(r_diag, row_expo, d, i * block_size, (i + 1) * block_size);
  return potential;
}
// adjust_radius_to_gh_bound (:260-276): max_dist (a mantissa with exponent max_dist_expo) is lowered to
// gh_factor x the Gaussian heuristic of a block_size-dimensional lattice of that root determinant
extern "C" double fphip_gso_util_adjust_radius_to_gh_bound(double max_dist, long max_dist_expo, int block_size,
                                                           double root_det, double gh_factor)
{
  double t = (double)block_size / 2.0 + 1;
  t        = std::lgamma(t);
  t        = std::pow(M_E, t * 2.0 / (double)block_size);
  t        = t / M_PI;
  double f = t;
  f        = f * root_det;
  f        = std::ldexp(f, (int)-max_dist_expo);
  f        = f * gh_factor;
  return f < max_dist ? f : max_dist;
}
// is_lll_reduced<ZT, FT = double>(m, delta, eta) (lll.cpp:226-258) on the STORED mu / r (d x d row-major,
// as fphip_gso_get_mu / fphip_gso_get_r hand them out) and the row exponents: every |mu(i,j)| <= eta, then
// r(i,i) >= (delta - mu(i,i-1)^2) r(i-1,i-1) for every row — in the precision of the GSO itself, like the
// reference's predicate (which is why the test infrastructure judges a 180-dimensional tour at 256 bits).
extern "C" int fphip_gso_util_is_lll_reduced(const double *mu, const double *r, const int64_t *row_expo, int d,
                                             double delta, double eta)
{
  auto ex    = [&](int i) { return row_expo ? (int)row_expo[i] : 0; };
  auto get_mu = [&](int i, int j) { return std::ldexp(mu[(size_t)i * d + j], ex(i) - ex(j)); };
  auto get_r  = [&](int i, int j) { return std::ldexp(r[(size_t)i * d + j], ex(i) + ex(j)); };
  for (int i = 0; i < d; i++)
    for (int j = 0; j < i; j++)
      if (std::fabs(get_mu(i, j)) > eta)
        return 0;
  for (int i = 1; i < d; i++)
  {
    double t = get_mu(i, i - 1);
    t        = t * t;
    t        = delta - t;
    t        = get_r(i - 1, i - 1) * t;
    if (get_r(i, i) < t)
      return 0;
  }
  return 1;
}
