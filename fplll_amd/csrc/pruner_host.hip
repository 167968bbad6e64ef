// pruner_host.hip — the pruner: cost and success probability of a pruned
// enumeration and the search for good pruning coefficients, host code like the reference's.
//
// Reference behaviour reproduced (fplll v5.5.0, FT = FP_NR<double>: plain doubles and the host libm):
//   prune<FP_NR<double>>(pruning, radius, preproc_cost, gso_r, target, metric, flags)   pruner/pruner.cpp:190-203
//   svp_probability<FP_NR<double>>(pr)                                                  pruner.cpp:166-176
//   Pruner::load_basis_shape / load_coefficients / save_coefficients / gaussian_heuristic   pruner_util.cpp
//   Pruner::enforce                                                                     pruner.h:1009-1054
//   eval_poly / integrate_poly / relative_volume (volume of the even simplex)           pruner_simplex.h
//   single_enum_cost(_evec/_lower/_upper), target_function, its numerical gradient, repeated_enum_cost
//                                                                                       pruner_cost.cpp
//   svp_probability(_evec/...), expected_solutions(...), measure_metric                 pruner_prob.cpp
//   optimize_coefficients, _cost_vary_prob, _cost_fixed_prob                            pruner_optimize.cpp
//   _preparation, _evec_core, _full_core, _local_adjust_decr_single / _incr_prob / _smooth, greedy,
//   gradient_descent(_step)                                                             pruner_optimize_tc.cpp
//   optimize_coefficients_incr_prob / _decr_prob / _local_adjust_prob                   pruner_optimize_tp.cpp
// Every expression keeps the reference's operation order (the coefficients come out of thousands of
// comparisons of floating-point cost values: the parity bar is bit-identical coefficients,
// tests/test_pruner_cpu.py against the real reference), including its integer divisions and float casts.
// Not offered: PRUNER_VERBOSE, FT other than double.
//
// Why it lives in this library: the strategy-BKZ host service uses single_enum_cost as its measure of a
// block's tree when it decides which enumerations go to the multi-wave enumerator (gso_host.hip), and a
// caller can build its strategies without fplll's CPU library (default.json is not shipped with the
// reference tree).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "../../include/fplll_hip.h"
#include "pruner_tables.h"

namespace
{
typedef std::vector<double> vec;

enum
{
  PR_CVP              = 0x1,
  PR_START_FROM_INPUT = 0x2,
  PR_GRADIENT         = 0x4,
  PR_NELDER_MEAD      = 0x8,
  PR_HALF             = 0x20,
  PR_SINGLE           = 0x40
};

struct Pruner
{
  double enumeration_radius = 0, preproc_cost = 0, target = 0;
  int metric = 0;  // 0 PRUNER_METRIC_PROBABILITY_OF_SHORTEST, 1 PRUNER_METRIC_EXPECTED_SOLUTIONS
  bool shape_loaded = false;
  int flags = 0, n = 0, d = 0;
  vec min_pruning_coefficients;
  bool opt_single = false;
  double epsilon         = std::pow(2., -7);
  double min_step        = std::pow(2., -6);
  double min_cf_decrease = .995;
  double step_factor     = std::pow(2, .5);
  double shell_ratio     = .995;
  double symmetry_factor = .5;
  vec r, ipv, r_old;
  double normalization_factor = 0, normalized_radius = 0, logvol = 0;

  // (limited) constructor: only for svp_probability, pruner.h
  explicit Pruner(int n_) : metric(0), flags(0), n(n_)
  {
    d = n / 2;
    min_pruning_coefficients.assign(d, 0.);
  }
  Pruner(double radius, double preproc, const vec &gso_r, double target_, int metric_, int flags_)
      : enumeration_radius(radius), preproc_cost(preproc), target(target_), metric(metric_), flags(flags_)
  {
    check_and_size((int)gso_r.size());
    load_basis_shape(gso_r);
  }
  // several bases at once (pruner.h:325-331): the cost is averaged over their shapes
  Pruner(double radius, double preproc, const std::vector<vec> &gso_rs, double target_, int metric_, int flags_)
      : enumeration_radius(radius), preproc_cost(preproc), target(target_), metric(metric_), flags(flags_)
  {
    if (gso_rs.empty())
      throw std::invalid_argument("no basis");
    check_and_size((int)gso_rs[0].size());
    load_basis_shapes(gso_rs);
  }
  void check_and_size(int n_)
  {
    n = n_;
    d = n / 2;
    if (flags & PR_CVP)
      symmetry_factor = 1;
    min_pruning_coefficients.assign(d, 0.);
    if (flags & PR_SINGLE)
    {
      opt_single = true;
      if (flags & PR_HALF)
        throw std::invalid_argument("flags PRUNER_HALF and PRUNER_SINGLE are mutually exclusive");
    }
    if (metric == 0)
    {
      if (target >= 1.0 || target <= 0.0)
        throw std::invalid_argument("need 0 < target < 1 with PRUNER_METRIC_PROBABILITY_OF_SHORTEST");
    }
    else if (metric == 1)
    {
      if (target <= 0.0)
        throw std::invalid_argument("need 0 < target with PRUNER_METRIC_EXPECTED_SOLUTIONS");
    }
    else
      throw std::invalid_argument("unknown metric");
  }

  // ---- pruner_util.cpp ------------------------------------------------------------------------
  void load_basis_shape(const vec &gso_r, bool reset_normalization = true)
  {
    shape_loaded = true;
    double tmp;
    logvol = 0.0;
    r.resize(n);
    ipv.resize(n);
    r_old.resize(n);
    for (int i = 0; i < n; ++i)
    {
      r[i]     = gso_r[n - 1 - i];
      r_old[i] = gso_r[i];
      logvol += std::log(r[i]);
    }
    if (reset_normalization)
    {
      normalization_factor = std::exp(logvol / ((float)(-n)));
      normalized_radius    = std::sqrt(enumeration_radius * normalization_factor);
    }
    for (int i = 0; i < n; ++i)
      r[i] *= normalization_factor;
    tmp = 1.;
    for (int i = 0; i < 2 * d; ++i)
    {
      tmp *= std::sqrt(r[i]);
      ipv[i] = 1.0 / tmp;
    }
  }
  // pruner_util.cpp:66-92: every basis is loaded in turn — normalised by the FIRST one's volume — and
  // the inverse partial volumes are averaged; r / r_old / logvol stay those of the last basis loaded
  void load_basis_shapes(const std::vector<vec> &gso_rs)
  {
    vec sum_ipv(n, 0.);
    const int count = (int)gso_rs.size();
    for (int k = 0; k < count; ++k)
    {
      if ((int)gso_rs[k].size() != n)
        throw std::runtime_error("loading several bases with different dimensions");
      load_basis_shape(gso_rs[k], k == 0);
      for (int i = 0; i < n; ++i)
        sum_ipv[i] += ipv[i];
    }
    for (int i = 0; i < n; ++i)
      ipv[i] = sum_ipv[i] / (1.0 * count);
  }
  double gaussian_heuristic() const
  {
    return std::exp(2. * std::log(fphip_pruner_ball_vol[n]) / ((float)-n)) / normalization_factor;
  }
  void save_coefficients(vec &pr, const vec &b) const
  {
    pr.resize(n);
    const int dn = (int)b.size();
    if (dn == d)
      for (int i = 0; i < d; ++i)
      {
        pr[n - 1 - 2 * i] = b[i];
        pr[n - 2 - 2 * i] = b[i];
      }
    else
      for (int i = 0; i < n; ++i)
        pr[n - 1 - i] = b[i];
    pr[0] = 1.;
  }
  void load_coefficients(vec &b, const vec &pr) const
  {
    const int dn = (int)b.size();
    const int c  = (dn == d) ? 2 : 1;
    for (int i = 0; i < dn; ++i)
      b[i] = pr[n - c * i - 1];
  }
  // pruner.h:1009-1054
  bool enforce(vec &b, const int j = 0) const
  {
    const int dn = (int)b.size();
    const int c  = (dn == d) ? 1 : 2;
    bool status  = false;
    if ((b[dn - 1] < .999) & (j != dn - 1))
    {
      status    = 1;
      b[dn - 1] = 1.;
    }
    for (int i = 0; i < dn; ++i)
    {
      status |= (b[i] > 1.0001);
      b[i] = b[i] > 1 ? 1. : b[i];
      if (i / c < d && b[i] <= min_pruning_coefficients[i / c])
        b[i] = min_pruning_coefficients[i / c];
    }
    for (int i = j; i < dn - 1; ++i)
      if (b[i + 1] < b[i])
      {
        status |= (b[i + 1] + .000001 < b[i]);
        b[i + 1] = b[i];
      }
    for (int i = std::min(j - 1, dn - 2); i >= 0; --i)
      if (b[i + 1] < b[i])
      {
        status |= (b[i + 1] + .000001 < b[i]);
        b[i] = b[i + 1];
      }
    return status;
  }

  // ---- pruner_simplex.h -------------------------------------------------------------------------
  static double eval_poly(const int ld, const vec &p, const double x)
  {
    double acc = 0.0;
    for (int i = ld; i >= 0; --i)
    {
      acc = acc * x;
      acc = acc + p[i];
    }
    return acc;
  }
  static void integrate_poly(const int ld, vec &p)
  {
    for (int i = ld; i >= 0; --i)
    {
      double tmp = i + 1.;
      p[i + 1]   = p[i] / tmp;
    }
    p[0] = 0.0;
  }
  static double relative_volume(const int rd, const vec &b)
  {
    vec P(rd + 1);
    P[0]   = 1;
    int ld = 0;
    for (int i = rd - 1; i >= 0; --i)
    {
      integrate_poly(ld, P);
      ld++;
      P[0] = -1.0 * eval_poly(ld, P, b[i] / b[rd - 1]);
    }
    double res = P[0] * fphip_pruner_factorial[rd];
    return (rd % 2) ? -res : res;
  }

  // ---- pruner_cost.cpp ----------------------------------------------------------------------------
  double single_enum_cost_evec(const vec &b, vec *detailed_cost) const
  {
    if (!shape_loaded)
      throw std::invalid_argument("no basis shape was loaded");
    if (detailed_cost)
      detailed_cost->resize(n);
    vec rv(n);
    for (int i = 0; i < d; ++i)
      rv[2 * i + 1] = relative_volume(i + 1, b);
    rv[0] = 1;
    for (int i = 1; i < d; ++i)
      rv[2 * i] = std::sqrt(rv[2 * i - 1] * rv[2 * i + 1]);
    double total                 = 0.0;
    double normalized_radius_pow = normalized_radius;
    for (int i = 0; i < 2 * d; ++i)
    {
      double tmp = normalized_radius_pow * rv[i] * fphip_pruner_ball_vol[i + 1] *
                   std::sqrt(std::pow(b[i / 2], static_cast<double>(1 + i))) * ipv[i];
      tmp *= symmetry_factor;
      if (detailed_cost)
        (*detailed_cost)[2 * d - (i + 1)] = tmp;
      total += tmp;
      normalized_radius_pow *= normalized_radius;
    }
    if (!std::isfinite(total))
      throw std::range_error("NaN or inf in single_enum_cost");
    return total;
  }
  double single_enum_cost(const vec &b, vec *detailed_cost = nullptr) const
  {
    if (b.size() == (unsigned int)d)
      return single_enum_cost_evec(b, detailed_cost);
    vec b_lower(d), b_upper(d);
    for (int i = 0; i < d; ++i)
      b_lower[i] = b[2 * i];
    const double cl = single_enum_cost_evec(b_lower, detailed_cost);
    for (int i = 0; i < d; ++i)
      b_upper[i] = b[2 * i + 1];
    const double cu = single_enum_cost_evec(b_upper, detailed_cost);
    return (cl + cu) / 2.0;
  }

  // ---- pruner_prob.cpp ----------------------------------------------------------------------------
  double svp_probability_evec(const vec &b) const
  {
    vec b_minus_db(d);
    const double dx = shell_ratio;
    for (int i = 0; i < d; ++i)
    {
      b_minus_db[i] = b[i] / (dx * dx);
      if (b_minus_db[i] > 1)
        b_minus_db[i] = 1;
    }
    const double vol  = relative_volume(d, b);
    const double dxn  = std::pow(dx, static_cast<double>(2 * d));
    const double dvol = dxn * relative_volume(d, b_minus_db) - vol;
    const double res  = dvol / (dxn - 1.);
    if (!std::isfinite(res))
      throw std::range_error("NaN or inf in svp_probability");
    return res;
  }
  double svp_probability(const vec &b) const
  {
    if (b.size() == (unsigned int)d)
      return svp_probability_evec(b);
    vec b_lower(d), b_upper(d);
    for (int i = 0; i < d; ++i)
      b_lower[i] = b[2 * i];
    const double pl = svp_probability_evec(b_lower);
    for (int i = 0; i < d; ++i)
      b_upper[i] = b[2 * i + 1];
    const double pu = svp_probability_evec(b_upper);
    return (pl + pu) / 2.0;
  }
  double expected_solutions_evec(const vec &b) const
  {
    const int j = d * 2 - 1;
    double tmp  = std::log(relative_volume(d, b));
    tmp += std::log(fphip_pruner_ball_vol[j + 1]);
    tmp += (std::log(normalized_radius) + std::log(b[j / 2]) / 2.0) * (j + 1);
    tmp += std::log(ipv[j]);
    tmp += std::log(symmetry_factor);
    tmp = std::exp(tmp);
    if (!std::isfinite(tmp))
      throw std::range_error("NaN or inf in expected_solutions");
    return tmp;
  }
  double expected_solutions(const vec &b) const
  {
    if (!shape_loaded)
      throw std::invalid_argument("no basis shape was loaded");
    if (b.size() == (unsigned int)d)
      return expected_solutions_evec(b);
    vec b_lower(d), b_upper(d);
    for (int i = 0; i < d; ++i)
      b_lower[i] = b[2 * i];
    const double pl = expected_solutions_evec(b_lower);
    for (int i = 0; i < d; ++i)
      b_upper[i] = b[2 * i + 1];
    const double pu = expected_solutions_evec(b_upper);
    return (pl + pu) / 2.0;
  }
  double measure_metric(const vec &b) const
  {
    if (metric == 0)
      return svp_probability(b);
    if (metric == 1)
      return expected_solutions(b);
    throw std::invalid_argument("unknown metric");
  }
  // the public overloads that take coefficients in the CALLER's order (pruner.h: vector<double> &pr)
  double measure_metric_pr(const vec &pr) const
  {
    vec b(n);
    load_coefficients(b, pr);
    return measure_metric(b);
  }
  double single_enum_cost_pr(const vec &pr, vec *detailed_cost) const
  {
    vec b(d);
    load_coefficients(b, pr);
    return single_enum_cost(b, detailed_cost);
  }

  double target_function(const vec &b) const
  {
    if (metric == 0)
    {
      const double probability = svp_probability(b);
      double trials            = std::log(1.0 - target) / std::log(1.0 - probability);
      if (!std::isfinite(trials))
        throw std::range_error("NaN or inf in target_function (METRIC_PROBABILITY_OF_SHORTEST)");
      trials = trials < 1.0 ? 1.0 : trials;
      return single_enum_cost(b) * trials + preproc_cost * (trials - 1.0);
    }
    if (metric == 1)
    {
      const double expected = expected_solutions(b);
      double trials         = target / expected;
      if (!std::isfinite(trials))
        throw std::range_error("NaN or inf in target_function (METRIC_EXPECTED_SOLUTION)");
      trials = trials < 1.0 ? 1.0 : trials;
      return single_enum_cost(b) * trials + preproc_cost * (trials - 1.0);
    }
    throw std::invalid_argument("unknown metric");
  }
  void target_function_gradient(const vec &b, vec &res) const
  {
    const int dn = (int)b.size();
    vec b_plus_db(dn);
    res[dn - 1] = 0.0;
    for (int i = 0; i < dn - 1; ++i)
    {
      b_plus_db = b;
      b_plus_db[i] *= (1.0 - epsilon);
      enforce(b_plus_db, i);
      const double X = target_function(b_plus_db);
      b_plus_db      = b;
      b_plus_db[i] *= (1.0 + epsilon);
      enforce(b_plus_db, i);
      const double Y = target_function(b_plus_db);
      res[i]         = (std::log(X) - std::log(Y)) / epsilon;
    }
  }

  // ---- pruner_optimize_tc.cpp ---------------------------------------------------------------------
  void greedy(vec &b)
  {
    if (!shape_loaded)
      throw std::invalid_argument("no basis shape was loaded");
    std::fill(min_pruning_coefficients.begin(), min_pruning_coefficients.end(), 0.);
    b.resize(d);
    std::fill(b.begin(), b.end(), 1.);
    double nodes;
    for (int j = 1; j < 2 * d - 1; j += 2)
    {
      const int i = j / 2;
      if (i > 1)
        b[i] = b[i - 1] > .9 ? 1 : 1.1 * b[i - 1];
      // (the second term is INTEGER arithmetic in the reference: 4 * j * (n - j) / (n * n * n))
      const double goal_factor = 1. / (3. * n) + 4 * j * (n - j) / (n * n * n);
      nodes                    = 1. + 1e10 * preproc_cost;
      while ((nodes > goal_factor * preproc_cost) & (b[i] > .001))
      {
        b[i] *= .98;
        for (int k = 0; k < i; ++k)
          b[k] = b[k] < b[i] ? b[k] : b[i];
        nodes = relative_volume((j + 1) / 2, b);
        nodes *= fphip_pruner_ball_vol[j + 1];
        nodes *= std::pow(normalized_radius * std::sqrt(b[i]), static_cast<double>(j + 1));
        nodes *= ipv[j];
        nodes *= symmetry_factor;
      }
    }
  }
  int gradient_descent_step(vec &b)
  {
    const int dn = (int)b.size();
    double cf    = target_function(b);
    const double old_cf = cf;
    vec new_b(dn), gradient(dn);
    target_function_gradient(b, gradient);
    double norm = 0.0;
    for (int i = 0; i < dn; ++i)
    {
      norm += gradient[i] * gradient[i];
      new_b[i] = b[i];
    }
    norm /= (double)dn;
    norm = std::sqrt(norm);
    if (norm <= 0.)
      return 0;
    for (int i = 0; i < dn; ++i)
      gradient[i] /= norm;
    double new_cf;
    double step = min_step;
    int j;
    for (j = 0;; ++j)
    {
      if (step > dn)
        return -1;
      for (int i = 0; i < dn; ++i)
        new_b[i] = new_b[i] + step * gradient[i];
      enforce(new_b);
      new_cf = target_function(new_b);
      if (new_cf >= cf)
        break;
      b  = new_b;
      cf = new_cf;
      step *= step_factor;
    }
    if (cf > old_cf * min_cf_decrease)
      return 0;
    return j;
  }
  int gradient_descent(vec &b)
  {
    const double old_epsilon = epsilon, old_min_step = min_step;
    int trials = 0;
    while (1)
    {
      const int ret = gradient_descent_step(b);
      if (ret == 0)
        break;
      else if (ret < 0)
      {
        epsilon  = epsilon * 0.9;
        min_step = min_step * 0.9;
        trials++;
        if (trials >= 5)
          break;
      }
      else
      {
        trials--;
        continue;
      }
    }
    epsilon  = old_epsilon;
    min_step = old_min_step;
    return 0;
  }
  // PRUNER_NELDER_MEAD (pruner_optimize_tc.cpp:581-825): one run of the downhill-simplex search from b,
  // returns whether the cost went down by the factor min_cf_decrease (the caller repeats while it does).
  // The search is ordinary Nelder-Mead (reflection 1, expansion 2, contraction 1/2, shrink 1/2); what
  // has to match the reference for identical coefficients are its particulars: the start simplex
  // (b and, per coordinate, b with that coordinate moved by 0.01 towards 1/2), `enforce` after every
  // move, the way worst / second worst / best are picked (the worst is compared with the CURRENT best
  // while the best is still being searched), and the stop rule (every dim + 1 steps: stop unless the
  // worst vertex improved by min_cf_decrease since the last check).
  int nelder_mead_step(vec &b)
  {
    const int dn = (int)b.size(), nv = dn + 1;
    std::vector<vec> vert(nv);
    vec val(nv);
    for (int i = 0; i < nv; ++i)
    {
      vert[i] = b;
      if (i < dn)
        vert[i][i] += (vert[i][i] < .5) ? 0.01 : -0.01;
      enforce(vert[i]);
      val[i] = target_function(vert[i]);
    }
    const double start_value = val[nv - 1];
    vec centre(dn), moved(dn);
    double worst_at_last_check = val[0];
    unsigned steps             = 0;
    int best = 0, worst = 0, second = 0;
    for (;;)
    {
      best = worst = second = 0;
      for (int i = 0; i < dn; ++i)
        centre[i] = vert[0][i];
      for (int i = 1; i < nv; ++i)
      {
        best  = (val[i] < val[best]) ? i : best;
        worst = (val[i] > val[best]) ? i : worst;
        for (int j = 0; j < dn; ++j)
          centre[j] += vert[i][j];
      }
      const double count = nv;
      for (int i = 0; i < dn; ++i)
        centre[i] /= count;  // (of ALL vertices, the worst included: the reference's centroid)
      if (!steps)
        worst_at_last_check = val[worst];
      second += (!worst);
      for (int i = 1; i < nv; ++i)
        second = ((val[i] > val[second]) && (i != worst)) ? i : second;
      if (enforce(centre))
        throw std::runtime_error("Concavity says that should not happen.");
      ++steps;
      if (!(steps % (unsigned)nv))
      {
        if (val[worst] > worst_at_last_check * min_cf_decrease)
          break;
        worst_at_last_check = val[worst];
      }
      for (int i = 0; i < nv; ++i)
        if ((val[i] > val[second]) && (i != worst))
          second = i;
      // reflection of the worst vertex through the centroid
      for (int i = 0; i < dn; ++i)
        moved[i] = centre[i] + 1.0 * (centre[i] - vert[worst][i]);
      enforce(moved);
      const double reflected = target_function(moved);
      if ((val[best] <= reflected) && (reflected < val[second]))
      {
        vert[worst] = moved;
        val[worst]  = reflected;
        continue;
      }
      if (reflected < val[best])
      {  // expansion: twice as far; keep the better of the two
        vec farther(dn);
        for (int i = 0; i < dn; ++i)
          farther[i] = centre[i] + 2.0 * (moved[i] - centre[i]);
        enforce(farther);
        const double expanded = target_function(farther);
        if (expanded < reflected)
        {
          vert[worst] = farther;
          val[worst]  = expanded;
        }
        else
        {
          vert[worst] = moved;
          val[worst]  = reflected;
        }
        continue;
      }
      if (!(reflected >= val[second]))
        throw std::runtime_error("Something certain is false in Nelder-Mead.");
      // contraction towards the worst vertex
      vec nearer(dn);
      for (int i = 0; i < dn; ++i)
        nearer[i] = centre[i] + 0.5 * (vert[worst][i] - centre[i]);
      enforce(nearer);
      const double contracted = target_function(nearer);
      if (contracted < val[worst])
      {
        vert[worst] = nearer;
        val[worst]  = contracted;
        continue;
      }
      // shrink every vertex towards the best one
      for (int j = 0; j < nv; ++j)
      {
        for (int i = 0; i < dn; ++i)
          vert[j][i] = vert[best][i] + 0.5 * (vert[j][i] - vert[best][i]);
        enforce(vert[j]);
        val[j] = target_function(vert[j]);
      }
    }
    b = vert[best];
    return (start_value * min_cf_decrease) > val[best];
  }
  void optimize_coefficients_evec_core(vec &pr)
  {
    vec b(d);
    load_coefficients(b, pr);
    if (flags & PR_GRADIENT)
      gradient_descent(b);
    if (flags & PR_NELDER_MEAD)
      while (nelder_mead_step(b))
      {
      }
    save_coefficients(pr, b);
  }
  void optimize_coefficients_full_core(vec &pr)
  {
    vec b(n);
    load_coefficients(b, pr);
    if (flags & PR_GRADIENT)
      gradient_descent(b);
    if (flags & PR_NELDER_MEAD)
      while (nelder_mead_step(b))
      {
      }
    save_coefficients(pr, b);
  }
  void optimize_coefficients_preparation(vec &pr)
  {
    vec b(d);
    if (flags & PR_START_FROM_INPUT)
      load_coefficients(b, pr);
    if (!(flags & PR_START_FROM_INPUT))
      greedy(b);
    if (flags & (PR_GRADIENT | PR_NELDER_MEAD))
    {
      preproc_cost *= .1;
      greedy(min_pruning_coefficients);
      if (!opt_single)
      {
        vec pr_min(n);
        save_coefficients(pr_min, min_pruning_coefficients);
        if (measure_metric(min_pruning_coefficients) > target)
        {
          std::fill(min_pruning_coefficients.begin(), min_pruning_coefficients.end(), 0.);
          optimize_coefficients_decr_prob(pr_min);
        }
        load_coefficients(min_pruning_coefficients, pr_min);
      }
      preproc_cost *= 10;
    }
    save_coefficients(pr, b);
  }
  void optimize_coefficients_local_adjust_decr_single(vec &pr)
  {
    int maxi, lasti, consecutive_fails;
    double improved_ratio, current_max = 0.0;
    double old_cf, old_cfs, new_cf, old_b;
    vec detailed_cost(n);
    vec slices(n, 10.0);
    std::vector<int> thresholds(n, 3);
    vec b(n);
    load_coefficients(b, pr);
    lasti             = -1;
    consecutive_fails = 0;
    improved_ratio    = 0.995;
    while (1)
    {
      old_cf  = target_function(b);
      old_cfs = single_enum_cost(b, &(detailed_cost));
      if (old_cfs < std::sqrt(old_cf) / 10.0)  // BALANCE_HEURISTIC_PRUNER_OPTIMIZE
        break;
      current_max = 0.0;
      maxi        = 0;
      for (int i = 0; i < n; i++)
        if ((i != (n - lasti - 1)) && (thresholds[n - i - 1] > 0))
          if (detailed_cost[i] > current_max)
          {
            current_max = detailed_cost[i];
            maxi        = i;
          }
      const int ind = n - maxi - 1;
      old_b         = b[ind];
      if (ind != 0)
        b[ind] = b[ind] - (b[ind] - b[ind - 1]) / slices[ind];
      else
        break;
      new_cf = target_function(b);
      if (new_cf >= (old_cf * improved_ratio))
      {
        b[ind] = old_b;
        lasti  = ind;
        thresholds[lasti]--;
        consecutive_fails++;
      }
      else
      {
        if (slices[ind] < 1024)
          slices[ind] = slices[ind] * 1.05;
        consecutive_fails = 0;
      }
      if (consecutive_fails > 10)
        break;
    }
    save_coefficients(pr, b);
  }
  void optimize_coefficients_local_adjust_incr_prob(vec &pr)
  {
    int trials, tours, maxi, ind;
    double old_cf, old_cf0, old_cfs, new_cf, old_b;
    double current_max;
    vec detailed_cost(n);
    vec slices(n, 10.0);
    vec b(n);
    load_coefficients(b, pr);
    old_cf0 = target_function(b);
    tours   = 0;
    while (1)
    {
      tours++;
      old_cf      = target_function(b);
      old_cfs     = single_enum_cost(b, &(detailed_cost));
      current_max = 0.0;
      maxi        = 0;
      for (int i = 0; i < n; i++)
        if (detailed_cost[i] > current_max)
        {
          current_max = detailed_cost[i];
          maxi        = i;
        }
      ind = n - maxi - 1;
      if (ind <= 1)
        break;
      if (old_cfs > std::sqrt(old_cf) / 10.0)  // BALANCE_HEURISTIC_PRUNER_OPTIMIZE
        break;
      for (int i = ind; i >= 1; --i)
      {
        if (b[i] <= b[i - 1])
          continue;
        trials = 0;
        while (1)
        {
          old_cf   = target_function(b);
          old_b    = b[i - 1];
          b[i - 1] = b[i - 1] + (b[i] - b[i - 1]) / slices[i - 1];
          new_cf   = target_function(b);
          if (new_cf >= (old_cf * 1.2))
          {
            b[i - 1] = old_b;
            break;
          }
          else
          {
            if (slices[i - 1] < 1024)
              slices[i - 1] = slices[i - 1] * 1.2;
          }
          trials++;
          if (trials >= 10)
            break;
        }
      }
      new_cf = target_function(b);
      if (new_cf > (old_cf0 * 1.1) || tours > 4)
        break;
    }
    save_coefficients(pr, b);
  }
  void optimize_coefficients_local_adjust_smooth(vec &pr)
  {
    vec b(n);
    double lr, rr;
    const double th = 1.0 / n;
    load_coefficients(b, pr);
    for (int i = 1; i < n - 1; ++i)
    {
      lr = b[i] / b[i - 1];
      rr = b[i + 1] / b[i];
      if ((rr / lr > 1.25) || (rr / lr < 0.8))
        b[i] = std::sqrt(b[i - 1] * b[i + 1]);
      if ((b[i + 1] - b[i]) > th || (b[i] - b[i - 1]) > th)
        b[i] = (b[i - 1] + b[i + 1]) / 2.0;
    }
    save_coefficients(pr, b);
  }

  // ---- pruner_optimize_tp.cpp ---------------------------------------------------------------------
  void optimize_coefficients_incr_prob(vec &pr)
  {
    const int dn = (int)pr.size();
    int tours;
    double normalized;
    double old_prob;
    vec b(dn), old_b(dn);
    vec detailed_cost(dn), weight(dn);
    bool not_changed;
    load_coefficients(b, pr);
    tours = 0;
    while (1)
    {
      if (tours > 1e4)  // OPTIMIZE_PROB_MAXSTEP
        break;
      tours++;
      old_prob = measure_metric(b);
      if (old_prob >= target)
        break;
      (void)single_enum_cost(b, &(detailed_cost));
      normalized = 0.0;
      for (int i = 0; i < dn; i++)
      {
        weight[i] = 0.0;
        for (int j = i; j < dn; j++)
          weight[i] = weight[i] + detailed_cost[j];
        weight[i] = 1.0 / weight[i];
        if (weight[i] < 1e-4)  // OPTIMIZE_PROB_MINSTEP
          weight[i] = 1e-4;
        normalized += weight[i];
      }
      for (int i = 0; i < dn; i++)
        weight[i] = weight[i] / normalized;
      for (int i = dn - 1; i >= 0; --i)
      {
        old_b[i] = b[i];
        b[i]     = b[i] + weight[i];
        if (b[i] >= 1.0)
          b[i] = 1.0;
      }
      enforce(b);
      not_changed = true;
      for (int i = dn - 1; i >= 0; --i)
        if (b[i] != old_b[i])
          not_changed = false;
      if (not_changed)
        break;
    }
    save_coefficients(pr, b);
  }
  void optimize_coefficients_decr_prob(vec &pr)
  {
    const int dn = (int)pr.size();
    int tours;
    double normalized;
    double old_prob;
    vec b(dn), old_b(dn);
    vec detailed_cost(dn), weight(dn);
    bool not_changed;
    load_coefficients(b, pr);
    tours = 0;
    while (1)
    {
      if (tours > 1e4)
        break;
      tours++;
      old_prob = measure_metric(b);
      if (old_prob <= target)
        break;
      (void)single_enum_cost(b, &(detailed_cost));
      normalized = 0.0;
      for (int i = 0; i < dn; i++)
      {
        weight[i] = 0.0;
        for (int j = i; j < dn; j++)
          weight[i] = weight[i] + detailed_cost[j];
        weight[i] = 1.0 / weight[i];
        if (weight[i] < 1e-4)
          weight[i] = 1e-4;
        normalized += weight[i];
      }
      for (int i = 0; i < dn; i++)
        weight[i] = weight[i] / normalized;
      for (int i = dn - 1; i >= 0; --i)
      {
        old_b[i] = b[i];
        b[i]     = b[i] - weight[i];
        if (b[i] < 1e-4)
          b[i] = 1e-4;
      }
      enforce(b);
      not_changed = true;
      for (int i = dn - 1; i >= 0; --i)
        if (b[i] != old_b[i])
          not_changed = false;
      if (not_changed)
        break;
    }
    save_coefficients(pr, b);
  }
  void optimize_coefficients_local_adjust_prob(vec &pr)
  {
    const int dn = (int)pr.size();
    double prob, ratio;
    vec b(dn), old_b(dn);
    bool not_changed;
    load_coefficients(b, pr);
    while (1)
    {
      prob  = measure_metric(b);
      ratio = prob / target;
      if (ratio < 1.05 && ratio > 0.95)
        break;
      if (ratio < 1)
      {
        for (int i = dn - 1; i >= 0; --i)
        {
          old_b[i] = b[i];
          b[i]     = b[i] + 1e-4;
          if (b[i] >= 1.0)
            b[i] = 1.0;
        }
      }
      else
      {
        for (int i = dn - 1; i >= 0; --i)
        {
          old_b[i] = b[i];
          b[i]     = b[i] - 1e-4;
          if (b[i] < 1e-4)
            b[i] = 1e-4;
        }
      }
      enforce(b);
      not_changed = true;
      for (int i = dn - 1; i >= 0; --i)
        if (b[i] != old_b[i])
          not_changed = false;
      if (not_changed)
        break;
    }
    save_coefficients(pr, b);
  }

  // ---- pruner_optimize.cpp ------------------------------------------------------------------------
  void optimize_coefficients_cost_vary_prob(vec &pr)
  {
    double old_c0, old_c1, new_c, min_c;
    vec b(n), best_b(n);
    optimize_coefficients_preparation(pr);
    optimize_coefficients_evec_core(pr);
    load_coefficients(b, pr);
    best_b = b;
    old_c0 = target_function(b);
    min_c  = old_c0;
    if (!(flags & PR_HALF))
    {
      int tours = 0;
      while (1)
      {
        tours++;
        load_coefficients(b, pr);
        old_c0 = target_function(b);
        optimize_coefficients_local_adjust_decr_single(pr);
        optimize_coefficients_local_adjust_incr_prob(pr);
        optimize_coefficients_local_adjust_smooth(pr);
        load_coefficients(b, pr);
        old_c1 = target_function(b);
        if (old_c1 < min_c)
        {
          min_c  = old_c1;
          best_b = b;
        }
        optimize_coefficients_full_core(pr);
        load_coefficients(b, pr);
        new_c = target_function(b);
        if (new_c < min_c)
        {
          min_c  = new_c;
          best_b = b;
        }
        if (new_c / old_c0 > 0.995 and tours > 3)  // NUM_OPTIMIZATION_TOURS
          break;
      }
      save_coefficients(pr, best_b);
    }
    else
      save_coefficients(pr, b);
  }
  void optimize_coefficients_cost_fixed_prob(vec &pr)
  {
    vec b(n);
    double prob;
    optimize_coefficients_preparation(pr);
    optimize_coefficients_evec_core(pr);
    optimize_coefficients_local_adjust_smooth(pr);
    optimize_coefficients_full_core(pr);
    optimize_coefficients_local_adjust_smooth(pr);
    load_coefficients(b, pr);
    prob = measure_metric(b);
    if (prob <= target)
      optimize_coefficients_incr_prob(pr);
    else
      optimize_coefficients_decr_prob(pr);
    optimize_coefficients_local_adjust_smooth(pr);
    optimize_coefficients_local_adjust_prob(pr);
  }
  void optimize_coefficients(vec &pr)
  {
    if (opt_single)
      optimize_coefficients_cost_fixed_prob(pr);
    else
      optimize_coefficients_cost_vary_prob(pr);
  }
};
}  // namespace

// ---- C ABI ------------------------------------------------------------------------------------------
extern "C" int fphip_pruner_prune(int n, const double *gso_r, double enumeration_radius, double preproc_cost,
                                  double target, int metric, int flags, double *coefficients,
                                  double *expectation, double *gh_factor, double *detailed_cost)
{
  if (n < 2 || n >= FPHIP_PRUNER_TABLE_N || !gso_r || !coefficients)
    return FPHIP_ERROR;
  if (flags & 0x10 /* PRUNER_VERBOSE */)
    return FPHIP_UNSUPPORTED;
  try
  {
    vec r(gso_r, gso_r + n);
    Pruner pruner(enumeration_radius, preproc_cost, r, target, metric, flags);
    vec pr;
    if (flags & PR_START_FROM_INPUT)
      pr.assign(coefficients, coefficients + n);
    pruner.optimize_coefficients(pr);
    vec dc;
    pruner.single_enum_cost_pr(pr, &dc);
    for (int i = 0; i < n; ++i)
      coefficients[i] = pr[i];
    if (detailed_cost)
      for (int i = 0; i < n; ++i)
        detailed_cost[i] = i < (int)dc.size() ? dc[i] : 0.0;
    if (gh_factor)
      *gh_factor = enumeration_radius / pruner.gaussian_heuristic();
    if (expectation)
      *expectation = pruner.measure_metric_pr(pr);
    return FPHIP_OK;
  }
  catch (const std::exception &)
  {
    return FPHIP_ERROR;  // (the reference throws on NaN / inf: "using a higher precision sometimes helps")
  }
}

extern "C" int fphip_pruner_prune_multi(int n, int count, const double *gso_rs, double enumeration_radius,
                                        double preproc_cost, double target, int metric, int flags,
                                        double *coefficients, double *expectation, double *gh_factor,
                                        double *detailed_cost)
{
  if (n < 2 || n >= FPHIP_PRUNER_TABLE_N || count < 1 || !gso_rs || !coefficients)
    return FPHIP_ERROR;
  if (flags & 0x10 /* PRUNER_VERBOSE */)
    return FPHIP_UNSUPPORTED;
  try
  {
    std::vector<vec> rs;
    for (int c = 0; c < count; ++c)
      rs.emplace_back(gso_rs + (size_t)c * n, gso_rs + (size_t)(c + 1) * n);
    Pruner pruner(enumeration_radius, preproc_cost, rs, target, metric, flags);
    vec pr;
    if (flags & PR_START_FROM_INPUT)
      pr.assign(coefficients, coefficients + n);
    pruner.optimize_coefficients(pr);
    vec dc;
    pruner.single_enum_cost_pr(pr, &dc);
    for (int i = 0; i < n; ++i)
      coefficients[i] = pr[i];
    if (detailed_cost)
      for (int i = 0; i < n; ++i)
        detailed_cost[i] = i < (int)dc.size() ? dc[i] : 0.0;
    if (gh_factor)
      *gh_factor = enumeration_radius / pruner.gaussian_heuristic();
    if (expectation)
      *expectation = pruner.measure_metric_pr(pr);
    return FPHIP_OK;
  }
  catch (const std::exception &)
  {
    return FPHIP_ERROR;
  }
}

extern "C" int fphip_pruner_svp_probability(int n, const double *pr, double *probability)
{
  if (n < 2 || n >= FPHIP_PRUNER_TABLE_N || !pr || !probability)
    return FPHIP_ERROR;
  try
  {
    Pruner pru(n);
    *probability = pru.measure_metric_pr(vec(pr, pr + n));
    return FPHIP_OK;
  }
  catch (const std::exception &)
  {
    return FPHIP_ERROR;
  }
}

extern "C" int fphip_pruner_enum_cost(int n, const double *gso_r, double enumeration_radius, const double *pr,
                                      int metric, double *cost, double *metric_value, double *detailed_cost)
{
  if (n < 2 || n >= FPHIP_PRUNER_TABLE_N || !gso_r || !pr)
    return FPHIP_ERROR;
  try
  {
    vec r(gso_r, gso_r + n);
    Pruner pruner(enumeration_radius, 0.0, r, metric == 0 ? 0.5 : 1.0, metric, 0);
    vec p(pr, pr + n), dc;
    const double c = pruner.single_enum_cost_pr(p, &dc);
    if (cost)
      *cost = c;
    if (detailed_cost)
      for (int i = 0; i < n; ++i)
        detailed_cost[i] = i < (int)dc.size() ? dc[i] : 0.0;
    if (metric_value)
      *metric_value = pruner.measure_metric_pr(p);
    return FPHIP_OK;
  }
  catch (const std::exception &)
  {
    return FPHIP_ERROR;
  }
}
