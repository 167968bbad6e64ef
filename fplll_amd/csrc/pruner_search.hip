// pruner_search.hip — the pruner of SURVEY 8(f) N2, built around BATCHES of candidates.
//
// What it computes is fplll's prune<FP_NR<double>>() (pruner/pruner.h:187-243): pruning coefficients
// that minimise  cost(one enumeration) x trials + preproc_cost x (trials - 1)  for a block's
// Gram-Schmidt profile, with the reference's model (volumes of cylinder intersections by symbolic
// integration of even simplices, Gama-Nguyen-Regev) and the reference's searches (greedy start,
// gradient descent with a numerical gradient, the Nelder-Mead simplex search, the local tuning passes).
// The coefficients come out of thousands of comparisons between cost values, so "the reference's
// result" means the reference's doubles at every comparison: each VALUE below is computed by the same
// sequence of IEEE operations as there (host libm for log / exp / pow / sqrt), and the tests check
// the output bit for bit (tests/test_pruner_cpu.py, tests/test_pruner_gpu.py).
//
// How it computes it is not the reference's: there every cost value is one call chain that ends in
// an O(n^3) scalar recurrence.  Here the searches are written against   score(batch of candidates):
//   * a numerical gradient is ONE batch of 2 (n - 1) + 1 perturbed vectors,
//   * a line search / the greedy shrink loop / a Nelder-Mead move evaluates its next `lookahead`
//     candidates at once and then consumes them in order (the candidates of these loops do not
//     depend on the scores, only the STOPPING point does),
//   * a candidate is scored once for everything that is asked about it (cost, per-level cost,
//     metric, target value; the volume V_half is shared between the cost and the metric),
// and a batch goes to a VolumeEngine (pruner_engine.h) as a matrix of bound vectors plus a job list:
// with a context that is the device kernel of pruner_volume.hip — one lane per (vector, k) —, the
// O(n) rest (square roots, powers, logarithms: the host libm's roundings) is finished here.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "../../include/fplll_hip.h"
#include "pruner_engine.h"
#include "pruner_tables.h"

namespace fphip_pruner
{
const double *factorial_table() { return fphip_pruner_factorial; }

namespace
{
typedef std::vector<double> dvec;

// the reference reports these conditions by throwing (NaN / inf in a cost value, a bad target):
// the C ABI turns them into FPHIP_ERROR
struct Failure : std::runtime_error
{
  explicit Failure(const char *what) : std::runtime_error(what) {}
};

enum
{
  OPT_CVP        = 0x1,
  OPT_FROM_INPUT = 0x2,
  OPT_GRADIENT   = 0x4,
  OPT_NELDER     = 0x8,
  OPT_VERBOSE    = 0x10,
  OPT_HALF       = 0x20,
  OPT_SINGLE     = 0x40
};
enum
{
  ASK_COST   = 1,  // expected nodes of one enumeration
  ASK_LEVELS = 2,  // ... per level
  ASK_METRIC = 4,  // success probability / expected number of solutions
  ASK_VALUE  = 8   // the objective: cost x trials + preproc x (trials - 1)
};

struct Score
{
  bool ok = true;     // false: a value came out NaN / inf (the reference throws when it meets it)
  double cost = 0, metric = 0, value = 0;
  dvec levels;
};

// ---------------------------------------------------------------------------------------------
// The block and its cost model (pruner.h:297-347 constructor, pruner_util.cpp:27-104 shape,
// pruner_cost.cpp / pruner_prob.cpp formulas)
// ---------------------------------------------------------------------------------------------
class BlockModel
{
public:
  int n = 0, half = 0;
  int metric = 0, flags = 0;
  double radius = 0, preproc = 0, target = 0;
  double sym = .5;     // SVP counts a vector and its negative once
  double scale = 0;    // det^(-2/n): the profile is normalised to volume 1
  double nrad = 0;     // the radius in that normalisation (not squared)
  dvec ipv;            // 1 / volume of the first i+1 normalised lengths
  dvec floor_;         // lower bounds of the half coefficients (enforce)
  VolumeEngine *eng = nullptr;
  // search parameters (pruner.h:611-617); the descent driver shrinks the first two temporarily
  double eps = std::pow(2., -7), min_step = std::pow(2., -6);
  const double keep_ratio = .995, step_growth = std::pow(2, .5), shell = .995;

  BlockModel(int n_, VolumeEngine *e) : n(n_), half(n_ / 2), floor_(n_ / 2, 0.), eng(e) {}

  void configure(double radius_, double preproc_, double target_, int metric_, int flags_)
  {
    radius = radius_, preproc = preproc_, target = target_, metric = metric_, flags = flags_;
    if (flags & OPT_CVP)
      sym = 1;
    if ((flags & OPT_SINGLE) && (flags & OPT_HALF))
      throw Failure("PRUNER_HALF and PRUNER_SINGLE exclude each other");
    if (metric == 0 ? !(target < 1.0 && target > 0.0) : metric == 1 ? !(target > 0.0) : true)
      throw Failure("bad target / metric");
  }

  // one profile; `first` fixes the normalisation (the overload for several bases keeps the first's)
  void load_profile(const double *gso_r, bool first)
  {
    dvec r(n);
    double logvol = 0.0;
    for (int i = 0; i < n; ++i)
    {
      r[i] = gso_r[n - 1 - i];
      logvol += std::log(r[i]);
    }
    if (first)
    {
      scale = std::exp(logvol / (double)((float)(-n)));
      nrad  = std::sqrt(radius * scale);
    }
    ipv.assign(n, 0.0);
    double run = 1.;
    for (int i = 0; i < 2 * half; ++i)
    {
      run *= std::sqrt(r[i] * scale);
      ipv[i] = 1.0 / run;
    }
  }
  void load_profiles(const double *gso_rs, int count)
  {
    dvec sum(n, 0.);
    for (int c = 0; c < count; ++c)
    {
      load_profile(gso_rs + (size_t)c * n, c == 0);
      for (int i = 0; i < n; ++i)
        sum[i] += ipv[i];
    }
    for (int i = 0; i < n; ++i)
      ipv[i] = sum[i] / (1.0 * count);
  }
  double gaussian_heuristic() const
  {
    return std::exp(2. * std::log(fphip_pruner_ball_vol[n]) / (double)((float)-n)) / scale;
  }

  // ---- coefficient vectors: external order (pr[0] = 1 at the full-length end) <-> internal ------
  void from_external(dvec &b, const dvec &pr) const
  {
    const int stride = ((int)b.size() == half) ? 2 : 1;
    for (size_t i = 0; i < b.size(); ++i)
      b[i] = pr[n - stride * (int)i - 1];
  }
  void to_external(dvec &pr, const dvec &b) const
  {
    pr.resize(n);
    if ((int)b.size() == half)
      for (int i = 0; i < half; ++i)
        pr[n - 1 - 2 * i] = pr[n - 2 - 2 * i] = b[i];
    else
      for (int i = 0; i < n; ++i)
        pr[n - 1 - i] = b[i];
    pr[0] = 1.;
  }

  // the feasible set: non-decreasing, within [floor, 1], last one 1; coordinate `keep` is the one a
  // caller has just moved — the repair runs away from it in both directions (pruner.h:1012-1054)
  bool enforce(dvec &b, int keep = 0) const
  {
    const int dn  = (int)b.size();
    const int per = (dn == half) ? 1 : 2;
    bool moved    = false;
    if ((b[dn - 1] < .999) & (keep != dn - 1))
    {
      moved     = true;
      b[dn - 1] = 1.;
    }
    for (int i = 0; i < dn; ++i)
    {
      moved |= (b[i] > 1.0001);
      if (b[i] > 1)
        b[i] = 1.;
      if (i / per < half && b[i] <= floor_[i / per])
        b[i] = floor_[i / per];
    }
    for (int i = keep; i < dn - 1; ++i)
      if (b[i + 1] < b[i])
      {
        moved |= (b[i + 1] + .000001 < b[i]);
        b[i + 1] = b[i];
      }
    for (int i = std::min(keep - 1, dn - 2); i >= 0; --i)
      if (b[i + 1] < b[i])
      {
        moved |= (b[i + 1] + .000001 < b[i]);
        b[i] = b[i + 1];
      }
    return moved;
  }

  // ---- scoring a batch --------------------------------------------------------------------------
  // A candidate of length `half` is one bound vector; one of length n is scored as the mean of its
  // even-indexed and odd-indexed halves (pruner_cost.cpp:77-118, pruner_prob.cpp:34-76).
  void score(const std::vector<dvec> &cands, unsigned ask, std::vector<Score> &out)
  {
    if (ask & ASK_VALUE)
      ask |= ASK_COST | ASK_METRIC;
    if (ask & ASK_LEVELS)
      ask |= ASK_COST;
    const int m        = half;
    const bool shrunk  = (ask & ASK_METRIC) && metric == 0;
    const int per_half = shrunk ? 2 : 1;  // rows per half: the bounds, and the bounds of the thinner shell
    rows.clear();
    jobs.clear();
    std::vector<int> first_row(cands.size());
    for (size_t c = 0; c < cands.size(); ++c)
    {
      const dvec &b   = cands[c];
      const int parts = ((int)b.size() == half) ? 1 : 2;
      first_row[c]    = (int)(rows.size() / (size_t)m);
      for (int p = 0; p < parts; ++p)
      {
        const int row = (int)(rows.size() / (size_t)m);
        for (int i = 0; i < m; ++i)
          rows.push_back(parts == 1 ? b[i] : b[2 * i + p]);
        if (ask & ASK_COST)
          for (int k = 1; k <= m; ++k)
            jobs.push_back(VolumeJob{row, k});
        else
          jobs.push_back(VolumeJob{row, m});
        if (shrunk)
        {
          const double dx = shell;
          for (int i = 0; i < m; ++i)
          {
            double v = rows[(size_t)row * m + i] / (dx * dx);
            if (v > 1)
              v = 1;
            rows.push_back(v);
          }
          jobs.push_back(VolumeJob{row + 1, m});
        }
      }
    }
    vols.resize(jobs.size());
    if (!eng->run(rows.data(), (int)(rows.size() / (size_t)m), m, jobs.data(), (int)jobs.size(), vols.data()))
      throw Failure(eng->error());
    out.assign(cands.size(), Score());
    const int jobs_per_half = ((ask & ASK_COST) ? m : 1) + (shrunk ? 1 : 0);
    size_t jpos             = 0;
    for (size_t c = 0; c < cands.size(); ++c)
    {
      const dvec &b   = cands[c];
      const int parts = ((int)b.size() == half) ? 1 : 2;
      Score &s        = out[c];
      double cost_p[2] = {0, 0}, met_p[2] = {0, 0};
      for (int p = 0; p < parts; ++p, jpos += (size_t)jobs_per_half)
      {
        const double *h  = rows.data() + (size_t)(first_row[c] + p * per_half) * m;
        const double *v  = vols.data() + jpos;                      // V_1 .. V_m (or V_m alone)
        const double vm  = (ask & ASK_COST) ? v[m - 1] : v[0];      // V_m of the bounds
        const double vsh = shrunk ? v[jobs_per_half - 1] : 0.0;     // V_m of the thinner shell
        if (ask & ASK_COST)
          s.ok &= half_cost(h, v, (ask & ASK_LEVELS) ? &s.levels : nullptr, cost_p[p]);
        if (ask & ASK_METRIC)
          s.ok &= metric == 0 ? half_probability(vm, vsh, met_p[p]) : half_solutions(h, vm, met_p[p]);
      }
      s.cost   = parts == 1 ? cost_p[0] : (cost_p[0] + cost_p[1]) / 2.0;
      s.metric = parts == 1 ? met_p[0] : (met_p[0] + met_p[1]) / 2.0;
      if ((ask & ASK_VALUE) && s.ok)
      {
        double trials = metric == 0 ? std::log(1.0 - target) / std::log(1.0 - s.metric) : target / s.metric;
        if (!std::isfinite(trials))
          s.ok = false;
        if (trials < 1.0)
          trials = 1.0;
        s.value = s.cost * trials + preproc * (trials - 1.0);
      }
    }
  }
  // one candidate, the reference's failure behaviour (it throws on the first non-finite value)
  Score score1(const dvec &b, unsigned ask)
  {
    one[0] = b;
    score(one, ask, one_out);
    if (!one_out[0].ok)
      throw Failure("NaN or inf in the cost model");
    return one_out[0];
  }
  double value(const dvec &b) { return score1(b, ASK_VALUE).value; }

private:
  dvec rows, vols;
  std::vector<VolumeJob> jobs;
  std::vector<dvec> one = std::vector<dvec>(1);
  std::vector<Score> one_out;

  // sqrt(pow(x, i + 1)) of the cost model.  The candidates of a batch differ from one another in a coordinate or
  // two (a gradient perturbs one), so level i sees the same x again and again: the last two (x, value) pairs of
  // every level are kept — libm's own doubles, returned instead of computed a second time
  struct PowMemo
  {
    unsigned long long key[2] = {~0ull, ~0ull};  // (the bit pattern of a NaN no bound has)
    double val[2]             = {0.0, 0.0};
    int last                  = 0;
  };
  mutable std::vector<PowMemo> pow_memo;
  double root_of_power(double x, int i) const
  {
    if ((int)pow_memo.size() < 2 * half)
      pow_memo.resize((size_t)2 * half);
    PowMemo &M = pow_memo[i];
    unsigned long long kx;
    std::memcpy(&kx, &x, sizeof kx);
    if (M.key[M.last] == kx)
      return M.val[M.last];
    const int other = M.last ^ 1;
    if (M.key[other] == kx)
    {
      M.last = other;
      return M.val[other];
    }
    const double v = std::sqrt(std::pow(x, (double)(1 + i)));
    M.key[other]   = kx;
    M.val[other]   = v;
    M.last         = other;
    return v;
  }

  // nodes per level and in all for one bound vector h, from V_1 .. V_m (pruner_cost.cpp:11-72): the
  // odd-dimensional volumes are the geometric means of their neighbours
  bool half_cost(const double *h, const double *v, dvec *levels, double &total) const
  {
    const int m = half;
    if (levels)
      levels->resize(n);
    double rpow = nrad;
    total       = 0.0;
    for (int i = 0; i < 2 * m; ++i)
    {
      double rv;
      if (i & 1)
        rv = v[i / 2];
      else
        rv = i == 0 ? 1.0 : std::sqrt(v[i / 2 - 1] * v[i / 2]);
      double t = rpow * rv * fphip_pruner_ball_vol[i + 1] * root_of_power(h[i / 2], i) * ipv[i];
      t *= sym;
      if (levels)
        (*levels)[2 * m - (i + 1)] = t;
      total += t;
      rpow *= nrad;
    }
    return std::isfinite(total);
  }
  // probability that the shortest vector survives: the mass of a thin shell (pruner_prob.cpp:5-28)
  bool half_probability(double vol, double vol_shrunk, double &p) const
  {
    const double dxn  = std::pow(shell, (double)(2 * half));
    const double dvol = dxn * vol_shrunk - vol;
    p                 = dvol / (dxn - 1.);
    return std::isfinite(p);
  }
  // expected number of solutions inside the radius (pruner_prob.cpp:78-94)
  bool half_solutions(const double *h, double vol, double &e) const
  {
    const int j = 2 * half - 1;
    double t    = std::log(vol);
    t += std::log(fphip_pruner_ball_vol[j + 1]);
    t += (std::log(nrad) + std::log(h[j / 2]) / 2.0) * (j + 1);
    t += std::log(ipv[j]);
    t += std::log(sym);
    e = std::exp(t);
    return std::isfinite(e);
  }
};

// ---------------------------------------------------------------------------------------------
// The searches
// ---------------------------------------------------------------------------------------------
class Search
{
public:
  explicit Search(BlockModel &model) : M(model) {}

  // prune(): external coefficients in (OPT_FROM_INPUT) and out
  void run(dvec &pr)
  {
    if (M.flags & OPT_SINGLE)
      fixed_probability(pr);
    else
      free_probability(pr);
  }

private:
  BlockModel &M;

  // ---- greedy start (pruner_optimize_tc.cpp:404-452): level by level, shrink the bound of the level
  // (and whatever lies below it) by 2 % until the level holds at most its share of the preprocessing
  // cost.  The shrinking sequence does not depend on the node counts — only where it stops does — so
  // the next `lookahead` candidates of a level are scored together.
  void greedy(dvec &b)
  {
    const int n = M.n, m = M.half;
    std::fill(M.floor_.begin(), M.floor_.end(), 0.);
    b.assign(m, 1.);  // (b may BE the floor vector: the reference's second call aliases them)
    const int ahead = std::max(1, M.eng->lookahead());
    dvec rows, vols;
    std::vector<VolumeJob> jobs;
    for (int j = 1; j < 2 * m - 1; j += 2)
    {
      const int i = j / 2;
      if (i > 1)
        b[i] = b[i - 1] > .9 ? 1 : 1.1 * b[i - 1];
      const double share = 1. / (3. * n) + (double)(4 * j * (n - j) / (n * n * n));  // (integer quotient)
      double nodes       = 1. + 1e10 * M.preproc;
      while ((nodes > share * M.preproc) & (b[i] > .001))
      {
        // candidates t = 1 .. T of this level: bound b[i] .98^t, the lower levels capped by it
        int T = 0;
        rows.clear();
        jobs.clear();
        dvec cur(b.begin(), b.begin() + i + 1);
        double bi = b[i];
        while (T < ahead && (T == 0 || bi > .001))
        {
          bi *= .98;
          cur[i] = bi;
          for (int k = 0; k < i; ++k)
            cur[k] = cur[k] < bi ? cur[k] : bi;
          rows.insert(rows.end(), cur.begin(), cur.end());
          rows.resize(rows.size() + (size_t)(m - (i + 1)), 1.0);  // (unused tail of the row)
          jobs.push_back(VolumeJob{T, i + 1});
          ++T;
        }
        vols.resize(T);
        if (!M.eng->run(rows.data(), T, m, jobs.data(), T, vols.data()))
          throw Failure(M.eng->error());
        for (int t = 0; t < T; ++t)
        {
          for (int k = 0; k <= i; ++k)
            b[k] = rows[(size_t)t * m + k];
          nodes = vols[t];
          nodes *= fphip_pruner_ball_vol[j + 1];
          nodes *= std::pow(M.nrad * std::sqrt(b[i]), (double)(j + 1));
          nodes *= M.ipv[j];
          nodes *= M.sym;
          if (!((nodes > share * M.preproc) & (b[i] > .001)))
            break;
        }
      }
    }
  }

  // ---- numerical gradient of log(objective) (pruner_cost.cpp:120-140) as ONE batch ----------------
  void gradient(const dvec &b, dvec &g, double &value_at_b)
  {
    const int dn = (int)b.size();
    batch.clear();
    batch.push_back(b);
    for (int i = 0; i < dn - 1; ++i)
      for (int side = 0; side < 2; ++side)
      {
        dvec p = b;
        p[i] *= side == 0 ? (1.0 - M.eps) : (1.0 + M.eps);
        M.enforce(p, i);
        batch.push_back(p);
      }
    M.score(batch, ASK_VALUE, scores);
    for (const Score &s : scores)
      if (!s.ok)
        throw Failure("NaN or inf in the objective");
    value_at_b = scores[0].value;
    g.assign(dn, 0.0);  // (the last coordinate is pinned to 1: no gradient there)
    for (int i = 0; i < dn - 1; ++i)
      g[i] = (std::log(scores[1 + 2 * i].value) - std::log(scores[2 + 2 * i].value)) / M.eps;
  }

  // ---- one descent step (pruner_optimize_tc.cpp:500-579): normalised gradient, then a line search
  // with growing steps until the objective stops falling.  Returns the number of accepted mini-steps,
  // 0 when the gain is below 0.5 %, -1 when the step outgrew the dimension.
  int descent_step(dvec &b)
  {
    const int dn = (int)b.size();
    dvec g;
    double cf;
    gradient(b, g, cf);
    const double start = cf;
    double norm        = 0.0;
    for (int i = 0; i < dn; ++i)
      norm += g[i] * g[i];
    norm /= (double)dn;
    norm = std::sqrt(norm);
    if (norm <= 0.)
      return 0;
    for (int i = 0; i < dn; ++i)
      g[i] /= norm;
    const int ahead = std::max(1, M.eng->lookahead() / 4);
    dvec probe      = b;
    double step     = M.min_step;
    int accepted    = 0;
    for (;;)
    {
      // the next candidates of the walk: each is the previous one moved and repaired
      batch.clear();
      dvec walker  = probe;
      double s     = step;
      bool overrun = false;
      for (int t = 0; t < ahead; ++t)
      {
        if (s > dn)
        {
          overrun = true;
          break;
        }
        for (int i = 0; i < dn; ++i)
          walker[i] = walker[i] + s * g[i];
        M.enforce(walker);
        batch.push_back(walker);
        s *= M.step_growth;
      }
      if (batch.empty())
        return -1;
      M.score(batch, ASK_VALUE, scores);
      bool stopped = false;
      for (size_t t = 0; t < batch.size(); ++t)
      {
        if (!scores[t].ok)
          throw Failure("NaN or inf in the objective");
        if (scores[t].value >= cf)
        {
          stopped = true;
          break;
        }
        b     = batch[t];
        probe = batch[t];
        cf    = scores[t].value;
        step *= M.step_growth;
        ++accepted;
      }
      if (stopped)
        break;
      if (overrun)
        return -1;
    }
    if (cf > start * M.keep_ratio)
      return 0;
    return accepted;
  }

  // the driver (pruner_optimize_tc.cpp:457-495): finer differences after a failed step, at most five
  void descend(dvec &b)
  {
    const double eps0 = M.eps, step0 = M.min_step;
    int failures = 0;
    for (;;)
    {
      const int r = descent_step(b);
      if (r == 0)
        break;
      if (r < 0)
      {
        M.eps      = M.eps * 0.9;
        M.min_step = M.min_step * 0.9;
        if (++failures >= 5)
          break;
      }
      else
        --failures;
    }
    M.eps      = eps0;
    M.min_step = step0;
  }

  // ---- the downhill simplex (pruner_optimize_tc.cpp:581-825).  What makes the result the
  // reference's are its particulars: vertices start 0.01 towards 1/2, every point is repaired before it
  // is scored, the "worst" vertex is picked against the best found SO FAR in the same scan, the centroid
  // runs over ALL vertices, progress is checked every dn + 1 moves.  A move's three possible points
  // (reflection, expansion, contraction) depend on the simplex only, so with a device engine they are
  // scored together.  Returns whether the pass gained at least 0.5 %.
  bool simplex_pass(dvec &b)
  {
    const int dn = (int)b.size(), L = dn + 1;
    std::vector<dvec> V(L, b);
    for (int i = 0; i < L; ++i)
    {
      if (i < dn)
        V[i][i] += (V[i][i] < .5) ? 0.01 : -0.01;
      M.enforce(V[i]);
    }
    dvec f(L);
    score_all(V, f);
    const double entry = f[L - 1];
    double worst_seen  = f[0];
    const bool eager   = M.eng->lookahead() > 1;
    unsigned moves     = 0;
    int lo = 0, hi = 0, hi2 = 0;
    dvec centre(dn), refl(dn), expd(dn), cont(dn);
    for (;;)
    {
      lo = hi = hi2 = 0;
      for (int i = 0; i < dn; ++i)
        centre[i] = V[0][i];
      for (int i = 1; i < L; ++i)
      {
        lo = (f[i] < f[lo]) ? i : lo;
        hi = (f[i] > f[lo]) ? i : hi;
        for (int j = 0; j < dn; ++j)
          centre[j] += V[i][j];
      }
      const double count = L;
      for (int i = 0; i < dn; ++i)
        centre[i] /= count;
      if (!moves)
        worst_seen = f[hi];
      hi2 += (!hi);
      for (int i = 1; i < L; ++i)
        hi2 = ((f[i] > f[hi2]) && (i != hi)) ? i : hi2;
      if (M.enforce(centre))
        throw Failure("the centroid of feasible points is infeasible");
      ++moves;
      if (!(moves % L))
      {
        if (f[hi] > worst_seen * M.keep_ratio)
          break;
        worst_seen = f[hi];
      }
      for (int i = 0; i < L; ++i)
        if ((f[i] > f[hi2]) && (i != hi))
          hi2 = i;
      for (int i = 0; i < dn; ++i)
        refl[i] = centre[i] + 1 * (centre[i] - V[hi][i]);
      M.enforce(refl);
      double fr, fe = 0, fc = 0;
      bool have_rest = false;
      if (eager)
      {
        for (int i = 0; i < dn; ++i)
          expd[i] = centre[i] + 2 * (refl[i] - centre[i]);
        M.enforce(expd);
        for (int i = 0; i < dn; ++i)
          cont[i] = centre[i] + 0.5 * (V[hi][i] - centre[i]);
        M.enforce(cont);
        batch.assign({refl, expd, cont});
        M.score(batch, ASK_VALUE, scores);
        if (!scores[0].ok)
          throw Failure("NaN or inf in the objective");
        fr = scores[0].value, fe = scores[1].value, fc = scores[2].value;
        have_rest = true;
      }
      else
        fr = M.value(refl);
      if ((f[lo] <= fr) && (fr < f[hi2]))
      {
        V[hi] = refl, f[hi] = fr;
        continue;
      }
      if (fr < f[lo])
      {
        if (!have_rest)
        {
          for (int i = 0; i < dn; ++i)
            expd[i] = centre[i] + 2 * (refl[i] - centre[i]);
          M.enforce(expd);
          fe = M.value(expd);
        }
        else if (!scores[1].ok)
          throw Failure("NaN or inf in the objective");
        if (fe < fr)
          V[hi] = expd, f[hi] = fe;
        else
          V[hi] = refl, f[hi] = fr;
        continue;
      }
      if (!(fr >= f[hi2]))
        throw Failure("simplex ordering violated");
      if (!have_rest)
      {
        for (int i = 0; i < dn; ++i)
          cont[i] = centre[i] + 0.5 * (V[hi][i] - centre[i]);
        M.enforce(cont);
        fc = M.value(cont);
      }
      else if (!scores[2].ok)
        throw Failure("NaN or inf in the objective");
      if (fc < f[hi])
      {
        V[hi] = cont, f[hi] = fc;
        continue;
      }
      // nothing helped: pull every vertex half way towards the best one
      for (int j = 0; j < L; ++j)
      {
        for (int i = 0; i < dn; ++i)
          V[j][i] = V[lo][i] + 0.5 * (V[j][i] - V[lo][i]);
        M.enforce(V[j]);
      }
      score_all(V, f);
    }
    b = V[lo];
    return (entry * M.keep_ratio) > f[lo];
  }
  void score_all(const std::vector<dvec> &V, dvec &f)
  {
    M.score(V, ASK_VALUE, scores);
    for (size_t i = 0; i < V.size(); ++i)
    {
      if (!scores[i].ok)
        throw Failure("NaN or inf in the objective");
      f[i] = scores[i].value;
    }
  }

  // ---- the stages of prune() ---------------------------------------------------------------------
  // start vector (greedy or the caller's) and the floor of enforce() (pruner_optimize_tc.cpp:11-60)
  void prepare(dvec &pr)
  {
    dvec b(M.half);
    if (M.flags & OPT_FROM_INPUT)
      M.from_external(b, pr);
    else
      greedy(b);
    if (M.flags & (OPT_GRADIENT | OPT_NELDER))
    {
      M.preproc *= .1;
      greedy(M.floor_);
      if (!(M.flags & OPT_SINGLE))
      {
        // a floor that already exceeds the target would make the target unreachable: lower it
        dvec pr_floor(M.n);
        M.to_external(pr_floor, M.floor_);
        if (M.score1(M.floor_, ASK_METRIC).metric > M.target)
        {
          std::fill(M.floor_.begin(), M.floor_.end(), 0.);
          lower_metric(pr_floor);
        }
        M.from_external(M.floor_, pr_floor);
      }
      M.preproc *= 10;
    }
    M.to_external(pr, b);
  }
  void refine(dvec &pr, bool full)
  {
    dvec b(full ? M.n : M.half);
    M.from_external(b, pr);
    if (M.flags & OPT_GRADIENT)
      descend(b);
    if (M.flags & OPT_NELDER)
      while (simplex_pass(b))
      {
      }
    M.to_external(pr, b);
  }

  // pruner_optimize.cpp:8-101
  void free_probability(dvec &pr)
  {
    prepare(pr);
    refine(pr, false);
    dvec b(M.n), best;
    M.from_external(b, pr);
    best        = b;
    double c0   = M.value(b);
    double cmin = c0;
    if (M.flags & OPT_HALF)
    {
      M.to_external(pr, b);
      return;
    }
    for (int rounds = 1;; ++rounds)
    {
      M.from_external(b, pr);
      c0 = M.value(b);
      trim_bottleneck(pr);
      raise_cheap_levels(pr);
      smooth(pr);
      M.from_external(b, pr);
      const double c1 = M.value(b);
      if (c1 < cmin)
        cmin = c1, best = b;
      refine(pr, true);
      M.from_external(b, pr);
      const double c2 = M.value(b);
      if (c2 < cmin)
        cmin = c2, best = b;
      if (c2 / c0 > 0.995 && rounds > 3)
        break;
    }
    M.to_external(pr, best);
  }
  // pruner_optimize.cpp:103-146
  void fixed_probability(dvec &pr)
  {
    prepare(pr);
    refine(pr, false);
    smooth(pr);
    refine(pr, true);
    smooth(pr);
    dvec b(M.n);
    M.from_external(b, pr);
    if (M.score1(b, ASK_METRIC).metric <= M.target)
      raise_metric(pr);
    else
      lower_metric(pr);
    smooth(pr);
    nudge_metric(pr);
  }

  // ---- local tuning on all n coefficients --------------------------------------------------------
  // pruner_optimize_tc.cpp:163-263: pull the bound of the most expensive level towards its neighbour
  // while that pays at least 0.5 %; three strikes per coordinate, ten failures in a row end the pass
  void trim_bottleneck(dvec &pr)
  {
    const int n = M.n;
    dvec b(n), fine(n, 10.0);
    std::vector<int> strikes(n, 3);
    M.from_external(b, pr);
    int last = -1, failures = 0;
    for (;;)
    {
      const Score s = M.score1(b, ASK_VALUE | ASK_LEVELS);
      if (s.cost < std::sqrt(s.value) / 10.0)
        break;
      double peak = 0.0;
      int at      = 0;
      for (int i = 0; i < n; i++)
        if ((i != (n - last - 1)) && (strikes[n - i - 1] > 0) && s.levels[i] > peak)
          peak = s.levels[i], at = i;
      const int ind     = n - at - 1;
      const double kept = b[ind];
      if (ind == 0)
        break;
      b[ind] = b[ind] - (b[ind] - b[ind - 1]) / fine[ind];
      if (M.value(b) >= (s.value * 0.995))
      {
        b[ind] = kept;
        last   = ind;
        strikes[last]--;
        failures++;
      }
      else
      {
        if (fine[ind] < 1024)
          fine[ind] = fine[ind] * 1.05;
        failures = 0;
      }
      if (failures > 10)
        break;
    }
    M.to_external(pr, b);
  }
  // pruner_optimize_tc.cpp:269-367: below the most expensive level, raise bounds towards their upper
  // neighbour as long as the objective does not grow by 20 %
  void raise_cheap_levels(dvec &pr)
  {
    const int n = M.n;
    dvec b(n), fine(n, 10.0);
    M.from_external(b, pr);
    const double entry = M.value(b);
    for (int rounds = 1;; ++rounds)
    {
      const Score s = M.score1(b, ASK_VALUE | ASK_LEVELS);
      double peak   = 0.0;
      int at        = 0;
      for (int i = 0; i < n; i++)
        if (s.levels[i] > peak)
          peak = s.levels[i], at = i;
      const int ind = n - at - 1;
      if (ind <= 1)
        break;
      if (s.cost > std::sqrt(s.value) / 10.0)
        break;
      double before = s.value;
      for (int i = ind; i >= 1; --i)
      {
        if (b[i] <= b[i - 1])
          continue;
        for (int tries = 0; tries < 10; ++tries)
        {
          before            = M.value(b);
          const double kept = b[i - 1];
          b[i - 1]          = b[i - 1] + (b[i] - b[i - 1]) / fine[i - 1];
          if (M.value(b) >= (before * 1.2))
          {
            b[i - 1] = kept;
            break;
          }
          if (fine[i - 1] < 1024)
            fine[i - 1] = fine[i - 1] * 1.2;
        }
      }
      if (M.value(b) > (entry * 1.1) || rounds > 4)
        break;
    }
    M.to_external(pr, b);
  }
  // pruner_optimize_tc.cpp:373-399: iron out kinks between neighbours
  void smooth(dvec &pr)
  {
    const int n = M.n;
    dvec b(n);
    const double gap = 1.0 / n;
    M.from_external(b, pr);
    for (int i = 1; i < n - 1; ++i)
    {
      const double below = b[i] / b[i - 1], above = b[i + 1] / b[i];
      if ((above / below > 1.25) || (above / below < 0.8))
        b[i] = std::sqrt(b[i - 1] * b[i + 1]);
      if ((b[i + 1] - b[i]) > gap || (b[i] - b[i - 1]) > gap)
        b[i] = (b[i - 1] + b[i + 1]) / 2.0;
    }
    M.to_external(pr, b);
  }

  // ---- moving the metric to the target (pruner_optimize_tp.cpp) -------------------------------------
  // every coefficient moves by a weight inversely proportional to the cost of the levels above it
  void shift_metric(dvec &pr, bool up)
  {
    const int dn = (int)pr.size();
    dvec b(dn), prev(dn), w(dn);
    M.from_external(b, pr);
    for (int rounds = 0; rounds <= 10000; ++rounds)
    {
      const Score s = M.score1(b, ASK_METRIC | ASK_LEVELS);
      if (up ? s.metric >= M.target : s.metric <= M.target)
        break;
      double total = 0.0;
      for (int i = 0; i < dn; i++)
      {
        w[i] = 0.0;
        for (int j = i; j < dn; j++)
          w[i] = w[i] + s.levels[j];
        w[i] = 1.0 / w[i];
        if (w[i] < 1e-4)
          w[i] = 1e-4;
        total += w[i];
      }
      for (int i = 0; i < dn; i++)
        w[i] = w[i] / total;
      for (int i = dn - 1; i >= 0; --i)
      {
        prev[i] = b[i];
        if (up)
        {
          b[i] = b[i] + w[i];
          if (b[i] >= 1.0)
            b[i] = 1.0;
        }
        else
        {
          b[i] = b[i] - w[i];
          if (b[i] < 1e-4)
            b[i] = 1e-4;
        }
      }
      M.enforce(b);
      if (b == prev)
        break;
    }
    M.to_external(pr, b);
  }
  void raise_metric(dvec &pr) { shift_metric(pr, true); }
  void lower_metric(dvec &pr) { shift_metric(pr, false); }
  // pruner_optimize_tp.cpp:142-203: uniform steps of 1e-4 until the metric is within 5 % of the target
  void nudge_metric(dvec &pr)
  {
    const int dn = (int)pr.size();
    dvec b(dn), prev(dn);
    M.from_external(b, pr);
    for (;;)
    {
      const double ratio = M.score1(b, ASK_METRIC).metric / M.target;
      if (ratio < 1.05 && ratio > 0.95)
        break;
      for (int i = dn - 1; i >= 0; --i)
      {
        prev[i] = b[i];
        if (ratio < 1)
        {
          b[i] = b[i] + 1e-4;
          if (b[i] >= 1.0)
            b[i] = 1.0;
        }
        else
        {
          b[i] = b[i] - 1e-4;
          if (b[i] < 1e-4)
            b[i] = 1e-4;
        }
      }
      M.enforce(b);
      if (b == prev)
        break;
    }
    M.to_external(pr, b);
  }

  std::vector<dvec> batch;
  std::vector<Score> scores;
};

// prune() and its by-products (pruner.cpp:190-227): coefficients, per-level cost of the HALF
// coefficients (the public single_enum_cost overload loads every second one, pruner.h:552-559),
// metric of all of them
int prune_with(VolumeEngine *eng, int n, int count, const double *gso_rs, double radius, double preproc,
               double target, int metric, int flags, double *coefficients, double *expectation,
               double *gh_factor, double *detailed_cost)
{
  if (n < 2 || n >= FPHIP_PRUNER_TABLE_N || count < 1 || !gso_rs || !coefficients)
    return FPHIP_ERROR;
  if (flags & OPT_VERBOSE)
    return FPHIP_UNSUPPORTED;
  try
  {
    BlockModel M(n, eng);
    M.configure(radius, preproc, target, metric, flags);
    if (count == 1)
      M.load_profile(gso_rs, true);
    else
      M.load_profiles(gso_rs, count);
    dvec pr;
    if (flags & OPT_FROM_INPUT)
      pr.assign(coefficients, coefficients + n);
    Search(M).run(pr);
    dvec hb(M.half), fb(n);
    M.from_external(hb, pr);
    M.from_external(fb, pr);
    const Score sc = M.score1(hb, ASK_COST | ASK_LEVELS);
    const Score sm = M.score1(fb, ASK_METRIC);
    for (int i = 0; i < n; ++i)
      coefficients[i] = pr[i];
    if (detailed_cost)
      for (int i = 0; i < n; ++i)
        detailed_cost[i] = i < (int)sc.levels.size() ? sc.levels[i] : 0.0;
    if (gh_factor)
      *gh_factor = radius / M.gaussian_heuristic();
    if (expectation)
      *expectation = sm.metric;
    return FPHIP_OK;
  }
  catch (const std::exception &)
  {
    return FPHIP_ERROR;
  }
}
}  // namespace

// hidden entry points for the BKZ service of gso_host.hip (in-loop pruning, hand-off decision)
__attribute__((visibility("hidden"))) int prune_block(VolumeEngine *eng, int n, const double *gso_r, double radius,
                                                      double preproc, double target, int flags,
                                                      double *coefficients, double *expectation)
{
  return prune_with(eng ? eng : host_volume_engine(), n, 1, gso_r, radius, preproc, target, 0, flags, coefficients,
                    expectation, nullptr, nullptr);
}
}  // namespace fphip_pruner

using namespace fphip_pruner;

// ---------------------------------------------------------------------------------------------
// C ABI (include/fplll_hip.h)
// ---------------------------------------------------------------------------------------------
struct fphip_pruner_engine
{
  VolumeEngine *e;
  char err[256];
};

extern "C" int fphip_pruner_engine_create(int device, fphip_pruner_engine **out)
{
  if (!out)
    return FPHIP_ERROR;
  fphip_pruner_engine *h = new fphip_pruner_engine;
  h->err[0]              = 0;
  h->e                   = create_device_volume_engine(device, h->err, sizeof h->err);
  if (!h->e)
  {
    delete h;
    *out = nullptr;
    return FPHIP_ERROR;
  }
  *out = h;
  return FPHIP_OK;
}
extern "C" void fphip_pruner_engine_destroy(fphip_pruner_engine *h)
{
  if (!h)
    return;
  destroy_volume_engine(h->e);
  delete h;
}
extern "C" int fphip_pruner_engine_stats(const fphip_pruner_engine *h, unsigned long long *device_jobs,
                                         unsigned long long *host_jobs, unsigned long long *launches)
{
  if (!h)
    return FPHIP_ERROR;
  if (device_jobs)
    *device_jobs = h->e->device_jobs;
  if (host_jobs)
    *host_jobs = h->e->host_jobs;
  if (launches)
    *launches = h->e->launches;
  return FPHIP_OK;
}
extern "C" int fphip_pruner_volumes(fphip_pruner_engine *h, int m, int nvec, const double *bounds, int njobs,
                                    const int *job_vec, const int *job_k, double *out)
{
  if (m < 1 || m > 255 || nvec < 1 || !bounds || njobs < 0 || !job_vec || !job_k || !out)
    return FPHIP_ERROR;
  std::vector<VolumeJob> jobs((size_t)njobs);
  for (int j = 0; j < njobs; ++j)
  {
    if (job_vec[j] < 0 || job_vec[j] >= nvec || job_k[j] < 1 || job_k[j] > m)
      return FPHIP_ERROR;
    jobs[j] = VolumeJob{job_vec[j], job_k[j]};
  }
  VolumeEngine *e = h ? h->e : host_volume_engine();
  return e->run(bounds, nvec, m, jobs.data(), njobs, out) ? FPHIP_OK : FPHIP_ERROR;
}

extern "C" int fphip_pruner_prune_on(fphip_pruner_engine *h, int n, int count, const double *gso_rs,
                                     double enumeration_radius, double preproc_cost, double target, int metric,
                                     int flags, double *coefficients, double *expectation, double *gh_factor,
                                     double *detailed_cost)
{
  return prune_with(h ? h->e : host_volume_engine(), n, count, gso_rs, enumeration_radius, preproc_cost, target,
                    metric, flags, coefficients, expectation, gh_factor, detailed_cost);
}
extern "C" int fphip_pruner_prune(int n, const double *gso_r, double enumeration_radius, double preproc_cost,
                                  double target, int metric, int flags, double *coefficients, double *expectation,
                                  double *gh_factor, double *detailed_cost)
{
  return prune_with(host_volume_engine(), n, 1, gso_r, enumeration_radius, preproc_cost, target, metric, flags,
                    coefficients, expectation, gh_factor, detailed_cost);
}
extern "C" int fphip_pruner_prune_multi(int n, int count, const double *gso_rs, double enumeration_radius,
                                        double preproc_cost, double target, int metric, int flags,
                                        double *coefficients, double *expectation, double *gh_factor,
                                        double *detailed_cost)
{
  return prune_with(host_volume_engine(), n, count, gso_rs, enumeration_radius, preproc_cost, target, metric, flags,
                    coefficients, expectation, gh_factor, detailed_cost);
}

// svp_probability(pr) (pruner.cpp:166-176): a model without a profile — the probability needs none
extern "C" int fphip_pruner_svp_probability(int n, const double *pr, double *probability)
{
  if (n < 2 || n >= FPHIP_PRUNER_TABLE_N || !pr || !probability)
    return FPHIP_ERROR;
  try
  {
    BlockModel M(n, host_volume_engine());
    dvec b(n), ext(pr, pr + n);
    M.from_external(b, ext);
    *probability = M.score1(b, ASK_METRIC).metric;
    return FPHIP_OK;
  }
  catch (const std::exception &)
  {
    return FPHIP_ERROR;
  }
}

// Pruner(radius, ., gso_r, ., metric).single_enum_cost(pr, &detailed_cost) / .measure_metric(pr)
extern "C" int fphip_pruner_enum_cost(int n, const double *gso_r, double enumeration_radius, const double *pr,
                                      int metric, double *cost, double *metric_value, double *detailed_cost)
{
  if (n < 2 || n >= FPHIP_PRUNER_TABLE_N || !gso_r || !pr)
    return FPHIP_ERROR;
  try
  {
    BlockModel M(n, host_volume_engine());
    M.configure(enumeration_radius, 0.0, metric == 0 ? 0.5 : 1.0, metric, 0);
    M.load_profile(gso_r, true);
    dvec ext(pr, pr + n), hb(M.half), fb(n);
    M.from_external(hb, ext);
    M.from_external(fb, ext);
    const Score sc = M.score1(hb, ASK_COST | ASK_LEVELS);
    if (cost)
      *cost = sc.cost;
    if (detailed_cost)
      for (int i = 0; i < n; ++i)
        detailed_cost[i] = i < (int)sc.levels.size() ? sc.levels[i] : 0.0;
    if (metric_value)
      *metric_value = M.score1(fb, ASK_METRIC).metric;
    return FPHIP_OK;
  }
  catch (const std::exception &)
  {
    return FPHIP_ERROR;
  }
}
