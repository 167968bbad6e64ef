/*
 * ref_driver.cpp — TEST INFRASTRUCTURE ONLY.
 *
 * Drives the REAL reference (oracle/_ref/libfplll.so, built from /root/reference by
 * oracle/Makefile) through its public API to
 *   (1) generate golden fixtures for tests/golden/ (command `enumfix`, `gsofix`), and
 *   (2) run the reference's drivers with OUR external enumerator plugged in through
 *       fplll::set_external_enumerator (command `plugin`), the reference's own test axis
 *       ("swap the enumerator, results must still pass").
 * Only public reference headers are included; no reference source is copied.
 * Built into oracle/_ref/ref_driver (git-ignored).
 */
#include <fplll/fplll.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <dlfcn.h>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

using namespace fplll;
using std::vector;

typedef Z_NR<mpz_t> ZT;
typedef FP_NR<double> FT;

// ---- a recording external enumerator: grabs exactly what a plugin is handed, then declines ----
struct Recorded
{
  int d = 0;
  double maxdist = 0;
  bool dual = false, findsubsols = false;
  vector<double> mut, rdiag, pruning;
  int calls = 0;
} g_rec;

static std::array<uint64_t, FPLLL_EXTENUM_MAX_EXTENUM_DIM>
recording_enumerator(const int dim, double maxdist, std::function<extenum_cb_set_config> cbfunc,
                     std::function<extenum_cb_process_sol>, std::function<extenum_cb_process_subsol>,
                     bool dual, bool findsubsols)
{
  g_rec.d           = dim;
  g_rec.maxdist     = maxdist;
  g_rec.dual        = dual;
  g_rec.findsubsols = findsubsols;
  g_rec.mut.assign((size_t)dim * dim, 0.0);
  g_rec.rdiag.assign(dim, 0.0);
  g_rec.pruning.assign(dim, 0.0);
  cbfunc(g_rec.mut.data(), dim, true, g_rec.rdiag.data(), g_rec.pruning.data());
  g_rec.calls++;
  if (getenv("REFDRV_RECORD") && getenv("REFDRV_RECORD")[0] == '2')
    fprintf(stderr, "call %d dim %d maxdist %a r0 %a\n", g_rec.calls, dim, maxdist, g_rec.rdiag[0]);
  std::array<uint64_t, FPLLL_EXTENUM_MAX_EXTENUM_DIM> out{};
  if (getenv("REFDRV_INPUT_ONLY"))
    return out;           // pretend "done, nothing found": only the plugin inputs are wanted
  out[0] = ~uint64_t(0);  // decline → fplll falls back to its own enumerator (enumerate_ext.cpp:88)
  return out;
}

// ---- an evaluator that logs every eval_sol call (normalised dist + coefficients) ----
struct LoggingEvaluator : public FastEvaluator<FT>
{
  vector<std::pair<double, vector<double>>> log;
  LoggingEvaluator(size_t n, EvaluatorStrategy s, bool subsols = false)
      : FastEvaluator<FT>(n, s, subsols)
  {
  }
  void eval_sol(const vector<FT> &c, const enumf &dist, enumf &max_dist) override
  {
    vector<double> x(c.size());
    for (size_t i = 0; i < c.size(); ++i)
      x[i] = c[i].get_d();
    log.emplace_back(dist, x);
    FastEvaluator<FT>::eval_sol(c, dist, max_dist);
  }
};

static std::string hexd(double v)
{
  char buf[64];
  snprintf(buf, sizeof buf, "\"%a\"", v);
  return buf;
}
static void dump_vec(std::ostream &os, const char *name, const vector<double> &v, bool comma = true)
{
  os << "\"" << name << "\":[";
  for (size_t i = 0; i < v.size(); ++i)
    os << (i ? "," : "") << hexd(v[i]);
  os << "]" << (comma ? ",\n" : "\n");
}

static void make_basis(ZZ_mat<mpz_t> &A, int n, int k, int bits, int seed, int bkz_pre)
{
  RandGen::init_with_seed(seed);
  A.resize(n, n);
  A.gen_qary_prime(k, bits);
  lll_reduction(A, LLL_DEF_DELTA, LLL_DEF_ETA, LM_WRAPPER, FT_DEFAULT, 0, LLL_DEFAULT);
  if (bkz_pre > 0)
  {
    vector<Strategy> strategies;
    BKZParam par(bkz_pre, strategies);
    par.flags = BKZ_AUTO_ABORT;
    bkz_reduction(&A, NULL, par, FT_DOUBLE, 0);
  }
}

static vector<double> make_pruning(const std::string &spec, int d, MatGSO<ZT, FT> *M = nullptr,
                                   int first = 0, double radius = 0.0)
{
  vector<double> pr;
  if (spec == "none")
    return pr;
  if (spec.rfind("prune:", 0) == 0)
  {
    // the default.json substitute of SURVEY §8(d) C3: the reference's own pruner
    // (fplll/pruner/pruner.h:187-193) on this block's r-profile
    double target = atof(spec.c_str() + 6);
    vector<double> r;
    for (int i = 0; i < d; ++i)
    {
      FT t;
      M->get_r(t, first + i, first + i);
      r.push_back(t.get_d());
    }
    PruningParams pp;
    prune<FT>(pp, radius, 1e7, r, target, PRUNER_METRIC_PROBABILITY_OF_SHORTEST, PRUNER_GRADIENT);
    return pp.coefficients;
  }
  if (spec.rfind("linear:", 0) == 0)
  {
    int level = atoi(spec.c_str() + 7);
    return PruningParams::LinearPruningParams(d, level).coefficients;
  }
  fprintf(stderr, "bad pruning spec %s\n", spec.c_str());
  exit(2);
}

/* enumfix n k bits seed bkz_pre first d pruning max_sols strategy radius_factor
 *   → one JSON fixture on stdout: the plugin inputs + the reference's internal-enumerator outputs */
static int cmd_enumfix(int argc, char **argv)
{
  if (argc < 12)
  {
    fprintf(stderr, "usage: enumfix n k bits seed bkz_pre first d pruning max_sols strategy rfac\n");
    return 2;
  }
  int n = atoi(argv[2]), k = atoi(argv[3]), bits = atoi(argv[4]), seed = atoi(argv[5]);
  int bkz_pre = atoi(argv[6]), first = atoi(argv[7]), d = atoi(argv[8]);
  std::string prspec = argv[9];
  size_t max_sols    = (size_t)atol(argv[10]);
  int strategy       = atoi(argv[11]);
  double rfac        = argc > 12 ? atof(argv[12]) : 0.99;

  ZZ_mat<mpz_t> A, U, UT;
  if (n == 0)
  {  // basis from a file written by `dumpbasis` (argv[3])
    std::ifstream is(argv[3]);
    is >> A;
    if (A.get_rows() == 0)
    {
      fprintf(stderr, "cannot read basis %s\n", argv[3]);
      return 2;
    }
  }
  else
    make_basis(A, n, k, bits, seed, bkz_pre);
  MatGSO<ZT, FT> M(A, U, UT, GSO_ROW_EXPO);
  M.update_gso();

  // REFDRV_DUAL=1: a DUAL enumeration of the block, as svp_reduction(dual = true) runs it
  // (bkz.cpp:308-331): radius from 1 / r of the block's LAST row, enumerate(..., dual = true).  The
  // fixture then holds the TRANSFORMED inputs of EnumerationDyn::enumerate (enumerate.cpp:88-141),
  // recomputed below through MatGSO's public getters — the plugin hook is handed untransformed
  // mu / r for a dual call (enumerate_ext.cpp:57-74), so the recording enumerator is of no use here.
  const bool dual = getenv("REFDRV_DUAL") != nullptr;
  long expo;
  FT max_dist = M.get_r_exp(dual ? first + d - 1 : first, dual ? first + d - 1 : first, expo);
  if (dual)
  {
    max_dist.pow_si(max_dist, -1, GMP_RNDU);
    expo *= -1;
  }
  max_dist *= rfac;  // bkz.cpp:311-318 (delta)
  if (!dual && d > 30 && rfac <= 1.0)
  {
    FT root_det = M.get_root_det(first, first + d);
    adjust_radius_to_gh_bound(max_dist, expo, d, root_det, 1.1);  // bkz.cpp:319-323
  }
  if (getenv("REFDRV_RADIUS_SCALE"))
    max_dist *= atof(getenv("REFDRV_RADIUS_SCALE"));
  vector<double> pruning =
      make_pruning(prspec, d, &M, first, max_dist.get_d() * std::pow(2.0, (double)expo));

  set_external_enumerator(recording_enumerator);
  // REFDRV_SUBSOLS=1: the evaluator also collects sub-solutions (findsubsols, evaluator.h:185-205)
  LoggingEvaluator ev(max_sols, (EvaluatorStrategy)strategy, getenv("REFDRV_SUBSOLS") != nullptr);
  Enumeration<ZT, FT> E(M, ev);
  const FT max_dist0 = max_dist;  // (enumerate() returns the final bound in max_dist)
  auto t0 = std::chrono::steady_clock::now();
  E.enumerate(first, first + d, max_dist, expo, vector<FT>(), vector<enumxt>(), pruning, dual);
  if (dual)
  {  // what the internal enumerator worked on (enumerate.cpp:88-141, dual branch)
    long normexp = -1;
    for (int i = 0; i < d; ++i)
    {
      long rexpo;
      FT fr   = M.get_r_exp(i + first, i + first, rexpo);
      normexp = std::max(normexp, rexpo + fr.exponent());
    }
    normexp *= -1;
    FT fmd;
    fmd.mul_2si(max_dist0, expo - normexp);
    g_rec.d       = d;
    g_rec.maxdist = fmd.get_d(GMP_RNDU);
    g_rec.dual    = true;
    g_rec.mut.assign((size_t)d * d, 0.0);
    g_rec.rdiag.assign(d, 0.0);
    g_rec.pruning.assign(d, 1.0);
    for (int i = 0; i < d && i < (int)pruning.size(); ++i)
      g_rec.pruning[i] = pruning[i];
    for (int i = 0; i < d; ++i)
    {
      long rexpo;
      FT fr = M.get_r_exp(i + first, i + first, rexpo);
      fr.mul_2si(fr, rexpo + normexp);
      g_rec.rdiag[d - i - 1] = 1.0 / fr.get_d();
    }
    for (int i = 0; i < d; ++i)
      for (int j = i + 1; j < d; ++j)
      {
        FT fmu;
        M.get_mu(fmu, j + first, i + first);
        g_rec.mut[(size_t)(d - j - 1) * d + (d - i - 1)] = -fmu.get_d();
      }
  }
  double secs =
      std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  auto nodes = E.get_nodes_array();

  // final bound, normalised like the plugin sees it (enumerate.cpp:150-152 inverse)
  FT fin;
  fin.mul_2si(max_dist, expo - ev.normExp);

  std::ostringstream os;
  os << "{\n\"desc\":\"" << (n == 0 ? argv[3] : "") << " qary n=" << n << " k=" << k << " bits=" << bits << " seed=" << seed
     << " bkz_pre=" << bkz_pre << " first=" << first << " d=" << d << " pruning=" << prspec
     << " max_sols=" << max_sols << " strategy=" << strategy << " rfac=" << rfac << "\",\n";
  os << "\"d\":" << d << ",\n\"max_sols\":" << max_sols << ",\n\"strategy\":" << strategy << ",\n";
  if (dual)
    os << "\"dual\":1,\n";
  os << "\"maxdist\":" << hexd(g_rec.maxdist) << ",\n";
  dump_vec(os, "mut", g_rec.mut);
  dump_vec(os, "rdiag", g_rec.rdiag);
  dump_vec(os, "pruning", g_rec.pruning);
  os << "\"nodes\":[";
  uint64_t tot = 0;
  for (int i = 0; i <= d; ++i)
  {
    os << (i ? "," : "") << nodes[i];
    tot += nodes[i];
  }
  os << "],\n\"total_nodes\":" << tot << ",\n\"ref_seconds\":" << secs << ",\n";
  os << "\"final_maxdist\":" << hexd(fin.get_d()) << ",\n";
  os << "\"sol_log\":[";
  for (size_t s = 0; s < ev.log.size(); ++s)
  {
    os << (s ? ",\n" : "\n") << "{\"dist\":" << hexd(ev.log[s].first) << ",\"x\":[";
    for (int i = 0; i < d; ++i)
      os << (i ? "," : "") << (long)ev.log[s].second[i];
    os << "]}";
  }
  os << "]";
  if (getenv("REFDRV_SUBSOLS"))
  {  // final table: best sub-solution per offset, distance normalised like the plugin sees it
    os << ",\n\"subsols\":[";
    bool firsts = true;
    for (size_t o = 0; o < ev.sub_solutions.size(); ++o)
    {
      if (ev.sub_solutions[o].second.empty())
        continue;
      FT dn;
      dn.mul_2si(ev.sub_solutions[o].first, -ev.normExp);
      os << (firsts ? "\n" : ",\n") << "{\"offset\":" << o << ",\"dist\":" << hexd(dn.get_d())
         << ",\"x\":[";
      for (int i = 0; i < d; ++i)
        os << (i ? "," : "") << (long)ev.sub_solutions[o].second[i].get_d();
      os << "]}";
      firsts = false;
    }
    os << "]";
  }
  os << "\n}\n";
  std::cout << os.str();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// plugin: install OUR external enumerator (from libfplll_hip_extenum.so) and run reference drivers
// ---------------------------------------------------------------------------------------------
typedef std::array<uint64_t, FPLLL_EXTENUM_MAX_EXTENUM_DIM>(extenum_fn)(
    const int, double, std::function<extenum_cb_set_config>, std::function<extenum_cb_process_sol>,
    std::function<extenum_cb_process_subsol>, bool, bool);

static extenum_fn *load_plugin(const char *path)
{
  void *h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  if (!h)
  {
    fprintf(stderr, "dlopen %s: %s\n", path, dlerror());
    exit(3);
  }
  void *sym = dlsym(h, "fplll_hip_extenum_entry");
  if (!sym)
  {
    fprintf(stderr, "dlsym: %s\n", dlerror());
    exit(3);
  }
  // the .so exports a C getter returning the address of the C++ function with fplll's signature
  typedef void *(getter_t)();
  return (extenum_fn *)((getter_t *)sym)();
}

struct EnumOut
{
  uint64_t total = 0;
  vector<uint64_t> nodes;
  bool found = false;
  double dist = 0;  // de-normalised first solution norm
  vector<double> x;
  double secs = 0;
};

static EnumOut run_enum(MatGSO<ZT, FT> &M, int first, int d, const vector<double> &pruning,
                        double rfac, size_t max_sols, int strategy, bool dual = false)
{
  long expo;
  FT max_dist = M.get_r_exp(dual ? first + d - 1 : first, dual ? first + d - 1 : first, expo);
  if (dual)
  {  // the radius of svp_reduction(dual = true), bkz.cpp:308-318
    max_dist.pow_si(max_dist, -1, GMP_RNDU);
    expo *= -1;
  }
  max_dist *= rfac;
  if (!dual && d > 30 && rfac <= 1.0)
  {
    FT root_det = M.get_root_det(first, first + d);
    adjust_radius_to_gh_bound(max_dist, expo, d, root_det, 1.1);
  }
  if (getenv("REFDRV_RADIUS_SCALE"))
    max_dist *= atof(getenv("REFDRV_RADIUS_SCALE"));
  FastEvaluator<FT> ev(max_sols, (EvaluatorStrategy)strategy, false);
  Enumeration<ZT, FT> E(M, ev);
  auto t0 = std::chrono::steady_clock::now();
  E.enumerate(first, first + d, max_dist, expo, vector<FT>(), vector<enumxt>(), pruning, dual);
  EnumOut o;
  o.secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  auto na = E.get_nodes_array();
  for (int i = 0; i <= d; ++i)
  {
    o.nodes.push_back(na[i]);
    o.total += na[i];
  }
  if (!ev.empty())
  {
    o.found = true;
    o.dist  = ev.begin()->first.get_d();
    for (auto &c : ev.begin()->second)
      o.x.push_back(c.get_d());
  }
  return o;
}

/* plugin <so> n k bits seed bkz_pre first d pruning max_sols strategy rfac
 * Runs the same enumeration with (a) the internal enumerator and (b) our plugin; prints a JSON
 * comparison.  Exit 0 iff the parity contract holds: same best squared norm; if max_sols is large
 * (bound never shrinks) identical per-level node counts. */
static int cmd_plugin(int argc, char **argv)
{
  if (argc < 14)
  {
    fprintf(stderr, "usage: plugin so n k bits seed bkz_pre first d pruning max_sols strategy rfac\n");
    return 2;
  }
  const char *so = argv[2];
  int n = atoi(argv[3]), k = atoi(argv[4]), bits = atoi(argv[5]), seed = atoi(argv[6]);
  int bkz_pre = atoi(argv[7]), first = atoi(argv[8]), d = atoi(argv[9]);
  std::string prspec = argv[10];
  size_t max_sols    = (size_t)atol(argv[11]);
  int strategy       = atoi(argv[12]);
  double rfac        = atof(argv[13]);

  ZZ_mat<mpz_t> A, U, UT;
  make_basis(A, n, k, bits, seed, bkz_pre);
  // REFDRV_PLUGIN_DUAL=1: a DUAL call through the hook, on a MatGSO WITHOUT row exponents — the only
  // configuration in which the reference's adapter hands a plugin the right radius for a dual call
  // (enumerate_ext.cpp:75 against enumerate.cpp:100-106; see extenum_shim.cpp)
  const bool pdual = getenv("REFDRV_PLUGIN_DUAL") != nullptr;
  MatGSO<ZT, FT> M(A, U, UT, pdual ? 0 : GSO_ROW_EXPO);
  M.update_gso();
  vector<double> pruning = make_pruning(prspec, d);

  set_external_enumerator(nullptr);
  EnumOut ref = run_enum(M, first, d, pruning, rfac, max_sols, strategy, pdual);
  extenum_fn *fn = load_plugin(so);
  set_external_enumerator(fn);
  EnumOut ours = run_enum(M, first, d, pruning, rfac, max_sols, strategy, pdual);
  // warm second run for timing
  EnumOut ours2 = run_enum(M, first, d, pruning, rfac, max_sols, strategy, pdual);
  bool dual_vec_ok = true;
  if (pdual && ref.found && ours.found)
  {  // the ORIENTATION of the solution is part of the contract (the adapter does not reverse it)
    vector<double> neg(ref.x);
    for (auto &v : neg)
      v = -v;
    dual_vec_ok = (ours.x == ref.x) || (ours.x == neg);
  }

  bool ok = (ref.found == ours.found) && (!ref.found || ref.dist == ours.dist);
  if (pdual && ref.found && ours.found)
  {  // the adapter leaves the evaluator's exponent at +normexp for a dual call (enumerate_ext.cpp:79;
     // EnumerationDyn::enumerate sets -normexp, enumerate.cpp:100-106): the de-normalised distance it
     // records is the internal one times a power of four — same vector, same normalised norm
    int e2;
    ok = std::frexp(ours.dist / ref.dist, &e2) == 0.5 && ((e2 - 1) % 2 == 0);
  }
  bool counts_equal = ref.nodes == ours.nodes;
  if (max_sols >= 1000000 && strategy == 0)
    ok = ok && counts_equal;
  ok = ok && dual_vec_ok;
  printf("{\"ok\":%s,\"counts_equal\":%s,\"ref_found\":%d,\"ours_found\":%d,\"ref_dist\":%.17g,"
         "\"ours_dist\":%.17g,\"ref_nodes\":%llu,\"ours_nodes\":%llu,\"ref_secs\":%.6f,"
         "\"ours_secs\":%.6f,\"ours_secs_warm\":%.6f}\n",
         ok ? "true" : "false", counts_equal ? "true" : "false", (int)ref.found, (int)ours.found,
         ref.dist, ours.dist, (unsigned long long)ref.total, (unsigned long long)ours.total,
         ref.secs, ours.secs, ours2.secs);
  return ok ? 0 : 1;
}

// ---------------------------------------------------------------------------------------------
// gsofix: MatGSO<Z_NR<long>,FP_NR<double>> + LLLReduction::size_reduction golden vectors
// ---------------------------------------------------------------------------------------------
template <class M> static void dump_mat_hex(std::ostream &os, const char *name, const M &m, int d,
                                            bool lower_incl_diag)
{
  os << "\"" << name << "\":[";
  bool first = true;
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < d; ++j)
    {
      double v = (j < i || (lower_incl_diag && j == i)) ? m(i, j).get_d() : 0.0;
      os << (first ? "" : ",") << hexd(v);
      first = false;
    }
  os << "],\n";
}

/* gsofix n k bits seed perturb  → JSON: input basis (long), state after update_gso(), state after
 * LLLReduction::size_reduction(0,d) (eta = 0.51) */
static int cmd_gsofix(int argc, char **argv)
{
  if (argc < 7)
    return 2;
  int n = atoi(argv[2]), k = atoi(argv[3]), bits = atoi(argv[4]), seed = atoi(argv[5]);
  int perturb = atoi(argv[6]);
  ZZ_mat<mpz_t> A;
  make_basis(A, n, k, bits, seed, 0);
  ZZ_mat<long> b(n, n), u, ut;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
      b(i, j) = A(i, j).get_si();
  // deterministic unimodular perturbation so that rows are NOT size-reduced any more
  uint64_t lcg = 0x9E3779B97F4A7C15ull ^ (uint64_t)seed;
  for (int i = 1; i < n && perturb > 0; ++i)
    for (int t = 0; t < perturb; ++t)
    {
      lcg    = lcg * 6364136223846793005ull + 1442695040888963407ull;
      int j  = (int)((lcg >> 33) % (uint64_t)i);
      lcg    = lcg * 6364136223846793005ull + 1442695040888963407ull;
      long c = (long)((lcg >> 33) % 7) - 3;
      for (int col = 0; col < n; ++col)
        b(i, col) = b(i, col).get_si() + c * b(j, col).get_si();
    }
  std::ostringstream os;
  os << "{\n\"desc\":\"qary n=" << n << " k=" << k << " bits=" << bits << " seed=" << seed
     << " LLL then " << perturb << " random row ops per row\",\n\"d\":" << n << ",\n\"n\":" << n
     << ",\n\"b_in\":[";
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
      os << ((i || j) ? "," : "") << b(i, j).get_si();
  os << "],\n";

  MatGSO<Z_NR<long>, FP_NR<double>> M(b, u, ut, GSO_ROW_EXPO);
  M.update_gso();
  dump_mat_hex(os, "mu0", M.get_mu_matrix(), n, false);
  dump_mat_hex(os, "r0", M.get_r_matrix(), n, true);
  os << "\"row_expo0\":[";
  for (int i = 0; i < n; ++i)
    os << (i ? "," : "") << M.row_expo[i];
  os << "],\n";

  LLLReduction<Z_NR<long>, FP_NR<double>> L(M, LLL_DEF_DELTA, LLL_DEF_ETA, LLL_DEFAULT);
  bool ok = L.size_reduction(0, n, 0);
  os << "\"status\":" << (ok ? 1 : 0) << ",\n\"b_out\":[";
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
      os << ((i || j) ? "," : "") << b(i, j).get_si();
  os << "],\n";
  dump_mat_hex(os, "mu1", M.get_mu_matrix(), n, false);
  dump_mat_hex(os, "r1", M.get_r_matrix(), n, true);
  os << "\"row_expo1\":[";
  for (int i = 0; i < n; ++i)
    os << (i ? "," : "") << M.row_expo[i];
  os << "]\n}\n";
  std::cout << os.str();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// lllfix: LLLReduction<Z_NR<long>,FP_NR<double>>::lll on MatGSO(GSO_ROW_EXPO) golden vectors
// ---------------------------------------------------------------------------------------------
// test bases for lllfix / hlllfix (see cmd_lllfix for the meaning of the arguments)
static bool gen_long_basis(const std::string &type, int &d, int k, int bits, int seed, int zero_rows,
                           int dup_rows, ZZ_mat<long> &b0)
{
  RandGen::init_with_seed(seed);
  ZZ_mat<mpz_t> A;
  if (type.rfind("f:", 0) == 0 || type.rfind("F:", 0) == 0)
  {  // basis from a file in fplll's matrix format (d is taken from the file); "F:" LLL-reduces it
     // with the wrapper first, as bkz_reduction does (bkz.cpp:870-877), so that entries of the
     // reference's own test lattices (tests/lattices/dim55_in: 75 bits) fit a long afterwards
    std::ifstream in(type.substr(2));
    in >> A;
    if (A.get_rows() == 0)
      return false;
    d = A.get_rows();
    if (type[0] == 'F')
    {
      lll_reduction(A, LLL_DEF_DELTA, LLL_DEF_ETA, LM_WRAPPER, FT_DEFAULT, 0, LLL_DEFAULT);
      for (int i = 0; i < A.get_rows(); ++i)
        for (int j = 0; j < A.get_cols(); ++j)
          if (!mpz_fits_slong_p(A(i, j).get_data()))
            return false;
    }
  }
  else if (type == "q")
  {
    A.resize(d, d);
    A.gen_qary_prime(k, bits);
  }
  else if (type == "r")
  {
    A.resize(d, d + 1);
    A.gen_intrel(bits);
  }
  else if (type == "n")
  {  // NTRU-like [[I, Rot(h)],[0, qI]], d must be even (latticegen n, matrix.cpp:288-352)
    A.resize(d, d);
    A.gen_ntrulike_bits(bits);
  }
  else
  {
    A.resize(d, d);
    A.gen_uniform(bits);
  }
  const int n = A.get_cols();
  b0.resize(d, n);
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < n; ++j)
      b0(i, j) = A(i, j).get_si();
  for (int t = 0; t < zero_rows && t < d; ++t)
    for (int j = 0; j < n; ++j)
      b0(t, j) = 0;
  for (int t = 0; t < dup_rows; ++t)
    for (int j = 0; j < n; ++j)
      b0(d - 1 - t, j) = b0(zero_rows + t, j).get_si() + b0(zero_rows + t + 1, j).get_si();
  return true;
}

/* lllfix type d k bits seed kmin kstart kend zero_rows dup_rows [reps]
 *   type q: gen_qary_prime(k,bits) d x d (raw, unreduced);  r: gen_intrel(bits) d x (d+1);
 *        u: gen_uniform(bits) d x d;  f:<path>: read the basis from a file
 *   zero_rows: that many leading rows are zeroed;  dup_rows: row d-1-t is replaced by a copy of
 *   row t+zero_rows + row t+zero_rows+1 (linear dependencies: the "zeros" path of lll.cpp:144-150)
 *   reps > 1: also time `reps` reductions of the same input (cpu baseline)            */
static int cmd_lllfix(int argc, char **argv)
{
  if (argc < 12)
    return 2;
  std::string type = argv[2];
  int d = atoi(argv[3]), k = atoi(argv[4]), bits = atoi(argv[5]), seed = atoi(argv[6]);
  int kmin = atoi(argv[7]), kstart = atoi(argv[8]), kend = atoi(argv[9]);
  int zero_rows = atoi(argv[10]), dup_rows = atoi(argv[11]);
  int reps = argc > 12 ? atoi(argv[12]) : 1;
  ZZ_mat<long> b0, u, ut;
  if (!gen_long_basis(type, d, k, bits, seed, zero_rows, dup_rows, b0))
    return 3;
  const int n = b0.get_cols();
  if (kend < 0)
    kend = d;
  std::ostringstream os;
  os << "{\n\"desc\":\"lll type=" << type << " d=" << d << " k=" << k << " bits=" << bits
     << " seed=" << seed << " zero_rows=" << zero_rows << " dup_rows=" << dup_rows
     << "\",\n\"d\":" << d << ",\n\"n\":" << n << ",\n\"kmin\":" << kmin << ",\n\"kstart\":"
     << kstart << ",\n\"kend\":" << kend << ",\n\"delta\":" << hexd(LLL_DEF_DELTA) << ",\n\"eta\":"
     << hexd(LLL_DEF_ETA) << ",\n\"flags\":" << (getenv("LLLFIX_FLAGS") ? atoi(getenv("LLLFIX_FLAGS")) : 0) << ",\n\"b_in\":[";
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < n; ++j)
      os << ((i || j) ? "," : "") << b0(i, j).get_si();
  os << "],\n";
  ZZ_mat<long> b = b0;
  double secs = 0;
  int status = 0, final_kappa = 0, n_swaps = 0, zeros = 0;
  for (int rep = 0; rep < reps; ++rep)
  {
    b = b0;
    if (getenv("LLLFIX_U"))  // enable_transform: the run keeps u with b_out = u b_in (the fixture records it)
      u.gen_identity(d);
    if (getenv("LLLFIX_UINV"))  // enable_inverse_transform as well: u_inv_t, the inverse transpose of u (gso.cpp:84-158)
      ut.gen_identity(d);
    auto t0 = std::chrono::steady_clock::now();
    MatGSO<Z_NR<long>, FP_NR<double>> M(b, u, ut, GSO_ROW_EXPO);
    // (LLLFIX_FLAGS: fplll's LLLFlags for this run — LLL_SIEGEL = 4; the fixture records them)
    LLLReduction<Z_NR<long>, FP_NR<double>> L(M, LLL_DEF_DELTA, LLL_DEF_ETA,
                                              getenv("LLLFIX_FLAGS") ? atoi(getenv("LLLFIX_FLAGS")) : LLL_DEFAULT);
    if (kstart > 0)
    {  // the caller's precondition: rows below kappa_start are known to the GSO
      for (int i = 0; i < kstart; ++i)
        M.update_gso_row(i);
    }
    L.lll(kmin, kstart, kend, 0);
    secs += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    status = L.status; final_kappa = L.final_kappa; n_swaps = L.n_swaps; zeros = L.zeros;
  }
  os << "\"ref_status\":" << status << ",\n\"final_kappa\":" << final_kappa << ",\n\"n_swaps\":"
     << n_swaps << ",\n\"zeros\":" << zeros << ",\n\"reps\":" << reps << ",\n\"ref_seconds\":"
     << secs << ",\n\"b_out\":[";
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < n; ++j)
      os << ((i || j) ? "," : "") << b(i, j).get_si();
  os << "]";
  if (u.get_rows() > 0)
  {
    os << ",\n\"u_out\":[";
    for (int i = 0; i < d; ++i)
      for (int j = 0; j < d; ++j)
        os << ((i || j) ? "," : "") << u(i, j).get_si();
    os << "]";
  }
  if (ut.get_rows() > 0)
  {
    os << ",\n\"u_inv_t_out\":[";
    for (int i = 0; i < d; ++i)
      for (int j = 0; j < d; ++j)
        os << ((i || j) ? "," : "") << ut(i, j).get_si();
    os << "]";
  }
  os << "\n}\n";
  std::cout << os.str();
  return 0;
}


/* hlllfix type d k bits seed [reps]: HLLLReduction<Z_NR<long>,FP_NR<double>>::hlll() with the
 * LM_FAST Householder flags (wrapper.cpp:790-806), default delta/eta/theta/c */
static int cmd_hlllfix(int argc, char **argv)
{
  if (argc < 7)
    return 2;
  std::string type = argv[2];
  int d = atoi(argv[3]), k = atoi(argv[4]), bits = atoi(argv[5]), seed = atoi(argv[6]);
  int reps = argc > 7 ? atoi(argv[7]) : 1;
  ZZ_mat<long> b0, u, ut;
  if (!gen_long_basis(type, d, k, bits, seed, 0, 0, b0))
    return 3;
  const int n = b0.get_cols();
  std::ostringstream os;
  os << "{\n\"desc\":\"hlll type=" << type << " d=" << d << " k=" << k << " bits=" << bits
     << " seed=" << seed << "\",\n\"d\":" << d << ",\n\"n\":" << n << ",\n\"delta\":"
     << hexd(LLL_DEF_DELTA) << ",\n\"eta\":" << hexd(LLL_DEF_ETA) << ",\n\"theta\":"
     << hexd(HLLL_DEF_THETA) << ",\n\"c\":" << hexd(HLLL_DEF_C) << ",\n\"b_in\":[";
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < n; ++j)
      os << ((i || j) ? "," : "") << b0(i, j).get_si();
  os << "],\n";
  ZZ_mat<long> b = b0;
  double secs = 0;
  int status = 0;
  for (int rep = 0; rep < reps; ++rep)
  {
    b = b0;
    auto t0 = std::chrono::steady_clock::now();
    MatHouseholder<Z_NR<long>, FP_NR<double>> M(b, u, ut,
                                                HOUSEHOLDER_ROW_EXPO | HOUSEHOLDER_OP_FORCE_LONG);
    HLLLReduction<Z_NR<long>, FP_NR<double>> H(M, LLL_DEF_DELTA, LLL_DEF_ETA, HLLL_DEF_THETA,
                                               HLLL_DEF_C, LLL_DEFAULT);
    H.hlll();
    secs += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    status = H.get_status();
  }
  os << "\"ref_status\":" << status << ",\n\"reps\":" << reps << ",\n\"ref_seconds\":" << secs
     << ",\n\"b_out\":[";
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < n; ++j)
      os << ((i || j) ? "," : "") << b(i, j).get_si();
  os << "]\n}\n";
  std::cout << os.str();
  return 0;
}


/* In-loop pruning with the REAL reference (REFDRV_INLOOP="preproc_cost target min_block pruner_flags" of
 * bkzfix; the product's FPHIP_BKZ_PRUNE_IN_LOOP): the reference has no such mode, so its top-level loop is
 * driven from here through its PUBLIC members — bkz() (bkz.cpp:522-668, primal, BKZ_MAX_LOOPS / default /
 * BKZ_AUTO_ABORT), tour = trunc_tour + hkz (:360-441), svp_reduction (:274-358) — with ONE change: where
 * svp_reduction picks a pruning set of the strategies (:325) a top-level block of at least min_block rows is
 * pruned by the reference's own prune<FP_NR<double>>() on its current r-profile and radius.  Everything
 * inside stays the library's: svp_preprocessing (its recursive tours use the strategies' sets),
 * rerandomize_block, Enumeration, svp_postprocessing, size_reduction. */
struct InloopBKZ
{
  typedef Z_NR<long> ZT;
  typedef FP_NR<double> FT;
  MatGSO<ZT, FT> &m;
  LLLReduction<ZT, FT> &lll_obj;
  BKZReduction<ZT, FT> &B;
  const BKZParam &par;
  double preproc, target;
  int min_block, pflags;
  FastEvaluator<FT> evaluator;
  long prune_calls = 0, prune_failures = 0;
  int num_rows;

  InloopBKZ(MatGSO<ZT, FT> &m_, LLLReduction<ZT, FT> &l, BKZReduction<ZT, FT> &b, const BKZParam &p, double pc,
            double tg, int mb, int pf)
      : m(m_), lll_obj(l), B(b), par(p), preproc(pc), target(tg), min_block(mb), pflags(pf)
  {
    for (num_rows = m.d; num_rows > 0 && m.b_row_is_zero(num_rows - 1); num_rows--)
    {
    }
  }

  bool svp_reduction(int kappa, int block_size)
  {
    if (block_size < min_block || block_size < 4)
      return B.svp_reduction(kappa, block_size, par);
    const int first = kappa;
    if (!lll_obj.size_reduction(0, first + 1, 0))
      throw std::runtime_error(RED_STATUS_STR[lll_obj.status]);
    long old_first_expo;
    FT old_first                 = FT(m.get_r_exp(first, first, old_first_expo));
    bool rerandomize             = false;
    double remaining_probability = 1.0;
    while (remaining_probability > 1. - par.min_success_probability)
    {
      if (rerandomize)
        B.rerandomize_block(kappa + 1, kappa + block_size, par.rerandomization_density);
      B.svp_preprocessing(kappa, block_size, par);
      long max_dist_expo;
      FT max_dist = m.get_r_exp(first, first, max_dist_expo);
      FT delta    = par.delta;
      max_dist *= delta;
      if ((par.flags & BKZ_GH_BND) && block_size > 30)
      {
        FT root_det = m.get_root_det(kappa, kappa + block_size);
        adjust_radius_to_gh_bound(max_dist, max_dist_expo, block_size, root_det, par.gh_factor);
      }
      // ---- the one change: prune THIS block (instead of get_pruning, bkz.cpp:82-98) ----
      PruningParams pruning;
      bool pruned = false;
      {
        vector<double> r;
        for (int i = kappa; i < kappa + block_size; ++i)
        {
          FT x;
          m.get_r(x, i, i);
          r.push_back(x.get_d());
        }
        const double radius = max_dist.get_d() * pow(2, max_dist_expo);
        ++prune_calls;
        try
        {
          if (std::isfinite(radius) && radius > 0)
          {
            prune<FT>(pruning, radius, preproc, r, target, PRUNER_METRIC_PROBABILITY_OF_SHORTEST, pflags);
            pruned = true;
          }
        }
        catch (const std::exception &)
        {
        }
      }
      if (!pruned)
      {  // the strategies' choice, bkz.cpp:82-98
        ++prune_failures;
        long e2;
        FT md = m.get_r_exp(kappa, kappa, e2), gh = md, root_det = m.get_root_det(kappa, kappa + block_size);
        adjust_radius_to_gh_bound(gh, e2, block_size, root_det, 1.0);
        pruning = par.strategies[block_size].get_pruning(md.get_d() * pow(2, e2), gh.get_d() * pow(2, e2));
      }
      evaluator.solutions.clear();
      Enumeration<ZT, FT> enum_obj(m, evaluator);
      enum_obj.enumerate(kappa, kappa + block_size, max_dist, max_dist_expo, vector<FT>(), vector<enumxt>(),
                         pruning.coefficients, false);
      B.nodes += enum_obj.get_nodes();
      if (!evaluator.empty())
      {
        B.svp_postprocessing(kappa, block_size, evaluator.begin()->second, false);
        rerandomize = false;
      }
      else
        rerandomize = true;
      remaining_probability *= (1 - pruning.expectation);
    }
    if (!lll_obj.size_reduction(0, first + 1, 0))
      throw std::runtime_error(RED_STATUS_STR[lll_obj.status]);
    long new_first_expo;
    FT new_first = m.get_r_exp(first, first, new_first_expo);
    new_first.mul_2si(new_first, new_first_expo - old_first_expo);
    return old_first <= new_first;
  }
  bool tour(int min_row, int max_row)
  {
    bool clean = true;
    for (int kappa = min_row; kappa < max_row - par.block_size; ++kappa)  // trunc_tour
      clean &= svp_reduction(kappa, par.block_size);
    const int h0 = std::max(max_row - par.block_size, 0);                 // hkz
    for (int kappa = h0; kappa < max_row - 1; ++kappa)
      clean &= svp_reduction(kappa, max_row - kappa);
    lll_obj.size_reduction(max_row - 1, max_row, max_row - 2);
    return clean;
  }
  int run()
  {
    B.nodes = 0;
    if (par.block_size < 2)
      return RED_SUCCESS;
    BKZAutoAbort<ZT, FT> auto_abort(m, num_rows);
    m.discover_all_rows();
    int final_status = RED_SUCCESS;
    try
    {
      for (int i = 0;; ++i)
      {
        if ((par.flags & BKZ_MAX_LOOPS) && i >= par.max_loops)
        {
          final_status = RED_BKZ_LOOPS_LIMIT;
          break;
        }
        if ((par.flags & BKZ_AUTO_ABORT) && auto_abort.test_abort(par.auto_abort_scale, par.auto_abort_max_no_dec))
          break;
        const bool clean = tour(0, num_rows);
        if (clean || par.block_size >= num_rows)
          break;
      }
    }
    catch (const std::runtime_error &)
    {
      return lll_obj.status;
    }
    return final_status;
  }
};

/* bkzfix type d k bits seed block_size max_loops [reps] ; REFDRV_BKZ_AUTO_ABORT=1 adds BKZ_AUTO_ABORT:
 * BKZReduction<Z_NR<long>,FP_NR<double>>::bkz() on MatGSO(GSO_ROW_EXPO) exactly as bkz_reduction_f
 * sets it up after convert<long> (bkz.cpp:813-845), empty strategies (no pruning / preprocessing),
 * flags BKZ_DEFAULT (max_loops = 0) or BKZ_MAX_LOOPS.  The input is LLL-reduced first, as
 * bkz_reduction does (bkz.cpp:870-885). */
static int cmd_bkzfix(int argc, char **argv)
{
  if (argc < 9)
    return 2;
  std::string type = argv[2];
  int d = atoi(argv[3]), k = atoi(argv[4]), bits = atoi(argv[5]), seed = atoi(argv[6]);
  int block_size = atoi(argv[7]), max_loops = atoi(argv[8]);
  int reps = argc > 9 ? atoi(argv[9]) : 1;
  ZZ_mat<long> b0, u, ut;
  if (!gen_long_basis(type, d, k, bits, seed, 0, 0, b0))
    return 3;
  const int n = b0.get_cols();
  {  // LLL first (long/double, the same reduction the device LLL kernel reproduces)
    MatGSO<Z_NR<long>, FP_NR<double>> M(b0, u, ut, GSO_ROW_EXPO);
    LLLReduction<Z_NR<long>, FP_NR<double>> L(M, LLL_DEF_DELTA, LLL_DEF_ETA, LLL_DEFAULT);
    L.lll();
  }
  std::ostringstream os;
  os << "{\n\"desc\":\"bkz type=" << type << " d=" << d << " k=" << k << " bits=" << bits
     << " seed=" << seed << " LLL-reduced, then BKZ-" << block_size << " max_loops=" << max_loops
     << "\",\n\"d\":" << d << ",\n\"n\":" << n << ",\n\"block_size\":" << block_size
     << ",\n\"max_loops\":" << max_loops << ",\n\"auto_abort\":"
     << (getenv("REFDRV_BKZ_AUTO_ABORT") ? 1 : 0) << ",\n\"delta\":" << hexd(LLL_DEF_DELTA) << ",\n\"eta\":"
     << hexd(LLL_DEF_ETA) << ",\n\"b_in\":[";
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < n; ++j)
      os << ((i || j) ? "," : "") << b0(i, j).get_si();
  os << "],\n";
  // REFDRV_STRATEGIES=<json>: BKZ with preprocessing / pruning strategies (load_strategies_json);
  // REFDRV_BKZ_FLAGS: extra BKZFlags (0x80 GH_BND, 0x10 BOUNDED_LLL); REFDRV_RNG_SEED: state of
  // RandGen (rerandomize_block draws from it) at the start of bkz()
  vector<Strategy> loaded;
  if (getenv("REFDRV_STRATEGIES"))
  {
    loaded = load_strategies_json(getenv("REFDRV_STRATEGIES"));
    while ((int)loaded.size() <= block_size)
    {
      loaded.emplace_back();
      loaded.back().pruning_parameters.emplace_back(PruningParams());
    }
    // the strategies as the reference holds them, flattened (oracle.h: oracle_strategies)
    std::ostringstream po, pv, ro, rg, re, co, cv;
    int np = 0, npr = 0, nc = 0;
    for (size_t bs = 0; bs < loaded.size(); ++bs)
    {
      po << (bs ? "," : "") << np;
      ro << (bs ? "," : "") << npr;
      for (int pb : loaded[bs].preprocessing_block_sizes)
        pv << (np++ ? "," : "") << pb;
      for (const PruningParams &pp : loaded[bs].pruning_parameters)
      {
        rg << (npr ? "," : "") << hexd(pp.gh_factor);
        re << (npr ? "," : "") << hexd(pp.expectation);
        co << (npr ? "," : "") << nc;
        for (double c : pp.coefficients)
          cv << (nc++ ? "," : "") << hexd(c);
        ++npr;
      }
    }
    po << "," << np;
    ro << "," << npr;
    co << (npr ? "," : "") << nc;
    os << "\"strategies\":{\"max_block_size\":" << loaded.size() - 1 << ",\"pre_off\":[" << po.str()
       << "],\"pre\":[" << pv.str() << "],\"prune_off\":[" << ro.str() << "],\"prune_gh\":["
       << rg.str() << "],\"prune_exp\":[" << re.str() << "],\"coeff_off\":[" << co.str()
       << "],\"coeff\":[" << cv.str() << "]},\n";
  }
  const int extra_flags = getenv("REFDRV_BKZ_FLAGS") ? (int)strtol(getenv("REFDRV_BKZ_FLAGS"), 0, 0) : 0;
  const long rng_seed   = getenv("REFDRV_RNG_SEED") ? atol(getenv("REFDRV_RNG_SEED")) : 0;
  os << "\"flags\":" << ((max_loops > 0 ? BKZ_MAX_LOOPS : BKZ_DEFAULT) | extra_flags |
                         (getenv("REFDRV_BKZ_AUTO_ABORT") ? BKZ_AUTO_ABORT : 0) |
                         (getenv("REFDRV_DUMP_GSO") ? BKZ_DUMP_GSO : 0))
     << ",\n\"gh_factor\":" << hexd(BKZ_DEF_GH_FACTOR) << ",\n\"rng_seed\":" << rng_seed << ",\n";
  ZZ_mat<long> b = b0;
  double secs = 0;
  int status = 0;
  long nodes = 0;
  long inloop_calls = 0, inloop_fails = 0;
  double inloop_desc[4] = {0, 0, 0, 0};
  for (int rep = 0; rep < reps; ++rep)
  {
    b = b0;
    vector<Strategy> strategies = loaded;
    int flags = (max_loops > 0 ? BKZ_MAX_LOOPS : BKZ_DEFAULT) | extra_flags;
    if (getenv("REFDRV_BKZ_AUTO_ABORT"))
      flags |= BKZ_AUTO_ABORT;
    if (getenv("REFDRV_DUMP_GSO"))
      flags |= BKZ_DUMP_GSO;
    BKZParam par(block_size, strategies, LLL_DEF_DELTA, flags, max_loops);
    if (getenv("REFDRV_DUMP_GSO"))  // BKZ_DUMP_GSO into this file; its content goes into the fixture below
      par.dump_gso_filename = getenv("REFDRV_DUMP_GSO");
    RandGen::init_with_seed(rng_seed);
    auto t0 = std::chrono::steady_clock::now();
    MatGSO<Z_NR<long>, FP_NR<double>> M(b, u, ut, GSO_ROW_EXPO);
    LLLReduction<Z_NR<long>, FP_NR<double>> L(M, LLL_DEF_DELTA, LLL_DEF_ETA, LLL_DEFAULT);
    BKZReduction<Z_NR<long>, FP_NR<double>> B(M, L, par);
    // fplll's own enumerator (node counts by the fplll rule); the build's default external
    // enumerator is enumlib, which counts before the bound test
    set_external_enumerator(nullptr);
    if (getenv("REFDRV_RECORD"))
      set_external_enumerator(recording_enumerator);
    if (getenv("REFDRV_RECORD") && getenv("REFDRV_RECORD")[0] == '3')
    {  // debugging aid: one hkz pass by hand, node count after every svp_reduction
      set_external_enumerator(nullptr);
      for (int kappa = 0; kappa < d - 1; ++kappa)
      {
        long before = B.nodes;
        B.svp_reduction(kappa, std::min(block_size, d - kappa), par);
        fprintf(stderr, "kappa %d nodes %ld\n", kappa, B.nodes - before);
      }
    }
    else if (getenv("REFDRV_INLOOP"))
    {
      double pc = 1e6, tg = 0.5;
      int mb = 24, pf = PRUNER_GRADIENT;
      sscanf(getenv("REFDRV_INLOOP"), "%lf %lf %d %d", &pc, &tg, &mb, &pf);
      InloopBKZ I(M, L, B, par, pc, tg, mb, pf);
      B.status       = I.run();
      inloop_calls   = I.prune_calls;
      inloop_fails   = I.prune_failures;
      inloop_desc[0] = pc, inloop_desc[1] = tg, inloop_desc[2] = mb, inloop_desc[3] = pf;
    }
    else
    B.bkz();
    if (getenv("REFDRV_RECORD"))
      fprintf(stderr, "enumeration calls: %d\n", g_rec.calls);
    secs += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    status = B.status;
    nodes  = B.nodes;
  }
  if (getenv("REFDRV_INLOOP"))
    os << "\"inloop\":{\"preproc_cost\":" << hexd(inloop_desc[0]) << ",\"target\":" << hexd(inloop_desc[1])
       << ",\"min_block\":" << (int)inloop_desc[2] << ",\"pruner_flags\":" << (int)inloop_desc[3]
       << ",\"prune_calls\":" << inloop_calls << ",\"prune_failures\":" << inloop_fails << "},\n";
  if (getenv("REFDRV_DUMP_GSO"))
  {  // the reference's own dump (dump_gso, bkz.cpp:729-790: a JSON array), verbatim
    std::ifstream df(getenv("REFDRV_DUMP_GSO"));
    std::stringstream dss;
    dss << df.rdbuf();
    os << "\"gso_dump\":" << dss.str() << ",\n";
  }
  os << "\"ref_status\":" << status << ",\n\"nodes\":" << nodes << ",\n\"reps\":" << reps
     << ",\n\"ref_seconds\":" << secs << ",\n\"b_out\":[";
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < n; ++j)
      os << ((i || j) ? "," : "") << b(i, j).get_si();
  os << "]\n}\n";
  std::cout << os.str();
  return 0;
}


// ---------------------------------------------------------------------------------------------
// BKZ tour axis (SURVEY.md §8(d) metric (ii)): the reference's bkz_reduction, unchanged, with its
// internal enumerator or with OUR plugin installed through set_external_enumerator.
// ---------------------------------------------------------------------------------------------
typedef std::array<uint64_t, FPLLL_EXTENUM_MAX_EXTENUM_DIM>(extenum_fn_fw)(
    const int, double, std::function<extenum_cb_set_config>, std::function<extenum_cb_process_sol>,
    std::function<extenum_cb_process_subsol>, bool, bool);
static extenum_fn_fw *load_plugin(const char *path);
static bool read_basis(const char *path, ZZ_mat<mpz_t> &A)
{
  std::ifstream is(path);
  is >> A;
  return A.get_rows() > 0;
}

/* genstrat basisfile beta → strategies JSON (load_strategies_json format, bkz_param.cpp:92-146):
 * the default.json substitute of SURVEY Appendix A: for b in (24,beta] four pruning vectors from the
 * reference's prune<>() on the r-profile of the middle block, radius = ghf*GH, target 0.5;
 * preprocessing block b-24 for b >= 45. */
static int cmd_genstrat(int argc, char **argv)
{
  if (argc < 4)
    return 2;
  ZZ_mat<mpz_t> A, U, UT;
  if (!read_basis(argv[2], A))
    return 2;
  int beta = atoi(argv[3]);
  int n    = A.get_rows();
  MatGSO<ZT, FT> M(A, U, UT, GSO_ROW_EXPO);
  M.update_gso();
  std::ostringstream os;
  os << "[";
  bool first = true;
  for (int b = 25; b <= beta; ++b)
  {
    int lo = (n - b) / 2;
    vector<double> r;
    for (int i = 0; i < b; ++i)
    {
      FT t;
      M.get_r(t, lo + i, lo + i);
      r.push_back(t.get_d());
    }
    os << (first ? "" : ",\n") << "{\"block_size\":" << b << ",\"preprocessing_block_sizes\":[";
    first = false;
    if (b >= 45)
      os << (b - 24);
    os << "],\"pruning_parameters\":[";
    const double ghfs[4] = {1.0, 1.1, 1.2, 1.4};
    for (int g = 0; g < 4; ++g)
    {
      FT max_dist = 1e300;  // "huge": adjust_radius_to_gh_bound then returns ghf*GH (bkz.cpp:92-96)
      long expo   = 0;
      FT root_det = M.get_root_det(lo, lo + b);
      adjust_radius_to_gh_bound(max_dist, expo, b, root_det, ghfs[g]);
      PruningParams pp;
      prune<FT>(pp, max_dist.get_d(), 1e7, r, 0.5, PRUNER_METRIC_PROBABILITY_OF_SHORTEST,
                PRUNER_GRADIENT);
      os << (g ? "," : "") << "[" << ghfs[g] << ",[";
      for (size_t i = 0; i < pp.coefficients.size(); ++i)
      {
        char buf[40];
        snprintf(buf, sizeof buf, "%.17g", pp.coefficients[i]);
        os << (i ? "," : "") << buf;
      }
      char pb[40];
      snprintf(pb, sizeof pb, "%.17g", std::min(1.0, std::max(1e-9, pp.expectation)));
      os << "]," << pb << "]";
    }
    os << "]}";
  }
  os << "]\n";
  std::cout << os.str();
  return 0;
}

/* teststrat block_size linear(0|1) → the strategies tests/test_bkz.cpp builds by hand
 * (test_bkz_param :69-105: preprocessing [5] / [10] / [15] at block sizes 10 / 20 / 30, no pruning;
 * test_bkz_param_linear_pruning :116-152: the same plus LinearPruningParams(b, b/2) at block_size),
 * in load_strategies_json's format */
static int cmd_teststrat(int argc, char **argv)
{
  if (argc < 4)
    return 2;
  int block_size = atoi(argv[2]), linear = atoi(argv[3]);
  std::ostringstream os;
  os << "[";
  for (int b = 0; b <= block_size; ++b)
  {
    os << (b ? ",\n" : "") << "{\"block_size\":" << b << ",\"preprocessing_block_sizes\":[";
    if (b == 10 && !(linear && b == block_size))
      os << 5;
    else if (b == 20 && !(linear && b == block_size))
      os << 10;
    else if (b == 30 && !(linear && b == block_size))
      os << 15;
    os << "],\"pruning_parameters\":[";
    if (linear && b == block_size)
    {
      PruningParams pp = PruningParams::LinearPruningParams(block_size, block_size / 2);
      char buf[40];
      os << "[" << pp.gh_factor << ",[";
      for (size_t i = 0; i < pp.coefficients.size(); ++i)
      {
        snprintf(buf, sizeof buf, "%.17g", pp.coefficients[i]);
        os << (i ? "," : "") << buf;
      }
      snprintf(buf, sizeof buf, "%.17g", pp.expectation);
      os << "]," << buf << "]";
    }
    os << "]}";
  }
  os << "]\n";
  std::cout << os.str();
  return 0;
}

/* bkztour basisfile strategies.json beta plugin.so|none → JSON: wall time of ONE BKZ-beta tour
 * (BKZ_MAX_LOOPS=1, BKZ_GH_BND 1.1) and a fingerprint of the result */
static int cmd_bkztour(int argc, char **argv)
{
  if (argc < 6)
    return 2;
  ZZ_mat<mpz_t> A;
  if (!read_basis(argv[2], A))
    return 2;
  vector<Strategy> strategies = load_strategies_json(argv[3]);
  int beta                    = atoi(argv[4]);
  std::string plug            = argv[5];
  if (plug == "none")
    set_external_enumerator(nullptr);
  else if (plug == "enumlib")
    ;  // the reference's default plugin stays installed
  else
    set_external_enumerator(load_plugin(plug.c_str()));
  BKZParam par(beta, strategies);
  par.flags     = BKZ_MAX_LOOPS | BKZ_GH_BND;
  par.max_loops = 1;
  par.gh_factor = 1.1;
  typedef void (*tt_get_fn)(double *);
  typedef void (*tt_reset_fn)(void);
  tt_get_fn tt_get     = (tt_get_fn)dlsym(RTLD_DEFAULT, "tour_timers_get");
  tt_reset_fn tt_reset = (tt_reset_fn)dlsym(RTLD_DEFAULT, "tour_timers_reset");
  const bool have_timers = tt_get && tt_reset;
  double tt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (have_timers)
    tt_reset();
  auto t0       = std::chrono::steady_clock::now();
  int status    = bkz_reduction(&A, NULL, par, FT_DOUBLE, 0);
  double secs   = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (have_timers)
    tt_get(tt);
  ZZ_mat<mpz_t> U, UT;
  MatGSO<ZT, FT> M(A, U, UT, GSO_ROW_EXPO);
  M.update_gso();
  FT r0;
  M.get_r(r0, 0, 0);
  double slope = M.get_current_slope(0, A.get_rows());
  // order-independent fingerprint of the basis
  unsigned long long fp = 1469598103934665603ull;
  for (int i = 0; i < A.get_rows(); ++i)
    for (int j = 0; j < A.get_cols(); ++j)
      fp = (fp ^ (unsigned long long)A(i, j).get_si()) * 1099511628211ull;
  // the reference's own acceptance predicate on the OUTPUT basis (lll.cpp:226-257, what
  // tests/test_lll.cpp asserts), evaluated at 256 bits so that it judges the basis, not the floats
  int lll_red = -1;
  double logvol = 0.0;
  {
    int old_prec = FP_NR<mpfr_t>::set_prec(256);
    {
      ZZ_mat<mpz_t> U2, UT2;
      MatGSO<Z_NR<mpz_t>, FP_NR<mpfr_t>> M2(A, U2, UT2, 0);
      lll_red = is_lll_reduced<Z_NR<mpz_t>, FP_NR<mpfr_t>>(M2, LLL_DEF_DELTA, LLL_DEF_ETA) ? 1 : 0;
      FP_NR<mpfr_t> f;
      for (int i = 0; i < A.get_rows(); ++i)
      {
        M2.get_r(f, i, i);
        f.log(f);
        logvol += f.get_d();
      }
    }
    FP_NR<mpfr_t>::set_prec(old_prec);
  }
  if (getenv("REFDRV_DUMP_BASIS"))
  {
    std::ofstream os(getenv("REFDRV_DUMP_BASIS"));
    os << A << std::endl;
  }
  // the split of the tour's wall time, when oracle/_ref/libtour_timers.so is preloaded (tour_timers.cpp)
  char split[512] = "";
  if (have_timers)
  {
    snprintf(split, sizeof split,
             ",\"breakdown\":{\"host_lll_s\":%.3f,\"host_lll_calls\":%.0f,\"cpu_enum_s\":%.3f,\"cpu_enum_calls\":%.0f,"
             "\"plugin_enum_s\":%.3f,\"plugin_enum_calls\":%.0f,\"plugin_declined_s\":%.3f,\"plugin_declined_calls\":%.0f,"
             "\"other_s\":%.3f}",
             tt[0], tt[1], tt[2], tt[3], tt[4], tt[5], tt[6], tt[7], secs - tt[0] - tt[2] - tt[4] - tt[6]);
  }
  printf("{\"plugin\":\"%s\",\"beta\":%d,\"status\":%d,\"tour_seconds\":%.3f,\"r00\":%.17g,"
         "\"slope\":%.9f,\"is_lll_reduced\":%d,\"log_volume\":%.12g,\"basis_fnv\":\"%016llx\"%s}\n",
         plug.c_str(), beta, status, secs, r0.get_d(), slope, lll_red, logvol, fp, split);
  return (status == RED_SUCCESS || status == RED_BKZ_LOOPS_LIMIT) ? 0 : 1;
}

/* prunefix basisfile first d gh_factor preproc_cost target metric flags
 *   → JSON: the reference's prune<FP_NR<double>> (pruner/pruner.cpp:190-203) on the r-profile of block
 *   [first, first + d) of the basis, radius = gh_factor x Gaussian heuristic of the block (the way the
 *   survey built its strategies); plus svp_probability<FP_NR<double>> and Pruner::single_enum_cost /
 *   measure_metric of the result AND of LinearPruningParams(d, d/2) through the public API.  Doubles in
 *   hex: the pruner of the product (fplll_amd/csrc/pruner_search.hip) is compared bit for bit. */
static void put_hex(const char *name, const vector<double> &v, bool last = false)
{
  printf("\"%s\":[", name);
  for (size_t i = 0; i < v.size(); ++i)
    printf("%s\"%a\"", i ? "," : "", v[i]);
  printf("]%s", last ? "" : ",");
}
static int cmd_prunefix(int argc, char **argv)
{
  if (argc < 10)
  {
    fprintf(stderr, "usage: prunefix basisfile first d gh_factor preproc_cost target metric flags\n");
    return 2;
  }
  ZZ_mat<mpz_t> A, U, UT;
  if (!read_basis(argv[2], A))
    return 2;
  const int first = atoi(argv[3]), d = atoi(argv[4]);
  const double ghf = atof(argv[5]), preproc = atof(argv[6]), target = atof(argv[7]);
  const int metric = atoi(argv[8]), flags = atoi(argv[9]);
  MatGSO<ZT, FT> M(A, U, UT, GSO_ROW_EXPO);
  M.update_gso();
  vector<double> r;
  for (int i = 0; i < d; ++i)
  {
    FT t;
    M.get_r(t, first + i, first + i);
    r.push_back(t.get_d());
  }
  long expo;
  FT max_dist = M.get_r_exp(first, first, expo);
  max_dist *= 1e10;  // a huge radius: the Gaussian-heuristic bound decides
  FT root_det = M.get_root_det(first, first + d);
  adjust_radius_to_gh_bound(max_dist, expo, d, root_det, ghf);
  const double radius = max_dist.get_d() * std::pow(2.0, (double)expo);
  PruningParams pp;
  auto t0 = std::chrono::steady_clock::now();
  prune<FT>(pp, radius, preproc, r, target, (PrunerMetric)metric, flags);
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("{\"d\":%d,\"radius\":\"%a\",\"preproc_cost\":\"%a\",\"target\":\"%a\",\"metric\":%d,\"flags\":%d,"
         "\"ref_seconds\":%.6f,\"expectation\":\"%a\",\"gh_factor\":\"%a\",",
         d, radius, preproc, target, metric, flags, secs, pp.expectation, pp.gh_factor);
  put_hex("gso_r", r);
  put_hex("coefficients", pp.coefficients);
  put_hex("detailed_cost", pp.detailed_cost);
  // the evaluation functions on a second coefficient vector, through the public API
  vector<double> lin = PruningParams::LinearPruningParams(d, d / 2).coefficients;
  Pruner<FT> ev(radius, preproc, r, target, (PrunerMetric)metric, 0);
  vector<double> lin_detail;
  const double lin_cost = ev.single_enum_cost(lin, &lin_detail);
  const double lin_metric = ev.measure_metric(lin);
  const double lin_svp    = svp_probability<FT>(lin).get_d();
  printf("\"lin_cost\":\"%a\",\"lin_metric\":\"%a\",\"lin_svp_probability\":\"%a\",", lin_cost, lin_metric, lin_svp);
  put_hex("lin", lin);
  put_hex("lin_detailed_cost", lin_detail, true);
  printf("}\n");
  return 0;
}

/* gsoutil basisfile  → JSON: stored r(i,i) and row_expo of MatGSO<long,double> (GSO_ROW_EXPO) after
 * update_gso, and the reference's get_current_slope / get_log_det / get_root_det / get_slide_potential
 * (gso_interface.cpp:197-258) on a list of ranges, plus adjust_radius_to_gh_bound (:260-276) — what
 * fphip_gso_util_* (fplll_amd/csrc/gso_util_host.hip) is compared with, bit for bit. */
static int cmd_gsoutil(int argc, char **argv)
{
  if (argc < 3)
    return 2;
  ZZ_mat<mpz_t> A;
  if (!read_basis(argv[2], A))
    return 2;
  ZZ_mat<long> b, u, ut;
  b.resize(A.get_rows(), A.get_cols());
  for (int i = 0; i < A.get_rows(); ++i)
    for (int j = 0; j < A.get_cols(); ++j)
      b(i, j) = A(i, j).get_si();
  MatGSO<Z_NR<long>, FP_NR<double>> M(b, u, ut, GSO_ROW_EXPO);
  M.update_gso();
  const int d = M.d;
  printf("{\"d\":%d,\"r_diag\":[", d);
  for (int i = 0; i < d; ++i)
  {
    long e;
    const FP_NR<double> &f = M.get_r_exp(i, i, e);
    printf("%s\"%a\"", i ? "," : "", f.get_d());
  }
  printf("],\"row_expo\":[");
  for (int i = 0; i < d; ++i)
  {
    long e;
    M.get_r_exp(i, i, e);
    printf("%s%ld", i ? "," : "", e / 2);
  }
  printf("],\"is_lll_reduced\":%d,\"is_lll_reduced_d0999_e0501\":%d", (int)is_lll_reduced<Z_NR<long>, FP_NR<double>>(M, LLL_DEF_DELTA, LLL_DEF_ETA),
         (int)is_lll_reduced<Z_NR<long>, FP_NR<double>>(M, 0.999, 0.501));
  if (d <= 64)
  {  // the stored matrices (what fphip_gso_get_mu / _get_r hand out): mu(i,j), r(i,j) for j <= i, else 0
    printf(",\"mu\":[");
    for (int i = 0; i < d; ++i)
      for (int j = 0; j < d; ++j)
      {
        long e;
        printf("%s\"%a\"", (i || j) ? "," : "", j < i ? M.get_mu_exp(i, j, e).get_d() : 0.0);
      }
    printf("],\"r\":[");
    for (int i = 0; i < d; ++i)
      for (int j = 0; j < d; ++j)
      {
        long e;
        printf("%s\"%a\"", (i || j) ? "," : "", j <= i ? M.get_r_exp(i, j, e).get_d() : 0.0);
      }
    printf("]");
  }
  printf(",\"queries\":[");
  const int ranges[][3] = {{0, d, 10}, {0, d, 20}, {5, d - 3, 7}, {d / 2, d, 12}, {0, 30, 30}, {10, 11, 1},
                           {3, 60, 19}, {0, d, d}, {-4, d + 9, 25}};
  bool first = true;
  for (auto &q : ranges)
  {
    const int a = q[0], e = q[1], bs = q[2];
    const int ca = std::max(0, a), ce = std::min(d, e);
    const double slope = (ce - ca >= 2) ? M.get_current_slope(ca, ce) : 0.0;
    const double ld    = M.get_log_det(a, e).get_d();
    const double rd    = M.get_root_det(a, e).get_d();
    const double pot   = M.get_slide_potential(ca, ce, bs).get_d();
    long expo;
    FP_NR<double> max_dist = M.get_r_exp(ca, ca, expo);
    FP_NR<double> root     = M.get_root_det(a, e);
    FP_NR<double> adj      = max_dist;
    adjust_radius_to_gh_bound(adj, expo, ce - ca, root, 1.1);
    FP_NR<double> big = max_dist;
    big *= 1e10;
    adjust_radius_to_gh_bound(big, expo, ce - ca, root, 1.05);
    printf("%s{\"start\":%d,\"end\":%d,\"block_size\":%d,\"slope\":\"%a\",\"log_det\":\"%a\",\"root_det\":\"%a\","
           "\"slide_potential\":\"%a\",\"max_dist\":\"%a\",\"expo\":%ld,\"adjusted_1.1\":\"%a\","
           "\"adjusted_big_1.05\":\"%a\"}",
           first ? "" : ",", a, e, bs, slope, ld, rd, pot, max_dist.get_d(), expo, adj.get_d(), big.get_d());
    first = false;
  }
  printf("]}\n");
  return 0;
}

/* stratdump strategies.json  → JSON: the strategies as load_strategies_json (bkz_param.cpp:82-157) holds
 * them, flattened like the "strategies" member of the BKZ fixtures (block sizes the file skips get the
 * empty strategy with one default PruningParams) — what fplll_amd.strategies.load_strategies_json is
 * compared with. */
static int cmd_stratdump(int argc, char **argv)
{
  if (argc < 3)
    return 2;
  vector<Strategy> loaded = load_strategies_json(argv[2]);
  auto hx = [](double v) { char b[64]; snprintf(b, sizeof b, "\"%a\"", v); return std::string(b); };
  std::ostringstream po, pv, ro, rg, re, co, cv;
  int np = 0, npr = 0, nc = 0;
  for (size_t bs = 0; bs < loaded.size(); ++bs)
  {
    po << (bs ? "," : "") << np;
    ro << (bs ? "," : "") << npr;
    for (int pb : loaded[bs].preprocessing_block_sizes)
      pv << (np++ ? "," : "") << pb;
    for (const PruningParams &pp : loaded[bs].pruning_parameters)
    {
      rg << (npr ? "," : "") << hx(pp.gh_factor);
      re << (npr ? "," : "") << hx(pp.expectation);
      co << (npr ? "," : "") << nc;
      for (double c : pp.coefficients)
        cv << (nc++ ? "," : "") << hx(c);
      ++npr;
    }
  }
  po << "," << np;
  ro << "," << npr;
  co << (npr ? "," : "") << nc;
  printf("{\"max_block_size\":%d,\"pre_off\":[%s],\"pre\":[%s],\"prune_off\":[%s],\"prune_gh\":[%s],"
         "\"prune_exp\":[%s],\"coeff_off\":[%s],\"coeff\":[%s]",
         (int)loaded.size() - 1, po.str().c_str(), pv.str().c_str(), ro.str().c_str(), rg.str().c_str(),
         re.str().c_str(), co.str().c_str(), cv.str().c_str());
  // Strategy::get_pruning (bkz_param.cpp:62-79) on a few (radius, gh) pairs per block size
  printf(",\"get_pruning\":[");
  bool first = true;
  const double ratios[] = {0.5, 0.97, 1.0, 1.05, 1.12, 1.27, 1.6, 3.0};
  for (size_t bs = 0; bs < loaded.size(); ++bs)
    for (double q : ratios)
    {
      const PruningParams &pp = loaded[bs].get_pruning(q * 1234.5, 1234.5);
      printf("%s[%d,%s,%d]", first ? "" : ",", (int)bs, hx(q).c_str(), (int)(&pp - &loaded[bs].pruning_parameters[0]));
      first = false;
    }
  printf("]}\n");
  return 0;
}

/* prunemulti basisfile first d count stride gh_factor preproc_cost target metric flags
 *   → JSON: the reference's prune<FP_NR<double>> over SEVERAL bases (pruner/pruner.cpp:214-227,
 *   Pruner::load_basis_shapes pruner_util.cpp:66-92): the r-profiles of the `count` blocks
 *   [first + i stride, first + i stride + d), radius from the first block as in prunefix. */
static int cmd_prunemulti(int argc, char **argv)
{
  if (argc < 12)
  {
    fprintf(stderr, "usage: prunemulti basisfile first d count stride gh_factor preproc_cost target metric flags\n");
    return 2;
  }
  ZZ_mat<mpz_t> A, U, UT;
  if (!read_basis(argv[2], A))
    return 2;
  const int first = atoi(argv[3]), d = atoi(argv[4]), count = atoi(argv[5]), stride = atoi(argv[6]);
  const double ghf = atof(argv[7]), preproc = atof(argv[8]), target = atof(argv[9]);
  const int metric = atoi(argv[10]), flags = atoi(argv[11]);
  MatGSO<ZT, FT> M(A, U, UT, GSO_ROW_EXPO);
  M.update_gso();
  vector<vector<double>> rs;
  for (int c = 0; c < count; ++c)
  {
    vector<double> r;
    for (int i = 0; i < d; ++i)
    {
      FT t;
      M.get_r(t, first + c * stride + i, first + c * stride + i);
      r.push_back(t.get_d());
    }
    rs.push_back(r);
  }
  long expo;
  FT max_dist = M.get_r_exp(first, first, expo);
  max_dist *= 1e10;
  FT root_det = M.get_root_det(first, first + d);
  adjust_radius_to_gh_bound(max_dist, expo, d, root_det, ghf);
  const double radius = max_dist.get_d() * std::pow(2.0, (double)expo);
  PruningParams pp;
  prune<FT>(pp, radius, preproc, rs, target, (PrunerMetric)metric, flags);
  printf("{\"d\":%d,\"count\":%d,\"radius\":\"%a\",\"preproc_cost\":\"%a\",\"target\":\"%a\",\"metric\":%d,"
         "\"flags\":%d,\"expectation\":\"%a\",\"gh_factor\":\"%a\",",
         d, count, radius, preproc, target, metric, flags, pp.expectation, pp.gh_factor);
  for (int c = 0; c < count; ++c)
  {
    char name[32];
    snprintf(name, sizeof name, "gso_r_%d", c);
    put_hex(name, rs[c]);
  }
  put_hex("coefficients", pp.coefficients);
  put_hex("detailed_cost", pp.detailed_cost, true);
  printf("}\n");
  return 0;
}

/* basisstat basisfile  → JSON: the reference's is_lll_reduced (256-bit GSO), slope of log r_ii
 * (gso_interface.cpp:198-218), log-volume, r_00 — the acceptance test of a tour whose enumerations
 * ran in another order than the reference's (a pruned shrinking-radius walk is order dependent) */
static int cmd_basisstat(int argc, char **argv)
{
  if (argc < 3)
    return 2;
  ZZ_mat<mpz_t> A;
  if (!read_basis(argv[2], A))
    return 2;
  // everything at 256 bits: a lattice whose GSO is beyond plain doubles (BASELINE config 5's) must not
  // turn the figures into NaN
  int lll_red;
  double logvol = 0.0, r00 = 0.0, slope = 0.0;
  int old_prec  = FP_NR<mpfr_t>::set_prec(256);
  {
    ZZ_mat<mpz_t> U2, UT2;
    MatGSO<Z_NR<mpz_t>, FP_NR<mpfr_t>> M2(A, U2, UT2, 0);
    lll_red = is_lll_reduced<Z_NR<mpz_t>, FP_NR<mpfr_t>>(M2, LLL_DEF_DELTA, LLL_DEF_ETA) ? 1 : 0;
    FP_NR<mpfr_t> f;
    for (int i = 0; i < A.get_rows(); ++i)
    {
      M2.get_r(f, i, i);
      if (i == 0)
        r00 = f.get_d();
      f.log(f);
      logvol += f.get_d();
    }
    slope = M2.get_current_slope(0, A.get_rows());
  }
  FP_NR<mpfr_t>::set_prec(old_prec);
  printf("{\"d\":%d,\"r00\":%.17g,\"slope\":%.9f,\"is_lll_reduced\":%d,\"log_volume\":%.12g}\n",
         A.get_rows(), r00, slope, lll_red, logvol);
  return 0;
}

/* enumtime basisfile first d pruning rfac mode threads [radius_scale]
 *   CPU-baseline leg: one Enumeration::enumerate call of block [first, first+d)
 *   of the basis in `basisfile`, radius min(rfac r, 1.1 GH) (x radius_scale), timed with a steady
 *   clock.  mode = internal (set_external_enumerator(nullptr): fplll's own recursive enumerator,
 *   fplll counting rule) | enumlib (the reference's default plugin, fplll/enum-parallel/, which
 *   counts nodes BEFORE the bound test) with set_threads(threads). */
static int cmd_enumtime(int argc, char **argv)
{
  if (argc < 9)
  {
    fprintf(stderr, "usage: enumtime basisfile first d pruning rfac internal|enumlib threads [scale]\n");
    return 2;
  }
  ZZ_mat<mpz_t> A, U, UT;
  if (!read_basis(argv[2], A))
    return 2;
  int first = atoi(argv[3]), d = atoi(argv[4]);
  std::string prspec = argv[5];
  double rfac        = atof(argv[6]);
  std::string mode   = argv[7];
  int threads        = atoi(argv[8]);
  double scale       = argc > 9 ? atof(argv[9]) : 1.0;
  MatGSO<ZT, FT> M(A, U, UT, GSO_ROW_EXPO);
  M.update_gso();
  long expo;
  FT max_dist = M.get_r_exp(first, first, expo);
  max_dist *= rfac;
  if (d > 30 && rfac <= 1.0)
  {
    FT root_det = M.get_root_det(first, first + d);
    adjust_radius_to_gh_bound(max_dist, expo, d, root_det, 1.1);
  }
  max_dist *= scale;
  vector<double> pruning =
      make_pruning(prspec, d, &M, first, max_dist.get_d() * std::pow(2.0, (double)expo));
  int used_threads = 1;
  if (mode == "internal")
    set_external_enumerator(nullptr);
  else
    used_threads = set_threads(threads);  // enumlib is the default plugin of this build
  FastEvaluator<FT> ev(1, EVALSTRATEGY_BEST_N_SOLUTIONS, false);
  Enumeration<ZT, FT> E(M, ev);
  auto t0 = std::chrono::steady_clock::now();
  E.enumerate(first, first + d, max_dist, expo, vector<FT>(), vector<enumxt>(), pruning);
  double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  uint64_t tot = 0;
  auto nodes   = E.get_nodes_array();
  for (int i = 0; i <= d; ++i)
    tot += nodes[i];
  double best = ev.empty() ? 0.0 : ev.begin()->first.get_d();
  printf("{\"mode\":\"%s\",\"threads\":%d,\"first\":%d,\"d\":%d,\"pruning\":\"%s\",\"scale\":%g,"
         "\"total_nodes\":%llu,\"seconds\":%.6f,\"best_sqnorm\":%.17g}\n",
         mode.c_str(), used_threads, first, d, prspec.c_str(), scale, (unsigned long long)tot, secs, best);
  return 0;
}

/* sweeptime basisfile reps
 *   CPU-baseline leg of the GSO roofline: LLLReduction::size_reduction(0, n)
 *   (lll.h:107-122) on a freshly built MatGSO<Z_NR<long>, FP_NR<double>>(GSO_ROW_EXPO) — the types
 *   BKZ runs on (bkz.cpp:816-829) — of the basis in `basisfile`, wall clock, mean of `reps`. */
static int cmd_sweeptime(int argc, char **argv)
{
  if (argc < 4)
    return 2;
  ZZ_mat<mpz_t> A;
  if (!read_basis(argv[2], A))
    return 2;
  int reps = atoi(argv[3]);
  int n = A.get_rows(), m = A.get_cols();
  double tot = 0.0;
  int ok     = 1;
  unsigned long long fp = 0;
  for (int r = 0; r < reps; ++r)
  {
    ZZ_mat<long> B(n, m), U, UT;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < m; ++j)
        B(i, j) = A(i, j).get_si();
    MatGSO<Z_NR<long>, FP_NR<double>> M(B, U, UT, GSO_ROW_EXPO);
    LLLReduction<Z_NR<long>, FP_NR<double>> L(M, LLL_DEF_DELTA, LLL_DEF_ETA, LLL_DEFAULT);
    auto t0 = std::chrono::steady_clock::now();
    bool good = L.size_reduction(0, n);
    tot += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    ok &= good ? 1 : 0;
    fp = 1469598103934665603ull;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < m; ++j)
        fp = (fp ^ (unsigned long long)B(i, j).get_si()) * 1099511628211ull;
  }
  printf("{\"rows\":%d,\"cols\":%d,\"reps\":%d,\"ok\":%d,\"seconds_per_sweep\":%.9f,"
         "\"basis_fnv\":\"%016llx\"}\n", n, m, reps, ok, tot / reps, fp);
  return ok ? 0 : 1;
}

/* hhfix n k bits seed perturb row_expo → JSON: input basis (long) and the reference's Householder
 * R-factor after refresh_R_bf() + update_R() (MatHouseholder<Z_NR<long>,FP_NR<double>>) */
static int cmd_hhfix(int argc, char **argv)
{
  if (argc < 8)
    return 2;
  int n = atoi(argv[2]), k = atoi(argv[3]), bits = atoi(argv[4]), seed = atoi(argv[5]);
  int perturb = atoi(argv[6]), rexp = atoi(argv[7]);
  ZZ_mat<mpz_t> A;
  make_basis(A, n, k, bits, seed, 0);
  ZZ_mat<long> b(n, n), u, ut;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
      b(i, j) = A(i, j).get_si();
  uint64_t lcg = 0x9E3779B97F4A7C15ull ^ (uint64_t)seed;
  for (int i = 1; i < n && perturb > 0; ++i)
    for (int t = 0; t < perturb; ++t)
    {
      lcg    = lcg * 6364136223846793005ull + 1442695040888963407ull;
      int j  = (int)((lcg >> 33) % (uint64_t)i);
      lcg    = lcg * 6364136223846793005ull + 1442695040888963407ull;
      long c = (long)((lcg >> 33) % 7) - 3;
      for (int col = 0; col < n; ++col)
        b(i, col) = b(i, col).get_si() + c * b(j, col).get_si();
    }
  MatHouseholder<Z_NR<long>, FP_NR<double>> H(b, u, ut, rexp ? HOUSEHOLDER_ROW_EXPO : 0);
  H.refresh_R_bf();
  H.update_R();
  std::ostringstream os;
  os << "{\n\"desc\":\"qary n=" << n << " k=" << k << " bits=" << bits << " seed=" << seed
     << " LLL then " << perturb << " row ops per row, row_expo=" << rexp << "\",\n\"d\":" << n
     << ",\n\"n\":" << n << ",\n\"row_expo_on\":" << rexp << ",\n\"b_in\":[";
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
      os << ((i || j) ? "," : "") << b(i, j).get_si();
  os << "],\n";
  vector<long> expo;
  const Matrix<FP_NR<double>> &R = H.get_R(expo);
  dump_mat_hex(os, "R", R, n, true);
  os << "\"row_expo\":[";
  for (int i = 0; i < n; ++i)
    os << (i ? "," : "") << expo[i];
  os << "]\n}\n";
  std::cout << os.str();
  return 0;
}

/* hhsr n k bits seed perturb row_expo kappa end start → JSON: the hhfix input, then the reference's
 * MatHouseholder::size_reduce(kappa, end, start) (householder.cpp:402-451) on the state update_R() left: the flag it
 * returns, row kappa of the basis and row kappa of R afterwards (R(kappa, c) for c < end is what line 5 of
 * Algorithm 3 leaves — "not the correct R[k]", the row is invalidated) */
static int cmd_hhsr(int argc, char **argv)
{
  if (argc < 11)
    return 2;
  int n = atoi(argv[2]), k = atoi(argv[3]), bits = atoi(argv[4]), seed = atoi(argv[5]);
  int perturb = atoi(argv[6]), rexp = atoi(argv[7]);
  int kappa = atoi(argv[8]), end = atoi(argv[9]), start = atoi(argv[10]);
  ZZ_mat<mpz_t> A;
  make_basis(A, n, k, bits, seed, 0);
  ZZ_mat<long> b(n, n), u, ut;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
      b(i, j) = A(i, j).get_si();
  uint64_t lcg = 0x9E3779B97F4A7C15ull ^ (uint64_t)seed;
  for (int i = 1; i < n && perturb > 0; ++i)
    for (int t = 0; t < perturb; ++t)
    {
      lcg    = lcg * 6364136223846793005ull + 1442695040888963407ull;
      int j  = (int)((lcg >> 33) % (uint64_t)i);
      lcg    = lcg * 6364136223846793005ull + 1442695040888963407ull;
      long c = (long)((lcg >> 33) % 7) - 3;
      for (int col = 0; col < n; ++col)
        b(i, col) = b(i, col).get_si() + c * b(j, col).get_si();
    }
  std::ostringstream os;
  os << "{\n\"desc\":\"qary n=" << n << " k=" << k << " bits=" << bits << " seed=" << seed << " LLL then "
     << perturb << " row ops per row, row_expo=" << rexp << "; update_R(), size_reduce(" << kappa << "," << end
     << "," << start << ")\",\n\"d\":" << n << ",\n\"n\":" << n << ",\n\"row_expo_on\":" << rexp
     << ",\n\"kappa\":" << kappa << ",\n\"end\":" << end << ",\n\"start\":" << start << ",\n\"b_in\":[";
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
      os << ((i || j) ? "," : "") << b(i, j).get_si();
  os << "],\n";
  MatHouseholder<Z_NR<long>, FP_NR<double>> H(b, u, ut, rexp ? HOUSEHOLDER_ROW_EXPO : 0);
  H.refresh_R_bf();
  H.update_R();
  const bool reduced = H.size_reduce(kappa, end, start);
  os << "\"reduced\":" << (reduced ? 1 : 0) << ",\n\"b_row\":[";
  for (int j = 0; j < n; ++j)
    os << (j ? "," : "") << b(kappa, j).get_si();
  os << "],\n\"R_row\":[";
  vector<long> expo;
  const Matrix<FP_NR<double>> &R = H.get_R(expo);
  for (int j = 0; j < n; ++j)
    os << (j ? "," : "") << hexd(R(kappa, j).get_d());
  os << "],\n\"row_expo\":[";
  for (int i = 0; i < n; ++i)
    os << (i ? "," : "") << expo[i];
  os << "]\n}\n";
  std::cout << os.str();
  return 0;
}

/* hhmp basisfile prec → JSON: the reference's Householder R-factor of the basis in `basisfile`
 * computed with FP_NR<mpfr_t> at `prec` bits (MatHouseholder<Z_NR<mpz_t>, FP_NR<mpfr_t>>,
 * householder.cpp:587-589; prec = 106 is PREC_DD, defs.h:140): R(i, j <= i) as decimal strings with
 * 40 significant digits.  The golden values the double-double device path is checked against (libqd,
 * hence FP_NR<dd_real>, is absent: SURVEY.md 8(c)). */
static int cmd_hhmp(int argc, char **argv)
{
  if (argc < 4)
    return 2;
  ZZ_mat<mpz_t> A, U, UT;
  if (!read_basis(argv[2], A))
    return 2;
  const int prec = atoi(argv[3]);
  const int old  = FP_NR<mpfr_t>::set_prec(prec);
  {
    MatHouseholder<Z_NR<mpz_t>, FP_NR<mpfr_t>> H(A, U, UT, 0);
    H.refresh_R_bf();
    H.update_R();
    const int d = A.get_rows();
    std::ostringstream os;
    os << "{\"prec\":" << prec << ",\"d\":" << d << ",\"n\":" << A.get_cols() << ",\"R\":[";
    char buf[128];
    for (int i = 0; i < d; ++i)
      for (int j = 0; j <= i; ++j)
      {
        FP_NR<mpfr_t> f;
        H.get_R(f, i, j);
        mpfr_snprintf(buf, sizeof buf, "%.40Re", f.get_data());
        os << ((i || j) ? "," : "") << "\"" << buf << "\"";
      }
    os << "]}\n";
    std::cout << os.str();
  }
  FP_NR<mpfr_t>::set_prec(old);
  return 0;
}

/* hlllmp basisfile prec → JSON: the reference's hlll_reduction of the basis in `basisfile` with
 * FP_NR<mpfr_t> at `prec` bits (LM_PROVED, FT_MPFR — `fplll -a hlll -m proved -f mpfr -p prec`):
 * status, seconds, the output basis.  prec = 106 is the stand-in for the dd_real run the reference
 * cannot make here (no libqd). */
static int cmd_hlllmp(int argc, char **argv)
{
  if (argc < 4)
    return 2;
  ZZ_mat<mpz_t> A;
  if (!read_basis(argv[2], A))
    return 2;
  const int prec = atoi(argv[3]);
  auto t0        = std::chrono::steady_clock::now();
  int status = hlll_reduction(A, LLL_DEF_DELTA, LLL_DEF_ETA, HLLL_DEF_THETA, HLLL_DEF_C, LM_PROVED, FT_MPFR, prec,
                              LLL_DEFAULT, false);
  double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("{\"prec\":%d,\"status\":%d,\"seconds\":%.3f,\"d\":%d,\"n\":%d,\"b_out\":[", prec, status, secs,
         A.get_rows(), A.get_cols());
  for (int i = 0; i < A.get_rows(); ++i)
    for (int j = 0; j < A.get_cols(); ++j)
      printf("%s%ld", (i || j) ? "," : "", A(i, j).get_si());
  printf("]}\n");
  return 0;
}

/* ishlll basisfile prec → {"reduced":0|1}: the reference's own predicate is_hlll_reduced
 * (hlll.cpp:507-585; what tests/test_hlll.cpp asserts), evaluated with FP_NR<mpfr_t> at `prec` bits
 * through hlll_reduction(..., nolll = true) (wrapper.cpp: "just verify if the basis is reduced"). */
static int cmd_ishlll(int argc, char **argv)
{
  if (argc < 4)
    return 2;
  ZZ_mat<mpz_t> A;
  if (!read_basis(argv[2], A))
    return 2;
  int status = hlll_reduction(A, LLL_DEF_DELTA, LLL_DEF_ETA, HLLL_DEF_THETA, HLLL_DEF_C, LM_PROVED, FT_MPFR,
                              atoi(argv[3]), LLL_DEFAULT, true);
  printf("{\"reduced\":%d,\"status\":%d}\n", status == RED_SUCCESS ? 1 : 0, status);
  return 0;
}

/* dumpbasis n k bits seed bkz_pre  → the reduced basis in fplll's text format on stdout */
static int cmd_dumpbasis(int argc, char **argv)
{
  if (argc < 7)
    return 2;
  ZZ_mat<mpz_t> A;
  make_basis(A, atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]));
  std::cout << A << std::endl;
  return 0;
}

int main(int argc, char **argv)
{
  if (argc < 2)
  {
    fprintf(stderr, "commands: enumfix | plugin\n");
    return 2;
  }
  std::string cmd = argv[1];
  if (cmd == "enumfix")
    return cmd_enumfix(argc, argv);
  if (cmd == "plugin")
    return cmd_plugin(argc, argv);
  if (cmd == "dumpbasis")
    return cmd_dumpbasis(argc, argv);
  if (cmd == "gsofix")
    return cmd_gsofix(argc, argv);
  if (cmd == "hhfix")
    return cmd_hhfix(argc, argv);
  if (cmd == "lllfix")
    return cmd_lllfix(argc, argv);
  if (cmd == "hlllfix")
    return cmd_hlllfix(argc, argv);
  if (cmd == "bkzfix")
    return cmd_bkzfix(argc, argv);
  if (cmd == "genstrat")
    return cmd_genstrat(argc, argv);
  if (cmd == "teststrat")
    return cmd_teststrat(argc, argv);
  if (cmd == "ishlll")
    return cmd_ishlll(argc, argv);
  if (cmd == "hlllmp")
    return cmd_hlllmp(argc, argv);
  if (cmd == "hhsr")
    return cmd_hhsr(argc, argv);
  if (cmd == "hhmp")
    return cmd_hhmp(argc, argv);
  if (cmd == "enumtime")
    return cmd_enumtime(argc, argv);
  if (cmd == "sweeptime")
    return cmd_sweeptime(argc, argv);
  if (cmd == "bkztour")
    return cmd_bkztour(argc, argv);
  if (cmd == "basisstat")
    return cmd_basisstat(argc, argv);
  if (cmd == "prunefix")
    return cmd_prunefix(argc, argv);
  if (cmd == "prunemulti")
    return cmd_prunemulti(argc, argv);
  if (cmd == "stratdump")
    return cmd_stratdump(argc, argv);
  if (cmd == "gsoutil")
    return cmd_gsoutil(argc, argv);
  fprintf(stderr, "unknown command %s\n", cmd.c_str());
  return 2;
}
