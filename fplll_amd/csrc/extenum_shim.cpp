// extenum_shim.cpp — host C++ adapter that lets fplll use the HIP enumerator UNCHANGED:
//
//     fplll::set_external_enumerator(fplll_hip_extenum);      // fplll/enum/enumerate_ext.h:100
//
// fplll's hook is a C++ ABI (std::function / std::array by value,
// fplll/enum/enumerate_ext_api.h:52-92), so this thin shim — compiled with the same libstdc++ as
// fplll — forwards to the C ABI of libfplll_hip.so (include/fplll_hip.h).  It needs no fplll
// header: the signature below is spelled with the std types the reference's typedef
// `extenum_fc_enumerate` uses (enumerate_ext_api.h:25-26, 88-92).
//
// Protocol implemented (enumerate_ext.cpp:48-167): call cbfunc once with mutranspose=true to
// receive mu^T / rdiag / pruning; report candidates through cbsol, which returns the new bound;
// return per-level node counts, or [0] = ~0 to decline so fplll falls back to its own enumerator
// (dual, dim > 128, or any device error).

#include <array>
#include <cstdint>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <vector>

#include "../../include/fplll_hip.h"

typedef double enumf;
typedef void(cb_set_config_t)(enumf *mu, std::size_t mudim, bool mutranspose, enumf *rdiag,
                              enumf *pruning);
typedef enumf(cb_process_sol_t)(enumf dist, enumf *sol);
typedef void(cb_process_subsol_t)(enumf dist, enumf *subsol, int offset);
typedef std::array<std::uint64_t, 1024> nodes_array_t;  // FPLLL_EXTENUM_MAX_EXTENUM_DIM

namespace
{
fphip_ctx *g_ctx = nullptr;
struct Totals  // FPLLL_HIP_STATS=1: printed when the process exits
{
  double secs = 0, kernel_ms = 0;
  unsigned long long calls = 0, declined = 0, nodes = 0;
  ~Totals()
  {
    if (getenv("FPLLL_HIP_STATS") && (calls || declined))
      fprintf(stderr, "[fplll_hip] %llu enumerations on the device (%llu declined): %.3f s in the plugin, "
                      "%.3f s of kernels, %.3e nodes\n",
              calls, declined, secs, kernel_ms * 1e-3, (double)nodes);
  }
} g_totals;
std::mutex g_mutex;  // fplll's global hook is process-wide and unsynchronised (enumerate_ext.cpp:32-37)

struct Trampoline
{
  std::function<cb_process_sol_t> *cbsol;
  std::function<cb_process_subsol_t> *cbsubsol;
  int dim;
  long delivered;  // candidates handed to fplll's evaluator so far
};

void subsol_trampoline(void *user, double dist, const double *subsol, int offset)
{
  Trampoline *t = static_cast<Trampoline *>(user);
  double buf[FPHIP_ENUM_MAX_DIM];
  for (int i = 0; i < t->dim; ++i)
    buf[i] = subsol[i];
  try
  {
    (*t->cbsubsol)(dist, buf, offset);
  }
  catch (...)
  {
  }
}

double sol_trampoline(void *user, double dist, const double *sol)
{
  Trampoline *t = static_cast<Trampoline *>(user);
  ++t->delivered;
  double buf[FPHIP_ENUM_MAX_DIM];
  for (int i = 0; i < t->dim; ++i)
    buf[i] = sol[i];
  try
  {
    return (*t->cbsol)(dist, buf);
  }
  catch (...)
  {
    return 0.0;  // no exception may cross the C boundary; a zero bound stops the enumeration
  }
}

fphip_ctx *context()
{
  if (!g_ctx)
  {
    const char *dev = getenv("FPLLL_HIP_DEVICE");
    if (fphip_create(dev ? atoi(dev) : 0, &g_ctx) != FPHIP_OK)
    {
      fprintf(stderr, "[fplll_hip] cannot create device context: %s\n", fphip_last_error(g_ctx));
      // keep g_ctx: later calls decline quickly; fplll falls back to its CPU enumerator
    }
  }
  return g_ctx;
}
}  // namespace

nodes_array_t fplll_hip_extenum(const int dim, enumf maxdist, std::function<cb_set_config_t> cbfunc,
                                std::function<cb_process_sol_t> cbsol,
                                std::function<cb_process_subsol_t> cbsubsol, bool dual,
                                bool findsubsols)
{
  nodes_array_t out{};
  out[0] = ~std::uint64_t(0);
  if (dim < 2 || dim > FPHIP_ENUM_MAX_DIM || dual)
    return out;
  std::lock_guard<std::mutex> lock(g_mutex);
  fphip_ctx *ctx = context();
  if (!ctx)
    return out;

  std::vector<double> mu((size_t)dim * dim, 0.0), rdiag(dim, 0.0), pruning(dim, 0.0);
  cbfunc(mu.data(), (size_t)dim, true, rdiag.data(), pruning.data());

  fphip_enum_opts opts{};
  const char *mn         = getenv("FPLLL_HIP_MIN_NODES");
  opts.min_nodes_decline = mn ? atoi(mn) : 0;
  opts.findsubsols       = findsubsols ? 1 : 0;
  Trampoline tr{&cbsol, &cbsubsol, dim, 0};
  std::vector<std::uint64_t> nodes(dim + 1, 0);
  fphip_enum_stats stats{};
  const auto t0 = std::chrono::steady_clock::now();
  int rc = fphip_enum_run(ctx, dim, maxdist, mu.data(), rdiag.data(), pruning.data(), &opts,
                          sol_trampoline, findsubsols ? subsol_trampoline : nullptr, &tr,
                          nodes.data(), &stats);
  if (rc == FPHIP_UNSUPPORTED)
  {
    g_totals.declined++;
    return out;
  }
  g_totals.secs += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  g_totals.kernel_ms += stats.kernel_ms;
  g_totals.calls++;
  g_totals.nodes += stats.total_nodes;
  if (rc != FPHIP_OK)
  {
    // fplll's protocol has one error channel: decline, upon which Enumeration::enumerate runs its
    // own enumerator from the ORIGINAL radius (enumerate.h:104-110).  If candidates had already
    // been handed to the evaluator, that second walk meets them again: harmless for BKZ's
    // FastEvaluator(max_sols = 1) (the shortest survives), but an evaluator that keeps N > 1
    // solutions may then hold the same vector twice — say so instead of failing silently.
    fprintf(stderr, "[fplll_hip] enumeration failed%s, fplll's enumerator takes over: %s\n",
            tr.delivered ? " AFTER candidates were delivered (evaluators with max_sols > 1 may now hold "
                           "duplicates)"
                         : "",
            fphip_last_error(ctx));
    return out;
  }
  out.fill(0);
  for (int i = 0; i <= dim; ++i)
    out[i] = nodes[i];
  return out;
}

// C getters so a host program can dlopen the shim without knowing the mangled name.
extern "C" void *fplll_hip_extenum_entry(void) { return (void *)&fplll_hip_extenum; }
extern "C" void fplll_hip_extenum_shutdown(void)
{
  std::lock_guard<std::mutex> lock(g_mutex);
  if (g_ctx)
    fphip_destroy(g_ctx);
  g_ctx = nullptr;
}
