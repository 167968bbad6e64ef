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
// In-process multi-GPU (FPLLL_HIP_DEVICES=0,1,2,… or "all"): one context per listed device, one host
// thread per context for the duration of an enumeration call; the subtree tasks are dealt to the
// contexts exactly as bench.py's ranks deal them (content-sorted snake, enum_host.hip), the
// collective exchange points are a host barrier + MIN, and BETWEEN them every bound fplll's
// evaluator returns is published to all other contexts at once (fphip_enum_lower_bound) — so a
// bkz_reduction running in ONE fplll process scales over the GPUs of the node without RCCL.
// Callbacks into fplll are serialised by a mutex (the evaluator is a std::multimap; enumlib does the
// same, fplll/enum-parallel/enumeration.h:286).  A device may be listed twice (two contexts on one
// GPU): that is how the path is tested on a one-GPU box.
//
// Protocol implemented (enumerate_ext.cpp:48-167): call cbfunc once with mutranspose=true to
// receive mu^T / rdiag / pruning; report candidates through cbsol, which returns the new bound;
// return per-level node counts, or [0] = ~0 to decline so fplll falls back to its own enumerator
// (dim > 128, any device error, and dual calls unless FPLLL_HIP_DUAL=1).
//
// Dual calls (FPLLL_HIP_DUAL=1).  The reference's adapter hands a plugin the UNTRANSFORMED mu / r of
// the block for a dual enumeration and does not reverse the solutions afterwards
// (enumerate_ext.cpp:57-89, against EnumerationDyn::enumerate's enumerate.cpp:100-123,154-158), so
// the plugin does both itself: r'_{d-1-i} = 1 / r_i, mu'^T[d-1-j][d-1-i] = -mu(j,i), the dualenum
// walk (alpha instead of x in the centre sums), every solution reversed before it reaches fplll's
// evaluator.  It is opt-in because the adapter's radius is only right when the caller's radius
// exponent is zero: it scales by 2^(normexp - fmaxdistexpo) (enumerate_ext.cpp:75) where
// EnumerationDyn::enumerate scales by 2^(fmaxdistexpo + normexp) (enumerate.cpp:100-106) — the same
// number for fmaxdistexpo = 0 (a MatGSO without GSO_ROW_EXPO), a radius off by 4^fmaxdistexpo
// otherwise (BKZ in double precision runs WITH row exponents, bkz.cpp:816-820; the reference's own
// plugin declines every dual call, enum-parallel/enumlib.cpp:99).

#include <array>
#include <cstdint>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
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
  unsigned long long calls = 0, declined = 0, nodes = 0, moved = 0;
  ~Totals()
  {
    if (getenv("FPLLL_HIP_STATS") && (calls || declined))
      fprintf(stderr, "[fplll_hip] %llu enumerations on the device (%llu declined): %.3f s in the plugin, "
                      "%.3f s of kernels, %.3e nodes, %llu subtree tasks moved between devices\n",
              calls, declined, secs, kernel_ms * 1e-3, (double)nodes, moved);
  }
} g_totals;
std::mutex g_mutex;  // fplll's global hook is process-wide and unsynchronised (enumerate_ext.cpp:32-37)

struct Trampoline
{
  std::function<cb_process_sol_t> *cbsol;
  std::function<cb_process_subsol_t> *cbsubsol;
  int dim;
  long delivered;  // candidates handed to fplll's evaluator so far
  bool dual;       // reverse every solution (enumerate.cpp:154-158 does it after the run)
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
    buf[i] = t->dual ? sol[t->dim - 1 - i] : sol[i];
  try
  {
    return (*t->cbsol)(dist, buf);
  }
  catch (...)
  {
    return 0.0;  // no exception may cross the C boundary; a zero bound stops the enumeration
  }
}

// ---- in-process multi-GPU ------------------------------------------------------------------
std::vector<fphip_ctx *> g_multi;  // contexts of FPLLL_HIP_DEVICES (empty: single-device mode)
bool g_multi_tried = false;

const std::vector<fphip_ctx *> &multi_contexts()
{
  if (g_multi_tried)
    return g_multi;
  g_multi_tried   = true;
  const char *lst = getenv("FPLLL_HIP_DEVICES");
  if (!lst || !*lst)
    return g_multi;
  std::vector<int> devs;
  if (std::string(lst) == "all")
    for (int i = 0; i < fphip_device_count(); ++i)
      devs.push_back(i);
  else
    for (const char *p = lst; *p;)
    {
      devs.push_back(atoi(p));
      while (*p && *p != ',')
        ++p;
      if (*p == ',')
        ++p;
    }
  if (devs.size() < 2)
    return g_multi;
  for (int dv : devs)
  {
    fphip_ctx *c = nullptr;
    if (fphip_create(dv, &c) != FPHIP_OK)
    {
      fprintf(stderr, "[fplll_hip] FPLLL_HIP_DEVICES: no context on device %d (%s); single-device mode\n", dv,
              c ? fphip_last_error(c) : "?");
      if (c)
        fphip_destroy(c);
      for (fphip_ctx *o : g_multi)
        fphip_destroy(o);
      g_multi.clear();
      return g_multi;
    }
    g_multi.push_back(c);
  }
  return g_multi;
}

// The collective of the multi-GPU protocol among host threads: every participant deposits (bound,
// active) and leaves with (min bound, any active).  A participant that has LEFT the protocol (its
// fphip_enum_run returned, normally or not) no longer counts, so a failing shard cannot strand the
// others at the barrier.
struct HostExchange
{
  std::mutex m;
  std::condition_variable cv;
  int expected = 0, arrived = 0;
  unsigned long long generation = 0;
  double acc_bound = 0, out_bound = 0;
  int acc_any = 0, out_any = 0;
  void release_locked()
  {
    out_bound = acc_bound;
    out_any   = acc_any;
    arrived   = 0;
    ++generation;
    cv.notify_all();
  }
  double exchange(double bound, int active, int *any)
  {
    std::unique_lock<std::mutex> lk(m);
    if (arrived == 0)
    {
      acc_bound = bound;
      acc_any   = active;
    }
    else
    {
      acc_bound = bound < acc_bound ? bound : acc_bound;
      acc_any |= active;
    }
    ++arrived;
    const unsigned long long gen = generation;
    if (arrived >= expected)
      release_locked();
    else
      cv.wait(lk, [&] { return generation != gen; });
    if (any)
      *any = out_any;
    return out_bound;
  }
  void leave()
  {
    std::lock_guard<std::mutex> lk(m);
    --expected;
    if (arrived > 0 && arrived >= expected)
      release_locked();
  }
};

// The all-gather of byte blocks of the work movement (fphip_enum_opts::gather) among the host threads of the
// in-process multi-device mode: every participant deposits its block, leaves with the blocks of all of them in
// shard order.  Blocks are staged in buffers of the collective (two sets, by the parity of the round: a thread is
// never more than one round ahead of the slowest); a participant that has left contributes an empty block.
struct HostGather
{
  std::mutex m;
  std::condition_variable cv;
  int world = 0, expected = 0, arrived = 0;
  unsigned long long generation = 0;
  std::vector<std::vector<char>> stage[2];
  std::vector<size_t> size[2];
  void init(int w)
  {
    world = expected = w;
    for (int p = 0; p < 2; ++p)
    {
      stage[p].assign(w, std::vector<char>());
      size[p].assign(w, 0);
    }
  }
  void release_locked()
  {
    arrived = 0;
    ++generation;
    cv.notify_all();
  }
  int gather(int index, const void *send, size_t send_bytes, void *recv, size_t recv_cap, size_t *sizes)
  {
    std::unique_lock<std::mutex> lk(m);
    const unsigned long long gen = generation;
    const int par                = (int)(gen & 1);
    if (arrived == 0)
      for (int r = 0; r < world; ++r)
        size[par][r] = 0;
    stage[par][index].assign((const char *)send, (const char *)send + send_bytes);
    size[par][index] = send_bytes;
    ++arrived;
    if (arrived >= expected)
      release_locked();
    else
      cv.wait(lk, [&] { return generation != gen; });
    size_t off = 0;
    for (int r = 0; r < world; ++r)
    {
      const size_t n = size[par][r];
      if (off + n > recv_cap)
        return 1;
      if (n)
        memcpy((char *)recv + off, stage[par][r].data(), n);
      sizes[r] = n;
      off += n;
    }
    return 0;
  }
  void leave()
  {
    std::lock_guard<std::mutex> lk(m);
    --expected;
    if (arrived > 0 && arrived >= expected)
      release_locked();
  }
};

struct MultiShared
{
  std::mutex cb_mutex;  // fplll's callbacks, one at a time
  std::function<cb_process_sol_t> *cbsol;
  std::function<cb_process_subsol_t> *cbsubsol;
  int dim;
  bool dual = false;
  long delivered = 0;
  const std::vector<fphip_ctx *> *ctxs;
  HostExchange ex;
  HostGather ga;
};
struct ShardUser
{
  MultiShared *sh;
  int index;
};

void multi_subsol(void *user, double dist, const double *subsol, int offset)
{
  ShardUser *u = static_cast<ShardUser *>(user);
  double buf[FPHIP_ENUM_MAX_DIM];
  memcpy(buf, subsol, sizeof(double) * u->sh->dim);
  std::lock_guard<std::mutex> lk(u->sh->cb_mutex);
  try
  {
    (*u->sh->cbsubsol)(dist, buf, offset);
  }
  catch (...)
  {
  }
}
double multi_sol(void *user, double dist, const double *sol)
{
  ShardUser *u = static_cast<ShardUser *>(user);
  double buf[FPHIP_ENUM_MAX_DIM];
  for (int i = 0; i < u->sh->dim; ++i)
    buf[i] = u->sh->dual ? sol[u->sh->dim - 1 - i] : sol[i];
  double nb = 0.0;
  {
    std::lock_guard<std::mutex> lk(u->sh->cb_mutex);
    ++u->sh->delivered;
    try
    {
      nb = (*u->sh->cbsol)(dist, buf);
    }
    catch (...)
    {
      nb = 0.0;
    }
  }
  // the other GPUs learn the new bound NOW, not at their next exchange point
  for (size_t j = 0; j < u->sh->ctxs->size(); ++j)
    if ((int)j != u->index)
      fphip_enum_lower_bound((*u->sh->ctxs)[j], nb);
  return nb;
}
double multi_exchange(void *user, double local_bound, int local_active, int *any_active)
{
  return static_cast<ShardUser *>(user)->sh->ex.exchange(local_bound, local_active, any_active);
}
int multi_gather(void *user, const void *send, size_t send_bytes, void *recv, size_t recv_cap, size_t *sizes)
{
  ShardUser *u = static_cast<ShardUser *>(user);
  return u->sh->ga.gather(u->index, send, send_bytes, recv, recv_cap, sizes);
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
  if (dim < 2 || dim > FPHIP_ENUM_MAX_DIM)
    return out;
  if (dual)
  {
    const char *dv = getenv("FPLLL_HIP_DUAL");
    if (!dv || atoi(dv) == 0 || findsubsols)
    {
      g_totals.declined++;
      return out;
    }
  }
  std::lock_guard<std::mutex> lock(g_mutex);
  const std::vector<fphip_ctx *> &multi = multi_contexts();
  fphip_ctx *ctx                        = multi.empty() ? context() : multi[0];
  if (!ctx)
    return out;

  std::vector<double> mu((size_t)dim * dim, 0.0), rdiag(dim, 0.0), pruning(dim, 0.0);
  cbfunc(mu.data(), (size_t)dim, true, rdiag.data(), pruning.data());
  if (dual)
  {  // EnumerationDyn::enumerate's transformation (enumerate.cpp:107-123) of the primal inputs
    std::vector<double> mu2((size_t)dim * dim, 0.0), r2(dim, 0.0);
    for (int i = 0; i < dim; ++i)
      r2[dim - 1 - i] = 1.0 / rdiag[i];
    for (int i = 0; i < dim; ++i)
      for (int j = i + 1; j < dim; ++j)  // mu[i*dim + j] = mu(j,i)
        mu2[(size_t)(dim - 1 - j) * dim + (dim - 1 - i)] = -mu[(size_t)i * dim + j];
    mu.swap(mu2);
    rdiag.swap(r2);
  }

  fphip_enum_opts opts{};
  opts.dual = dual ? 1 : 0;
  const char *mn         = getenv("FPLLL_HIP_MIN_NODES");
  opts.min_nodes_decline = mn ? atoi(mn) : 0;
  opts.findsubsols       = findsubsols ? 1 : 0;
  if (!multi.empty())
  {
    const int W = (int)multi.size();
    MultiShared sh;
    sh.cbsol       = &cbsol;
    sh.cbsubsol    = &cbsubsol;
    sh.dim         = dim;
    sh.dual        = dual;
    sh.ctxs        = &multi;
    sh.ex.expected = W;
    sh.ga.init(W);
    // work movement between the devices (donated subtrees levelled at every round boundary, like the ranks of the
    // torch.distributed mode): FPLLL_HIP_MOVE=0 keeps every donated subtree on its device
    const char *mv   = getenv("FPLLL_HIP_MOVE");
    const bool move  = !(mv && mv[0] == '0');
    std::vector<ShardUser> users(W);
    std::vector<std::vector<std::uint64_t>> nodes(W, std::vector<std::uint64_t>(dim + 1, 0));
    std::vector<fphip_enum_stats> stats(W);
    std::vector<int> rcs(W, FPHIP_ERROR);
    std::vector<std::thread> th;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < W; ++i)
    {
      users[i] = ShardUser{&sh, i};
      th.emplace_back(
          [&, i]()
          {
            fphip_enum_opts o = opts;
            o.shard_index     = i;
            o.shard_count     = W;
            o.exchange        = multi_exchange;
            o.exchange_user   = &users[i];
            o.exchange_chunks = 4;
            o.gather          = move ? multi_gather : nullptr;
            o.gather_user     = &users[i];
            rcs[i] = fphip_enum_run(multi[i], dim, maxdist, mu.data(), rdiag.data(), pruning.data(), &o,
                                    multi_sol, findsubsols ? multi_subsol : nullptr, &users[i], nodes[i].data(),
                                    &stats[i]);
            sh.ex.leave();
            sh.ga.leave();
          });
    }
    for (auto &t : th)
      t.join();
    bool declined = true, failed = false;
    for (int i = 0; i < W; ++i)
    {
      declined = declined && rcs[i] == FPHIP_UNSUPPORTED;
      failed   = failed || (rcs[i] != FPHIP_OK && rcs[i] != FPHIP_UNSUPPORTED);
    }
    if (declined)
    {  // every shard takes the same decision from the same estimate (before anything is launched)
      g_totals.declined++;
      return out;
    }
    g_totals.secs += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    g_totals.calls++;
    double kmax = 0;
    for (int i = 0; i < W; ++i)
    {
      kmax = stats[i].kernel_ms > kmax ? stats[i].kernel_ms : kmax;
      g_totals.nodes += stats[i].total_nodes;
      g_totals.moved += stats[i].moved_tasks;
    }
    g_totals.kernel_ms += kmax;
    if (failed)
    {
      for (int i = 0; i < W; ++i)
        if (rcs[i] != FPHIP_OK)
          fprintf(stderr, "[fplll_hip] shard %d of %d failed%s, fplll's enumerator takes over: %s\n", i, W,
                  sh.delivered ? " AFTER candidates were delivered" : "", fphip_last_error(multi[i]));
      return out;
    }
    out.fill(0);
    for (int i = 0; i < W; ++i)
      for (int k = 0; k <= dim; ++k)
        out[k] += nodes[i][k];
    return out;
  }
  Trampoline tr{&cbsol, &cbsubsol, dim, 0, dual};
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
  for (fphip_ctx *c : g_multi)
    fphip_destroy(c);
  g_multi.clear();
  g_multi_tried = false;
}
