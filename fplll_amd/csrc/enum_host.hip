// enum_host.hip — host side of the C ABI for enumeration (include/fplll_hip.h): context,
// top-of-tree phase planning, launches, and the solution ring consumer that runs the caller's
// callback while the kernel is in flight.
//
// Reference counterparts: ExternalEnumeration::enumerate (fplll/enum/enumerate_ext.cpp:48-89) is
// the caller; enumlib's enumerate_dim_detail (enum-parallel/enumlib_dim.cpp:47-104) is the
// CPU plugin this replaces.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/fplll_hip.h"
#include "dev_mem.h"
#include "trace.h"
#include "enum_device.h"


namespace fphip
{
template <bool MU_LDS, bool SUBS, bool DUAL>
__global__ void enum_phase_kernel(DevShared *g, HostCtl *h, TaskBuf in, TaskBuf out, int d,
                                  int Lmax, int stop, unsigned task_lo, unsigned task_hi,
                                  const unsigned *idxlist, int launch_idx, int count_nodes,
                                  unsigned budget, const double *xhi_root, double *gstk, int Tsplit,
                                  unsigned *qh, const unsigned *rcnt, unsigned rcap,
                                  unsigned long long bound_init);
template <int NQT, bool SUBS, bool DUAL>
__global__ void enum_top_kernel(DevShared *g, HostCtl *h, TopBuf in, unsigned n_in, TopBuf out_top,
                                int stop, TaskBuf out, double *xhi_root, int d, double maxdist,
                                int count_nodes, int launch_idx, double *gtop);
__global__ void task_key_kernel(TaskBuf in, unsigned n, int d, unsigned long long *keys,
                                const double *xhi_root, const unsigned *slots);
template <bool MU_LDS, bool DUAL>
__global__ void enum_walk_kernel(DevShared *g, HostCtl *h, TaskBuf in, TaskBuf out, int d, int Lmax, unsigned task_lo,
                                 unsigned task_hi, const unsigned *idxlist, int launch_idx, int count_nodes,
                                 unsigned budget, const double *xhi_root, double *gstk, int Tsplit, unsigned *qh,
                                 const unsigned *rcnt, unsigned rcap, unsigned long long bound_init);
// enum_deal.hip: the content-sorted snake deal of a multi-rank call on the device
size_t deal_work_bytes(unsigned n);
unsigned deal_tasks_device(hipStream_t s, const unsigned long long *keys, const double *pd, const unsigned *slot_of,
                           unsigned n, unsigned W, unsigned rank, unsigned *mine, void *work, size_t work_bytes);
__global__ void task_pack_kernel(TaskBuf in, unsigned lo, unsigned n, double *rec, const double *xhi_root, int xstr);
__global__ void task_unpack_kernel(TaskBuf out, unsigned lo, unsigned n, const double *rec, double *xhi_root, int xstr,
                                   unsigned root_base);
template <bool DUAL>
__global__ void enum_bfs_kernel(DevShared *g, double maxdist, QueueMem *qm, TaskBuf f0, TaskBuf f1,
                                TaskBuf fin, int L0, int nlev, int floor_level, float heavy, int count_nodes,
                                int compact_n, int shard_index, int shard_count);
// Closes the breadth-first stage: number of final tasks and the error flags into the pinned control
// block (the host reads them without another device round trip); with `slots` (multi-GPU partition:
// the host sorts the tasks by content) also the compact list of the occupied slots and their
// partial distances.
__global__ void enum_bfs_epilogue(const DevShared *g, HostCtl *h, const QueueMem *qm, unsigned rcap,
                                  unsigned *slots, double *pdc, const double *pd)
{
  __shared__ unsigned base[FPHIP_NQ + 1];
  if (threadIdx.x == 0)
  {
    unsigned tot = 0;
    for (int q = 0; q < FPHIP_NQ; ++q)
    {
      base[q] = tot;
      const unsigned c = qm->fin[q * FPHIP_QS];
      tot += c < rcap ? c : rcap;
    }
    base[FPHIP_NQ] = tot;
    h->pad[0]      = tot;
    h->pad[1]      = g->error_flags;
    __threadfence_system();
  }
  __syncthreads();
  if (slots)
    for (int q = 0; q < FPHIP_NQ; ++q)
      for (unsigned i = threadIdx.x; i < base[q + 1] - base[q]; i += blockDim.x)
      {
        slots[base[q] + i] = (unsigned)q * rcap + i;
        pdc[base[q] + i]   = pd[(size_t)q * rcap + i];
      }
}
}
using namespace fphip;

// ---- process-wide cache of pinned host buffers (dev_mem.h) ---------------------------------------
#include <mutex>
#include <vector>
namespace
{
struct PinnedBuf
{
  void *p;
  size_t bytes;
  bool busy;
};
std::mutex g_pinned_mutex;
std::vector<PinnedBuf> g_pinned;
}  // namespace
#include "dev_cache.h"  // fphip_dev_alloc / fphip_dev_free: the process-wide cache of device blocks (dev_mem.h)

void *fphip_pinned_get(size_t bytes)
{
  std::lock_guard<std::mutex> lk(g_pinned_mutex);
  PinnedBuf *best = nullptr;
  for (PinnedBuf &b : g_pinned)
    if (!b.busy && b.bytes >= bytes && (!best || b.bytes < best->bytes))
      best = &b;
  if (best)
  {
    best->busy = true;
    return best->p;
  }
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess)
    return nullptr;
  g_pinned.push_back(PinnedBuf{p, bytes, true});
  return p;
}
void fphip_pinned_put(void *p)
{
  std::lock_guard<std::mutex> lk(g_pinned_mutex);
  for (PinnedBuf &b : g_pinned)
    if (b.p == p)
      b.busy = false;
}

struct fphip_ctx
{
  int device        = 0;
  int num_cus       = 256;
  hipStream_t stream = nullptr;
  hipEvent_t ev[2]  = {nullptr, nullptr};
  DevShared *g      = nullptr;  // device
  DevShared *stage  = nullptr;  // pinned host staging copy
  HostCtl *h        = nullptr;  // pinned, coherent; same pointer is valid on the device
  TaskBuf buf[3] = {};  // [0], [1]: task lists of the launches; [2]: second frontier of the breadth-first stage
  unsigned cap                 = 0;
  unsigned long long ring_next = 0;
  unsigned long long *keys     = nullptr;  // device: content key per task (multi-GPU partition)
  unsigned *idxlist            = nullptr;  // device: this rank's task indices, heaviest first
  void *deal_work              = nullptr;  // device: scratch of the device-side deal (enum_deal.hip)
  size_t deal_work_bytes       = 0;
  double *xhi_root             = nullptr;  // device: cap * 64 doubles: the coefficients of levels >= 64 per level-64
                                           // ancestor (blocks larger than 64; 64 doubles per started chunk of levels)
  QueueMem *qm                 = nullptr;  // device: ticket / emission counters of the current call
  unsigned *slots              = nullptr;  // device: compact list of the occupied slots of a regioned buffer
  double *pdc                  = nullptr;  // device: their partial distances (multi-GPU partition)
  double *gstk                 = nullptr;  // device: per-wave scratch of the tall stack slots (split stack)
  size_t gstk_doubles          = 0;
  int xhi_row                  = 64;       // doubles per level-64 ancestor xhi_root has room for (64, or 192 once a
                                           // block above 128 rows has been seen)
  double *wire                 = nullptr;  // device: task records on their way to / from other ranks (work movement)
  size_t wire_doubles          = 0;
  double *gtop                 = nullptr;  // device: per-wave column stacks of the top walk of blocks above 128 rows
  size_t gtop_doubles          = 0;
  TopBuf top[2] = {};                           // top tasks of blocks larger than 64 (allocated on demand)
  char err[512]                = {0};
  // GSO state lives in gso_host.hip, linked through this opaque slot
  void *gso = nullptr;
};

static int fail(fphip_ctx *ctx, const char *fmt, ...)
{
  if (ctx)
  {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx->err, sizeof ctx->err, fmt, ap);
    va_end(ap);
  }
  return FPHIP_ERROR;
}

#define HIPCHK(ctx, call)                                                                          \
  do                                                                                               \
  {                                                                                                \
    hipError_t e_ = (call);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return fail(ctx, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

extern "C" int fphip_abi_version(void) { return 1; }

extern "C" int fphip_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess)
    return 0;
  return n;
}

static int env_int(const char *name, int dflt)
{
  const char *s = getenv(name);
  return (s && *s) ? atoi(s) : dflt;
}

// Streams and hardware queues.  The runtime maps streams onto a small pool of hardware queues PER PRIORITY
// (four by default) and lets streams share a queue once the pool is used up.  Some kernels of this library
// are PERSISTENT — the strategy-BKZ schedule kernel sits on its stream until every mailbox request of the
// run has been answered — and whatever shares their queue waits behind them: a helper kernel that has to
// finish BEFORE the mailbox is answered (the pruner's volume kernel, a handed-off enumeration) must never
// do that (round 4: eight volume engines next to one schedule kernel deadlocked; two did not, by the luck
// of the round-robin).  Hence the priority classes: helper streams are HIGH (their own queue pool), a
// context that is known to run minutes-long launches beside others can be created LOW.

// The task buffers of the enumeration (three generations of cap tasks with their 64 partial sums and 64
// coefficients, keys, slots: ~3.6 GB at the default cap) are allocated by the first fphip_enum_run of a context —
// a context that only carries GSO objects (MatGSOHip, MatHouseholderHip, the batched reductions) never needs
// them.  Called with the context's device current.
static int ensure_task_buffers(fphip_ctx *ctx)
{
  if (ctx->qm)
    return FPHIP_OK;
#define ONCE(ptr, bytes) \
  if (!(ptr))             \
  HIPCHK(ctx, fphip_dev_alloc((void **)&(ptr), (bytes), ctx->stream))
  ONCE(ctx->slots, (size_t)ctx->cap * sizeof(unsigned));
  ONCE(ctx->pdc, (size_t)ctx->cap * sizeof(double));
  for (int b = 0; b < 3; ++b)
  {
    ONCE(ctx->buf[b].col, (size_t)ctx->cap * 64 * sizeof(double));
    ONCE(ctx->buf[b].x, (size_t)ctx->cap * 64 * sizeof(double));
    ONCE(ctx->buf[b].pd, (size_t)ctx->cap * sizeof(double));
    ONCE(ctx->buf[b].level, (size_t)ctx->cap * sizeof(int));
    ONCE(ctx->buf[b].root, (size_t)ctx->cap * sizeof(int));
    ONCE(ctx->buf[b].count, 64);
    ctx->buf[b].cap = ctx->cap;
  }
  ONCE(ctx->keys, (size_t)ctx->cap * sizeof(unsigned long long));
  ONCE(ctx->idxlist, (size_t)ctx->cap * sizeof(unsigned));
  ONCE(ctx->xhi_root, (size_t)ctx->cap * 64 * sizeof(double));
  HIPCHK(ctx, hipMemsetAsync(ctx->xhi_root, 0, 64 * sizeof(double), ctx->stream));
  // (qm last: it is the "buffers are there" flag, so a failed allocation above is retried by the next call)
  ONCE(ctx->qm, sizeof(QueueMem));
#undef ONCE
  return FPHIP_OK;
}

extern "C" int fphip_create(int device, fphip_ctx **out) { return fphip_create_ex(device, 0, out); }

extern "C" int fphip_create_ex(int device, int priority, fphip_ctx **out)
{
  if (!out)
    return FPHIP_ERROR;
  *out           = nullptr;
  fphip_ctx *ctx = new fphip_ctx();
  ctx->device    = device;
  int ndev       = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
  {
    // fail loudly: there is no CPU fallback in the product path
    fail(ctx, "no HIP device visible (hipGetDeviceCount)");
    *out = ctx;
    return FPHIP_ERROR;
  }
  *out = ctx;
  HIPCHK(ctx, hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(ctx, hipGetDeviceProperties(&prop, device));
  ctx->num_cus = prop.multiProcessorCount;
  if (priority == 0)
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  else
  {
    int least = 0, greatest = 0;  // (numerically: greatest priority = the smaller number)
    HIPCHK(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIPCHK(ctx, hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, priority > 0 ? greatest : least));
  }
  HIPCHK(ctx, hipEventCreate(&ctx->ev[0]));
  HIPCHK(ctx, hipEventCreate(&ctx->ev[1]));
  HIPCHK(ctx, fphip_dev_alloc((void **)&ctx->g, sizeof(DevShared), ctx->stream));
  ctx->stage = (DevShared *)fphip_pinned_get(sizeof(DevShared));
  ctx->h     = (HostCtl *)fphip_pinned_get(sizeof(HostCtl));
  if (!ctx->stage || !ctx->h)
    HIPCHK(ctx, hipErrorOutOfMemory);
  memset(ctx->h, 0, sizeof(HostCtl));
  ctx->cap = (unsigned)env_int("FPHIP_TASK_CAP", 1 << 20);
  ctx->cap = (ctx->cap + FPHIP_NQ - 1) / FPHIP_NQ * FPHIP_NQ;  // regions of cap / FPHIP_NQ slots
  HIPCHK(ctx, hipFuncSetAttribute((const void *)enum_phase_kernel<true, false, false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(ctx, hipFuncSetAttribute((const void *)enum_phase_kernel<false, false, false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(ctx, hipFuncSetAttribute((const void *)enum_phase_kernel<true, true, false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(ctx, hipFuncSetAttribute((const void *)enum_phase_kernel<false, true, false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(ctx, hipFuncSetAttribute((const void *)enum_phase_kernel<true, false, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(ctx, hipFuncSetAttribute((const void *)enum_phase_kernel<false, false, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  return FPHIP_OK;
}

extern "C" __attribute__((visibility("hidden"))) void fphip_gso_release_all(fphip_ctx *ctx);  // gso_host.hip

extern "C" void fphip_destroy(fphip_ctx *ctx)
{
  if (!ctx)
    return;
  fphip_gso_release_all(ctx);
  if (ctx->stream)
    hipStreamSynchronize(ctx->stream);
  for (int b = 0; b < 3; ++b)
  {
    if (ctx->buf[b].col)
      fphip_dev_free(ctx->buf[b].col, ctx->stream);
    if (ctx->buf[b].x)
      fphip_dev_free(ctx->buf[b].x, ctx->stream);
    if (ctx->buf[b].pd)
      fphip_dev_free(ctx->buf[b].pd, ctx->stream);
    if (ctx->buf[b].level)
      fphip_dev_free(ctx->buf[b].level, ctx->stream);
    if (ctx->buf[b].root)
      fphip_dev_free(ctx->buf[b].root, ctx->stream);
    if (ctx->buf[b].count)
      fphip_dev_free(ctx->buf[b].count, ctx->stream);
  }
  if (ctx->xhi_root)
    fphip_dev_free(ctx->xhi_root, ctx->stream);
  if (ctx->gstk)
    fphip_dev_free(ctx->gstk, ctx->stream);
  if (ctx->gtop)
    fphip_dev_free(ctx->gtop, ctx->stream);
  if (ctx->wire)
    fphip_dev_free(ctx->wire, ctx->stream);
  for (int b = 0; b < 2; ++b)
  {
    if (ctx->top[b].col)
      fphip_dev_free(ctx->top[b].col, ctx->stream);
    if (ctx->top[b].xhi)
      fphip_dev_free(ctx->top[b].xhi, ctx->stream);
    if (ctx->top[b].pd)
      fphip_dev_free(ctx->top[b].pd, ctx->stream);
    if (ctx->top[b].level)
      fphip_dev_free(ctx->top[b].level, ctx->stream);
    if (ctx->top[b].count)
      fphip_dev_free(ctx->top[b].count, ctx->stream);
  }
  if (ctx->qm)
    fphip_dev_free(ctx->qm, ctx->stream);
  if (ctx->slots)
    fphip_dev_free(ctx->slots, ctx->stream);
  if (ctx->pdc)
    fphip_dev_free(ctx->pdc, ctx->stream);
  if (ctx->keys)
    fphip_dev_free(ctx->keys, ctx->stream);
  if (ctx->idxlist)
    fphip_dev_free(ctx->idxlist, ctx->stream);
  if (ctx->deal_work)
    fphip_dev_free(ctx->deal_work, ctx->stream);
  if (ctx->g)
    fphip_dev_free(ctx->g, ctx->stream);
  // Wait for the stream BEFORE the pinned buffers go back to the process-wide cache: after an error
  // path (ring timeout, failed HIP call) a kernel may still be resident and writing its ring / bound /
  // sequence words — another thread's fphip_create must not be handed that HostCtl meanwhile.
  if (ctx->stream)
    hipStreamSynchronize(ctx->stream);  // also completes the stream-ordered frees above
  if (ctx->stage)
    fphip_pinned_put(ctx->stage);
  if (ctx->h)
    fphip_pinned_put(ctx->h);
  if (ctx->ev[0])
    hipEventDestroy(ctx->ev[0]);
  if (ctx->ev[1])
    hipEventDestroy(ctx->ev[1]);
  if (ctx->stream)
    hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" const char *fphip_last_error(const fphip_ctx *ctx) { return ctx ? ctx->err : "null ctx"; }

// accessors for sibling translation units
hipStream_t fphip_ctx_stream(fphip_ctx *ctx) { return ctx->stream; }
void **fphip_ctx_gso_slot(fphip_ctx *ctx) { return &ctx->gso; }
char *fphip_ctx_errbuf(fphip_ctx *ctx) { return ctx->err; }
int fphip_ctx_num_cus(fphip_ctx *ctx) { return ctx->num_cus; }
int fphip_ctx_device(fphip_ctx *ctx) { return ctx->device; }
// the task buffers NOW (gso_host.hip's hand-off set-up: before the persistent schedule kernel is launched, never from
// the worker thread that answers its mailbox — a multi-gigabyte allocation there would stall the kernel at best)
// a context whose enumerations are known to be small (the extra hand-off contexts of a batch of tours: blocks of at
// most 64 rows) can hold its three task generations in a fraction of the default 3.6 GB; before the first run only
int fphip_ctx_set_task_cap(fphip_ctx *ctx, unsigned cap)
{
  if (!ctx || ctx->qm || cap < 4096)
    return FPHIP_ERROR;
  ctx->cap = (cap + FPHIP_NQ - 1) / FPHIP_NQ * FPHIP_NQ;
  return FPHIP_OK;
}
int fphip_ctx_ensure_task_buffers(fphip_ctx *ctx)
{
  if (hipSetDevice(ctx->device) != hipSuccess)
    return FPHIP_ERROR;
  return ensure_task_buffers(ctx);
}

// ---------------------------------------------------------------------------------------------
// ring consumer
// ---------------------------------------------------------------------------------------------
static inline unsigned long long dbits(double v)
{
  unsigned long long b;
  memcpy(&b, &v, 8);
  return b;
}
static inline double bdbl(unsigned long long b)
{
  double v;
  memcpy(&v, &b, 8);
  return v;
}

// lower the pinned bound word to min(current, nb): bit patterns of non-negative doubles order like
// integers.  Safe against the serving thread and against other host threads (fphip_enum_lower_bound).
static void publish_bound_min(fphip_ctx *ctx, double nb)
{
  if (!(nb >= 0.0))
    nb = 0.0;
  const unsigned long long want = dbits(nb);
  unsigned long long cur        = __atomic_load_n(&ctx->h->bound_bits, __ATOMIC_ACQUIRE);
  while (want < cur &&
         !__atomic_compare_exchange_n(&ctx->h->bound_bits, &cur, want, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE))
  {
  }
}

// Lower the enumeration bound of a context from ANY host thread while fphip_enum_run is in flight on
// it: the waves poll the device mirror of this word (every 64 steps) and the word itself (rarely), so
// a bound found on another GPU takes effect within microseconds instead of at the next chunk
// boundary.  Never raises the bound; a no-op when no enumeration is running.
extern "C" int fphip_enum_lower_bound(fphip_ctx *ctx, double bound)
{
  if (!ctx || !ctx->h)
    return FPHIP_ERROR;
  publish_bound_min(ctx, bound);
  return FPHIP_OK;
}

static void drain(fphip_ctx *ctx, int dim, fphip_sol_cb cb, fphip_subsol_cb subcb, void *user,
                  uint64_t *nsol)
{
  for (;;)
  {
    SolRec *r            = &ctx->h->ring[ctx->ring_next % FPHIP_RING_CAP];
    unsigned long long s = __atomic_load_n(&r->seq, __ATOMIC_ACQUIRE);
    if (s != ctx->ring_next + 1)
      return;
    double x[FPHIP_ENUM_MAX_DIM];
    double dist = r->dist;
    memcpy(x, (const void *)r->x, sizeof(double) * FPHIP_ENUM_MAX_DIM);
    if (r->kind == 1)
    {  // extenum_cb_process_subsol (enumerate_ext_api.h:70-71): no effect on the radius
      if (subcb)
        subcb(user, dist, x, r->offset);
      ctx->ring_next++;
      __atomic_store_n(&ctx->h->consumed, ctx->ring_next, __ATOMIC_RELEASE);
      continue;
    }
    double nb = cb(user, dist, x);  // extenum_cb_process_sol: returns the new bound
    // (the reference's evaluators only ever shrink the bound; min() also keeps a smaller bound that
    // another host thread published meanwhile — fphip_enum_lower_bound)
    publish_bound_min(ctx, nb);
    ctx->ring_next++;
    __atomic_store_n(&ctx->h->consumed, ctx->ring_next, __ATOMIC_RELEASE);
    (void)dim;
    (*nsol)++;
  }
}

// wait for the stream while serving the ring
static int wait_serving(fphip_ctx *ctx, int dim, fphip_sol_cb cb, fphip_subsol_cb subcb, void *user,
                        uint64_t *nsol)
{
  for (;;)
  {
    drain(ctx, dim, cb, subcb, user, nsol);
    hipError_t q = hipStreamQuery(ctx->stream);
    if (q == hipSuccess)
    {
      drain(ctx, dim, cb, subcb, user, nsol);
      return FPHIP_OK;
    }
    if (q != hipErrorNotReady)
      return fail(ctx, "hipStreamQuery: %s", hipGetErrorString(q));
  }
}

// ---------------------------------------------------------------------------------------------
// Gaussian-heuristic node estimate per level (only used to place the phase cuts; never affects
// results).  logN[k] = log( 1/2 · V_{d-k}(R_k) / prod_{i>=k} sqrt(rdiag_i) ),  R_k^2 = pruning_k·maxdist.
// ---------------------------------------------------------------------------------------------
static void estimate_levels(int d, const double *rdiag, const double *pruning, double maxdist,
                            double *logN /* d+1 */)
{
  double sumlog = 0.0;
  logN[d]       = 0.0;
  for (int k = d - 1; k >= 0; --k)
  {
    sumlog += 0.5 * std::log(rdiag[k] > 0 ? rdiag[k] : 1e-300);
    int n       = d - k;
    double R2   = (pruning ? pruning[k] : 1.0) * maxdist;
    double logV = 0.5 * n * std::log(M_PI) - std::lgamma(0.5 * n + 1.0);
    logN[k]     = std::log(0.5) + logV + 0.5 * n * std::log(R2 > 0 ? R2 : 1e-300) - sumlog;
  }
}

// level at which to cut next, or -1 to walk to the leaves
static int choose_stop(const double *logN, int L, double C, double target_final, double growth)
{
  double want = std::min(target_final, C * growth);
  int argmax  = -1;
  double best = -1e300;
  for (int k = L - 1; k >= 1; --k)
  {
    double est = std::log(C) + logN[k] - logN[L];
    if (est >= std::log(want))
      return k;
    if (est > best)
    {
      best   = est;
      argmax = k;
    }
  }
  if (argmax >= 1 && best >= std::log(4.0 * C))
    return argmax;
  return -1;
}

// Work movement between the ranks at a round boundary of the final phase (fphip_enum_opts::gather): the ranks
// learn each other's numbers of donated tasks; those above the average pack their surplus (the tail of their
// list), the pool of all surpluses is gathered everywhere, and the ranks below the average take consecutive
// slices of it in rank order.  Every rank computes the same plan from the same counts: no task is lost or walked
// twice, whatever the order of the lists.  *cnt is this rank's number of tasks in `buf` before and after.
// moved_out (nullable): tasks that left or reached this rank.
// d > 64: a task points into its rank's table of level-64 ancestors (xhi_root: the coefficients of levels >= 64,
// xstr doubles per row; the top walk that fills it is replicated, but in an order of its own on every rank), so the
// record carries that row and the receiver appends it to its table: *xhi_used rows are taken, cap is the table's size.
static int rebalance_tasks(fphip_ctx *ctx, const fphip_enum_opts &o, TaskBuf buf, unsigned *cnt, unsigned *moved_out,
                           int d, unsigned *xhi_used)
{
  const int W = o.shard_count, me = o.shard_index;
  const int xstr = d > 64 ? 64 * ((d - 1) >> 6) : 0;
  std::vector<unsigned long long> both((size_t)W * 2, 0), counts((size_t)W, 0), room((size_t)W, 0);
  std::vector<size_t> sizes((size_t)W, 0);
  unsigned long long mine[2] = {*cnt, xstr > 0 ? (unsigned long long)(ctx->cap - std::min(ctx->cap, *xhi_used)) : ~0ull};
  if (o.gather(o.gather_user, mine, sizeof mine, both.data(), sizeof(unsigned long long) * 2 * (size_t)W, sizes.data()) != 0)
    return fail(ctx, "work movement: the gather callback failed (counts)");
  unsigned long long total = 0;
  for (int r = 0; r < W; ++r)
  {
    if (sizes[r] != sizeof mine)
      return fail(ctx, "work movement: rank %d sent %zu bytes of counts, %zu expected", r, sizes[r], sizeof mine);
    counts[r] = both[2 * r];
    room[r]   = both[2 * r + 1];
    total += counts[r];
  }
  std::vector<unsigned long long> surplus((size_t)W, 0), deficit((size_t)W, 0);
  unsigned long long moved = 0;
  for (int r = 0; r < W; ++r)
  {
    const unsigned long long target = total / W + ((unsigned long long)r < total % W ? 1 : 0);
    if (counts[r] > target)
      surplus[r] = counts[r] - target;
    else
      deficit[r] = target - counts[r];
    moved += surplus[r];
  }
  if (moved_out)
    *moved_out = 0;
  // not worth a transfer: (nearly) balanced already, or a target beyond this context's buffers (the same
  // decision on every rank: it only depends on the counts)
  // (FPHIP_MOVE_FRACTION: move when at least total / that many tasks would; 16 by default — tests lower the bar)
  const unsigned long long frac = (unsigned long long)std::max(1, env_int("FPHIP_MOVE_FRACTION", 16));
  if (moved == 0 || moved * frac < total || total / W + 1 > ctx->cap)
    return FPHIP_OK;
  for (int r = 0; r < W; ++r)  // (a receiver without room for the ancestors' rows: nobody moves this round)
    if (deficit[r] > room[r])
      return FPHIP_OK;
  const size_t recd = (size_t)FPHIP_TASK_REC + (size_t)xstr;
  const size_t recb = recd * sizeof(double);
  if (ctx->wire_doubles < (size_t)moved * recd)
  {
    if (ctx->wire)
      fphip_dev_free(ctx->wire, ctx->stream);
    ctx->wire         = nullptr;
    ctx->wire_doubles = 0;
    HIPCHK(ctx, fphip_dev_alloc((void **)&ctx->wire, (size_t)moved * recb, ctx->stream));
    ctx->wire_doubles = (size_t)moved * recd;
  }
  std::vector<double> send((size_t)surplus[me] * recd), pool((size_t)moved * recd);
  if (surplus[me] > 0)
  {
    const unsigned n  = (unsigned)surplus[me];
    const unsigned lo = *cnt - n;  // the tail of the list leaves
    hipLaunchKernelGGL(task_pack_kernel, dim3(std::min<unsigned>((n + 3) / 4, (unsigned)ctx->num_cus * 8u)), dim3(256), 0,
                       ctx->stream, buf, lo, n, ctx->wire, ctx->xhi_root, xstr);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(send.data(), ctx->wire, (size_t)n * recb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *cnt = lo;
  }
  if (o.gather(o.gather_user, send.data(), (size_t)surplus[me] * recb, pool.data(), (size_t)moved * recb, sizes.data()) != 0)
    return fail(ctx, "work movement: the gather callback failed (tasks)");
  for (int r = 0; r < W; ++r)
    if (sizes[r] != (size_t)surplus[r] * recb)
      return fail(ctx, "work movement: rank %d sent %zu bytes, %zu expected", r, sizes[r], (size_t)surplus[r] * recb);
  if (deficit[me] > 0)
  {
    unsigned long long off = 0;
    for (int r = 0; r < me; ++r)
      off += deficit[r];
    const unsigned n = (unsigned)deficit[me];
    HIPCHK(ctx, hipMemcpyAsync(ctx->wire, pool.data() + (size_t)off * recd, (size_t)n * recb,
                               hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(task_unpack_kernel, dim3(std::min<unsigned>((n + 3) / 4, (unsigned)ctx->num_cus * 8u)), dim3(256), 0,
                       ctx->stream, buf, *cnt, n, ctx->wire, ctx->xhi_root, xstr, *xhi_used);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *cnt += n;
    if (xstr > 0)
      *xhi_used += n;
  }
  if (moved_out)
    *moved_out = (unsigned)(surplus[me] + deficit[me]);
  return FPHIP_OK;
}

extern "C" int fphip_enum_run(fphip_ctx *ctx, int dim, double maxdist, const double *mut,
                              const double *rdiag, const double *pruning,
                              const fphip_enum_opts *opts_in, fphip_sol_cb cb,
                              fphip_subsol_cb subcb, void *user, uint64_t *nodes_out,
                              fphip_enum_stats *stats)
{
  FPHIP_RANGE("fphip_enum_run");
  if (!ctx)
    return FPHIP_ERROR;
  if (!ctx->g)
    return fail(ctx, "context has no device (creation failed): %s", ctx->err);
  if (!mut || !rdiag || !cb || !nodes_out)
    return fail(ctx, "null argument");
  auto t_begin = std::chrono::steady_clock::now();
  fphip_enum_opts o;
  memset(&o, 0, sizeof o);
  if (opts_in)
    o = *opts_in;
  if (o.shard_count <= 0)
  {
    o.shard_count = 1;
    o.shard_index = 0;
  }
  if (o.exchange_chunks <= 0)
    o.exchange_chunks = 1;
  const int d = dim;
  if (d < 2 || d > FPHIP_ENUM_MAX_DIM || (o.findsubsols && !subcb))
    return FPHIP_UNSUPPORTED;  // → ~uint64_t(0): fplll falls back (enumerate_ext.cpp:88)
  // dual enumeration (enumerate_base.cpp:57-61, 103-105; the caller passes the transformed mu / r of
  // enumerate.cpp:107-123): without sub-solutions, as the reference's dual caller has it
  // (BKZReduction's evaluator is built with find_subsolutions = false, bkz.cpp:327-331)
  const bool dual = o.dual != 0;
  if (dual && o.findsubsols)
    return FPHIP_UNSUPPORTED;
  const bool subs = o.findsubsols != 0;
  if (!(maxdist >= 0.0))
    return FPHIP_UNSUPPORTED;
  for (int i = 0; i < d; ++i)
    if (!(rdiag[i] > 0.0) || !std::isfinite(rdiag[i]))
      return FPHIP_UNSUPPORTED;

  // tasks per task from one split launch to the next.  A small factor means more, shorter split
  // launches: each is a handful of lone waves walking the top of their subtree as one dependent
  // chain, so its duration is its longest chain (measured, tests/perf/pruner_variants.sh: 12…48
  // instead of the round-1 value 96 takes 15 % off a 2.7·10⁶-node call and 10–20 % off the wall
  // time of a 10⁹-node call — the tasks are better balanced and the radius shrinks sooner; the
  // nodes/s of the walk itself are unchanged)
  const double growth = o.phase_growth > 0 ? o.phase_growth : env_int("FPHIP_PHASE_GROWTH", 12);
  const int wpb_final =
      std::max(1, std::min(8, o.waves_per_block > 0 ? o.waves_per_block
                                                    : env_int("FPHIP_WAVES_PER_BLOCK", 2)));

  double logN[FPHIP_ENUM_MAX_DIM + 1];
  estimate_levels(d, rdiag, pruning, maxdist, logN);
  // number of subtree tasks the walk launches start from: more for big trees (finer balance, the
  // radius found by one task prunes the others sooner: 5.2 vs 4.8·10^9 nodes/s on the 3·10^9-node
  // blocks), fewer for small ones (the split launches are their serial part)
  double gh_nodes = 0;
  for (int k = 0; k < d; ++k)
    gh_nodes += std::exp(std::min(logN[k], 60.0));
  double target_final =
      o.target_tasks > 0 ? o.target_tasks
                         : env_int("FPHIP_TARGET_TASKS", gh_nodes > 1e8 ? 65536 : 32768);
  int overflow_retries = 0;  // split-launch overflows under sharding answered by a smaller task target
  if (o.min_nodes_decline > 0)
  {
    double tot = 0;
    for (int k = 0; k < d; ++k)
      tot += std::exp(std::min(logN[k], 700.0));
    if (tot < (double)o.min_nodes_decline)
      return FPHIP_UNSUPPORTED;
  }

  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (int rc_ = ensure_task_buffers(ctx))
    return rc_;
  // Breadth-first stage instead of the depth-first split launches (see enum_bfs_kernel): the default;
  // sub-solution calls keep the split launches (the expansion does not report sub-solutions), and a
  // call whose expansion overflowed a buffer starts over with them.
  bool use_bfs = env_int("FPHIP_BFS", 1) != 0 && !subs;
  int bfs_restarts = 0;
restart:
  // ---- upload the block: rdiag, pruning, mu rows (triangular) ---------------------------------
  DevShared *st = ctx->stage;
  // (mu_tri is the struct's last member: a block of d rows uses — and uploads — its first d (d - 1) / 2 entries)
  const size_t st_used = offsetof(DevShared, mu_tri) + (((size_t)d * (d - 1) / 2 * sizeof(double) + 63) & ~(size_t)63);
  memset(st, 0, st_used);
  for (int i = 0; i < d; ++i)
  {
    st->rdiag[i]   = rdiag[i];
    st->pruning[i] = pruning ? pruning[i] : 1.0;
    st->rp[i][0]   = st->rdiag[i];  // the walk kernels read (r_ii, pruning_i) as one scalar load
    st->rp[i][1]   = st->pruning[i];
    if (i < 64)
    {
      st->mu_sq[i][64] = st->rdiag[i];
      st->mu_sq[i][65] = st->pruning[i];
    }
    st->sub_bits[i] = dbits(rdiag[i]);  // subsoldists = rdiag, enumerate.cpp:143
  }
  for (int k = 1; k < d; ++k)
    for (int i = 0; i < k; ++i)
    {
      st->mu_tri[(k * (k - 1)) / 2 + i] = mut[(size_t)i * d + k];  // mu(k,i)
      if (k < 64)
        st->mu_sq[k][i] = mut[(size_t)i * d + k];
    }
  // table of the subtree-size estimate of the breadth-first stage (levels below 64):
  // A[L][k] = log V_{L-k}(1) - sum_{i=k}^{L-1} log sqrt(r_ii)
  float bfs_est0[65];  // estimate for a node of partial distance 0 at level L (the heaviest there is)
  if (use_bfs)
  {
    const int Lh = d < 64 ? d : 64;
    for (int Lv = 1; Lv <= Lh; ++Lv)
    {
      double sumlog = 0.0, tot = 0.0;
      for (int k = Lv - 1; k >= 0; --k)
      {
        sumlog += 0.5 * std::log(rdiag[k]);
        const int n       = Lv - k;
        const double logV = 0.5 * n * std::log(M_PI) - std::lgamma(0.5 * n + 1.0);
        const double A    = logV - sumlog;
        if (Lv < 64)
          st->bfs_A[Lv][k] = (float)A;
        const double R2 = (pruning ? pruning[k] : 1.0) * maxdist;
        tot += std::exp(std::min(A + 0.5 * n * std::log(R2 > 0 ? R2 : 1e-300), 80.0));
      }
      bfs_est0[Lv] = (float)std::min(tot, 1e30);
    }
  }
  st->sol_head   = ctx->ring_next;
  st->bound_bits = dbits(maxdist);
  __atomic_store_n(&ctx->h->bound_bits, dbits(maxdist), __ATOMIC_RELEASE);
  __atomic_store_n(&ctx->h->consumed, ctx->ring_next, __ATOMIC_RELEASE);
  HIPCHK(ctx, hipMemcpyAsync(ctx->g, st, st_used, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(ctx->qm, 0, sizeof(QueueMem), ctx->stream));
  // root task: level d, zero partial sums, zero prefix, zero partial distance (the breadth-first
  // stage reads its first frontier from buf[1] and leaves the final task list in buf[0])
  int cur = use_bfs ? 1 : 0;
  HIPCHK(ctx, hipMemsetAsync(ctx->buf[cur].col, 0, 64 * sizeof(double), ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(ctx->buf[cur].x, 0, 64 * sizeof(double), ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(ctx->buf[cur].pd, 0, sizeof(double), ctx->stream));
  HIPCHK(ctx, hipMemcpyAsync(ctx->buf[cur].level, &d, sizeof(int), hipMemcpyHostToDevice,
                             ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(ctx->buf[cur].root, 0, sizeof(int), ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(ctx->xhi_root, 0, 64 * sizeof(double), ctx->stream));
  // Blocks larger than 64: one wave walks levels 64..d-1 and leaves the surviving level-64 nodes
  // as the initial tasks (enum_top_kernel).  Nothing can have been reported yet, so an overfull
  // task buffer is still a clean decline.
  unsigned top_tasks = 0;
  double top_ms      = 0.0;
  int launch_idx     = 0;
  if (d > 64)
  {
    const unsigned cap2 = (unsigned)env_int("FPHIP_TOP_TASK_CAP", 32768);
    for (int b = 0; b < 2 && !ctx->top[1].col; ++b)
    {
      // (sized for 256 rows: the kernels of blocks up to 128 rows use row strides 128 / 64 inside them)
      HIPCHK(ctx, fphip_dev_alloc((void **)&ctx->top[b].col, (size_t)cap2 * 256 * sizeof(double), ctx->stream));
      HIPCHK(ctx, fphip_dev_alloc((void **)&ctx->top[b].xhi, (size_t)cap2 * 192 * sizeof(double), ctx->stream));
      HIPCHK(ctx, fphip_dev_alloc((void **)&ctx->top[b].pd, (size_t)cap2 * sizeof(double), ctx->stream));
      HIPCHK(ctx, fphip_dev_alloc((void **)&ctx->top[b].level, (size_t)cap2 * sizeof(int), ctx->stream));
      HIPCHK(ctx, fphip_dev_alloc((void **)&ctx->top[b].count, 64, ctx->stream));
      ctx->top[b].cap = cap2;
    }
    // root top task: level d, zero column, no coefficient chosen, zero partial distance
    HIPCHK(ctx, hipMemsetAsync(ctx->top[0].col, 0, 256 * sizeof(double), ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->top[0].xhi, 0, 192 * sizeof(double), ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->top[0].pd, 0, sizeof(double), ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->top[0].level, &d, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->top[1].count, 0, 4, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(ctx->buf[cur].count, 0, 4, ctx->stream));
    // the column stack of the levels above 64: in LDS up to 128 rows (49 KB there), above that in a per-wave
    // region of global memory (246 KB at 256 rows)
    const bool wide     = d > 128;
    const size_t tstack = (size_t)((d + 1) * d / 2 - 65 * 64 / 2);  // doubles per wave
    const size_t tlds   = wide ? 0 : tstack * sizeof(double);
    const unsigned top_waves_max = (unsigned)ctx->num_cus * 4u;
    if (wide && ctx->gtop_doubles < tstack * top_waves_max)
    {
      if (ctx->gtop)
        fphip_dev_free(ctx->gtop, ctx->stream);
      ctx->gtop         = nullptr;
      ctx->gtop_doubles = 0;
      HIPCHK(ctx, fphip_dev_alloc((void **)&ctx->gtop, tstack * top_waves_max * sizeof(double), ctx->stream));
      ctx->gtop_doubles = tstack * top_waves_max;
    }
    // xhi_root holds cap rows of 64 doubles; a block above 128 rows needs 128 or 192 per level-64 node: grown
    // once, by the first such call of the context (1.6 GB at the default cap)
    TaskBuf top_out = ctx->buf[cur];
    if (wide && ctx->xhi_row < 192)
    {
      fphip_dev_free(ctx->xhi_root, ctx->stream);
      ctx->xhi_root = nullptr;
      ctx->xhi_row  = 64;
      HIPCHK(ctx, fphip_dev_alloc((void **)&ctx->xhi_root, (size_t)ctx->cap * 192 * sizeof(double), ctx->stream));
      HIPCHK(ctx, hipMemsetAsync(ctx->xhi_root, 0, 192 * sizeof(double), ctx->stream));
      ctx->xhi_row = 192;
    }
    // cut level of the first (single-wave) top launch: where the Gaussian heuristic expects a few
    // thousand nodes; the second launch walks those subtrees in parallel down to level 64
    int cut = 64;
    for (int k = d - 1; k > 64; --k)
      if (logN[k] - logN[d] >= std::log(2048.0))
      {
        cut = k;
        break;
      }
    // the top walk is replicated on every rank (like the split launches below): only shard 0
    // counts its nodes, so that the per-level counts summed over the ranks stay the reference's
    const int top_count = (o.shard_index == 0) ? 1 : 0;
    auto top_launch = [&](int in_idx, unsigned n_in, int stop, unsigned grid) -> int
    {
      HIPCHK(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
#define FPHIP_TOP_LAUNCH(N, S_, D_)                                                                                \
  hipLaunchKernelGGL((enum_top_kernel<N, S_, D_>), dim3(grid), dim3(64), tlds, ctx->stream, ctx->g, ctx->h,          \
                     ctx->top[in_idx], n_in, ctx->top[in_idx ^ 1], stop, top_out, ctx->xhi_root, d, maxdist,        \
                     top_count, launch_idx, ctx->gtop)
      if (wide)
      {
        if (dual)
          FPHIP_TOP_LAUNCH(4, false, true);
        else if (subs)
          FPHIP_TOP_LAUNCH(4, true, false);
        else
          FPHIP_TOP_LAUNCH(4, false, false);
      }
      else if (dual)
        FPHIP_TOP_LAUNCH(2, false, true);
      else if (subs)
        FPHIP_TOP_LAUNCH(2, true, false);
      else
        FPHIP_TOP_LAUNCH(2, false, false);
#undef FPHIP_TOP_LAUNCH
      HIPCHK(ctx, hipGetLastError());
      HIPCHK(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
      uint64_t nsub = 0;  // sub-solutions of the top levels arrive through the ring meanwhile
      int rcw       = wait_serving(ctx, d, cb, subcb, user, &nsub);
      if (rcw != FPHIP_OK)
        return rcw;
      float tms = 0;
      HIPCHK(ctx, hipEventElapsedTime(&tms, ctx->ev[0], ctx->ev[1]));
      top_ms += tms;
      ++launch_idx;
      return FPHIP_OK;
    };
    int rct = top_launch(0, 1u, cut, 1u);
    if (rct != FPHIP_OK)
      return rct;
    if (cut > 64)
    {
      unsigned c1 = 0;
      HIPCHK(ctx, hipMemcpy(&c1, ctx->top[1].count, 4, hipMemcpyDeviceToHost));
      if (c1 > cap2)
      {
        snprintf(ctx->err, sizeof ctx->err, "more than %u top tasks: the block stays on the CPU enumerator", cap2);
        return FPHIP_UNSUPPORTED;
      }
      if (c1 > 0)
      {
        const unsigned per_cu = wide ? 4u : (unsigned)std::max<size_t>(1, (160 * 1024) / std::max<size_t>(tlds, 1));
        const unsigned grid   = std::min<unsigned>(c1, (unsigned)ctx->num_cus * std::min(per_cu, wide ? 4u : 8u));
        rct                   = top_launch(1, c1, 64, grid);
        if (rct != FPHIP_OK)
          return rct;
      }
    }
    HIPCHK(ctx, hipMemcpy(&top_tasks, ctx->buf[cur].count, 4, hipMemcpyDeviceToHost));
    if (top_tasks > top_out.cap)
    {
      snprintf(ctx->err, sizeof ctx->err,
               "more than %u surviving nodes at level 64: the block stays on the CPU enumerator", top_out.cap);
      return FPHIP_UNSUPPORTED;
    }
  }

  const int debug     = env_int("FPHIP_DEBUG", 0);
  const bool walk2    = env_int("FPHIP_WALK2", 1) != 0;
  uint64_t nsol       = 0;
  double kernel_ms    = top_ms, final_ms = 0.0;
  int launches        = 0;
  int L               = d > 64 ? 64 : d;  // highest root level among the current tasks (LDS geometry)
  unsigned C          = d > 64 ? top_tasks : 1;  // number of current tasks
  int final_tasks     = 0, final_L = d;
  const int max_split = env_int("FPHIP_MAX_SPLIT_PHASES", 6);
  // Work-donation budget (loop iterations a task may run before it sheds its upper subtrees).
  // Sized from the Gaussian-heuristic node estimate so that a wave sees ~8 budgets of work.
  double est_nodes = 0;
  for (int k = 0; k < d; ++k)
    est_nodes += std::exp(std::min(logN[k], 60.0));
  // hard per-task cap; the operative trigger is the drained-queue signal inside the kernel
  unsigned budget       = (unsigned)env_int("FPHIP_BUDGET", 1 << 22);
  const int max_rounds  = env_int("FPHIP_MAX_ROUNDS", 24);
  if (debug)
    fprintf(stderr, "[fphip] d=%d est_nodes=%.3e budget=%u\n", d, est_nodes, budget);
  unsigned long long moved_tasks = 0;  // work movement between ranks: tasks that left or reached this rank
  unsigned xhi_used = d > 64 ? top_tasks : 0;  // rows of xhi_root taken (level-64 ancestors; received tasks append theirs)
  bool in_final       = false;  // false: level-cut splitting phases; true: budgeted walk rounds
  int round           = 0;
  unsigned prevC      = 0;

  // ---- breadth-first stage: root (or the level-64 tasks of the top walk) -> final task list ------
  bool regioned = false;  // the current task list is a regioned buffer (what the stage leaves in buf[0])
  unsigned n_slots = 0;   // multi-GPU: length of the compact slot list of the regioned buffer
  bool bfs_sharded = false;  // multi-GPU: the stage ended with a task list of this rank's own (nothing to deal)
  if (use_bfs && C > 0)
  {
    FPHIP_RANGE("enum: breadth-first stage");
    const int L0        = L;
    const unsigned rcap = ctx->cap / FPHIP_NQ;
    // a subtree counts as heavy above `heavy` estimated nodes: enough final tasks to balance the
    // walk, few enough to fit the buffers (the estimate runs 2-5x high on pruned trees)
    const double want_tasks = o.target_tasks > 0 ? o.target_tasks : env_int("FPHIP_BFS_TASKS", 65536);
    float heavy = (float)std::max((double)env_int("FPHIP_BFS_HEAVY", 1024), est_nodes / want_tasks);
    // levels: down to the first one where even a node of partial distance 0 is light
    int Lend = 1;
    for (int Lv = L0 - 1; Lv >= 1; --Lv)
      if (bfs_est0[Lv] <= heavy)
      {
        Lend = Lv;
        break;
      }
    Lend = std::max(Lend, std::min(L0 - 1, env_int("FPHIP_BFS_FLOOR", 4)));
    const int cnt_bfs = (o.shard_index == 0) ? 1 : 0;  // the replicated part (on every rank): shard 0 counts
    // Multi-GPU: the thin top (one workgroup) is replicated; from the first level that is launched over the chip
    // every rank expands only ITS share of the frontier — the parents whose coefficient prefix hashes to it — and
    // ends with a task list of its own: nothing is dealt afterwards, the lists are levelled by the work movement
    // (gather) like donated subtrees.  Blocks above 64 rows keep the replicated stage (their tasks are told apart
    // by an index into a table whose order is the rank's own).
    const bool shard_bfs = o.shard_count > 1 && d <= 64 && o.exchange && env_int("FPHIP_BFS_SHARD", 0) != 0;
    // the thin top in ONE single-workgroup launch (a barrier between levels), then a launch per level
    int n_single = 0;
    for (int Lv = L0; Lv > Lend; --Lv)
    {
      const double est_parents = (Lv == L0 ? (double)C : std::exp(std::min(logN[Lv] - logN[L0], 40.0)) * C);
      if (est_parents > env_int("FPHIP_BFS_SINGLE_MAX", 256))
        break;
      ++n_single;
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    int Lv = L0, par = 0;  // par: which of buf[1] / buf[2] holds the frontier of level Lv
    int compact_n = (int)C;  // the first frontier is a compact list in buf[1] (root / top-walk tasks)
    auto bfs_launch = [&](int nlev, unsigned grid, unsigned threads)
    {
      const TaskBuf &fa = ctx->buf[1 + par], &fb = ctx->buf[2 - par];
      // (the filter acts on ONE launch — the first over the chip; behind it the frontier is the rank's own)
      const bool filter = shard_bfs && grid > 1u && !bfs_sharded;
      // shard mode of the launch: 0 none, 1 = the parents are this rank's share (the one launch that splits the
      // frontier), 2 = replicated parents, but a LIGHT child — a final task — stays only on the rank it hashes to
      // (the launches in front of the split: their final tasks must not be walked by everybody)
      const int smode   = filter ? 1 : ((shard_bfs && !bfs_sharded) ? 2 : 0);
      const int scount  = smode ? o.shard_count * 4 + smode : 0;  // (count and mode in one argument; 0: no sharding)
      const int cnt     = (bfs_sharded || filter) ? 1 : cnt_bfs;
      if (filter)
        bfs_sharded = true;
      if (dual)
        hipLaunchKernelGGL((enum_bfs_kernel<true>), dim3(grid), dim3(threads), 0, ctx->stream, ctx->g, maxdist,
                           ctx->qm, fa, fb, ctx->buf[0], Lv, nlev, Lend, heavy, cnt, compact_n, o.shard_index, scount);
      else
        hipLaunchKernelGGL((enum_bfs_kernel<false>), dim3(grid), dim3(threads), 0, ctx->stream, ctx->g, maxdist,
                           ctx->qm, fa, fb, ctx->buf[0], Lv, nlev, Lend, heavy, cnt, compact_n, o.shard_index, scount);
      Lv -= nlev;
      par ^= nlev & 1;
      compact_n = -1;
      ++launches;
    };
    if (n_single > 0)
      bfs_launch(n_single, 1u, 1024u);
    // (waves per launch: a multiple of FPHIP_NQ — each wave reads one region)
    const unsigned bgrid = std::max(32u, ((unsigned)ctx->num_cus * (unsigned)env_int("FPHIP_BFS_WG_PER_CU", 1)) / 32u * 32u);
    while (Lv > Lend)
      bfs_launch(1, bgrid, 256u);
    const bool want_slots = o.shard_count > 1 && !bfs_sharded;
    hipLaunchKernelGGL(enum_bfs_epilogue, dim3(1), dim3(1024), 0, ctx->stream, ctx->g, ctx->h, ctx->qm, rcap,
                       want_slots ? ctx->slots : nullptr, want_slots ? ctx->pdc : nullptr, ctx->buf[0].pd);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    float bms = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&bms, ctx->ev[0], ctx->ev[1]));
    kernel_ms += bms;
    const unsigned nfin = (unsigned)__atomic_load_n(&ctx->h->pad[0], __ATOMIC_ACQUIRE);
    const unsigned flg  = (unsigned)__atomic_load_n(&ctx->h->pad[1], __ATOMIC_ACQUIRE);
    if (debug)
      fprintf(stderr, "[fphip s%d] bfs levels %d..%d (%d in one workgroup), heavy > %.0f nodes: %u tasks, "
                      "%.3f ms%s\n", o.shard_index, L0, Lend, n_single, heavy, nfin, bms,
              (flg & FPHIP_FLAG_BFS_OVERFLOW) ? " OVERFLOW: starting over with split launches" : "");
    bool bfs_over = (flg & FPHIP_FLAG_BFS_OVERFLOW) != 0;
    if (o.exchange)
    {
      // multi-GPU: WHICH region of a buffer overflows depends on the order of the atomics, so one
      // rank may overflow where another does not — and a rank that started over with the split
      // launches would hold another task set than the others.  One more collective (the bound
      // cannot have moved yet) makes the decision common: everybody starts over if anybody must.
      int any = 0;
      (void)o.exchange(o.exchange_user, bdbl(__atomic_load_n(&ctx->h->bound_bits, __ATOMIC_ACQUIRE)),
                       bfs_over ? 1 : 0, &any);
      bfs_over = any != 0;
    }
    if (bfs_over)
    {
      // start over with the split launches — nothing has been reported yet, the counters are uploaded again.  A
      // block above 64 rows repeats its top walk as well (milliseconds; until round 6 such a block was declined).
      use_bfs = false;
      ++bfs_restarts;
      goto restart;
    }
    cur         = 0;
    C           = nfin;
    L           = L0 - 1;
    in_final    = true;
    final_tasks = (int)C;
    final_L     = L;
    regioned    = true;
    n_slots     = want_slots ? nfin : 0u;
  }

  // multi-GPU: some other rank still has tasks (a rank whose share of a sharded stage came out empty still takes
  // part in every round boundary)
  bool others_active = bfs_sharded;
  while (C > 0 || others_active)
  {
    int stop = -1;
    if (!in_final)
    {
      if (launches < max_split && (double)C < 0.5 * target_final)
        stop = choose_stop(logN, L, (double)C, target_final, growth);
      if (stop >= L)
        stop = L - 1;
      if (stop < 1)
        stop = -1;
      if (stop < 0)
      {
        in_final    = true;
        final_tasks = (int)C;
        final_L     = L;
      }
    }
    const int wpb     = (in_final && C >= 1024) ? wpb_final : 1;
    const int triL    = L * (L + 1) / 2;
    // big walk launches read mu through L1 and spend the LDS on more resident waves
    // (only where the LDS limits residency the most — tall stacks — and the launch is large;
    // measured +7 % on a 10^9-node tree with L = 49, but -5..10 % on 3 M-node trees with L = 44)
    const bool mu_lds = C < (unsigned)env_int("FPHIP_MU_GLOBAL_MIN_TASKS", 8192) ||
                        L < env_int("FPHIP_MU_GLOBAL_MIN_LEVEL", 47);
    // Split stack: the slots k < Ts of the column stack stay in LDS, the tall ones go to a per-wave
    // scratch in global memory.  Ts is the largest level whose LDS part still lets 32 waves (8 per
    // SIMD) reside on a CU in the big walk launches (tri_off(34) + 64 = 625 doubles = 5 KB per wave); the
    // split launches and small trees keep the whole stack in LDS.
    // Sub-solution calls do not split the stack (FPHIP_SUBS_SPLIT=1 brings the split back): their walk with the tall
    // slots in global memory gave per-level counts that changed from run to run on a 130-row block.  The cause turned
    // out to be the allocator — the scratch had left hipMallocAsync microseconds earlier (dev_mem.h, DESIGN.md
    // section 6) — but the fix for that has no full suite behind it yet, so this stays as it was validated.
    int Ts = L + 1;
    if (in_final && C >= 1024 && !mu_lds && (!subs || env_int("FPHIP_SUBS_SPLIT", 0) != 0))
    {
      const int want = env_int("FPHIP_STACK_SPLIT", 34);
      if (want > 1 && want < Ts)
        Ts = want;
    }
    const int ldsRow = (Ts * (Ts - 1)) / 2;
    // (+ the pad per wave the lanes beyond a short stack row write to)
    const size_t lds  = ((size_t)(mu_lds ? triL : 0) + (size_t)wpb * (ldsRow + FPHIP_STACK_PAD)) * sizeof(double);
    if (lds > 160 * 1024)
      return fail(ctx, "LDS request too large (%zu)", lds);
    int blocks_per_cu = (int)std::max<size_t>(1, std::min<size_t>(32 / wpb, (160 * 1024) / lds));
    {
      const size_t per_wave = (size_t)(triL - ldsRow + 1);
      size_t need           = per_wave * (size_t)wpb * (size_t)ctx->num_cus * (size_t)blocks_per_cu;
      // Never smaller than what the largest default launch asks for (64 levels, split at 34, 32 waves per CU: 100 MB
      // on 256 CUs): a context allocates this scratch once, with its first big launch.
      need = std::max(need, (size_t)(64 * 65 / 2 - 34 * 33 / 2 + 1) * 32u * (size_t)ctx->num_cus);
      if (need > ctx->gstk_doubles)
      {
        if (ctx->gstk)
          fphip_dev_free(ctx->gstk, ctx->stream);
        ctx->gstk = nullptr;
        // + 256 doubles of padding: the STEP loop loads the column / mu row of level k + 1 one test
        // ahead, before it knows that k + 1 exists (enum_kernel.hip) — for the last wave that read
        // lands up to Lmax - 1 doubles behind its stack; the value is never used
        HIPCHK(ctx, fphip_dev_alloc((void **)&ctx->gstk, (need + 256) * sizeof(double), ctx->stream));
        ctx->gstk_doubles = need;
      }
    }
    const int nxt     = cur ^ 1;
    // only the first final round is sharded across GPUs; donated tasks stay on their GPU
    const bool shard_now = in_final && round == 0 && o.shard_count > 1 && !bfs_sharded;
    // (a regioned task list that is not dealt over ranks is drawn region by region: one launch)
    const int chunks     = (in_final && round == 0 && !(regioned && !shard_now)) ? o.exchange_chunks : 1;
    const unsigned *idxl = nullptr;
    unsigned n_list      = C;  // number of tasks this rank walks in this round
    if (shard_now)
    {
      // Partition by CONTENT, weight-aware: every rank holds the same task set (in its own
      // order).  Sort by (partial distance of the root, 64-bit key of the coefficient prefix) — a
      // total order that depends only on content; smaller partial distance = larger remaining
      // radius = heavier subtree — and deal the sorted list to the ranks in snake order.  The
      // shares are disjoint, complete and of nearly equal weight, and each rank walks its share
      // heaviest-first.
      const unsigned kgrid = std::min<unsigned>((C + 3) / 4, (unsigned)ctx->num_cus * 8u);
      // (a regioned list — the breadth-first stage's — goes through the compact list of its slots)
      const bool via_slots = regioned && n_slots == C;
      if (regioned && !via_slots)
        return fail(ctx, "internal: regioned task list without its slot list under sharding");
      hipLaunchKernelGGL(task_key_kernel, dim3(kgrid ? kgrid : 1), dim3(256), 0, ctx->stream,
                         ctx->buf[cur], C, d, ctx->keys, ctx->xhi_root, via_slots ? ctx->slots : nullptr);
      HIPCHK(ctx, hipGetLastError());
      if (env_int("FPHIP_DEAL_HOST", 0) == 0)
      {
        // sort and deal on the device (enum_deal.hip): no copy of the keys to the host, no host sort (6-9 ms for
        // 65 536 tasks, on every rank, in front of a walk of 41 ms at eight GPUs)
        const size_t need = deal_work_bytes(C);
        if (ctx->deal_work_bytes < need)
        {
          if (ctx->deal_work)
            fphip_dev_free(ctx->deal_work, ctx->stream);
          ctx->deal_work = nullptr;
          HIPCHK(ctx, fphip_dev_alloc(&ctx->deal_work, need, ctx->stream));
          ctx->deal_work_bytes = need;
        }
        n_list = deal_tasks_device(ctx->stream, ctx->keys, via_slots ? ctx->pdc : ctx->buf[cur].pd,
                                   via_slots ? ctx->slots : nullptr, C, (unsigned)o.shard_count, (unsigned)o.shard_index,
                                   ctx->idxlist, ctx->deal_work, ctx->deal_work_bytes);
        if (n_list == ~0u)
          return fail(ctx, "the device-side deal of the task list failed");
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        idxl = ctx->idxlist;
      }
      else
      {
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      std::vector<unsigned long long> keys(C);
      std::vector<double> pds(C);
      std::vector<unsigned> slot_of;
      HIPCHK(ctx, hipMemcpy(keys.data(), ctx->keys, (size_t)C * 8, hipMemcpyDeviceToHost));
      HIPCHK(ctx, hipMemcpy(pds.data(), via_slots ? ctx->pdc : ctx->buf[cur].pd, (size_t)C * 8, hipMemcpyDeviceToHost));
      if (via_slots)
      {
        slot_of.resize(C);
        HIPCHK(ctx, hipMemcpy(slot_of.data(), ctx->slots, (size_t)C * 4, hipMemcpyDeviceToHost));
      }
      std::vector<unsigned> order(C);
      for (unsigned i = 0; i < C; ++i)
        order[i] = i;
      std::sort(order.begin(), order.end(), [&](unsigned a, unsigned b) {
        if (pds[a] != pds[b])
          return pds[a] < pds[b];
        return keys[a] < keys[b];
      });
      std::vector<unsigned> mine;
      mine.reserve(C / o.shard_count + 2);
      const unsigned W = (unsigned)o.shard_count;
      for (unsigned p = 0; p < C; ++p)
      {
        const unsigned r = p % (2 * W);
        const unsigned owner = r < W ? r : 2 * W - 1 - r;  // snake: 0..W-1, W-1..0
        if (owner == (unsigned)o.shard_index)
          mine.push_back(via_slots ? slot_of[order[p]] : order[p]);
      }
      n_list = (unsigned)mine.size();
      if (n_list)
        HIPCHK(ctx, hipMemcpy(ctx->idxlist, mine.data(), (size_t)n_list * 4, hipMemcpyHostToDevice));
      idxl = ctx->idxlist;
      }
    }
    const int count_nodes = (in_final || o.shard_index == 0) ? 1 : 0;
    HIPCHK(ctx, hipMemsetAsync(ctx->buf[nxt].count, 0, 4, ctx->stream));

    for (int ch = 0; ch < chunks; ++ch)
    {
      unsigned lo = (unsigned)(((unsigned long long)n_list * ch) / chunks);
      unsigned hi = (unsigned)(((unsigned long long)n_list * (ch + 1)) / chunks);
      if (hi <= lo && !(in_final && o.exchange))
        continue;
      unsigned mine = hi - lo;
      unsigned grid = std::min<unsigned>((mine + wpb - 1) / wpb,
                                         (unsigned)(ctx->num_cus * blocks_per_cu));
      if (grid == 0)
        grid = 1;
      if (launch_idx >= FPHIP_MAX_LAUNCHES)
        return fail(ctx, "too many launches");
      if (hi > lo)
      {
        FPHIP_RANGE(in_final ? "enum: walk launch" : "enum: split launch");
        HIPCHK(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
        // (FPHIP_SUBS_DONATE=0: sub-solution calls walk without work donation — an A/B switch from the hunt for the
        //  run-to-run differences of the split stack above; donation was not their cause)
        const unsigned bud = (in_final && round < max_rounds && (!subs || env_int("FPHIP_SUBS_DONATE", 1) != 0)) ? budget : 0u;
#define FPHIP_LAUNCH(M, S, D)                                                                       \
  hipLaunchKernelGGL((enum_phase_kernel<M, S, D>), dim3(grid), dim3(wpb * 64), lds, ctx->stream, ctx->g, \
                     ctx->h, ctx->buf[cur], ctx->buf[nxt], d, L, stop, lo, hi, idxl, launch_idx,     \
                     count_nodes, bud, ctx->xhi_root, ctx->gstk, Ts, &ctx->qm->head[launch_idx][0],          \
                     (regioned && !shard_now) ? &ctx->qm->fin[0] : (const unsigned *)nullptr, ctx->cap / FPHIP_NQ, \
                     (unsigned long long)__atomic_load_n(&ctx->h->bound_bits, __ATOMIC_ACQUIRE))
        // the walk launches (no sub-solutions): the second-generation walk — all children of a node in one vector
        // test (enum_walk.hip); FPHIP_WALK2=0 keeps enum_phase_kernel (the A/B partner)
#define FPHIP_LAUNCH2(M, D)                                                                          \
  hipLaunchKernelGGL((enum_walk_kernel<M, D>), dim3(grid), dim3(wpb * 64), lds, ctx->stream, ctx->g,  \
                     ctx->h, ctx->buf[cur], ctx->buf[nxt], d, L, lo, hi, idxl, launch_idx,           \
                     count_nodes, bud, ctx->xhi_root, ctx->gstk, Ts, &ctx->qm->head[launch_idx][0],  \
                     (regioned && !shard_now) ? &ctx->qm->fin[0] : (const unsigned *)nullptr, ctx->cap / FPHIP_NQ, \
                     (unsigned long long)__atomic_load_n(&ctx->h->bound_bits, __ATOMIC_ACQUIRE))
        if (in_final && !subs && walk2)
        {
          if (dual && mu_lds)
            FPHIP_LAUNCH2(true, true);
          else if (dual)
            FPHIP_LAUNCH2(false, true);
          else if (mu_lds)
            FPHIP_LAUNCH2(true, false);
          else
            FPHIP_LAUNCH2(false, false);
        }
        else if (dual && mu_lds)
          FPHIP_LAUNCH(true, false, true);
        else if (dual)
          FPHIP_LAUNCH(false, false, true);
        else if (mu_lds && !subs)
          FPHIP_LAUNCH(true, false, false);
        else if (!mu_lds && !subs)
          FPHIP_LAUNCH(false, false, false);
        else if (mu_lds)
          FPHIP_LAUNCH(true, true, false);
        else
          FPHIP_LAUNCH(false, true, false);
#undef FPHIP_LAUNCH
#undef FPHIP_LAUNCH2
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
        ++launch_idx;
        int rc = wait_serving(ctx, d, cb, subcb, user, &nsol);
        if (rc != FPHIP_OK)
          return rc;
        float ms = 0.f;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
        kernel_ms += ms;
        if (in_final)
          final_ms += ms;
        if (debug)
        {
          unsigned long long it = 0;
          hipMemcpy(&it, &ctx->g->iters, 8, hipMemcpyDeviceToHost);
          fprintf(stderr,
                  "[fphip s%d] launch %d %s tasks [%u,%u) L=%d stop=%d budget=%u grid=%u wpb=%d lds=%zu "
                  "-> %.3f ms, iters so far %llu\n",
                  o.shard_index, launch_idx - 1, in_final ? "walk" : "split", lo, hi, L, stop,
                  in_final ? budget : 0u, grid, wpb, lds, ms, it);
        }
      }
      if (in_final && o.exchange && ch + 1 < chunks)
      {  // multi-GPU: agree on the best bound at every chunk boundary (collective: every rank
         // makes the same sequence of exchange calls)
        double local = bdbl(__atomic_load_n(&ctx->h->bound_bits, __ATOMIC_ACQUIRE));
        int any      = 1;
        double glob  = o.exchange(o.exchange_user, local, 1, &any);
        if (glob < local)
          publish_bound_min(ctx, glob);  // never raises: another host thread may have lowered it since
      }
    }
    ++launches;
    unsigned cnt = 0;
    if (C > 0)
    {
      // The count was zeroed with a memset on OUR stream; a plain hipMemcpy runs on the null stream,
      // which does not order against a non-blocking stream — it is only safe because every launch
      // above was waited for.  Make that explicit (an idle multi-GPU iteration launches nothing:
      // reading the count there returned the stale value of two rounds ago and re-ran old tasks).
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      HIPCHK(ctx, hipMemcpy(&cnt, ctx->buf[nxt].count, 4, hipMemcpyDeviceToHost));
    }
    if (!in_final && o.shard_count > 1 && cnt > ctx->cap)
    {
      // The multi-GPU partition deals the tasks of the replicated split launches by content; that
      // needs every rank to hold the SAME task set.  After an overflow, which tasks made it into
      // the buffer depends on the order of the atomics and differs per rank: a subtree could be
      // walked twice or by nobody.  (A single GPU walks the overflow inline and stays exact.)
      // The COUNT, however, is a property of the replicated tree — the same on every rank — so every
      // rank takes the same decision here without talking: start over with an eighth of the task
      // target (nothing has been reported yet: candidates only appear in the walk launches).
      if (d > 64 || ++overflow_retries > 3)
        return fail(ctx, "task buffer overflow in a split launch (%u tasks, capacity %u) under sharding: "
                         "raise FPHIP_TASK_CAP or lower the task target", cnt, ctx->cap);
      target_final = std::max(64.0, target_final / 8.0);
      use_bfs      = false;
      goto restart;
    }
    if (cnt > ctx->cap)
      cnt = ctx->cap;
    if (in_final && o.exchange)
    {  // round boundary: exchange the bound and learn whether any rank still has tasks
      double local = bdbl(__atomic_load_n(&ctx->h->bound_bits, __ATOMIC_ACQUIRE));
      int any      = cnt > 0;
      double glob  = o.exchange(o.exchange_user, local, cnt > 0, &any);
      if (glob < local)
        publish_bound_min(ctx, glob);
      others_active = any != 0;
      // work movement: while anybody still has tasks, the ranks level their lists (a task of a block above 64 rows
      // travels with the row of its level-64 ancestor)
      if (o.gather && o.shard_count > 1 && any != 0)
      {
        unsigned mv = 0;
        const int rcm = rebalance_tasks(ctx, o, ctx->buf[nxt], &cnt, &mv, d, &xhi_used);
        if (rcm != FPHIP_OK)
          return rcm;
        moved_tasks += mv;
        if (debug && mv)
          fprintf(stderr, "[fphip s%d] round %d: %u tasks moved, %u here now\n", o.shard_index, round, mv, cnt);
      }
    }
    if (in_final)
    {
      // donated tasks are rooted strictly below the level their donor was rooted at
      ++round;
      prevC = cnt;
      if (L > 1)
        L = L - 1;
    }
    else
    {
      L = stop;
    }
    C        = cnt;
    cur      = nxt;
    regioned = false;  // what a launch emits is a compact list
  }
  const int phases = launches;

  // ---- results --------------------------------------------------------------------------------
  HIPCHK(ctx, hipMemcpy(st, ctx->g, offsetof(DevShared, task_head), hipMemcpyDeviceToHost));
  if (st->error_flags & FPHIP_ERR_RING_TIMEOUT)
    return fail(ctx, "device timed out waiting for the host ring consumer");
  uint64_t total = 0;
  for (int k = 0; k <= d; ++k)
    nodes_out[k] = (k < d) ? st->nodes[k] : 0;
  if (o.shard_index == 0)
    for (int k = 1; k < d; ++k)
      nodes_out[k]--;  // enumerate_base.cpp:181-184: the initial descent is not counted
  for (int k = 0; k < d; ++k)
    total += nodes_out[k];
  if (stats)
  {
    stats->total_nodes      = total;
    stats->solutions        = nsol;
    stats->kernel_ms        = kernel_ms;
    stats->final_kernel_ms  = final_ms;
    stats->phases           = phases;
    stats->final_tasks      = final_tasks;
    stats->final_root_level = final_L;
    stats->overflowed       = (st->error_flags & FPHIP_FLAG_TASK_OVERFLOW) ? 1 : 0;
    stats->bfs_restarts     = bfs_restarts;
    stats->moved_tasks      = moved_tasks;
    stats->wall_ms =
        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin)
            .count();
  }
  return FPHIP_OK;
}
