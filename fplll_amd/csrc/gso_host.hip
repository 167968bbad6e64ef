// gso_host.hip — C ABI (include/fplll_hip.h) for the batched device-resident GSO:
// MatGSO<Z_NR<long>, FP_NR<double>> look-alike entry points at SWEEP granularity
// (update_gso / size_reduction over a row range), because a per-row device call would be
// launch-bound (SURVEY.md §7 "Granularity").

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <limits>
#include <cstring>
#include <utility>
#include <algorithm>
#include <vector>
#include <string>
#include <chrono>
#include <fstream>
#include <sstream>
#include <iomanip>

#include "../../include/fplll_hip.h"
#include "../../include/fplll_hip_debug.h"
#include "gso_device.h"
#include "gso_sweep2.h"
#include "dev_mem.h"
#include "trace.h"
#include "ftx.h"
#include "pruner_engine.h"
namespace fphip_pruner
{
// pruner_search.hip: prune<FP_NR<double>> of one block under PRUNER_METRIC_PROBABILITY_OF_SHORTEST on the
// given volume engine (nullptr: the host loop)
__attribute__((visibility("hidden"))) int prune_block(VolumeEngine *eng, int n, const double *gso_r, double radius,
                                                      double preproc, double target, int flags,
                                                      double *coefficients, double *expectation);
}  // namespace fphip_pruner

#ifndef FPHIP_GSO_RING
#define FPHIP_GSO_RING 6
#endif

namespace fphip
{
template <int NQ> __global__ void gso_sweep_kernel(GsoBatch P, int kmin, int kend, double eta, int mode);
template <int NQ>
__global__ void bkz_kernel(GsoBatch P, int block_size, double delta, double eta, double logdelta,
                           int use_max_loops, int max_loops, int stack_doubles);
template <int NQ>
__global__ void bkzs_kernel(GsoBatch P, BkzStrat S, BkzMail *mailbox, int *abort_flag, int block_size,
                            int top_flags, double delta, double eta, double logdelta, int max_loops,
                            int stack_doubles);
namespace sdv
{
template <int NQ>
__global__ void bkzd_kernel(GsoBatch P, BkzStrat S, BkzMail *mailbox, int *abort_flag, int block_size,
                            int top_flags, double delta, double eta, double logdelta, int max_loops,
                            int stack_doubles, int run_mode);
}
template <int NQ> __global__ void hlll_kernel(HhBatch P, double delta, double theta, long long iter_cap);
template <int NQ> __global__ void hh_blocked_kernel(HhBatch P, double *Tbuf);
struct HlllX
{
  double *Rlo, *Vlo;
  double *sc;
  long long *prevE;
  double delta, theta;
  long long iter_cap;
  const int *only_failed;
  double *Thi, *Tlo;
  double *Rx, *Vx;
};
template <int NQ, class FT> __global__ void hlll_x_kernel(HhBatch P, HlllX X);
struct LllX
{
  int batch, d, n, ldn, ldd, row_expo;
  long long *b, *b2;
  double *bf;
  double *mu_hi, *mu_lo;
  double *r_hi, *r_lo;
  double *gf_hi, *gf_lo;
  double *mu_x, *r_x, *gf_x;
  long long *rexp;
  int *status, *info;
  const int *only_failed;
  int kmin, kstart, kend;
  double delta, eta;
};
template <int NQ, class FT> __global__ void lll_x_kernel(LllX A);
__global__ void dd_op_kernel(const double *ahi, const double *alo, const double *bhi, const double *blo,
                             double *ohi, double *olo, int op, int count);
template <int NQ, bool EARLY>
__global__ void lll_kernel(GsoBatch P, int kmin, int kstart, int kend, double delta, double eta,
                           double logdelta);
}
using namespace fphip;

hipStream_t fphip_ctx_stream(fphip_ctx *ctx);
char *fphip_ctx_errbuf(fphip_ctx *ctx);
int fphip_ctx_num_cus(fphip_ctx *ctx);
int fphip_ctx_device(fphip_ctx *ctx);
int fphip_ctx_ensure_task_buffers(fphip_ctx *ctx);  // enum_host.hip
int fphip_ctx_set_task_cap(fphip_ctx *ctx, unsigned cap);

struct fphip_gso
{
  fphip_ctx *ctx;
  GsoBatch P;
  hipEvent_t ev[2];
  float last_ms;
  int waves_per_block;
  int blocks_per_cu;
  bool dirty;  // the integer basis was uploaded after the last (re)float of the rows
  // planes of mu for gso_sweep2_kernel (low / high words; row-major and transposed), [batch][2][d][ldd]
  double *muA;  // anchored transpose of mu for gso_sweep2_kernel's recurrence, [batch][d][ldd] (gso_sweep2.hip)
  short *m16;   // 2-byte mirrors: [batch][n*ldd + d*ldn] (bT16 then b16 of each lattice)
  int *flag16;  // [batch][d]
  int sweep_version;  // 2 (default) or 1 (FPHIP_GSO_SWEEP=1: the first-generation kernel)
  fphip_ctx *ectx;    // hand-off mode of the strategy-BKZ kernel: the enumeration context (same device)
  std::vector<fphip_ctx *> ectx_more;  // … and the further ones of a BATCH of tours (one per hand-off worker)
  double *xbuf;       // lll_x.hip workspace: bf rows [B][d][ldn], then the low planes of mu, r, gf [B][d][ldd] each
  int xbuf_planes = 0;  // ... 3 of them, or 9 once a quad-double run has asked for components 2 and 3
  // in-loop pruning of the strategy-BKZ service (FPHIP_BKZ_PRUNE_IN_LOOP; fphip_gso_bkz_inloop_pruning)
  double il_preproc = 1e6, il_target = 0.5;
  int il_min_block = 24, il_flags = 0x4 /* PRUNER_GRADIENT */, il_device = 1;
  unsigned long long il_calls = 0, il_device_jobs = 0, il_host_jobs = 0, il_launches = 0;
  // BKZ_MAX_TIME / BKZ_DUMP_GSO of the device BKZ drivers (fphip_gso_bkz_limits): seconds, file name
  double bkz_max_time = 0.0;
  std::string bkz_dump_path = "gso.json";  // (BKZParam::dump_gso_filename's default, bkz_param.h:120)
  // resident LLL session (fphip_gso_session_lll): the rows live in the kernel's slots while it is active
  bool session_active  = false;
  bool u_in_slots      = false;  // a session left the rows of u in the kernel's slots (restore_position_order)
  bool b_in_slots      = false;  // ... and those of b
  long long *sess_in_d = nullptr;  // device: positions then rows of the caller's row operations
  long long *sess_in_h = nullptr;  // pinned staging of the same
  char *sess_out_h     = nullptr;  // pinned: the state in position order after the last session call
};

static int gfail(fphip_ctx *ctx, const char *what, hipError_t e)
{
  snprintf(fphip_ctx_errbuf(ctx), 512, "%s failed: %s", what, hipGetErrorString(e));
  return FPHIP_ERROR;
}
static int gfail_msg(fphip_ctx *ctx, const char *msg)
{
  snprintf(fphip_ctx_errbuf(ctx), 512, "%s", msg);
  return FPHIP_ERROR;
}
#define GCHK(call)                            \
  do                                          \
  {                                           \
    hipError_t e_ = (call);                   \
    if (e_ != hipSuccess)                     \
      return gfail(g->ctx, #call, e_);        \
  } while (0)

// (internal hook of fphip_destroy, enum_host.hip: not exported)
extern "C" __attribute__((visibility("hidden"))) void fphip_gso_release_all(fphip_ctx *ctx) { (void)ctx; }

static int gso_allocate(fphip_gso *g);

extern "C" int fphip_gso_create(fphip_ctx *ctx, int batch, int d, int n, int row_expo,
                                fphip_gso **out)
{
  if (!ctx || !out)
    return FPHIP_ERROR;
  *out = nullptr;
  if (batch <= 0 || d <= 0 || n <= 0)
  {
    snprintf(fphip_ctx_errbuf(ctx), 512, "fphip_gso_create: bad shape");
    return FPHIP_ERROR;
  }
  if (d > 256 || n > 256)
    return FPHIP_UNSUPPORTED;  // caller keeps the lattice on the CPU path (fplll's MatGSO)
  if (!fphip_ctx_stream(ctx))
  {
    snprintf(fphip_ctx_errbuf(ctx), 512, "fphip_gso_create: context has no device");
    return FPHIP_ERROR;
  }
  fphip_gso *g = new fphip_gso();  // (value-initialised: every plain member zero, the defaults of the class applied)
  g->il_preproc = 1e6, g->il_target = 0.5, g->il_min_block = 24, g->il_flags = 0x4, g->il_device = 1;
  g->ctx        = ctx;
  g->P.batch    = batch;
  g->P.d        = d;
  g->P.n        = n;
  g->P.row_expo = row_expo ? 1 : 0;
  const int rc  = gso_allocate(g);
  if (rc != FPHIP_OK)
  {
    fphip_gso_destroy(g);  // frees whatever was allocated before the failure
    return rc;
  }
  *out = g;
  return FPHIP_OK;
}

static int gso_allocate(fphip_gso *g)
{
  fphip_ctx *ctx = g->ctx;
  const int batch = g->P.batch, d = g->P.d, n = g->P.n;
  // leading dimensions padded to 16 elements: every row starts on a 128-byte line, so the DMA
  // windows that begin at column 0 waste no partial line (FPHIP_GSO_LD_ALIGN=2 keeps them dense)
  const char *la_s = getenv("FPHIP_GSO_LD_ALIGN");
  const int la     = la_s ? atoi(la_s) : 16;
  g->P.ldd         = (d + la - 1) / la * la;
  g->P.ldn         = (n + la - 1) / la * la;
  if (g->P.ldd > 256)
    g->P.ldd = 256;
  if (g->P.ldn > 256)
    g->P.ldn = 256;
  const size_t B   = (size_t)batch;
  const size_t ldd = g->P.ldd, ldn = g->P.ldn;
  const size_t pad = 4096;  // the DMA ring reads whole 16-byte lanes past the end of a row
  GCHK(fphip_dev_alloc((void **)&g->P.b, B * d * ldn * sizeof(long long) + pad, fphip_ctx_stream(g->ctx)));
  GCHK(fphip_dev_alloc((void **)&g->P.bfT, B * n * ldd * sizeof(double) + pad, fphip_ctx_stream(g->ctx)));
  GCHK(fphip_dev_alloc((void **)&g->P.mu, B * d * ldd * sizeof(double) + pad, fphip_ctx_stream(g->ctx)));
  GCHK(fphip_dev_alloc((void **)&g->P.muT, B * d * ldd * sizeof(double) + pad, fphip_ctx_stream(g->ctx)));
  GCHK(fphip_dev_alloc((void **)&g->P.r, B * d * ldd * sizeof(double) + pad, fphip_ctx_stream(g->ctx)));
  GCHK(fphip_dev_alloc((void **)&g->P.rdg, B * d * sizeof(double), fphip_ctx_stream(g->ctx)));
  GCHK(fphip_dev_alloc((void **)&g->P.rexp, B * d * sizeof(long long), fphip_ctx_stream(g->ctx)));
  GCHK(fphip_dev_alloc((void **)&g->P.status, B * sizeof(int), fphip_ctx_stream(g->ctx)));
  GCHK(fphip_dev_alloc((void **)&g->P.bfT32, B * n * ldd * sizeof(float) + pad, fphip_ctx_stream(g->ctx)));
  GCHK(fphip_dev_alloc((void **)&g->P.b32, B * d * ldn * sizeof(int) + pad, fphip_ctx_stream(g->ctx)));
  GCHK(fphip_dev_alloc((void **)&g->P.narrow, B * d * sizeof(int), fphip_ctx_stream(g->ctx)));
  GCHK(fphip_dev_alloc((void **)&g->muA, B * d * ldd * sizeof(double) + pad, fphip_ctx_stream(g->ctx)));
  hipStream_t s0 = fphip_ctx_stream(ctx);
  GCHK(hipMemsetAsync(g->muA, 0, B * d * ldd * sizeof(double) + pad, s0));
  GCHK(fphip_dev_alloc((void **)&g->m16, B * ((size_t)n * ldd + (size_t)d * ldn) * sizeof(short) + pad, fphip_ctx_stream(g->ctx)));
  GCHK(fphip_dev_alloc((void **)&g->flag16, B * d * sizeof(int), fphip_ctx_stream(g->ctx)));
  GCHK(hipMemsetAsync(g->m16, 0, B * ((size_t)n * ldd + (size_t)d * ldn) * sizeof(short) + pad, s0));
  GCHK(hipMemsetAsync(g->flag16, 0, B * d * sizeof(int), s0));
  {
    const char *sv   = getenv("FPHIP_GSO_SWEEP");
    g->sweep_version = (sv && atoi(sv) == 1) ? 1 : 2;
  }
  GCHK(hipMemsetAsync(g->P.bfT32, 0, B * n * ldd * sizeof(float) + pad, s0));
  GCHK(hipMemsetAsync(g->P.b32, 0, B * d * ldn * sizeof(int) + pad, s0));
  GCHK(hipMemsetAsync(g->P.narrow, 0, B * d * sizeof(int), s0));
  {
    const char *nv = getenv("FPHIP_GSO_NARROW");  // 0 keeps every pass on the 8-byte rows (A/B runs)
    // FPHIP_GSO_NARROW: 0 = 8-byte arrays only, 1 = 4-byte mirrors, 2 (default) = 2-byte mirrors too
    g->P.use_narrow = nv ? atoi(nv) : 3;  // 3 = 2-byte mirrors + the integer AXPY fused into the Gram pass (gso_sweep2.hip)
    {
      const char *wr = getenv("FPHIP_GSO_WIDE_RING");
      g->P.wide_ring = wr ? atoi(wr) : 1;
    }
  }
  GCHK(hipMemsetAsync(g->P.b, 0, B * d * ldn * sizeof(long long) + pad, s0));
  GCHK(hipMemsetAsync(g->P.bfT, 0, B * n * ldd * sizeof(double) + pad, s0));
  GCHK(hipMemsetAsync(g->P.mu, 0, B * d * ldd * sizeof(double) + pad, s0));
  GCHK(hipMemsetAsync(g->P.muT, 0, B * d * ldd * sizeof(double) + pad, s0));
  GCHK(hipMemsetAsync(g->P.r, 0, B * d * ldd * sizeof(double) + pad, s0));
  GCHK(hipStreamSynchronize(s0));  // uploads use blocking copies on the null stream
  GCHK(hipEventCreate(&g->ev[0]));
  GCHK(hipEventCreate(&g->ev[1]));
  const char *w      = getenv("FPHIP_GSO_WAVES_PER_BLOCK");
  g->waves_per_block = w ? atoi(w) : 4;
  const char *bp     = getenv("FPHIP_GSO_BLOCKS_PER_CU");
  g->blocks_per_cu   = bp ? atoi(bp) : 0;  // 0 = as many as the LDS ring allows
  return FPHIP_OK;
}

extern "C" void fphip_gso_destroy(fphip_gso *g)
{
  if (!g)
    return;
  if (g->ectx)
    fphip_destroy(g->ectx);
  for (auto *c : g->ectx_more)
    fphip_destroy(c);
  if (g->xbuf)
    fphip_dev_free(g->xbuf, fphip_ctx_stream(g->ctx));
  hipStreamSynchronize(fphip_ctx_stream(g->ctx));
  fphip_dev_free(g->P.b, fphip_ctx_stream(g->ctx));
  fphip_dev_free(g->P.bfT, fphip_ctx_stream(g->ctx));
  fphip_dev_free(g->P.mu, fphip_ctx_stream(g->ctx));
  fphip_dev_free(g->P.muT, fphip_ctx_stream(g->ctx));
  fphip_dev_free(g->P.r, fphip_ctx_stream(g->ctx));
  fphip_dev_free(g->P.rdg, fphip_ctx_stream(g->ctx));
  fphip_dev_free(g->P.rexp, fphip_ctx_stream(g->ctx));
  fphip_dev_free(g->P.status, fphip_ctx_stream(g->ctx));
  fphip_dev_free(g->P.bfT32, fphip_ctx_stream(g->ctx));
  fphip_dev_free(g->P.b32, fphip_ctx_stream(g->ctx));
  fphip_dev_free(g->P.narrow, fphip_ctx_stream(g->ctx));
  fphip_dev_free(g->muA, fphip_ctx_stream(g->ctx));
  fphip_dev_free(g->m16, fphip_ctx_stream(g->ctx));
  fphip_dev_free(g->flag16, fphip_ctx_stream(g->ctx));
  fphip_dev_free(g->P.sess_slots, fphip_ctx_stream(g->ctx));
  fphip_dev_free(g->P.sess_state, fphip_ctx_stream(g->ctx));
  fphip_dev_free(g->P.sess_out, fphip_ctx_stream(g->ctx));
  fphip_dev_free(g->sess_in_d, fphip_ctx_stream(g->ctx));
  if (g->sess_in_h)
    fphip_pinned_put(g->sess_in_h);
  if (g->sess_out_h)
    fphip_pinned_put(g->sess_out_h);
  if (g->P.gf)
    fphip_dev_free(g->P.gf, fphip_ctx_stream(g->ctx));
  if (g->P.vc)
    fphip_dev_free(g->P.vc, fphip_ctx_stream(g->ctx));
  if (g->P.b2)
    fphip_dev_free(g->P.b2, fphip_ctx_stream(g->ctx));
  if (g->P.u)
    fphip_dev_free(g->P.u, fphip_ctx_stream(g->ctx));
  if (g->P.u2)
    fphip_dev_free(g->P.u2, fphip_ctx_stream(g->ctx));
  if (g->P.lll_info)
    fphip_dev_free(g->P.lll_info, fphip_ctx_stream(g->ctx));
  if (g->P.enum_mu)
    fphip_dev_free(g->P.enum_mu, fphip_ctx_stream(g->ctx));
  if (g->P.bkz_active)
    fphip_dev_free(g->P.bkz_active, fphip_ctx_stream(g->ctx));
  if (g->P.bkz_rows)
    fphip_dev_free(g->P.bkz_rows, fphip_ctx_stream(g->ctx));
  if (g->ev[0])
    hipEventDestroy(g->ev[0]);
  if (g->ev[1])
    hipEventDestroy(g->ev[1]);
  delete g;
}

struct LllArgs
{
  int kstart;
  double delta, logdelta;
};

// A session leaves the rows of b (and of u) in the kernel's slots; its last call also wrote them in position order
// behind the caller's view of the state: that copy becomes b / u when the session ends — by fphip_gso_set_basis
// (which may re-upload only part of the batch: the other lattices get their rows back in position order) or by a
// call that ended with a status other than 1 (the stateless entry points are no longer blocked after it and must
// find b and u consistent).
static int restore_position_order(fphip_gso *g)
{
  if (!g->P.sess_out)
    return FPHIP_OK;
  const size_t B = (size_t)g->P.batch, d = g->P.d, ldd = g->P.ldd, ldn = g->P.ldn;
  if (g->b_in_slots)
  {
    for (size_t L = 0; L < B; ++L)
      GCHK(hipMemcpy(g->P.b + L * d * ldn, g->P.sess_out + L * fphip_session_out_stride(d, ldd, ldn),
                     d * ldn * sizeof(long long), hipMemcpyDeviceToDevice));
    g->b_in_slots = false;
    g->dirty      = true;
  }
  if (g->u_in_slots && g->P.u)
  {
    for (size_t L = 0; L < B; ++L)
      GCHK(hipMemcpy(g->P.u + L * d * ldd,
                     g->P.sess_out + L * fphip_session_out_stride(d, ldd, ldn) + fphip_session_out_bytes(d, ldd, ldn),
                     d * ldd * sizeof(long long), hipMemcpyDeviceToDevice));
  }
  g->u_in_slots = false;
  return FPHIP_OK;
}

static int session_guard(fphip_gso *g, const char *what)
{
  if (!g->session_active)
    return FPHIP_OK;
  snprintf(fphip_ctx_errbuf(g->ctx), 512, "%s: a resident LLL session is active (the rows live in the kernel's slots); "
           "fphip_gso_set_basis ends it", what);
  return FPHIP_ERROR;
}

// With a transformation matrix on the device only the entry points that keep it in step with b may change b
static int transform_guard(fphip_gso *g, const char *what)
{
  if (!g->P.u)
    return FPHIP_OK;
  snprintf(fphip_ctx_errbuf(g->ctx), 512, "%s: the transformation matrix u is tracked (fphip_gso_enable_transform) and "
           "this entry point does not update it; fphip_gso_lll / fphip_gso_lll_flags do", what);
  return FPHIP_UNSUPPORTED;
}

static int launch(fphip_gso *g, int kmin, int kend, double eta, int mode, const LllArgs *la = nullptr)
{
  if (mode == 1)
    if (int rct = transform_guard(g, "size_reduce"))
      return rct;
  if (g->P.sess_mode != 2)
    if (int rcg = session_guard(g, "gso"))
      return rcg;
  if (mode == 2)
    g->dirty = false;
  else if (g->dirty)
  {  // a sweep on a freshly uploaded basis: float the rows first (MatGSO::update_bf for every row)
    const int rc0 = launch(g, 0, g->P.d, 0.0, 2);
    if (rc0 != FPHIP_OK)
      return rc0;
  }
  const int need = (g->P.d > g->P.n ? g->P.d : g->P.n);
  const int nq   = (need + 63) / 64;
  const int wpb  = g->waves_per_block;
  int grid       = (g->P.batch + wpb - 1) / wpb;
  hipStream_t s = fphip_ctx_stream(g->ctx);
  if (!la && g->sweep_version == 2)
  {  // gso_sweep2_kernel: its own ring geometry (gso_sweep2.h)
    const int ring   = nq == 1 ? s2::Cfg<1>::RING : nq == 2 ? s2::Cfg<2>::RING : nq == 3 ? s2::Cfg<3>::RING : s2::Cfg<4>::RING;
    const int wps    = nq == 4 ? s2::Cfg<4>::WAVES_PER_SIMD : nq == 3 ? s2::Cfg<3>::WAVES_PER_SIMD : s2::Cfg<1>::WAVES_PER_SIMD;
    const size_t lds2 = (size_t)wpb * ring;
    // NQ = 3: three blocks of four waves per CU, not the four that fit — measured 100.8-101.9 ms against
    // 104.5-105.1 ms at batch 8192 (three alternating repetitions on one box, profiles/r04_gso_roof_ab.log):
    // the kernel is bound by the memory system, a quarter fewer lattices in flight thrash it less
    int bpc2          = g->blocks_per_cu > 0 ? g->blocks_per_cu : std::min((int)((160 * 1024) / lds2), nq == 3 ? 3 : 4);
    if (bpc2 * wpb > 4 * wps)
      bpc2 = 4 * wps / wpb;
    if (bpc2 < 1)
      bpc2 = 1;
    const int cap2 = fphip_ctx_num_cus(g->ctx) * bpc2;
    if (grid > cap2)
      grid = cap2;
    GCHK(hipEventRecord(g->ev[0], s));
    switch (nq)
    {
    case 1:
      hipLaunchKernelGGL(s2::gso_sweep2_kernel<1>, dim3(grid), dim3(wpb * 64), lds2, s, g->P, g->muA, g->m16, g->flag16, kmin, kend, eta, mode);
      break;
    case 2:
      hipLaunchKernelGGL(s2::gso_sweep2_kernel<2>, dim3(grid), dim3(wpb * 64), lds2, s, g->P, g->muA, g->m16, g->flag16, kmin, kend, eta, mode);
      break;
    case 3:
      hipLaunchKernelGGL(s2::gso_sweep2_kernel<3>, dim3(grid), dim3(wpb * 64), lds2, s, g->P, g->muA, g->m16, g->flag16, kmin, kend, eta, mode);
      break;
    default:
      hipLaunchKernelGGL(s2::gso_sweep2_kernel<4>, dim3(grid), dim3(wpb * 64), lds2, s, g->P, g->muA, g->m16, g->flag16, kmin, kend, eta, mode);
      break;
    }
    GCHK(hipGetLastError());
    GCHK(hipEventRecord(g->ev[1], s));
    GCHK(hipStreamSynchronize(s));
    GCHK(hipEventElapsedTime(&g->last_ms, g->ev[0], g->ev[1]));
    return FPHIP_OK;
  }
  // per-wave LDS-DMA ring: FPHIP_GSO_RING slots of IPS KiB (IPS = ceil(NQ/2))
  const size_t lds = la ? (size_t)wpb * fphip_reduce_ring_bytes(nq)
                        : (size_t)wpb * FPHIP_GSO_RING * (size_t)((nq + 1) / 2) * 1024;
  int bpc          = g->blocks_per_cu > 0 ? g->blocks_per_cu : (lds ? (int)((160 * 1024) / lds) : 32);  // (no LDS: the register streams)
  if (bpc * wpb > 32)
    bpc = 32 / wpb;
  const int cap = fphip_ctx_num_cus(g->ctx) * (bpc > 0 ? bpc : 1);
  if (grid > cap)
    grid = cap;
  if (lds > 64 * 1024)
  {
    snprintf(fphip_ctx_errbuf(g->ctx), 512, "gso: LDS ring of %zu bytes exceeds the 64 KiB the DMA base can address", lds);
    return FPHIP_ERROR;
  }
  GCHK(hipEventRecord(g->ev[0], s));
  if (la)
  {
    // (the LLL_EARLY_RED instantiations live in lll_kernel_early.hip)
#define FPHIP_LLL_LAUNCH(NQ_)                                                                                          \
  do                                                                                                                   \
  {                                                                                                                    \
    if (g->P.lll_early || g->P.u)                                                                                      \
      hipLaunchKernelGGL((lll_kernel<NQ_, true>), dim3(grid), dim3(wpb * 64), lds, s, g->P, kmin, la->kstart, kend,    \
                         la->delta, eta, la->logdelta);                                                                \
    else                                                                                                               \
      hipLaunchKernelGGL((lll_kernel<NQ_, false>), dim3(grid), dim3(wpb * 64), lds, s, g->P, kmin, la->kstart, kend,   \
                         la->delta, eta, la->logdelta);                                                                \
  } while (0)
    switch (nq)
    {
    case 1: FPHIP_LLL_LAUNCH(1); break;
    case 2: FPHIP_LLL_LAUNCH(2); break;
    case 3: FPHIP_LLL_LAUNCH(3); break;
    default: FPHIP_LLL_LAUNCH(4); break;
    }
#undef FPHIP_LLL_LAUNCH
  }
  else
  switch (nq)
  {
  case 1:
    hipLaunchKernelGGL(gso_sweep_kernel<1>, dim3(grid), dim3(wpb * 64), lds, s, g->P, kmin, kend, eta, mode);
    break;
  case 2:
    hipLaunchKernelGGL(gso_sweep_kernel<2>, dim3(grid), dim3(wpb * 64), lds, s, g->P, kmin, kend, eta, mode);
    break;
  case 3:
    hipLaunchKernelGGL(gso_sweep_kernel<3>, dim3(grid), dim3(wpb * 64), lds, s, g->P, kmin, kend, eta, mode);
    break;
  default:
    hipLaunchKernelGGL(gso_sweep_kernel<4>, dim3(grid), dim3(wpb * 64), lds, s, g->P, kmin, kend, eta, mode);
    break;
  }
  GCHK(hipGetLastError());
  GCHK(hipEventRecord(g->ev[1], s));
  if (g->P.sess_mode != 0)
    return FPHIP_OK;  // (a session call: its copies are queued behind the kernel, one wait for all of it)
  GCHK(hipStreamSynchronize(s));
  GCHK(hipEventElapsedTime(&g->last_ms, g->ev[0], g->ev[1]));
  return FPHIP_OK;
}

static int fetch_status(fphip_gso *g, int *status)
{
  if (status)
    GCHK(hipMemcpy(status, g->P.status, sizeof(int) * g->P.batch, hipMemcpyDeviceToHost));
  return FPHIP_OK;
}

extern "C" int fphip_gso_set_basis(fphip_gso *g, int first, int count, const int64_t *b)
{
  if (!g || !b || first < 0 || count <= 0 || first + count > g->P.batch)
    return FPHIP_ERROR;
  const size_t rows = (size_t)g->P.d * count;
  // a session ends here for every lattice of the batch: those outside [first, first + count) get their rows back
  // in position order (and so do the rows of u) before the new ones are written
  g->session_active = false;
  if (int rc = restore_position_order(g))
    return rc;
  GCHK(hipMemcpy2D(g->P.b + (size_t)first * g->P.d * g->P.ldn, (size_t)g->P.ldn * 8, b,
                   (size_t)g->P.n * 8, (size_t)g->P.n * 8, rows, hipMemcpyHostToDevice));
  // The kernels run on a NON-BLOCKING stream, which nothing orders against the null stream: the upload has to be
  // complete on the device, not merely handed to the DMA engine from its staging buffer, when this returns.  (The
  // API text allows a pageable upload to return at that point; one of 1536 lattices of a stress test once came back
  // reduced from an all-zero input.)
  GCHK(hipStreamSynchronize(nullptr));
  g->dirty          = true;
  return FPHIP_OK;
}

// replicate lattice `src` into every slot of the batch (device-side copies; used by benchmarks)
extern "C" int fphip_gso_broadcast_basis(fphip_gso *g, int src)
{
  if (!g || src < 0 || src >= g->P.batch)
    return FPHIP_ERROR;
  const size_t per = (size_t)g->P.d * g->P.ldn * sizeof(long long);
  for (int L = 0; L < g->P.batch; ++L)
    if (L != src)
      GCHK(hipMemcpyAsync(g->P.b + (size_t)L * g->P.d * g->P.ldn,
                          g->P.b + (size_t)src * g->P.d * g->P.ldn, per, hipMemcpyDeviceToDevice,
                          fphip_ctx_stream(g->ctx)));
  GCHK(hipStreamSynchronize(fphip_ctx_stream(g->ctx)));
  g->dirty = true;
  return FPHIP_OK;
}

// lattices count, count+1, … become copies of lattices 0 … count-1, cyclically (device-side
// copies: a batch of `count` distinct inputs replicated to the full batch; used by benchmarks)
extern "C" int fphip_gso_tile_basis(fphip_gso *g, int count)
{
  if (!g || count <= 0 || count > g->P.batch)
    return FPHIP_ERROR;
  const size_t per = (size_t)g->P.d * g->P.ldn;
  int have         = count;  // always a multiple of count: the cyclic pattern survives the doubling
  while (have < g->P.batch)
  {
    const int m = have < g->P.batch - have ? have : g->P.batch - have;
    GCHK(hipMemcpyAsync(g->P.b + (size_t)have * per, g->P.b, (size_t)m * per * sizeof(long long),
                        hipMemcpyDeviceToDevice, fphip_ctx_stream(g->ctx)));
    have += m;
  }
  GCHK(hipStreamSynchronize(fphip_ctx_stream(g->ctx)));
  g->dirty = true;
  return FPHIP_OK;
}

extern "C" int fphip_gso_get_basis(fphip_gso *g, int first, int count, int64_t *b)
{
  if (!g || !b || first < 0 || count <= 0 || first + count > g->P.batch)
    return FPHIP_ERROR;
  if (int rcg = session_guard(g, "fphip_gso_get_basis"))  // (fphip_gso_session_read has the basis in position order)
    return rcg;
  const size_t rows = (size_t)g->P.d * count;
  GCHK(hipMemcpy2D(b, (size_t)g->P.n * 8, g->P.b + (size_t)first * g->P.d * g->P.ldn,
                   (size_t)g->P.ldn * 8, (size_t)g->P.n * 8, rows, hipMemcpyDeviceToHost));
  return FPHIP_OK;
}

// MatGSO(b, u, ...) with a non-empty u: enable_transform (gso_interface.h:96-110).  u ([batch][d][d], or NULL for
// the identity) goes to the device; the LLL entry points then apply every row operation to its rows as well and
// rotate them with b's (gso.cpp:84-158, 289-366).
extern "C" int fphip_gso_enable_transform(fphip_gso *g, const int64_t *u)
{
  if (!g)
    return FPHIP_ERROR;
  if (int rcg = session_guard(g, "fphip_gso_enable_transform"))
    return rcg;
  const size_t B = (size_t)g->P.batch, d = g->P.d, ldd = g->P.ldd;
  hipStream_t s = fphip_ctx_stream(g->ctx);
  const size_t bytes = B * d * ldd * sizeof(long long);
  if (!g->P.u)
    GCHK(fphip_dev_alloc((void **)&g->P.u, bytes + 4096, s));
  if (!g->P.u2)
    GCHK(fphip_dev_alloc((void **)&g->P.u2, bytes + 4096, s));
  GCHK(hipMemsetAsync(g->P.u, 0, bytes + 4096, s));
  GCHK(hipMemsetAsync(g->P.u2, 0, bytes + 4096, s));
  GCHK(hipStreamSynchronize(s));
  if (u)
    GCHK(hipMemcpy2D(g->P.u, ldd * 8, u, d * 8, d * 8, B * d, hipMemcpyHostToDevice));
  else
  {
    std::vector<long long> id(d * ldd, 0);
    for (size_t i = 0; i < d; ++i)
      id[i * ldd + i] = 1;
    for (size_t L = 0; L < B; ++L)
      GCHK(hipMemcpy(g->P.u + L * d * ldd, id.data(), d * ldd * sizeof(long long), hipMemcpyHostToDevice));
  }
  GCHK(hipStreamSynchronize(nullptr));  // (see fphip_gso_set_basis)
  g->u_in_slots = false;
  return FPHIP_OK;
}

extern "C" int fphip_gso_get_transform(fphip_gso *g, int first, int count, int64_t *u)
{
  if (!g || !u || first < 0 || count <= 0 || first + count > g->P.batch)
    return FPHIP_ERROR;
  if (!g->P.u)
  {
    snprintf(fphip_ctx_errbuf(g->ctx), 512, "fphip_gso_get_transform: no transformation matrix (fphip_gso_enable_transform)");
    return FPHIP_ERROR;
  }
  if (int rcg = session_guard(g, "fphip_gso_get_transform"))
    return rcg;
  const size_t d = g->P.d, ldd = g->P.ldd;
  GCHK(hipMemcpy2D(u, d * 8, g->P.u + (size_t)first * d * ldd, ldd * 8, d * 8, d * (size_t)count, hipMemcpyDeviceToHost));
  return FPHIP_OK;
}

// (re)float every row from the integer basis: MatGSO::update_bf for all rows, gso.cpp:24-48
extern "C" int fphip_gso_refresh(fphip_gso *g)
{
  FPHIP_RANGE("fphip_gso_refresh");
  if (!g)
    return FPHIP_ERROR;
  return launch(g, 0, g->P.d, 0.0, 2);
}

// MatGSOInterface::update_gso(), gso_interface.h:767-775, for every lattice of the batch
extern "C" int fphip_gso_update(fphip_gso *g, int *status)
{
  FPHIP_RANGE("fphip_gso_update");
  if (!g)
    return FPHIP_ERROR;
  int rc = launch(g, 0, g->P.d, 0.0, 0);
  if (rc != FPHIP_OK)
    return rc;
  return fetch_status(g, status);
}

// LLLReduction::size_reduction(kappa_min, kappa_end), lll.h:107-122, for every lattice
extern "C" int fphip_gso_size_reduce(fphip_gso *g, int kappa_min, int kappa_end, double eta,
                                     int *status)
{
  FPHIP_RANGE("fphip_gso_size_reduce");
  if (!g)
    return FPHIP_ERROR;
  if (kappa_end < 0)
    kappa_end = g->P.d;
  if (kappa_min < 0 || kappa_min > kappa_end || kappa_end > g->P.d)
    return FPHIP_ERROR;
  int rc = launch(g, kappa_min, kappa_end, eta, 1);
  if (rc != FPHIP_OK)
    return rc;
  return fetch_status(g, status);
}

static int ensure_lll_buffers(fphip_gso *g);
extern "C" int fphip_gso_lll_flags(fphip_gso *g, int kappa_min, int kappa_start, int kappa_end,
                                   double delta, double eta, int flags, int *status, int *info);

// LLLReduction<Z_NR<long>, FP_NR<double>>(m, delta, eta, LLL_DEFAULT).lll(kappa_min, kappa_start,
// kappa_end, 0) on a fresh MatGSO(b, GSO_ROW_EXPO) of every lattice (lll.cpp:44-164,
// wrapper.cpp lll_reduction_zf with LM_FAST), then update_gso() so that mu / r are readable in
// place.  info (nullable): 4 ints per lattice — final_kappa, n_swaps, zeros, loop iterations.
extern "C" int fphip_gso_lll(fphip_gso *g, int kappa_min, int kappa_start, int kappa_end,
                             double delta, double eta, int *status, int *info)
{
  return fphip_gso_lll_flags(g, kappa_min, kappa_start, kappa_end, delta, eta, 0, status, info);
}

// fplll's LLLFlags (defs.h:222-227): LLL_VERBOSE (1) is ignored, LLL_EARLY_RED (2) and LLL_SIEGEL (4) run on the
// device (early reduction on the block streams only: FPHIP_UNSUPPORTED in a -DFPHIP_LLL_STREAM=0 build)
static int lll_flags_check(fphip_gso *g, int flags, int *siegel, int *early)
{
  if (flags & ~7)
  {
    snprintf(fphip_ctx_errbuf(g->ctx), 512, "lll: unknown flags 0x%x", flags);
    return FPHIP_ERROR;
  }
#if !FPHIP_LLL_STREAM
  if (flags & 2)
    return FPHIP_UNSUPPORTED;
#endif
  *early  = (flags & 2) ? 1 : 0;
  *siegel = (flags & 4) ? 1 : 0;
  return FPHIP_OK;
}

extern "C" int fphip_gso_lll_flags(fphip_gso *g, int kappa_min, int kappa_start, int kappa_end,
                                   double delta, double eta, int flags, int *status, int *info)
{
  FPHIP_RANGE("fphip_gso_lll");
  if (!g)
    return FPHIP_ERROR;
  int siegel = 0, early = 0;
  if (int rcf = lll_flags_check(g, flags, &siegel, &early))
    return rcf;
  if (kappa_end < 0)
    kappa_end = g->P.d;
  if (kappa_min < 0 || kappa_min > kappa_start || kappa_start >= kappa_end || kappa_end > g->P.d)
  {
    snprintf(fphip_ctx_errbuf(g->ctx), 512, "fphip_gso_lll: need 0 <= kappa_min <= kappa_start < kappa_end <= d");
    return FPHIP_ERROR;
  }
  const size_t B = (size_t)g->P.batch;
  {
    const int rc0 = ensure_lll_buffers(g);
    if (rc0 != FPHIP_OK)
      return rc0;
  }
  int rc = launch(g, 0, g->P.d, 0.0, 2);  // bf / row_expo of every row from b
  if (rc != FPHIP_OK)
    return rc;
  // (swap_threshold = siegel ? delta - eta * eta : delta, lll.cpp:40; the iteration limit keeps log(delta))
  LllArgs la{kappa_start, siegel ? delta - eta * eta : delta, std::log(delta)};
  g->P.lll_siegel = siegel;
  g->P.lll_early  = early;
  rc              = launch(g, kappa_min, kappa_end, eta, 3, &la);
  g->P.lll_siegel = 0;
  g->P.lll_early  = 0;
  if (rc != FPHIP_OK)
    return rc;
  const float lll_ms = g->last_ms;
  std::swap(g->P.b, g->P.b2);  // the kernel wrote the rows in position order into b2
  if (g->P.u)
    std::swap(g->P.u, g->P.u2);  // ... and those of u into u2
  std::vector<int> st(B);
  GCHK(hipMemcpy(st.data(), g->P.status, sizeof(int) * B, hipMemcpyDeviceToHost));
  if (info)
    GCHK(hipMemcpy(info, g->P.lll_info, sizeof(int) * 4 * B, hipMemcpyDeviceToHost));
  // identity-layout GSO of the reduced bases (same values: every entry is a function of b)
  rc = launch(g, 0, g->P.d, 0.0, 2);
  if (rc == FPHIP_OK)
    rc = launch(g, 0, g->P.d, 0.0, 0);
  g->last_ms = lll_ms;
  if (status)
    memcpy(status, st.data(), sizeof(int) * B);
  return rc;
}

// ---- resident LLL session --------------------------------------------------------------------------------
// What MatGSOHip (csrc/dropin/) keeps between the reference's lll() calls: the reference's MatGSO is an OBJECT —
// its rows, Gram cache, mu / r and gso_valid_cols persist from one lll() to the next (gso_interface.h:675-732 the
// accessors, gso_interface.cpp:26-53 the validity tracking), and BKZ's 16 000 lll() calls of a config-2 run each
// touch a few rows.  The stateless fphip_gso_lll rebuilds all of it per call (upload, re-float, every Gram row,
// update_gso, download: 6.2 ms); here the kernel's own state stays on the device: the rows in their slots, slot
// table, symmetric Gram cache, mu / r by slot, valid-column counts, verified prefix.
static int ensure_session_buffers(fphip_gso *g)
{
  const size_t B = (size_t)g->P.batch, d = g->P.d, ldd = g->P.ldd, ldn = g->P.ldn;
  hipStream_t s = fphip_ctx_stream(g->ctx);
  const size_t outb = fphip_session_out_stride(d, ldd, ldn);
  if (!g->P.sess_slots)
    GCHK(fphip_dev_alloc((void **)&g->P.sess_slots, B * 256 * sizeof(int), s));
  if (!g->P.sess_state)
    GCHK(fphip_dev_alloc((void **)&g->P.sess_state, B * 4 * sizeof(int), s));
  if (!g->P.sess_out)
    GCHK(fphip_dev_alloc((void **)&g->P.sess_out, B * outb + 4096, s));
  if (!g->sess_in_d)
    GCHK(fphip_dev_alloc((void **)&g->sess_in_d, (d + d * ldn + d * ldd) * sizeof(long long) + 4096, s));
  if (!g->sess_in_h)
    g->sess_in_h = (long long *)fphip_pinned_get((d + d * ldn + d * ldd) * sizeof(long long));
  if (!g->sess_out_h)
    g->sess_out_h = (char *)fphip_pinned_get(B * outb);
  if (!g->sess_in_h || !g->sess_out_h)
  {
    snprintf(fphip_ctx_errbuf(g->ctx), 512, "fphip_gso_session_lll: no pinned staging memory");
    return FPHIP_ERROR;
  }
  return FPHIP_OK;
}

extern "C" int fphip_gso_session_lll(fphip_gso *g, int resume, int kappa_min, int kappa_start, int kappa_end,
                                     double delta, double eta, int flags, int n_dirty, const int *dirty_pos,
                                     const int64_t *dirty_rows, int *status, int *info)
{
  FPHIP_RANGE("fphip_gso_session_lll");
  if (!g)
    return FPHIP_ERROR;
  int siegel = 0, early = 0;
  if (int rcf = lll_flags_check(g, flags, &siegel, &early))
    return rcf;
  if (kappa_end < 0)
    kappa_end = g->P.d;
  if (kappa_min < 0 || kappa_min > kappa_start || kappa_start >= kappa_end || kappa_end > g->P.d)
  {
    snprintf(fphip_ctx_errbuf(g->ctx), 512, "fphip_gso_session_lll: need 0 <= kappa_min <= kappa_start < kappa_end <= d");
    return FPHIP_ERROR;
  }
  const size_t B = (size_t)g->P.batch, d = g->P.d, ldn = g->P.ldn, n = g->P.n, ldd = g->P.ldd;
  const size_t rec = n + (g->P.u ? d : 0);  // a dirty row: its n integers, then its d integers of u when u is tracked
  if (resume && !g->session_active)
  {
    snprintf(fphip_ctx_errbuf(g->ctx), 512, "fphip_gso_session_lll: no session to resume (start one with resume = 0)");
    return FPHIP_ERROR;
  }
  if (n_dirty < 0 || n_dirty > (int)d || (n_dirty > 0 && (!resume || B != 1 || !dirty_pos || !dirty_rows)))
  {
    snprintf(fphip_ctx_errbuf(g->ctx), 512, "fphip_gso_session_lll: row operations need a running session of a batch of one");
    return FPHIP_ERROR;
  }
  int rc = ensure_lll_buffers(g);
  if (rc == FPHIP_OK)
    rc = ensure_session_buffers(g);
  if (rc != FPHIP_OK)
    return rc;
  hipStream_t s = fphip_ctx_stream(g->ctx);
  if (!resume)
  {
    g->session_active = false;
    rc = restore_position_order(g);  // (a new session on the rows an earlier one left in its slots)
    if (rc != FPHIP_OK)
      return rc;
    rc = launch(g, 0, g->P.d, 0.0, 2);  // bf / row_expo / narrow flags of every row from b
    if (rc != FPHIP_OK)
      return rc;
  }
  else if (n_dirty > 0)
  {
    for (int t = 0; t < n_dirty; ++t)
    {
      if (dirty_pos[t] < 0 || dirty_pos[t] >= (int)d)
        return FPHIP_ERROR;
      g->sess_in_h[t] = dirty_pos[t];
      long long *row = g->sess_in_h + n_dirty + (size_t)t * ldn;
      memcpy(row, dirty_rows + (size_t)t * rec, n * sizeof(long long));
      for (size_t c = n; c < ldn; ++c)
        row[c] = 0;
      if (g->P.u)
      {
        long long *urow = g->sess_in_h + n_dirty + (size_t)n_dirty * ldn + (size_t)t * ldd;
        memcpy(urow, dirty_rows + (size_t)t * rec + n, d * sizeof(long long));
        for (size_t c = d; c < ldd; ++c)
          urow[c] = 0;
      }
    }
    GCHK(hipMemcpyAsync(g->sess_in_d, g->sess_in_h,
                        ((size_t)n_dirty + (size_t)n_dirty * ldn + (g->P.u ? (size_t)n_dirty * ldd : 0)) * sizeof(long long),
                        hipMemcpyHostToDevice, s));
  }
  g->P.sess_mode   = resume ? 2 : 1;
  g->P.sess_ndirty = n_dirty;
  g->P.sess_in     = g->sess_in_d;
  LllArgs la{kappa_start, siegel ? delta - eta * eta : delta, std::log(delta)};
  g->P.lll_siegel  = siegel;
  g->P.lll_early   = early;
  rc               = launch(g, kappa_min, kappa_end, eta, 3, &la);
  g->P.lll_siegel  = 0;
  g->P.lll_early   = 0;
  g->P.sess_mode   = 0;
  g->P.sess_ndirty = 0;
  g->session_active = false;
  if (rc != FPHIP_OK)
    return rc;
  std::vector<int> st(B);
  const size_t outb = fphip_session_out_stride(d, g->P.ldd, ldn);
  GCHK(hipMemcpyAsync(g->sess_out_h, g->P.sess_out, B * outb, hipMemcpyDeviceToHost, s));
  GCHK(hipMemcpyAsync(st.data(), g->P.status, sizeof(int) * B, hipMemcpyDeviceToHost, s));
  if (info)
    GCHK(hipMemcpyAsync(info, g->P.lll_info, sizeof(int) * 4 * B, hipMemcpyDeviceToHost, s));
  GCHK(hipStreamSynchronize(s));
  GCHK(hipEventElapsedTime(&g->last_ms, g->ev[0], g->ev[1]));
  // a failed reduction leaves a state the reference would not continue from either; -2 (a multiplier beyond
  // 63 bits) has changed rows the caller's host copy does not have: either way the next call starts over
  bool all_ok = true;
  for (size_t L = 0; L < B; ++L)
    all_ok &= (st[L] == 1);
  g->session_active = all_ok;
  g->u_in_slots     = g->P.u != nullptr;
  g->b_in_slots     = true;
  if (!all_ok)
    if (int rc2 = restore_position_order(g))
      return rc2;
  if (status)
    memcpy(status, st.data(), sizeof(int) * B);
  return FPHIP_OK;
}

extern "C" int fphip_gso_session_read_transform(fphip_gso *g, int lattice, int64_t *u)
{
  if (!g || !u || lattice < 0 || lattice >= g->P.batch || !g->sess_out_h || !g->P.u)
    return FPHIP_ERROR;
  const size_t d = g->P.d, ldd = g->P.ldd, ldn = g->P.ldn;
  const long long *ou = (const long long *)(g->sess_out_h + (size_t)lattice * fphip_session_out_stride(d, ldd, ldn) +
                                            fphip_session_out_bytes(d, ldd, ldn));
  for (size_t i = 0; i < d; ++i)
    memcpy(u + i * d, ou + i * ldd, d * sizeof(long long));
  return FPHIP_OK;
}

extern "C" int fphip_gso_session_read(fphip_gso *g, int lattice, int64_t *b, double *mu, double *r,
                                      int *valid_cols, int64_t *row_expo)
{
  if (!g || lattice < 0 || lattice >= g->P.batch || !g->sess_out_h)
    return FPHIP_ERROR;
  const size_t d = g->P.d, ldd = g->P.ldd, ldn = g->P.ldn, n = g->P.n;
  const char *out      = g->sess_out_h + (size_t)lattice * fphip_session_out_stride(d, ldd, ldn);
  const long long *ob  = (const long long *)out;
  const double *omu    = (const double *)(out + d * ldn * 8);
  const double *orr    = omu + d * ldd;
  const long long *oex = (const long long *)(orr + d * ldd);
  const int *ovc       = (const int *)(oex + d);
  for (size_t i = 0; i < d; ++i)
  {
    if (b)
      memcpy(b + i * n, ob + i * ldn, n * sizeof(long long));
    if (mu)
      memcpy(mu + i * d, omu + i * ldd, d * sizeof(double));
    if (r)
      memcpy(r + i * d, orr + i * ldd, d * sizeof(double));
    if (row_expo)
      row_expo[i] = oex[i];
    if (valid_cols)
      valid_cols[i] = ovc[i];
  }
  return FPHIP_OK;
}

// LLLReduction::lll in a selectable floating-point type (lll_x.hip): precision 106 = double-double on
// the device (the stand-in for FP_NR<dd_real>: Wrapper::lll's fast_lll<dd_real>, wrapper.cpp:322-330),
// 53 = plain double with the same (per-lane) summation order.  Same statuses / info as fphip_gso_lll.
static int gso_lll_ex(fphip_gso *g, int kappa_min, int kappa_start, int kappa_end, double delta, double eta,
                      int precision, const int *d_only_failed, int *status, int *info)
{
  if (!g || (precision != 53 && precision != 106 && precision != 212))
    return FPHIP_ERROR;
  if (kappa_end < 0)
    kappa_end = g->P.d;
  if (kappa_min < 0 || kappa_min > kappa_start || kappa_start >= kappa_end || kappa_end > g->P.d || g->P.d > 256)
  {
    snprintf(fphip_ctx_errbuf(g->ctx), 512, "fphip_gso_lll_ex: need 0 <= kappa_min <= kappa_start < kappa_end <= d <= 256");
    return FPHIP_ERROR;
  }
  int rc = ensure_lll_buffers(g);
  if (rc != FPHIP_OK)
    return rc;
  const size_t B = (size_t)g->P.batch, d = g->P.d, ldd = g->P.ldd, ldn = g->P.ldn;
  hipStream_t s = fphip_ctx_stream(g->ctx);
  const size_t n_bf = B * d * ldn, n_pl = B * d * ldd;
  // (bf, then the low planes of mu / r / gf, then — quad-double — two more planes of each)
  const int planes = precision == 212 ? 9 : 3;
  if (g->xbuf && g->xbuf_planes < planes)
  {
    fphip_dev_free(g->xbuf, s);
    g->xbuf = nullptr;
  }
  if (!g->xbuf)
  {
    GCHK(fphip_dev_alloc((void **)&g->xbuf, (n_bf + planes * n_pl) * sizeof(double) + 4096, s));
    g->xbuf_planes = planes;
  }
  GCHK(hipMemsetAsync(g->xbuf, 0, (n_bf + planes * n_pl) * sizeof(double), s));
  const bool wide = precision >= 106;
  LllX A;
  A.batch    = g->P.batch;
  A.d        = g->P.d;
  A.n        = g->P.n;
  A.ldn      = g->P.ldn;
  A.ldd      = g->P.ldd;
  A.row_expo = g->P.row_expo;
  A.b        = g->P.b;
  A.b2       = g->P.b2;
  A.bf       = g->xbuf;
  A.mu_hi    = g->P.mu;
  A.mu_lo    = wide ? g->xbuf + n_bf : nullptr;
  A.r_hi     = g->P.r;
  A.r_lo     = wide ? g->xbuf + n_bf + n_pl : nullptr;
  A.gf_hi    = g->P.gf;
  A.gf_lo    = wide ? g->xbuf + n_bf + 2 * n_pl : nullptr;
  A.mu_x     = precision == 212 ? g->xbuf + n_bf + 3 * n_pl : nullptr;
  A.r_x      = precision == 212 ? g->xbuf + n_bf + 5 * n_pl : nullptr;
  A.gf_x     = precision == 212 ? g->xbuf + n_bf + 7 * n_pl : nullptr;
  A.rexp     = g->P.rexp;
  A.status   = g->P.status;
  A.info     = g->P.lll_info;
  A.only_failed = d_only_failed;
  A.kmin     = kappa_min;
  A.kstart   = kappa_start;
  A.kend     = kappa_end;
  A.delta    = delta;
  A.eta      = eta;
  if (d_only_failed)  // the lattices that are skipped keep their rows: b2 := b first
    GCHK(hipMemcpyAsync(g->P.b2, g->P.b, B * d * ldn * sizeof(long long), hipMemcpyDeviceToDevice, s));
  const int need = (g->P.d > g->P.n ? g->P.d : g->P.n);
  const int nq   = (need + 63) / 64;
  int grid       = g->P.batch;
  if (grid > fphip_ctx_num_cus(g->ctx) * 8)
    grid = fphip_ctx_num_cus(g->ctx) * 8;
  GCHK(hipEventRecord(g->ev[0], s));
  if (precision == 212)
    switch (nq)
    {
    case 1: hipLaunchKernelGGL((lll_x_kernel<1, QD>), dim3(grid), dim3(64), 0, s, A); break;
    case 2: hipLaunchKernelGGL((lll_x_kernel<2, QD>), dim3(grid), dim3(64), 0, s, A); break;
    case 3: hipLaunchKernelGGL((lll_x_kernel<3, QD>), dim3(grid), dim3(64), 0, s, A); break;
    default: hipLaunchKernelGGL((lll_x_kernel<4, QD>), dim3(grid), dim3(64), 0, s, A); break;
    }
  else if (precision == 106)
    switch (nq)
    {
    case 1: hipLaunchKernelGGL((lll_x_kernel<1, DD>), dim3(grid), dim3(64), 0, s, A); break;
    case 2: hipLaunchKernelGGL((lll_x_kernel<2, DD>), dim3(grid), dim3(64), 0, s, A); break;
    case 3: hipLaunchKernelGGL((lll_x_kernel<3, DD>), dim3(grid), dim3(64), 0, s, A); break;
    default: hipLaunchKernelGGL((lll_x_kernel<4, DD>), dim3(grid), dim3(64), 0, s, A); break;
    }
  else
    switch (nq)
    {
    case 1: hipLaunchKernelGGL((lll_x_kernel<1, double>), dim3(grid), dim3(64), 0, s, A); break;
    case 2: hipLaunchKernelGGL((lll_x_kernel<2, double>), dim3(grid), dim3(64), 0, s, A); break;
    case 3: hipLaunchKernelGGL((lll_x_kernel<3, double>), dim3(grid), dim3(64), 0, s, A); break;
    default: hipLaunchKernelGGL((lll_x_kernel<4, double>), dim3(grid), dim3(64), 0, s, A); break;
    }
  GCHK(hipGetLastError());
  GCHK(hipEventRecord(g->ev[1], s));
  GCHK(hipStreamSynchronize(s));
  float ms = 0;
  GCHK(hipEventElapsedTime(&ms, g->ev[0], g->ev[1]));
  std::swap(g->P.b, g->P.b2);  // the kernel wrote the rows in position order into b2
  if (status)
    GCHK(hipMemcpy(status, g->P.status, sizeof(int) * B, hipMemcpyDeviceToHost));
  if (info)
    GCHK(hipMemcpy(info, g->P.lll_info, sizeof(int) * 4 * B, hipMemcpyDeviceToHost));
  // identity-layout double GSO of the new bases for the getters (its values are functions of b; on a
  // lattice that needs more than 53 bits they are as good as doubles get)
  rc = launch(g, 0, g->P.d, 0.0, 2);
  if (rc == FPHIP_OK)
    rc = launch(g, 0, g->P.d, 0.0, 0);
  g->last_ms = ms;
  return rc;
}

extern "C" int fphip_gso_lll_ex(fphip_gso *g, int kappa_min, int kappa_start, int kappa_end, double delta,
                                double eta, int precision, int *status, int *info)
{
  FPHIP_RANGE("fphip_gso_lll_ex");
  if (g)
    if (int rct = transform_guard(g, "lll_ex"))
      return rct;
  return gso_lll_ex(g, kappa_min, kappa_start, kappa_end, delta, eta, precision, nullptr, status, info);
}

// The LLL-side precision ladder of the reference's wrapper (Wrapper::lll, wrapper.cpp:281-359:
// fast_lll<double>, then the wider types, each on the basis the failed attempt left) with both
// stages on the device: the exact-order double kernel (lll_kernel.hip) for the whole batch, then
// double-double (lll_x.hip) for the lattices that stopped with RED_GSO_FAILURE (0), RED_BABAI_FAILURE
// (-1) or RED_LLL_FAILURE (-3), then quad-double for those it gives up on (round 6).  stage[batch] (nullable): 53,
// 106 or 212.  A lattice that fails at 212 bits keeps its status: the caller's MPFR stage (fplll's CPU path) is next.
extern "C" int fphip_gso_lll_ladder(fphip_gso *g, int kappa_min, int kappa_start, int kappa_end, double delta,
                                    double eta, int *status, int *info, int *stage)
{
  FPHIP_RANGE("fphip_gso_lll_ladder");
  if (!g)
    return FPHIP_ERROR;
  if (int rct = transform_guard(g, "lll_ladder"))
    return rct;
  const size_t B = (size_t)g->P.batch;
  std::vector<int> st(B, 0), inf(4 * B, 0), stg(B, 53);
  int rc = fphip_gso_lll(g, kappa_min, kappa_start, kappa_end, delta, eta, st.data(), inf.data());
  if (rc != FPHIP_OK)
    return rc;
  float ms = g->last_ms;
  bool any = false;
  std::vector<int> mask(B);
  if (getenv("FPHIP_LLL_LADDER_TEST") && atoi(getenv("FPHIP_LLL_LADDER_TEST")) >= 1)
    for (size_t L = 1; L < B; L += 2)  // (tests only: the odd lattices as if the double stage had failed)
      if (st[L] == 1)
        st[L] = -1;
  for (size_t L = 0; L < B; ++L)
  {
    const bool failed = (st[L] == 0 || st[L] == -1 || st[L] == -3);
    mask[L]           = failed ? 0 : 1;
    any |= failed;
  }
  if (any)
  {
    std::vector<int> st2(B, 0), inf2(4 * B, 0);
    int *d_mask = nullptr;
    GCHK(fphip_dev_alloc((void **)&d_mask, B * sizeof(int), fphip_ctx_stream(g->ctx)));
    hipError_t e = hipMemcpy(d_mask, mask.data(), B * sizeof(int), hipMemcpyHostToDevice);
    // (the wider stage restarts the loop from the top of the range, like a fresh fast_lll<dd_real>)
    rc = (e == hipSuccess) ? gso_lll_ex(g, kappa_min, kappa_min, kappa_end, delta, eta, 106, d_mask, st2.data(), inf2.data())
                           : FPHIP_ERROR;
    fphip_dev_free(d_mask, fphip_ctx_stream(g->ctx));
    if (rc != FPHIP_OK)
      return rc;
    ms += g->last_ms;
    for (size_t L = 0; L < B; ++L)
      if (!mask[L])
      {
        st[L]  = st2[L];
        stg[L] = 106;
        for (int t = 0; t < 4; ++t)
          inf[4 * L + t] = (t == 0 || t == 2) ? inf2[4 * L + t] : inf[4 * L + t] + inf2[4 * L + t];
      }
    // third stage (Wrapper::lll goes on to fast_lll<qd_real>, wrapper.cpp:331-339): quad-double for the lattices
    // double-double gave up on (FPHIP_LLL_LADDER_TEST=2, tests only: every fourth lattice as if it had)
    if (getenv("FPHIP_LLL_LADDER_TEST") && atoi(getenv("FPHIP_LLL_LADDER_TEST")) == 2)
      for (size_t L = 3; L < B; L += 4)
        if (!mask[L] && st[L] == 1)
          st[L] = -1;
    std::vector<int> mask3(B);
    bool any3 = false;
    for (size_t L = 0; L < B; ++L)
    {
      const bool failed = !mask[L] && (st[L] == 0 || st[L] == -1 || st[L] == -3);
      mask3[L]          = failed ? 0 : 1;
      any3 |= failed;
    }
    if (any3)
    {
      std::vector<int> st3(B, 0), inf3(4 * B, 0);
      int *d_mask3 = nullptr;
      GCHK(fphip_dev_alloc((void **)&d_mask3, B * sizeof(int), fphip_ctx_stream(g->ctx)));
      hipError_t e3 = hipMemcpy(d_mask3, mask3.data(), B * sizeof(int), hipMemcpyHostToDevice);
      rc = (e3 == hipSuccess) ? gso_lll_ex(g, kappa_min, kappa_min, kappa_end, delta, eta, 212, d_mask3, st3.data(), inf3.data())
                              : FPHIP_ERROR;
      fphip_dev_free(d_mask3, fphip_ctx_stream(g->ctx));
      if (rc != FPHIP_OK)
        return rc;
      ms += g->last_ms;
      for (size_t L = 0; L < B; ++L)
        if (!mask3[L])
        {
          st[L]  = st3[L];
          stg[L] = 212;
          for (int t = 0; t < 4; ++t)
            inf[4 * L + t] = (t == 0 || t == 2) ? inf3[4 * L + t] : inf[4 * L + t] + inf3[4 * L + t];
        }
    }
  }
  g->last_ms = ms;
  if (status)
    memcpy(status, st.data(), B * sizeof(int));
  if (info)
    memcpy(info, inf.data(), 4 * B * sizeof(int));
  if (stage)
    memcpy(stage, stg.data(), B * sizeof(int));
  return FPHIP_OK;
}

static int ensure_lll_buffers(fphip_gso *g)
{
  const size_t B = (size_t)g->P.batch, d = g->P.d, ldd = g->P.ldd, ldn = g->P.ldn;
  // (each buffer on its own: a failed allocation must not leave a half-initialised set behind)
  if (!g->P.gf)
    GCHK(fphip_dev_alloc((void **)&g->P.gf, B * d * ldd * sizeof(double) + 4096, fphip_ctx_stream(g->ctx)));
  if (!g->P.vc)
    GCHK(fphip_dev_alloc((void **)&g->P.vc, B * d * sizeof(int), fphip_ctx_stream(g->ctx)));
  if (!g->P.lll_info)
    GCHK(fphip_dev_alloc((void **)&g->P.lll_info, B * 4 * sizeof(int), fphip_ctx_stream(g->ctx)));
  if (!g->P.b2)
  {
    GCHK(fphip_dev_alloc((void **)&g->P.b2, B * d * ldn * sizeof(long long) + 4096, fphip_ctx_stream(g->ctx)));
    // (stream-ordered: no device-wide synchronisation — other contexts may have kernels running)
    GCHK(hipMemsetAsync(g->P.b2, 0, B * d * ldn * sizeof(long long) + 4096, fphip_ctx_stream(g->ctx)));
  }
  return FPHIP_OK;
}

// BKZReduction<Z_NR<long>, FP_NR<double>>(m, lll_obj, BKZParam(block_size, {}, delta, flags,
// max_loops)).bkz() (bkz.cpp:522-668) on every lattice: primal BKZ with empty strategies (no
// pruning, no preprocessing — what bkz_reduction(b, beta, BKZ_DEFAULT, FT_DOUBLE) runs without a
// strategies file, BASELINE config 2).  flags: FPHIP_BKZ_MAX_LOOPS (0x4) with max_loops and / or
// FPHIP_BKZ_AUTO_ABORT (0x20), fplll's values.  Without AUTO_ABORT the whole reduction is one
// launch; with it the tours are launched one by one and BKZAutoAbort::test_abort (scale 1.0, 5
// tours, bkz.cpp:800-809) runs on the host between them — the slope needs the host's log(), the
// reference's — on the r_ii of every lattice that is still active.  The input is expected
// LLL-reduced, as bkz_reduction guarantees (bkz.cpp:870-885): call fphip_gso_lll first.
// status[batch]: 1 RED_SUCCESS, 8 RED_BKZ_LOOPS_LIMIT, <= 0 the failing LLL status.
// info (nullable) [batch][4]: tours, enumeration nodes (low, high 32 bits; fplll rule), enumeration calls.
static int bkz_launch(fphip_gso *g, int block_size, double delta, double eta, int use_loops,
                      int max_loops, float *ms, int *st_out, int *info_out)
{
  if (int rcg = session_guard(g, "bkz"))
    return rcg;
  if (int rct = transform_guard(g, "bkz"))
    return rct;
  const int need = (g->P.d > g->P.n ? g->P.d : g->P.n);
  const int nq   = (need + 63) / 64;
  const int wpb  = g->waves_per_block;
  const int bs   = block_size < 2 ? 2 : (block_size < g->P.d ? block_size : g->P.d);
  const int stack_doubles = (bs * (bs + 1)) / 2 + 2;
  const size_t ring_bytes = (size_t)wpb * fphip_reduce_ring_bytes(nq);
  const size_t lds        = ring_bytes + (size_t)wpb * stack_doubles * sizeof(double);
  if (ring_bytes > 64 * 1024 || lds > 160 * 1024)
  {
    snprintf(fphip_ctx_errbuf(g->ctx), 512, "bkz: %zu bytes of LDS per workgroup do not fit", lds);
    return FPHIP_ERROR;
  }
  int bpc = lds ? (int)((160 * 1024) / lds) : 32;
  if (bpc * wpb > 32)
    bpc = 32 / wpb;
  int grid      = (g->P.batch + wpb - 1) / wpb;
  const int cap = fphip_ctx_num_cus(g->ctx) * (bpc > 0 ? bpc : 1);
  if (grid > cap)
    grid = cap;
  hipStream_t s     = fphip_ctx_stream(g->ctx);
  const double logd = std::log(delta);
  if (lds > 64 * 1024)
  {
    switch (nq)
    {
    case 1: GCHK(hipFuncSetAttribute((const void *)bkz_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); break;
    case 2: GCHK(hipFuncSetAttribute((const void *)bkz_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); break;
    case 3: GCHK(hipFuncSetAttribute((const void *)bkz_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); break;
    default: GCHK(hipFuncSetAttribute((const void *)bkz_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); break;
    }
  }
  GCHK(hipEventRecord(g->ev[0], s));
  switch (nq)
  {
  case 1: hipLaunchKernelGGL(bkz_kernel<1>, dim3(grid), dim3(wpb * 64), lds, s, g->P, block_size, delta, eta, logd, use_loops, max_loops, stack_doubles); break;
  case 2: hipLaunchKernelGGL(bkz_kernel<2>, dim3(grid), dim3(wpb * 64), lds, s, g->P, block_size, delta, eta, logd, use_loops, max_loops, stack_doubles); break;
  case 3: hipLaunchKernelGGL(bkz_kernel<3>, dim3(grid), dim3(wpb * 64), lds, s, g->P, block_size, delta, eta, logd, use_loops, max_loops, stack_doubles); break;
  default: hipLaunchKernelGGL(bkz_kernel<4>, dim3(grid), dim3(wpb * 64), lds, s, g->P, block_size, delta, eta, logd, use_loops, max_loops, stack_doubles); break;
  }
  GCHK(hipGetLastError());
  GCHK(hipEventRecord(g->ev[1], s));
  GCHK(hipStreamSynchronize(s));
  GCHK(hipEventElapsedTime(ms, g->ev[0], g->ev[1]));
  // (the sweep launches below reuse P.status)
  GCHK(hipMemcpy(st_out, g->P.status, sizeof(int) * g->P.batch, hipMemcpyDeviceToHost));
  GCHK(hipMemcpy(info_out, g->P.lll_info, sizeof(int) * 4 * g->P.batch, hipMemcpyDeviceToHost));
  std::swap(g->P.b, g->P.b2);  // the kernel wrote the rows in position order into b2
  // identity-layout GSO of the current bases (same values: every entry is a function of b)
  int rc = launch(g, 0, g->P.d, 0.0, 2);
  if (rc == FPHIP_OK)
    rc = launch(g, 0, g->P.d, 0.0, 0);
  return rc;
}

// info accumulation across one-tour launches: tours, 64-bit node count (two 32-bit halves), calls
static void accumulate_info(int *inf, const int *one, size_t L)
{
  inf[4 * L + 0] += one[4 * L + 0];
  const unsigned long long a = ((unsigned long long)(unsigned)inf[4 * L + 2] << 32) | (unsigned)inf[4 * L + 1];
  const unsigned long long b = ((unsigned long long)(unsigned)one[4 * L + 2] << 32) | (unsigned)one[4 * L + 1];
  const unsigned long long t = a + b;
  inf[4 * L + 1] = (int)(unsigned)(t & 0xffffffffull);
  inf[4 * L + 2] = (int)(unsigned)(t >> 32);
  inf[4 * L + 3] += one[4 * L + 3];
}

// BKZ_AUTO_ABORT for both device BKZ drivers: one tour per launch, BKZAutoAbort::test_abort(1.0, 5)
// (bkz.cpp:800-809) on the host in between — the slope of log r_ii (MatGSOInterface::
// get_current_slope, gso_interface.cpp:198-218) needs the host's log(), the one the reference calls.
// run_tour(loop, &ms, s1, one) launches exactly one tour for the lattices with active[L] != 0 and
// leaves the identity-layout GSO (r_ii, row exponents) of the new bases on the device.
// slope_test: BKZ_AUTO_ABORT (the slope test between the tours).  slide: BKZ_SLD_RED — a launch is one
// slide_tour, and the tour's own progress test runs here: get_slide_potential (gso_interface.cpp:230-258:
// sum over the blocks of (p - i) log det, the host's log) of the new basis against the previous one,
// bkz.cpp:512-518 — "clean" (no progress) ends the loop with RED_SUCCESS like any clean tour.
// BKZ_MAX_TIME (bkz.cpp:563,588-592) and BKZ_DUMP_GSO (:373-377, 456-460, 508-512, 536-539, 667-670; dump_gso
// :729-790) of the tour loop below.  Time is the wall clock of the call (the reference reads the process's CPU time,
// cputime(): the same thing for its single thread).  The dump is the reference's hand-written JSON, one file per
// lattice (lattice 0: the name itself, lattice L > 0: name.L): "Input", one entry per tour, "Output".
struct TourHooks
{
  bool use_time = false, dump = false;
  double max_time = 0.0;
  const char *step = "End of BKZ loop";
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double seconds() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

static std::string dump_name(const fphip_gso *g, size_t L)
{
  return L == 0 ? g->bkz_dump_path : g->bkz_dump_path + "." + std::to_string(L);
}

// BKZReduction::dump_gso (bkz.cpp:729-790): norms = log(r_ii) + expo log 2 of the rows below num_rows, 8 digits
static int dump_gso_entry(const fphip_gso *g, size_t L, bool append, const char *step, int loop, double time,
                          const double *rdg, const long long *rex, int num_rows)
{
  std::ofstream dump;
  if (append)
    dump.open(dump_name(g, L).c_str(), std::ios_base::app);
  else
  {
    dump.open(dump_name(g, L).c_str());
    dump << "[" << std::endl;
  }
  if (!dump)
    return FPHIP_ERROR;
  dump << std::string(8, ' ') << "{" << std::endl;
  dump << std::string(16, ' ') << "\"step\": \"" << step << "\"," << std::endl;
  dump << std::string(16, ' ') << "\"loop\": " << loop << "," << std::endl;
  dump << std::string(16, ' ') << "\"time\": " << time << "," << std::endl;
  std::stringstream ss;
  for (int i = 0; i < num_rows; ++i)
  {
    const double log_f = std::log(rdg[i]);
    const long expo    = (long)(2 * rex[i]);  // get_r_exp(i, i, expo), gso_interface.h:711-716
    ss << std::setprecision(8) << log_f + expo * std::log(2.0) << ", ";
  }
  const std::string str = ss.str();
  dump << std::string(16, ' ') << "\"norms\": [" << str.substr(0, str.size() >= 2 ? str.size() - 2 : 0) << "]"
       << std::endl;
  dump << std::string(8, ' ') << "}";
  if (std::string(step) == "Output")
    dump << std::endl << "]";
  else
    dump << "," << std::endl;
  return dump ? FPHIP_OK : FPHIP_ERROR;
}

// the "Output" entry (bkz.cpp:667-670): the GSO norms of the bases as the call leaves them
static int dump_gso_output(fphip_gso *g, const TourHooks &hooks, const std::vector<int> &st, const std::vector<int> &rows)
{
  const size_t B = (size_t)g->P.batch, d = (size_t)g->P.d;
  int rc = launch(g, 0, g->P.d, 0.0, 0);
  if (rc != FPHIP_OK)
    return rc;
  std::vector<double> rdg(B * d);
  std::vector<long long> rex(B * d);
  GCHK(hipMemcpy(rdg.data(), g->P.rdg, sizeof(double) * B * d, hipMemcpyDeviceToHost));
  GCHK(hipMemcpy(rex.data(), g->P.rexp, sizeof(long long) * B * d, hipMemcpyDeviceToHost));
  for (size_t L = 0; L < B; ++L)
    if (st[L] == 1 || st[L] == 7 || st[L] == 8)  // (a failing tour throws past the dump, bkz.cpp:611-614)
      if (dump_gso_entry(g, L, true, "Output", -1, hooks.seconds(), &rdg[L * d], &rex[L * d], rows[L]) != FPHIP_OK)
        return gfail_msg(g->ctx, "BKZ_DUMP_GSO: cannot write the dump file");
  return FPHIP_OK;
}

template <class RunTour>
static int auto_abort_loop(fphip_gso *g, int block_size, bool use_loops, int max_loops, std::vector<int> &active,
                           std::vector<int> &st, std::vector<int> &inf, std::vector<int> &rows, float &total_ms,
                           RunTour run_tour, bool slope_test = true, bool slide = false,
                           const TourHooks &hooks = TourHooks())
{
  const size_t B = (size_t)g->P.batch, d = (size_t)g->P.d;
  std::vector<char> ran(B, 0);  // the lattice took part in the tour before this iteration
  int rc = launch(g, 0, g->P.d, 0.0, 0);  // r_ii of the input bases
  if (rc != FPHIP_OK)
    return rc;
  std::vector<double> rdg(B * d), old_slope(B, std::numeric_limits<double>::max());
  std::vector<long long> rex(B * d);
  std::vector<int> no_dec(B, -1), one(4 * B), s1(B);
  std::vector<double> sld_potential(B, 0.0);
  bool rows_known = false;
  for (int loop = 0;; ++loop)
  {
    size_t n_active = 0;
    GCHK(hipMemcpy(rdg.data(), g->P.rdg, sizeof(double) * B * d, hipMemcpyDeviceToHost));
    GCHK(hipMemcpy(rex.data(), g->P.rexp, sizeof(long long) * B * d, hipMemcpyDeviceToHost));
    if (!rows_known)
    {  // trailing zero rows (bkz.cpp:35-37) have r_ii == 0 exactly
      for (size_t L = 0; L < B; ++L)
      {
        int nr = (int)d;
        while (nr > 0 && rdg[L * d + nr - 1] == 0.0)
          --nr;
        rows[L] = nr;
      }
      rows_known = true;
    }
    for (size_t L = 0; L < B; ++L)
    {
      if (hooks.dump && (loop == 0 ? active[L] != 0 : ran[L] != 0))
      {  // "Input" (bkz.cpp:536-539) / the entry at the end of tour loop - 1 (:373-377)
        if (dump_gso_entry(g, L, loop != 0, loop == 0 ? "Input" : hooks.step, loop == 0 ? -1 : loop - 1,
                           loop == 0 ? 0.0 : hooks.seconds(), &rdg[L * d], &rex[L * d], rows[L]) != FPHIP_OK)
          return gfail_msg(g->ctx, "BKZ_DUMP_GSO: cannot write the dump file");
      }
      ran[L] = 0;
      if (!active[L])
        continue;
      if (block_size < 2)
      {
        active[L] = 0;
        continue;
      }
      if (slide)
      {
        const int n = rows[L];
        double potential = 0.0;
        int p            = n / block_size;
        if (n % block_size == 0)
          --p;
        for (int i = 0; i < p; ++i)
        {
          double log_det = 0.0;  // get_log_det(i bs, (i + 1) bs)
          for (int k = i * block_size; k < std::min(n, (i + 1) * block_size); ++k)
            log_det += std::log(std::ldexp(rdg[L * d + k], (int)(2 * rex[L * d + k])));
          potential += (p - i) * log_det;
        }
        if (loop == 0)
          sld_potential[L] = potential;  // bkz.cpp:567-571
        else if (potential >= sld_potential[L] || block_size >= n)
        {  // slide_tour returned clean (or one tour was all there is): the loop ends, RED_SUCCESS
          st[L]     = 1;
          active[L] = 0;
          continue;
        }
        else
          sld_potential[L] = potential;
      }
      if (use_loops && loop >= max_loops)
      {
        st[L]     = 8;
        active[L] = 0;
        continue;
      }
      if (hooks.use_time && hooks.seconds() >= hooks.max_time)
      {  // RED_BKZ_TIME_LIMIT, bkz.cpp:588-592
        st[L]     = 7;
        active[L] = 0;
        continue;
      }
      const int n = rows[L];
      if (slope_test)
      {
      double v1 = 0, v2 = (double)(n + 1) * n * (n - 1) / 12.0, weight = (1.0 - n) / 2.0;
      for (int i = 0; i < n; ++i)
      {
        const double logf = std::log(rdg[L * d + i]);
        const long expo   = (long)(2 * rex[L * d + i]);
        v1 += weight * (logf + expo * std::log(2.0));
        weight++;
      }
      const double new_slope = -(v1 / v2);
      if (no_dec[L] == -1 || new_slope < 1.0 * old_slope[L])
        no_dec[L] = 0;
      else
        no_dec[L]++;
      old_slope[L] = std::min(old_slope[L], new_slope);
      if (no_dec[L] >= 5)
      {
        active[L] = 0;  // abort: status stays RED_SUCCESS
        continue;
      }
      }
      ++n_active;
    }
    if (n_active == 0)
      break;
    GCHK(hipMemcpy(g->P.bkz_active, active.data(), sizeof(int) * B, hipMemcpyHostToDevice));
    for (size_t L = 0; L < B; ++L)
      ran[L] = active[L] ? 1 : 0;
    float ms = 0;
    rc       = run_tour(loop, &ms, s1.data(), one.data());
    if (rc != FPHIP_OK)
      return rc;
    total_ms += ms;
    for (size_t L = 0; L < B; ++L)
    {
      if (!active[L])
        continue;
      accumulate_info(inf.data(), one.data(), L);
      if (s1[L] == 8)
        continue;        // tour done, not clean: next loop
      st[L]     = s1[L];  // 1: clean (or block_size >= num_rows); <= 0: failure
      active[L] = 0;
    }
  }
  return FPHIP_OK;
}

// BKZParam::max_time (seconds) and BKZParam::dump_gso_filename (bkz_param.h:151,167) of the device BKZ drivers
extern "C" int fphip_gso_bkz_limits(fphip_gso *g, double max_time, const char *dump_gso_filename)
{
  if (!g)
    return FPHIP_ERROR;
  g->bkz_max_time = max_time;
  if (dump_gso_filename)
    g->bkz_dump_path = dump_gso_filename;
  return FPHIP_OK;
}

extern "C" int fphip_gso_bkz(fphip_gso *g, int block_size, double delta, double eta, int flags,
                             int max_loops, int *status, int *info)
{
  FPHIP_RANGE("fphip_gso_bkz");
  if (!g)
    return FPHIP_ERROR;
  if (block_size > 64 || (flags & ~(0x4 | 0x8 | 0x20 | 0x40)))
    return FPHIP_UNSUPPORTED;  // blocks beyond one wavefront / other BKZ variants: fplll's CPU code
  int rc = ensure_lll_buffers(g);
  if (rc != FPHIP_OK)
    return rc;
  const size_t B = (size_t)g->P.batch, d = g->P.d;
  if (!g->P.enum_mu)
    GCHK(fphip_dev_alloc((void **)&g->P.enum_mu, B * (64 * 63 / 2) * sizeof(double), fphip_ctx_stream(g->ctx)));
  if (!g->P.bkz_active)
    GCHK(fphip_dev_alloc((void **)&g->P.bkz_active, B * sizeof(int), fphip_ctx_stream(g->ctx)));
  if (!g->P.bkz_rows)
    GCHK(fphip_dev_alloc((void **)&g->P.bkz_rows, B * sizeof(int), fphip_ctx_stream(g->ctx)));
  std::vector<int> active(B, 1), st(B, 1), inf(4 * B, 0), one(4 * B), rows(B, (int)d);
  GCHK(hipMemcpy(g->P.bkz_active, active.data(), sizeof(int) * B, hipMemcpyHostToDevice));
  rc = launch(g, 0, g->P.d, 0.0, 2);  // bf / row_expo of every row from b
  if (rc != FPHIP_OK)
    return rc;
  const bool use_loops = (flags & 0x4) != 0, auto_abort = (flags & 0x20) != 0;
  // BKZ_MAX_TIME (0x8) / BKZ_DUMP_GSO (0x40), fplll's values: one tour per launch, the clock and the dump in between
  TourHooks hooks;
  hooks.use_time = (flags & 0x8) != 0;
  hooks.dump     = (flags & 0x40) != 0;
  hooks.max_time = g->bkz_max_time;
  float total_ms = 0, ms = 0;
  if (!auto_abort && !hooks.use_time && !hooks.dump)
  {
    rc = bkz_launch(g, block_size, delta, eta, use_loops ? 1 : 0, max_loops, &ms, st.data(), inf.data());
    if (rc != FPHIP_OK)
      return rc;
    total_ms = ms;
  }
  else
  {
    rc = auto_abort_loop(g, block_size, use_loops, max_loops, active, st, inf, rows, total_ms,
                         [&](int, float *tms, int *s1, int *one)
                         { return bkz_launch(g, block_size, delta, eta, 1, 1, tms, s1, one); },
                         auto_abort, false, hooks);
    if (rc == FPHIP_OK && hooks.dump)
      rc = dump_gso_output(g, hooks, st, rows);
    if (rc != FPHIP_OK)
      return rc;
  }
  g->last_ms = total_ms;
  if (status)
    memcpy(status, st.data(), sizeof(int) * B);
  if (info)
    memcpy(info, inf.data(), sizeof(int) * 4 * B);
  return FPHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// BKZ with strategies: BKZReduction::bkz() with a BKZParam(block_size, strategies, delta, flags,
// max_loops, ..., gh_factor) — recursive preprocessing tours, pruning sets chosen by the radius /
// Gaussian-heuristic ratio, the GH radius bound, the success-probability loop with
// rerandomize_block (bkz.cpp:43-124, 274-441, 522-668).  The wave of each lattice runs the whole
// reduction (bkzs_kernel.hip); this thread serves its mailbox requests while the kernel is in
// flight: the radius / pruning decision (host libm: log, exp, lgamma, pow — the reference's own
// roundings) and the rerandomisation plan (drawn from the caller's generator, rnd(user, lattice,
// n) = gmp_urandomm_ui(state of that lattice, n)).
#include <atomic>
#include <functional>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
static inline void *pinned_get(size_t bytes) { return fphip_pinned_get(bytes); }
static inline void pinned_put(void *p) { fphip_pinned_put(p); }

// ---------------------------------------------------------------------------------------------
namespace
{
// adjust_radius_to_gh_bound, gso_interface.cpp:260-276
void adjust_radius_to_gh_bound(double &max_dist, long max_dist_expo, int block_size, double root_det,
                               double gh_factor)
{
  double t = (double)block_size / 2.0 + 1;
  t        = lgamma(t);
  t        = pow(M_E, t * 2.0 / (double)block_size);
  t        = t / M_PI;
  double f = t;
  f        = f * root_det;
  f        = std::ldexp(f, (int)-max_dist_expo);
  f        = f * gh_factor;
  if (f < max_dist)
    max_dist = f;
}

struct BkzsHost
{
  const fphip_strategies *S;
  double gh_factor;
  fphip_rand_fn rnd;
  void *rnd_user;
  double handoff_nodes;  // > 0: blocks above that many estimated nodes go to the multi-wave enumerator
  // in-loop pruning: blocks of the top-level tour of at least il_min_block rows get coefficients from
  // prune() on their own r-profile instead of a set of the strategies table
  int inloop           = 0;
  double il_preproc    = 0, il_target = 0;
  int il_min_block     = 0, il_flags = 0;
  std::atomic<int> *il_errors = nullptr;  // prune() calls that failed for another reason than the profile
};
// the blocks in-loop pruning applies to: primal blocks of the top-level tour (not of a preprocessing
// tour: 0x10000, not dual: 0x20000)
static inline bool inloop_block(const BkzsHost &H, const BkzMail *m)
{
  return H.inloop && m->type == 1 && !(m->flags & 0x30000) && m->bs >= H.il_min_block && m->bs >= 4;
}

// Gaussian-heuristic size of the pruned tree of a block (what enum_host.hip's estimate_levels
// computes, in logarithms: the r_ii carry their row exponents here): sum over the levels k of
// 1/2 V_{bs-k}(R_k) / prod_{i>=k} sqrt(r_ii), R_k^2 = pruning_k * radius.  Scheduling only.
double estimate_block_nodes(int bs, const double *logr, double log_radius, const double *prune)
{
  double sumlog = 0.0, tot = 0.0;
  for (int k = bs - 1; k >= 0; --k)
  {
    sumlog += 0.5 * logr[k];
    const int n       = bs - k;
    const double logV = 0.5 * n * std::log(M_PI) - std::lgamma(0.5 * n + 1.0);
    const double pr   = prune ? prune[k] : 1.0;
    tot += std::exp(std::min(std::log(0.5) + logV + 0.5 * n * (std::log(pr > 0 ? pr : 1e-300) + log_radius) - sumlog, 80.0));
  }
  return tot;
}

// type 1: radius (bkz.cpp:309-323) and pruning set (get_pruning :82-98, Strategy::get_pruning
// bkz_param.cpp:64-80) of the block whose r_ii the wave has written
void serve_radius(const BkzsHost &H, BkzMail *m, fphip_pruner::VolumeEngine *engine = nullptr)
{
  const int bs      = m->bs;
  const bool dual   = (m->flags & 0x20000) != 0;  // a dual block of self-dual BKZ (bkzs_kernel.hip, DUALS)
  long expo         = m->e2[0];
  const double r0   = m->r[0];
  double max_dist   = r0;
  if (dual)
  {  // radius from the LAST row: max_dist.pow_si(max_dist, -1); max_dist_expo *= -1, bkz.cpp:311-316
    max_dist = ::pow(m->r[bs - 1], static_cast<double>(-1));
    expo     = -(long)m->e2[bs - 1];
  }
  max_dist = max_dist * m->delta;  // max_dist *= delta
  const long expo0 = m->e2[0];     // get_pruning always looks at r(kappa, kappa), bkz.cpp:89-97
  // MatGSOInterface::get_root_det / get_log_det, gso_interface.cpp:220-242
  double log_det = 0.0;
  for (int i = 0; i < bs; ++i)
  {
    const double h = std::ldexp(m->r[i], m->e2[i]);  // get_r(h, i, i)
    log_det += std::log(h);
  }
  double root_det = log_det / (double)bs;
  root_det        = std::exp(root_det);
  if ((m->flags & 0x80) && bs > 30)
    adjust_radius_to_gh_bound(max_dist, expo, bs, root_det,
                              (m->flags & 0x10000) ? 1.1 /* BKZ_DEF_GH_FACTOR of a preprocessing BKZParam */ : H.gh_factor);
  int best           = -1;
  double expectation = 1.0;
  if (H.S)
  {
    double gh_max_dist = r0;
    adjust_radius_to_gh_bound(gh_max_dist, expo0, bs, root_det, 1.0);
    const double radius    = r0 * pow(2, expo0);
    const double gh        = gh_max_dist * pow(2, expo0);
    const double gh_factor = radius / gh;
    double closest         = pow(2, 80);
    best                   = H.S->prune_off[bs];
    for (int p = H.S->prune_off[bs]; p < H.S->prune_off[bs + 1]; ++p)
      if (fabs(H.S->prune_gh[p] - gh_factor) < closest)
      {
        closest = fabs(H.S->prune_gh[p] - gh_factor);
        best    = p;
      }
    expectation = H.S->prune_exp[best];
  }
  const double *inloop_pr = nullptr;
  if (inloop_block(H, m))
  {
    // Pruning chosen where the reference chooses it (bkz.cpp:325, after the preprocessing and the radius)
    // — but computed for THIS block: prune<FP_NR<double>>(radius, preproc_cost, r_ii of the block, target,
    // PROBABILITY_OF_SHORTEST, flags) (pruner.h:187-193) on the profile the wave has just sent; the
    // strategies' set stays in force if the pruner fails (a degenerate profile)
    double rr[64], co[64], ex = 1.0;
    for (int i = 0; i < bs; ++i)
      rr[i] = std::ldexp(m->r[i], m->e2[i]);
    const double radius = max_dist * pow(2, expo);
    int prc = FPHIP_UNSUPPORTED;
    if (std::isfinite(radius) && radius > 0)
    {
      prc = fphip_pruner::prune_block(engine, bs, rr, radius, H.il_preproc, H.il_target, H.il_flags, co, &ex);
      // a device engine that failed (its message is in error()) is not a degenerate profile: the call reports it
      if (prc != FPHIP_OK && engine && engine->error()[0] && H.il_errors)
        H.il_errors->fetch_add(1);
    }
    if (prc == FPHIP_OK)
    {
      for (int i = 0; i < bs; ++i)
        m->prn[i] = co[i];
      best        = -2;  // "the coefficients are in the mailbox"
      expectation = ex;
      inloop_pr   = m->prn;
    }
  }
  m->max_dist    = max_dist;
  m->expectation = expectation;
  m->prune       = best;
  m->handoff     = 0;
  if (H.handoff_nodes > 0 && bs >= 24)
  {
    // the tree the wave is about to walk: primal block r_ii 2^e2_i, radius max_dist 2^expo; a dual
    // block is the index-reversed inverse (enumerate.cpp:107-123)
    double logr[64];
    for (int i = 0; i < bs; ++i)
    {
      const double lr = std::log(m->r[i]) + m->e2[i] * M_LN2;
      if (dual)
        logr[bs - 1 - i] = -lr;
      else
        logr[i] = lr;
    }
    const double *pr = inloop_pr;
    if (H.S && best >= 0 && H.S->coeff_off[best + 1] - H.S->coeff_off[best] == bs)
      pr = H.S->coeff + H.S->coeff_off[best];
    // expected number of nodes of this enumeration: the pruner's cost function (pruner_host.hip:
    // Pruner::single_enum_cost, volumes of the cylinder intersections — what the reference's pruner
    // optimises); the plain Gaussian-heuristic sum (2-5x high on pruned trees) only if it fails
    double est = -1.0;
    {
      double rr[64], ones[64], cost = 0.0;
      for (int i = 0; i < bs; ++i)
      {
        rr[i]   = std::exp(logr[i]);
        ones[i] = 1.0;
      }
      const double radius = std::exp(std::log(max_dist) + expo * M_LN2);
      if (std::isfinite(radius) && fphip_pruner_enum_cost(bs, rr, radius, pr ? pr : ones, 0, &cost, nullptr, nullptr) == FPHIP_OK &&
          std::isfinite(cost))
        est = cost;
    }
    if (est < 0.0)
      est = 0.25 * estimate_block_nodes(bs, logr, std::log(max_dist) + expo * M_LN2, pr);
    m->handoff = est >= H.handoff_nodes ? 1 : 0;
  }
}

// type 3: the enumeration of a large block on the multi-wave enumerator (enum_host.hip) with
// FastEvaluator(1) semantics (evaluator.h:122-156, max_sols = 1: every delivered solution replaces
// the last one and becomes the radius).  Runs on the hand-off worker thread.
struct HandoffSol
{
  double sol[64];
  int have;
  int dim;
};
double handoff_cb(void *user, double dist, const double *sol)
{
  HandoffSol *h = static_cast<HandoffSol *>(user);
  for (int i = 0; i < h->dim; ++i)
    h->sol[i] = sol[i];
  h->have = 1;
  return dist;
}
int serve_enumeration(fphip_ctx *ectx, BkzMail *m, const double *mu_tri)
{
  const int bs = m->bs;
  std::vector<double> mut((size_t)bs * bs, 0.0);
  for (int k = 1; k < bs; ++k)
    for (int l = 0; l < k; ++l)
      mut[(size_t)l * bs + k] = mu_tri[(k * (k - 1)) / 2 + l];  // mut[i*dim + j] = mu(j, i), j > i
  HandoffSol hs;
  hs.have = 0;
  hs.dim  = bs;
  fphip_enum_opts o;
  memset(&o, 0, sizeof o);
  o.dual = m->dual3;
  std::vector<uint64_t> nodes(bs + 1, 0);
  fphip_enum_stats stt;
  memset(&stt, 0, sizeof stt);
  const int rc = fphip_enum_run(ectx, bs, m->maxdist3, mut.data(), m->rd, m->prn, &o, handoff_cb, nullptr, &hs,
                                nodes.data(), &stt);
  if (rc != FPHIP_OK)
    return rc;
  m->have_sol = hs.have;
  for (int i = 0; i < 64; ++i)
    m->sol[i] = (i < bs && hs.have) ? hs.sol[i] : 0.0;
  m->nodes3 = stt.total_nodes;
  return FPHIP_OK;
}

// type 2: the random choices of rerandomize_block(min_row, max_row, density), bkz.cpp:43-80 — they
// do not depend on the basis, so the whole call is drawn at once, in the reference's order
int serve_plan(const BkzsHost &H, int lattice, BkzMail *m)
{
  const int min_row = m->lo, max_row = m->hi, density = m->density;
  int np = 0, n_moves = 0, n_ops = 0;
  // (a range of exactly two rows makes the reference spin for ever — gmp_urandomm_ui(state, 1) is
  // always 0, so `while (b == a)` never ends, bkz.cpp:53-58; the block is left as it is instead)
  if (max_row - min_row > 2 && H.rnd)
  {
    const size_t niter = 4 * (size_t)(max_row - min_row);
    for (size_t i = 0; i < niter && np < FPHIP_BKZS_PLAN_MAX; ++i)
    {
      const size_t a = H.rnd(H.rnd_user, lattice, (unsigned long)(max_row - min_row - 1)) + min_row;
      size_t b       = a;
      // (the reference's loop is unbounded, bkz.cpp:56-57; a generator that keeps returning the
      // same value — e.g. the 0 a failed Python callback yields through ctypes — must not hang the
      // service thread with a kernel waiting on it: give up after 4096 equal draws)
      for (int tries = 0; b == a; ++tries)
      {
        if (tries >= 4096)
          return -1;
        b = H.rnd(H.rnd_user, lattice, (unsigned long)(max_row - min_row - 1)) + min_row;
      }
      m->plan[np++] = (unsigned)b | ((unsigned)a << 8);
      ++n_moves;
    }
    for (long a = min_row; a < max_row - 2; ++a)
      for (long i = 0; i < density && np < FPHIP_BKZS_PLAN_MAX; i++)
      {
        const size_t b = H.rnd(H.rnd_user, lattice, (unsigned long)(max_row - (a + 1) - 1)) + a + 1;
        const unsigned add = H.rnd(H.rnd_user, lattice, 2) ? 1u : 0u;
        m->plan[np++] = (unsigned)a | ((unsigned)b << 8) | (add << 16);
        ++n_ops;
      }
  }
  m->n_moves = n_moves;
  m->n_ops   = n_ops;
  return np;
}
}  // namespace

// The two host-side decisions on their own (no device involved): what the mailbox service answers
// for a block with the given r_ii / exponents, and the plan it draws for a rerandomisation.  Used by
// the CPU test-suite to check this arithmetic against the oracle (tests/test_bkzs_host_cpu.py).
extern "C" int fphip_debug_bkz_radius(const fphip_strategies *S, double gh_factor, int bs, int flags,
                                      double delta, const double *r, const int *e2, double *max_dist,
                                      int *prune, double *expectation)
{
  if (bs < 1 || bs > 64 || !r || !e2)
    return FPHIP_ERROR;
  BkzMail m;
  memset(&m, 0, sizeof m);
  m.type  = 1;
  m.bs    = bs;
  m.flags = flags;
  m.delta = delta;
  for (int i = 0; i < bs; ++i)
  {
    m.r[i]  = r[i];
    m.e2[i] = e2[i];
  }
  BkzsHost H{S, gh_factor, nullptr, nullptr};
  serve_radius(H, &m);
  if (max_dist)
    *max_dist = m.max_dist;
  if (prune)
    *prune = m.prune;
  if (expectation)
    *expectation = m.expectation;
  return FPHIP_OK;
}

extern "C" int fphip_debug_bkz_plan(fphip_rand_fn rnd, void *rnd_user, int lattice, int lo, int hi,
                                    int density, unsigned *plan, int *n_moves, int *n_ops)
{
  if (!rnd || !plan || hi - lo > 63)
    return FPHIP_ERROR;
  BkzMail m;
  memset(&m, 0, sizeof m);
  m.type    = 2;
  m.lo      = lo;
  m.hi      = hi;
  m.density = density;
  BkzsHost H{nullptr, 1.1, rnd, rnd_user};
  const int np = serve_plan(H, lattice, &m);
  memcpy(plan, m.plan, sizeof(unsigned) * (size_t)np);
  if (n_moves)
    *n_moves = m.n_moves;
  if (n_ops)
    *n_ops = m.n_ops;
  return FPHIP_OK;
}

// One pass of slide reduction on a subset of its (disjoint) blocks: the unit of the block-parallel mode
// (slide_tour's p primal blocks and p - 1 dual blocks, bkz.cpp:475-480, 495-499).
extern "C" int fphip_gso_slide_pass(fphip_gso *g, int block_size, double delta, double eta, int flags,
                                    double gh_factor, const fphip_strategies *S, fphip_rand_fn rnd, void *rnd_user,
                                    int pass, unsigned long long block_mask, int *status, int *info)
{
  if (!g || pass < 1 || pass > 3 || block_size < 2)
    return FPHIP_ERROR;
  if (int rct = transform_guard(g, "slide_pass"))
    return rct;
  // without BKZ_BOUNDED_LLL every svp_reduction starts with an LLL from row 0: the blocks of a pass are
  // not independent then (bkz.cpp:107-108)
  if (!(flags & 0x10))
    return FPHIP_UNSUPPORTED;
  // the block mask has 64 bits (advisor, round 4: a wider pass would silently reduce nothing)
  if ((g->P.d + block_size - 1) / block_size > 64)
    return FPHIP_UNSUPPORTED;
  g->P.sld_pass = pass;
  g->P.sld_mask = block_mask;
  const int rc  = fphip_gso_bkz_strategies(g, block_size, delta, eta, (flags & (0x10 | 0x80 | 0x2000)) | 0x4 | 0x200, 1,
                                           gh_factor, S, rnd, rnd_user, status, info);
  g->P.sld_pass = 0;
  g->P.sld_mask = 0;
  return rc;
}

// The block-parallel slide TOUR over several batch-of-one objects (one per context / device), host threads in
// this process: what fplll_amd.distributed.slide_reduction_blocks does with LocalGather, behind the C ABI so
// that a C++ caller (an fplll process that holds several devices) needs no Python.  Same calls in the same
// order per participant, hence the same result: block i of a pass is reduced by gs[i % count] from the
// pass-start basis, the merged basis replaces every copy, the bounded LLL / the potential test / the closing
// hkz run on every copy alike.
extern "C" int fphip_gso_slide_reduction_blocks(fphip_gso **gs, int count, int block_size, double delta, double eta,
                                                int flags, int max_loops, double gh_factor,
                                                const fphip_strategies *S, fphip_rand_fn rnd, void *rnd_user,
                                                int *status, unsigned long long *nodes_out, int *tours_out)
{
  if (!gs || count < 1 || block_size < 2)
    return FPHIP_ERROR;
  for (int r = 0; r < count; ++r)
    if (!gs[r] || gs[r]->P.batch != 1 || gs[r]->P.d != gs[0]->P.d || gs[r]->P.n != gs[0]->P.n)
      return FPHIP_ERROR;
  if (!(flags & 0x10))
    return FPHIP_UNSUPPORTED;  // the blocks of a pass are independent only with BKZ_BOUNDED_LLL
  // A caller generator cannot be shared by the participants: the host threads would race on its state, and
  // every copy / rank would consume a different number of draws, so that the closing hkz's rerandomisations
  // (and with them the result) would depend on the number of devices (advisor, round 4).  The block-parallel
  // tour is defined for the deterministic case — no rerandomisation: rnd == NULL.
  if (rnd)
    return FPHIP_UNSUPPORTED;
  const int d = gs[0]->P.d, n = gs[0]->P.n;
  const int p = (d + block_size - 1) / block_size;
  if (p > 64)
    return FPHIP_UNSUPPORTED;  // 64-bit block masks
  // the GSO entry points launch on the calling thread's current device: gs[0]'s while this thread uses it
  int caller_dev0 = 0;
  (void)hipGetDevice(&caller_dev0);
  (void)hipSetDevice(fphip_ctx_device(gs[0]->ctx));
  struct RestoreDev
  {
    int dev;
    ~RestoreDev() { (void)hipSetDevice(dev); }
  } restore_dev{caller_dev0};
  struct Blk
  {
    int lo, hi;
  };
  std::vector<Blk> primal, dual;
  for (int i = 0; i < p; ++i)
    primal.push_back(Blk{i * block_size, std::min(d, (i + 1) * block_size)});
  for (int i = 0; i + 1 < p; ++i)
    dual.push_back(Blk{i * block_size + 1, (i + 1) * block_size + 1});
  const size_t rowsz = (size_t)n;
  std::vector<int64_t> start((size_t)d * n), merged((size_t)d * n);
  std::atomic<int> failed{0};
  unsigned long long total_nodes = 0;
  // every participant on a thread of its own: f(r) for r = 0 .. count-1
  int caller_dev = 0;
  (void)hipGetDevice(&caller_dev);
  auto on_all = [&](const std::function<void(int)> &f)
  {
    // (the GSO entry points launch on their context's stream with the calling thread's current device:
    //  every thread selects its object's device first)
    auto on_dev = [&](int r)
    {
      (void)hipSetDevice(fphip_ctx_device(gs[r]->ctx));
      f(r);
    };
    std::vector<std::thread> ts;
    for (int r = 1; r < count; ++r)
      ts.emplace_back(on_dev, r);
    on_dev(0);
    for (auto &t : ts)
      t.join();
    (void)hipSetDevice(caller_dev);
  };
  auto set_all = [&](const std::vector<int64_t> &b)
  {
    on_all([&](int r)
           {
             if (fphip_gso_set_basis(gs[r], 0, 1, b.data()) != FPHIP_OK || fphip_gso_refresh(gs[r]) != FPHIP_OK)
               failed = 1;
           });
  };
  auto run_pass = [&](int pass, const std::vector<Blk> &layout, bool &clean) -> unsigned long long
  {
    if (fphip_gso_get_basis(gs[0], 0, 1, start.data()) != FPHIP_OK)
      failed = 1;
    merged = start;
    std::atomic<int> all_clean{1};
    std::atomic<unsigned long long> nd{0};
    on_all([&](int r)
           {
             std::vector<int64_t> mine((size_t)d * n);
             for (int i = r; i < (int)layout.size(); i += count)
             {
               int st = 0, info[4] = {0, 0, 0, 0};
               if (fphip_gso_set_basis(gs[r], 0, 1, start.data()) != FPHIP_OK || fphip_gso_refresh(gs[r]) != FPHIP_OK ||
                   fphip_gso_slide_pass(gs[r], block_size, delta, eta, flags & (0x10 | 0x80 | 0x2000), gh_factor, S, rnd,
                                        rnd_user, pass, 1ull << i, &st, info) != FPHIP_OK ||
                   st <= 0 || fphip_gso_get_basis(gs[r], 0, 1, mine.data()) != FPHIP_OK)
               {
                 failed = 1;
                 return;
               }
               // (the blocks of a pass are disjoint row ranges: no two threads write the same rows)
               memcpy(&merged[(size_t)layout[i].lo * rowsz], &mine[(size_t)layout[i].lo * rowsz],
                      sizeof(int64_t) * rowsz * (size_t)(layout[i].hi - layout[i].lo));
               if (pass == 1 && !info[0])
                 all_clean = 0;
               nd += ((unsigned long long)(unsigned)info[2] << 32) | (unsigned)info[1];
             }
           });
    set_all(merged);
    clean = all_clean.load() != 0;
    return nd.load();
  };
  auto potential = [&]() -> double
  {
    int st = 0;
    std::vector<double> rm((size_t)d * d), diag(d);
    std::vector<int64_t> re(d);
    if (fphip_gso_update(gs[0], &st) != FPHIP_OK || st != 1 || fphip_gso_get_r(gs[0], 0, rm.data()) != FPHIP_OK ||
        fphip_gso_get_row_expo(gs[0], 0, re.data()) != FPHIP_OK)
    {
      failed = 1;
      return 0.0;
    }
    for (int i = 0; i < d; ++i)
      diag[i] = rm[(size_t)i * d + i];
    return fphip_gso_util_slide_potential(diag.data(), re.data(), d, 0, d, block_size);
  };
  int st_out = 1, tours = 0;
  double old = potential();
  while (!failed)
  {
    if (max_loops > 0 && tours >= max_loops)
    {
      st_out = 8;  // RED_BKZ_LOOPS_LIMIT
      break;
    }
    for (;;)
    {  // primal passes until one leaves every block, and the bounded LLL, unchanged (bkz.cpp:472-494)
      bool clean = true;
      total_nodes += run_pass(1, primal, clean);
      if (failed)
        break;
      std::atomic<int> lll_bad{0}, swaps{0};
      on_all([&](int r)
             {
               int st = 0, info[4] = {0, 0, 0, 0};
               if (fphip_gso_lll(gs[r], 0, 0, d, delta, eta, &st, info) != FPHIP_OK)
                 failed = 1;
               else if (st != 1)
                 lll_bad = st == 0 ? -100 : st;
               if (r == 0)
                 swaps = info[1];
             });
      if (failed)
        break;
      if (lll_bad.load())
      {
        st_out = lll_bad.load() == -100 ? 0 : lll_bad.load();
        failed = 2;  // (a reduction status, not a device error)
        break;
      }
      if (swaps.load() > 0)
        clean = false;
      if (clean)
        break;
    }
    if (failed)
      break;
    if (!dual.empty())
    {
      bool unused = true;
      total_nodes += run_pass(2, dual, unused);
      if (failed)
        break;
    }
    ++tours;
    const double now = potential();
    if (failed || now >= old || block_size >= d)
      break;
    old = now;
  }
  if (failed == 1)
  {
    snprintf(fphip_ctx_errbuf(gs[0]->ctx), 512, "fphip_gso_slide_reduction_blocks: a pass failed: %s",
             fphip_last_error(gs[0]->ctx));
    return FPHIP_ERROR;
  }
  if (failed == 0)
  {  // the closing hkz of every block (bkz.cpp:643-660) on every copy alike; counted once
    std::atomic<int> bad{0};
    std::atomic<unsigned long long> nd{0};
    on_all([&](int r)
           {
             int st = 0, info[4] = {0, 0, 0, 0};
             if (fphip_gso_slide_pass(gs[r], block_size, delta, eta, flags & (0x10 | 0x80 | 0x2000), gh_factor, S, rnd,
                                      rnd_user, 3, 0, &st, info) != FPHIP_OK)
               bad = 1;
             else if (r == 0)
             {
               nd = ((unsigned long long)(unsigned)info[2] << 32) | (unsigned)info[1];
               if (st <= 0)
                 bad = 2 + (st == 0 ? 100 : -st);
             }
           });
    if (bad.load() == 1)
      return FPHIP_ERROR;
    total_nodes += nd.load();
    if (bad.load() >= 2)
      st_out = bad.load() - 2 == 100 ? 0 : -(bad.load() - 2);
  }
  if (status)
    *status = st_out;
  if (nodes_out)
    *nodes_out = total_nodes;
  if (tours_out)
    *tours_out = tours;
  return FPHIP_OK;
}

extern "C" int fphip_gso_bkz_inloop_pruning(fphip_gso *g, double preproc_cost, double target, int min_block_size,
                                            int pruner_flags, int on_device)
{
  if (!g || !(target > 0.0 && target < 1.0) || !(preproc_cost >= 0.0) || (pruner_flags & 0x10))
    return FPHIP_ERROR;
  g->il_preproc   = preproc_cost;
  g->il_target    = target;
  g->il_min_block = min_block_size;
  g->il_flags     = pruner_flags;
  g->il_device    = on_device ? 1 : 0;
  return FPHIP_OK;
}
extern "C" int fphip_gso_bkz_inloop_stats(const fphip_gso *g, unsigned long long *prune_calls,
                                          unsigned long long *device_jobs, unsigned long long *host_jobs,
                                          unsigned long long *launches)
{
  if (!g)
    return FPHIP_ERROR;
  if (prune_calls)
    *prune_calls = g->il_calls;
  if (device_jobs)
    *device_jobs = g->il_device_jobs;
  if (host_jobs)
    *host_jobs = g->il_host_jobs;
  if (launches)
    *launches = g->il_launches;
  return FPHIP_OK;
}

extern "C" int fphip_gso_bkz_strategies(fphip_gso *g, int block_size, double delta, double eta,
                                        int flags, int max_loops, double gh_factor,
                                        const fphip_strategies *S, fphip_rand_fn rnd, void *rnd_user,
                                        int *status, int *info)
{
  if (g)
    if (int rcg = session_guard(g, "bkz_strategies"))
      return rcg;
  if (g)
    if (int rct = transform_guard(g, "bkz_strategies"))
      return rct;
  FPHIP_RANGE("fphip_gso_bkz_strategies");
  if (!g)
    return FPHIP_ERROR;
  // one wavefront enumerates a block: sizes up to 64; BKZ_MAX_LOOPS, BKZ_BOUNDED_LLL, BKZ_AUTO_ABORT,
  // BKZ_GH_BND
  // BKZ_SD_VARIANT (0x100): self-dual BKZ, bkzs_body<NQ, true>
  const bool sd  = (flags & 0x100) != 0;
  const bool sld = (flags & 0x200) != 0;  // BKZ_SLD_RED: slide reduction (slide_tour, bkz.cpp:465-520)
  if (block_size > 64 || (flags & ~(0x4 | 0x8 | 0x10 | 0x20 | 0x40 | 0x80 | 0x100 | 0x200 | 0x1000 | 0x2000)) ||
      (sd && sld))
    return FPHIP_UNSUPPORTED;
  // BKZ_MAX_TIME (0x8) / BKZ_DUMP_GSO (0x40): one tour per launch, the clock and the dump on the host in between
  TourHooks hooks;
  hooks.use_time = (flags & 0x8) != 0;
  hooks.dump     = (flags & 0x40) != 0;
  hooks.max_time = g->bkz_max_time;
  hooks.step     = sd ? "End of SD-BKZ loop" : (sld ? "End of SLD loop" : "End of BKZ loop");
  flags &= ~(0x8 | 0x40);
  // FPHIP_BKZ_PRUNE_IN_LOOP (0x2000): pruning per block from the mailbox service (serve_radius)
  const bool inloop = (flags & 0x2000) != 0;
  flags &= ~0x2000;
  // (a last block of one row — d = k bs + 1 — would be an svp_reduction of block size 1: not offered)
  if (sld && block_size >= 2 && g->P.d % block_size == 1)
    return FPHIP_UNSUPPORTED;
  // FPHIP_BKZ_HANDOFF (0x1000, or FPHIP_BKZ_HANDOFF=1 in the environment): blocks whose tree is large
  // are enumerated by the multi-wave enumerator on a second context instead of by the lattice's wave
  const bool handoff = (flags & 0x1000) != 0 || (getenv("FPHIP_BKZ_HANDOFF") && atoi(getenv("FPHIP_BKZ_HANDOFF")) != 0);
  flags &= ~0x1000;
  if (sd && !(flags & (0x4 | 0x20)) && !hooks.use_time)
    flags |= 0x20;  // "SD Variant of BKZ requires explicit termination condition", bkz.cpp:548-554
  const int bsz = block_size < g->P.d ? block_size : g->P.d;
  if (S)
  {
    if (!rnd)
    {
      snprintf(fphip_ctx_errbuf(g->ctx), 512, "fphip_gso_bkz_strategies: strategies need the caller's generator");
      return FPHIP_ERROR;
    }
    if (S->max_block_size < bsz || !S->pre_off || !S->prune_off || !S->coeff_off)
    {
      snprintf(fphip_ctx_errbuf(g->ctx), 512, "fphip_gso_bkz_strategies: strategies do not cover block size %d", bsz);
      return FPHIP_ERROR;
    }
    // every block size needs a pruning set whose coefficient vector is empty or of that size;
    // preprocessing block sizes must be proper (2 <= p < b) and nest at most MAX_DEPTH - 1 deep
    std::vector<int> depth(bsz + 1, 0);
    for (int b = 0; b <= bsz; ++b)
    {
      if (S->prune_off[b + 1] <= S->prune_off[b])
        return FPHIP_UNSUPPORTED;
      for (int p = S->prune_off[b]; p < S->prune_off[b + 1]; ++p)
      {
        const int len = S->coeff_off[p + 1] - S->coeff_off[p];
        if (len != 0 && len != b)
          return FPHIP_UNSUPPORTED;
      }
      for (int p = S->pre_off[b]; p < S->pre_off[b + 1]; ++p)
      {
        const int pb = S->pre[p];
        if (pb < 2 || pb >= b)
          return FPHIP_UNSUPPORTED;
        // a tour of block size pb reaches every block size <= pb through hkz
        int dmax = 0;
        for (int c = 2; c <= pb; ++c)
          dmax = std::max(dmax, depth[c]);
        depth[b] = std::max(depth[b], dmax + 1);
      }
    }
    int dtop = 0;
    for (int b = 2; b <= bsz; ++b)
      dtop = std::max(dtop, depth[b]);
    if (dtop + 1 > FPHIP_BKZS_MAX_DEPTH)
      return FPHIP_UNSUPPORTED;
  }
  int rc = ensure_lll_buffers(g);
  if (rc != FPHIP_OK)
    return rc;
  const size_t B = (size_t)g->P.batch;
  if (!g->P.enum_mu)
    GCHK(fphip_dev_alloc((void **)&g->P.enum_mu, B * (64 * 63 / 2) * sizeof(double), fphip_ctx_stream(g->ctx)));
  if (!g->P.bkz_active)
    GCHK(fphip_dev_alloc((void **)&g->P.bkz_active, B * sizeof(int), fphip_ctx_stream(g->ctx)));
  if (!g->P.bkz_rows)
    GCHK(fphip_dev_alloc((void **)&g->P.bkz_rows, B * sizeof(int), fphip_ctx_stream(g->ctx)));

  // device copy of what the kernel reads of the strategies; mailboxes; abort flag
  BkzStrat DS;
  memset(&DS, 0, sizeof DS);
  int *d_pre_off = nullptr, *d_pre = nullptr, *d_coeff_off = nullptr, *d_abort = nullptr;
  double *d_coeff = nullptr;
  BkzMail *mail   = nullptr;
  std::vector<fphip_pruner::VolumeEngine *> il_engines;  // in-loop pruning: one volume engine per service worker
  auto cleanup = [&]()
  {
    for (auto *x : il_engines)  // (every error return goes through here: streams and pinned staging go with it)
      fphip_pruner::destroy_volume_engine(x);
    il_engines.clear();
    fphip_dev_free(d_pre_off, fphip_ctx_stream(g->ctx));
    fphip_dev_free(d_pre, fphip_ctx_stream(g->ctx));
    fphip_dev_free(d_coeff_off, fphip_ctx_stream(g->ctx));
    fphip_dev_free(d_coeff, fphip_ctx_stream(g->ctx));
    fphip_dev_free(d_abort, fphip_ctx_stream(g->ctx));
    if (mail)
      pinned_put(mail);
    if (g->P.enum_mu_h)
    {
      pinned_put(g->P.enum_mu_h);
      g->P.enum_mu_h = nullptr;
    }
  };
#define BCHK(call)                         \
  do                                       \
  {                                        \
    hipError_t e_ = (call);                \
    if (e_ != hipSuccess)                  \
    {                                      \
      cleanup();                           \
      return gfail(g->ctx, #call, e_);     \
    }                                      \
  } while (0)
  if (S)
  {
    const int nb   = S->max_block_size + 2;
    const int npre = S->pre_off[S->max_block_size + 1];
    const int nset = S->prune_off[S->max_block_size + 1];
    const int ncoe = S->coeff_off[nset];
    BCHK(fphip_dev_alloc((void **)&d_pre_off, sizeof(int) * nb, fphip_ctx_stream(g->ctx)));
    BCHK(fphip_dev_alloc((void **)&d_pre, sizeof(int) * (npre > 0 ? npre : 1), fphip_ctx_stream(g->ctx)));
    BCHK(fphip_dev_alloc((void **)&d_coeff_off, sizeof(int) * (nset + 1), fphip_ctx_stream(g->ctx)));
    BCHK(fphip_dev_alloc((void **)&d_coeff, sizeof(double) * (ncoe > 0 ? ncoe : 1), fphip_ctx_stream(g->ctx)));
    BCHK(hipMemcpy(d_pre_off, S->pre_off, sizeof(int) * nb, hipMemcpyHostToDevice));
    if (npre > 0)
      BCHK(hipMemcpy(d_pre, S->pre, sizeof(int) * npre, hipMemcpyHostToDevice));
    BCHK(hipMemcpy(d_coeff_off, S->coeff_off, sizeof(int) * (nset + 1), hipMemcpyHostToDevice));
    if (ncoe > 0)
      BCHK(hipMemcpy(d_coeff, S->coeff, sizeof(double) * ncoe, hipMemcpyHostToDevice));
    DS.max_block_size = S->max_block_size;
    DS.pre_off        = d_pre_off;
    DS.pre            = d_pre;
    DS.coeff_off      = d_coeff_off;
    DS.coeff          = d_coeff;
  }
  BCHK(fphip_dev_alloc((void **)&d_abort, sizeof(int), fphip_ctx_stream(g->ctx)));  // (cleared before every launch)
  mail = (BkzMail *)pinned_get(B * sizeof(BkzMail));
  if (!mail)
  {
    cleanup();
    snprintf(fphip_ctx_errbuf(g->ctx), 512, "bkz_strategies: no pinned memory for %zu mailboxes", B);
    return FPHIP_ERROR;
  }
  memset(mail, 0, B * sizeof(BkzMail));

  const int need = (g->P.d > g->P.n ? g->P.d : g->P.n);
  const int nq   = (need + 63) / 64;
  int wpb        = g->waves_per_block;
  const int bs   = block_size < 2 ? 2 : bsz;
  size_t ring_bytes = (size_t)wpb * fphip_reduce_ring_bytes(nq);
  // The scaled mu rows of the block under enumeration go to LDS (behind the column stack) when few
  // lattices share a CU — every row's L1 latency is exposed to a lone wave — and when they fit;
  // large batches keep them in global memory and spend the LDS on resident waves.
  int stack_doubles = (bs * (bs + 1)) / 2 + 2;
  int mu_lds_flag   = 0;
  {
    const int with_mu  = stack_doubles + (bs * (bs - 1)) / 2;
    const size_t lds_m = ring_bytes + (size_t)wpb * with_mu * sizeof(double) + (size_t)wpb * FPHIP_BKZS_MAX_DEPTH * 64;
    const int def      = (B <= (size_t)4 * (size_t)fphip_ctx_num_cus(g->ctx)) ? 1 : 0;
    const char *e      = getenv("FPHIP_BKZ_MU_LDS");
    if ((e ? atoi(e) : def) && lds_m <= 160 * 1024)
    {
      stack_doubles = with_mu;
      mu_lds_flag   = 0x40000000;
    }
  }
  // A big batch of lattices of at most 64 columns (primal schedule): bkzs_kernel<1> is built for two waves per SIMD
  // (256 registers), and what then bounds the resident waves is the LDS per workgroup — ring + column stack, 22.6 KB
  // per wave for blocks of 40: one workgroup of four waves per CU, but three of two.  Measured (call r5z3, BKZ-40 on
  // 64-dim lattices): 327 reductions/s with four waves per workgroup, 397-399 with two (1536 / 3072 lattices); one
  // wave per workgroup: 259.
  if (nq == 1 && !sd && !sld && !mu_lds_flag && !getenv("FPHIP_GSO_WAVES_PER_BLOCK") && wpb == 4)
  {
    const size_t per_wave = fphip_reduce_ring_bytes(nq) + (size_t)stack_doubles * sizeof(double) + FPHIP_BKZS_MAX_DEPTH * 64;
    if ((160 * 1024) / (4 * per_wave) * 4 < (160 * 1024) / (2 * per_wave) * 2)
    {
      wpb        = 2;
      ring_bytes = (size_t)wpb * fphip_reduce_ring_bytes(nq);
    }
  }
  const size_t lds = ring_bytes + (size_t)wpb * stack_doubles * sizeof(double) +
                     (size_t)wpb * FPHIP_BKZS_MAX_DEPTH * 64;  // sizeof(BkzsFrame) == 64
  if (ring_bytes > 64 * 1024 || lds > 160 * 1024)
  {
    cleanup();
    snprintf(fphip_ctx_errbuf(g->ctx), 512, "bkz: %zu bytes of LDS per workgroup do not fit", lds);
    return FPHIP_ERROR;
  }
  int bpc = lds ? (int)((160 * 1024) / lds) : 32;
  if (bpc * wpb > 32)
    bpc = 32 / wpb;
  int grid      = (g->P.batch + wpb - 1) / wpb;
  const int cap = fphip_ctx_num_cus(g->ctx) * (bpc > 0 ? bpc : 1);
  if (grid > cap)
    grid = cap;
  hipStream_t s     = fphip_ctx_stream(g->ctx);
  const double logd = std::log(delta);
  if (lds > 64 * 1024)
  {
    switch (nq)
    {
    case 1: BCHK(hipFuncSetAttribute((const void *)bkzs_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); break;
    case 2: BCHK(hipFuncSetAttribute((const void *)bkzs_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); break;
    case 3: BCHK(hipFuncSetAttribute((const void *)bkzs_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); break;
    default: BCHK(hipFuncSetAttribute((const void *)bkzs_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); break;
    }
  }
  if ((sd || sld) && lds > 64 * 1024)
  {
    switch (nq)
    {
    case 1: BCHK(hipFuncSetAttribute((const void *)sdv::bkzd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); break;
    case 2: BCHK(hipFuncSetAttribute((const void *)sdv::bkzd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); break;
    case 3: BCHK(hipFuncSetAttribute((const void *)sdv::bkzd_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); break;
    default: BCHK(hipFuncSetAttribute((const void *)sdv::bkzd_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); break;
    }
  }
  BkzsHost H{S, gh_factor, rnd, rnd_user, 0.0};
  // in-loop pruning: a pool of worker threads (one prune() is tens of milliseconds; the lattices of a batch
  // ask at about the same time), each with a volume engine of its own (stream, staging, device buffers)
  int n_workers = 1, n_handoff = 1;
  if (inloop)
  {
    H.inloop       = 1;
    H.il_preproc   = g->il_preproc;
    H.il_target    = g->il_target;
    H.il_min_block = g->il_min_block;
    H.il_flags     = g->il_flags;
    const char *we = getenv("FPHIP_BKZ_PRUNE_WORKERS");
    n_workers      = we ? atoi(we) : 8;
    n_workers      = std::max(1, std::min(n_workers, (int)std::min<size_t>(B, 64)));
    for (int w = 0; w < n_workers && g->il_device; ++w)
    {
      char why[256] = {0};
      fphip_pruner::VolumeEngine *e = fphip_pruner::create_device_volume_engine(fphip_ctx_device(g->ctx), why, sizeof why);
      if (!e)
      {
        cleanup();
        snprintf(fphip_ctx_errbuf(g->ctx), 512, "bkz_strategies: %s", why);
        return FPHIP_ERROR;
      }
      il_engines.push_back(e);
    }
  }
  std::atomic<unsigned long long> il_calls{0};
  std::atomic<int> il_errors{0};
  H.il_errors = &il_errors;
  const unsigned long long il_host_jobs0 = inloop ? (unsigned long long)fphip_pruner::host_volume_engine()->host_jobs : 0ull;
  if (handoff)
  {
    // hand-off mode: a second context on this device for the enumerations, the blocks' mu rows in
    // pinned host memory (the wave writes them, the worker thread reads them without a HIP call)
    if (!g->ectx && fphip_create_ex(fphip_ctx_device(g->ctx), 1 /* never behind the schedule kernel */, &g->ectx) != FPHIP_OK)
    {
      snprintf(fphip_ctx_errbuf(g->ctx), 512, "bkz_strategies: no enumeration context for the hand-off: %s",
               g->ectx ? fphip_last_error(g->ectx) : "?");
      if (g->ectx)
        fphip_destroy(g->ectx);
      g->ectx = nullptr;
      cleanup();
      return FPHIP_ERROR;
    }
    // the enumeration's task buffers exist before the schedule kernel runs (advisor, round 4: the first
    // fphip_enum_run of a context allocates them — here that would be the worker thread, mid-launch)
    {
      int cur_dev = 0;
      (void)hipGetDevice(&cur_dev);
      const int erc = fphip_ctx_ensure_task_buffers(g->ectx);
      (void)hipSetDevice(cur_dev);
      if (erc != FPHIP_OK)
      {
        snprintf(fphip_ctx_errbuf(g->ctx), 512, "bkz_strategies: task buffers of the hand-off context: %s",
                 fphip_last_error(g->ectx));
        cleanup();
        return FPHIP_ERROR;
      }
    }
    // a BATCH of tours shares the hand-off service: one worker thread and one enumeration context per worker (the
    // extra contexts with a small task capacity: these blocks have at most 64 rows), so that the blocks of
    // different tours are enumerated side by side instead of queueing behind one context
    {
      const char *hw = getenv("FPHIP_BKZ_HANDOFF_WORKERS");
      const int want = std::max(1, std::min(hw ? atoi(hw) : 8, (int)std::min<size_t>(B, 16)));
      n_handoff      = want;
      int cur_dev = 0;
      (void)hipGetDevice(&cur_dev);
      while ((int)g->ectx_more.size() + 1 < want)
      {
        fphip_ctx *c = nullptr;
        if (fphip_create_ex(fphip_ctx_device(g->ctx), 1, &c) != FPHIP_OK || fphip_ctx_set_task_cap(c, 1u << 18) != FPHIP_OK ||
            fphip_ctx_ensure_task_buffers(c) != FPHIP_OK)
        {
          snprintf(fphip_ctx_errbuf(g->ctx), 512, "bkz_strategies: hand-off context %zu: %s", g->ectx_more.size() + 1,
                   c ? fphip_last_error(c) : "?");
          if (c)
            fphip_destroy(c);
          (void)hipSetDevice(cur_dev);
          cleanup();
          return FPHIP_ERROR;
        }
        g->ectx_more.push_back(c);
      }
      (void)hipSetDevice(cur_dev);
    }
    g->P.enum_mu_h = (double *)pinned_get(B * (64 * 63 / 2) * sizeof(double));
    if (!g->P.enum_mu_h)
    {
      cleanup();
      snprintf(fphip_ctx_errbuf(g->ctx), 512, "bkz_strategies: no pinned memory for the hand-off");
      return FPHIP_ERROR;
    }
    const char *hn  = getenv("FPHIP_BKZ_HANDOFF_NODES");
    H.handoff_nodes = hn ? atof(hn) : 800.0;
  }
  unsigned long long handoff_calls = 0;
  int handoff_rc                   = FPHIP_OK;
  fphip_ctx *handoff_ectx          = nullptr;  // the context whose enumeration failed (its error text)
  bool rnd_failed              = false;
  unsigned long long heartbeat = 0;
  std::vector<unsigned long long> handled(B, 0);
  std::vector<int> active(B, 1);
  BCHK(hipMemcpy(g->P.bkz_active, active.data(), sizeof(int) * B, hipMemcpyHostToDevice));
  // one launch (kflags / kloops as the kernel sees them) with the mailbox service; afterwards the
  // identity-layout GSO of the new bases (same values: every entry is a function of b)
  auto run_once = [&](int kflags, int kloops, float *ms, int *st_out, int *info_out, int run_mode = 7) -> int
  {
    int rc1 = launch(g, 0, g->P.d, 0.0, 2);  // bf / row_expo of every row from b
    if (rc1 != FPHIP_OK)
      return rc1;
    GCHK(hipMemsetAsync(d_abort, 0, sizeof(int), s));  // one launch's timeout must not poison the next
    GCHK(hipEventRecord(g->ev[0], s));
    if (sd || sld)
    {
      switch (nq)
      {
      case 1: hipLaunchKernelGGL(sdv::bkzd_kernel<1>, dim3(grid), dim3(wpb * 64), lds, s, g->P, DS, mail, d_abort, block_size, kflags | (sd ? 0x100 : 0) | mu_lds_flag, delta, eta, logd, kloops, stack_doubles, run_mode); break;
      case 2: hipLaunchKernelGGL(sdv::bkzd_kernel<2>, dim3(grid), dim3(wpb * 64), lds, s, g->P, DS, mail, d_abort, block_size, kflags | (sd ? 0x100 : 0) | mu_lds_flag, delta, eta, logd, kloops, stack_doubles, run_mode); break;
      case 3: hipLaunchKernelGGL(sdv::bkzd_kernel<3>, dim3(grid), dim3(wpb * 64), lds, s, g->P, DS, mail, d_abort, block_size, kflags | (sd ? 0x100 : 0) | mu_lds_flag, delta, eta, logd, kloops, stack_doubles, run_mode); break;
      default: hipLaunchKernelGGL(sdv::bkzd_kernel<4>, dim3(grid), dim3(wpb * 64), lds, s, g->P, DS, mail, d_abort, block_size, kflags | (sd ? 0x100 : 0) | mu_lds_flag, delta, eta, logd, kloops, stack_doubles, run_mode); break;
      }
    }
    else
    switch (nq)
    {
    case 1: hipLaunchKernelGGL(bkzs_kernel<1>, dim3(grid), dim3(wpb * 64), lds, s, g->P, DS, mail, d_abort, block_size, kflags | mu_lds_flag, delta, eta, logd, kloops, stack_doubles); break;
    case 2: hipLaunchKernelGGL(bkzs_kernel<2>, dim3(grid), dim3(wpb * 64), lds, s, g->P, DS, mail, d_abort, block_size, kflags | mu_lds_flag, delta, eta, logd, kloops, stack_doubles); break;
    case 3: hipLaunchKernelGGL(bkzs_kernel<3>, dim3(grid), dim3(wpb * 64), lds, s, g->P, DS, mail, d_abort, block_size, kflags | mu_lds_flag, delta, eta, logd, kloops, stack_doubles); break;
    default: hipLaunchKernelGGL(bkzs_kernel<4>, dim3(grid), dim3(wpb * 64), lds, s, g->P, DS, mail, d_abort, block_size, kflags | mu_lds_flag, delta, eta, logd, kloops, stack_doubles); break;
    }
    GCHK(hipGetLastError());
    GCHK(hipEventRecord(g->ev[1], s));
    // Serve the mailboxes while the kernel runs, from a thread of its own that makes NO HIP call:
    // the kernel's liveness test watches mail[0].heartbeat, and a HIP call (even hipStreamQuery)
    // can stall for seconds behind another host thread's runtime work in the same process
    // (measured: the config-3 tour timed out next to the rest of the GPU test suite).  This thread
    // only waits for the stream.
    std::atomic<bool> stop{false};
    // hand-off mode: type-3 requests are answered by a worker thread of their own (it makes HIP calls
    // — launches on the enumeration context's stream — and may take milliseconds per request; the
    // service thread goes on sweeping, so the heartbeat never stands still)
    std::mutex hq_m;
    std::condition_variable hq_cv;
    std::deque<std::pair<size_t, unsigned long long>> hq;
    bool hq_stop = false;
    // (every hand-off enumeration runs on a context of its own worker; workers beyond the contexts — in-loop pruning
    //  may want more threads than there are hand-off contexts — share the last one under a mutex)
    std::mutex enum_m;
    std::mutex hrc_m;
    std::vector<std::thread> workers;
    const int n_threads = (handoff || inloop) ? std::max(inloop ? n_workers : 1, handoff ? n_handoff : 1) : 0;
    for (int w = 0; w < n_threads; ++w)
      workers.emplace_back([&, w]()
      {
        fphip_pruner::VolumeEngine *engine = w < (int)il_engines.size() ? il_engines[w] : nullptr;
        const int nctx      = 1 + (int)g->ectx_more.size();
        const bool own_ctx  = handoff && w < nctx - 1;  // (the last context may be shared)
        const int ci        = std::min(w, nctx - 1);    // (0 = the object's own context — also every worker's when
        fphip_ctx *my_ectx  = !handoff ? nullptr : (ci == 0 ? g->ectx : g->ectx_more[ci - 1]);  //  there is no other)
        for (;;)
        {
          std::pair<size_t, unsigned long long> job;
          {
            std::unique_lock<std::mutex> lk(hq_m);
            hq_cv.wait(lk, [&] { return hq_stop || !hq.empty(); });
            if (hq.empty())
              return;
            job = hq.front();
            hq.pop_front();
          }
          BkzMail *m   = &mail[job.first];
          if (m->type == 1)
          {  // in-loop pruning: radius, prune() on the block's profile (its batches on the engine's stream)
            serve_radius(H, m, engine);
            ++il_calls;
            __atomic_store_n(&m->rsp_seq, job.second, __ATOMIC_RELEASE);
            continue;
          }
          int rc;
          if (own_ctx)
            rc = serve_enumeration(my_ectx, m, g->P.enum_mu_h + job.first * (64 * 63 / 2));
          else
          {
            std::lock_guard<std::mutex> lk(enum_m);
            rc = serve_enumeration(my_ectx, m, g->P.enum_mu_h + job.first * (64 * 63 / 2));
          }
          {
            std::lock_guard<std::mutex> lk(hrc_m);
            if (rc != FPHIP_OK)
            {  // the wave cannot walk the block itself any more: no solution, and the call reports the error
              m->have_sol = 0;
              m->nodes3   = 0;
              handoff_rc  = rc;
              handoff_ectx = my_ectx;
            }
            ++handoff_calls;
          }
          __atomic_store_n(&m->rsp_seq, job.second, __ATOMIC_RELEASE);
        }
      });
    std::thread server([&]()
    {
      for (;;)
      {
        const bool last = stop.load(std::memory_order_acquire);  // one more sweep after the kernel ended
        for (size_t L = 0; L < B; ++L)
        {
          BkzMail *m = &mail[L];
          const unsigned long long seq = __atomic_load_n(&m->req_seq, __ATOMIC_ACQUIRE);
          if (seq == handled[L])
            continue;
          if ((m->type == 3 && handoff) || inloop_block(H, m))
          {  // to the worker; it stores rsp_seq when the enumeration (or the block's prune()) is done
            handled[L] = seq;
            {
              std::lock_guard<std::mutex> lk(hq_m);
              hq.emplace_back(L, seq);
            }
            hq_cv.notify_one();
            continue;
          }
          if (m->type == 1)
            serve_radius(H, m);
          else if (m->type == 2 && serve_plan(H, (int)L, m) < 0)
          {  // unusable generator: answer with an empty plan, report the error after the launch
            m->n_moves = m->n_ops = 0;
            rnd_failed            = true;
          }
          handled[L] = seq;
          __atomic_store_n(&m->rsp_seq, seq, __ATOMIC_RELEASE);
        }
        __atomic_store_n(&mail[0].heartbeat, ++heartbeat, __ATOMIC_RELEASE);
        if (last)
          break;
      }
    });
    const hipError_t q = hipStreamSynchronize(s);
    stop.store(true, std::memory_order_release);
    server.join();
    if (handoff || inloop)
    {
      {
        std::lock_guard<std::mutex> lk(hq_m);
        hq_stop = true;
      }
      hq_cv.notify_all();
      for (auto &t : workers)
        t.join();
    }
    if (q != hipSuccess)
      return gfail(g->ctx, "bkzs_kernel", q);
    GCHK(hipEventElapsedTime(ms, g->ev[0], g->ev[1]));
    GCHK(hipMemcpy(st_out, g->P.status, sizeof(int) * B, hipMemcpyDeviceToHost));
    GCHK(hipMemcpy(info_out, g->P.lll_info, sizeof(int) * 4 * B, hipMemcpyDeviceToHost));
    std::swap(g->P.b, g->P.b2);  // the kernel wrote the rows in position order into b2
    rc1 = launch(g, 0, g->P.d, 0.0, 2);
    if (rc1 == FPHIP_OK)
      rc1 = launch(g, 0, g->P.d, 0.0, 0);
    return rc1;
  };

  std::vector<int> st(B, 1), inf(4 * B, 0), hook_rows(B, (int)g->P.d);
  float total_ms = 0, ms = 0;
  const bool use_loops = (flags & 0x4) != 0, auto_abort = (flags & 0x20) != 0;
  const int kbase = flags & (0x10 | 0x80);
  if (sld && g->P.sld_pass != 0)
  {
    // block-parallel mode (fphip_gso_slide_pass): ONE pass of a slide tour restricted to the blocks of
    // g->P.sld_mask (1 primal, 2 dual), or the closing hkz of every block (3) — the caller runs the tour
    rc       = run_once(kbase | 0x4 | 0x200, 1, &ms, st.data(), inf.data(), g->P.sld_pass == 3 ? 4 : 2);
    total_ms = ms;
  }
  else if (sld)
  {
    // slide reduction: one slide_tour per launch (the kernel's 0x200 frame), the potential test on the
    // host in between; then the closing hkz of every block (run_mode 4)
    rc = auto_abort_loop(g, block_size, use_loops, max_loops, active, st, inf, hook_rows, total_ms,
                         [&](int, float *tms, int *s1, int *one)
                         { return run_once(kbase | 0x4 | 0x200, 1, tms, s1, one, 2); },
                         auto_abort, true, hooks);
    std::vector<int> one(4 * B), s1(B);
    if (rc == FPHIP_OK)
    {
      for (size_t L = 0; L < B; ++L)
        active[L] = (st[L] == 1 || st[L] == 8 || st[L] == 7) ? 1 : 0;
      BCHK(hipMemcpy(g->P.bkz_active, active.data(), sizeof(int) * B, hipMemcpyHostToDevice));
      rc = run_once(kbase | 0x4 | 0x200, 1, &ms, s1.data(), one.data(), 4);
      total_ms += ms;
      for (size_t L = 0; L < B && rc == FPHIP_OK; ++L)
      {
        if (!active[L])
          continue;
        const unsigned long long a0 = ((unsigned long long)(unsigned)inf[4 * L + 2] << 32) | (unsigned)inf[4 * L + 1];
        const unsigned long long a1 = ((unsigned long long)(unsigned)one[4 * L + 2] << 32) | (unsigned)one[4 * L + 1];
        const unsigned long long t  = a0 + a1;
        inf[4 * L + 1] = (int)(unsigned)(t & 0xffffffffull);
        inf[4 * L + 2] = (int)(unsigned)(t >> 32);
        inf[4 * L + 3] += one[4 * L + 3];
        if (s1[L] <= 0)
          st[L] = s1[L];
      }
    }
  }
  else if (!auto_abort && !hooks.use_time && !hooks.dump)
  {
    rc       = run_once(kbase | (use_loops ? 0x4 : 0), max_loops, &ms, st.data(), inf.data());
    total_ms = ms;
  }
  else
  {
    // one tour per launch (self-dual BKZ: the prelude lll() goes with the first one only)
    rc = auto_abort_loop(g, block_size, use_loops, max_loops, active, st, inf, hook_rows, total_ms,
                         [&](int loop, float *tms, int *s1, int *one)
                         { return run_once(kbase | 0x4, 1, tms, s1, one, sd && loop == 0 ? 3 : 2); },
                         auto_abort, false, hooks);
    std::vector<int> one(4 * B), s1(B);
    if (sd && rc == FPHIP_OK)
    {
      // closing pass of self-dual BKZ on every lattice that ended regularly: hkz of the last window
      // (bkz.cpp:627-641), its own launch
      for (size_t L = 0; L < B; ++L)
        active[L] = (st[L] == 1 || st[L] == 8 || st[L] == 7) ? 1 : 0;
      BCHK(hipMemcpy(g->P.bkz_active, active.data(), sizeof(int) * B, hipMemcpyHostToDevice));
      rc = run_once(kbase | 0x4, 1, &ms, s1.data(), one.data(), 4);
      total_ms += ms;
      for (size_t L = 0; L < B && rc == FPHIP_OK; ++L)
      {
        if (!active[L])
          continue;
        const unsigned long long a0 = ((unsigned long long)(unsigned)inf[4 * L + 2] << 32) | (unsigned)inf[4 * L + 1];
        const unsigned long long a1 = ((unsigned long long)(unsigned)one[4 * L + 2] << 32) | (unsigned)one[4 * L + 1];
        const unsigned long long t  = a0 + a1;
        inf[4 * L + 1] = (int)(unsigned)(t & 0xffffffffull);
        inf[4 * L + 2] = (int)(unsigned)(t >> 32);
        inf[4 * L + 3] += one[4 * L + 3];
        if (s1[L] <= 0)
          st[L] = s1[L];
      }
    }
  }
#undef BCHK
  if (rc == FPHIP_OK && hooks.dump && !(sld && g->P.sld_pass != 0))
    rc = dump_gso_output(g, hooks, st, hook_rows);  // "Output", bkz.cpp:667-670 (after the closing hkz passes)
  if (inloop)
  {
    g->il_calls += il_calls.load();
    for (auto *e : il_engines)
    {
      g->il_device_jobs += e->device_jobs;
      g->il_host_jobs += e->host_jobs;
      g->il_launches += e->launches;
    }
    // without device engines the workers share the host engine: its jobs of this call
    if (il_engines.empty())
      g->il_host_jobs += fphip_pruner::host_volume_engine()->host_jobs - il_host_jobs0;
  }
  cleanup();  // (run_once's GCHKs return to this function, never past it: nothing leaks on a HIP error; the
              //  volume engines go here too)
  if (handoff && getenv("FPHIP_DEBUG"))
    fprintf(stderr, "[fphip] bkz_strategies: %llu block enumerations handed to the multi-wave enumerator\n",
            handoff_calls);
  if (handoff_rc != FPHIP_OK && rc == FPHIP_OK)
  {
    snprintf(fphip_ctx_errbuf(g->ctx), 512, "bkz_strategies: a handed-off enumeration failed: %s",
             handoff_ectx ? fphip_last_error(handoff_ectx) : (g->ectx ? fphip_last_error(g->ectx) : "?"));
    rc = FPHIP_ERROR;
  }
  if (il_errors.load() > 0 && rc == FPHIP_OK)
  {
    snprintf(fphip_ctx_errbuf(g->ctx), 512, "bkz_strategies: %d in-loop prune() calls lost their volume engine "
             "(the strategies' sets were used for those blocks)", il_errors.load());
    rc = FPHIP_ERROR;
  }
  if (rnd_failed && rc == FPHIP_OK)
  {
    snprintf(fphip_ctx_errbuf(g->ctx), 512,
             "bkz_strategies: the caller's rnd() kept returning the same value (rerandomize_block "
             "needs two different rows); for large batches rnd should be a C function");
    rc = FPHIP_ERROR;
  }
  g->last_ms = total_ms;
  if (status)
    memcpy(status, st.data(), sizeof(int) * B);
  if (info)
    memcpy(info, inf.data(), sizeof(int) * 4 * B);
  return rc;
}

extern "C" int fphip_gso_get_mu(fphip_gso *g, int lattice, double *mu)
{
  if (!g || !mu || lattice < 0 || lattice >= g->P.batch)
    return FPHIP_ERROR;
  if (int rcg = session_guard(g, "fphip_gso_get_mu"))
    return rcg;
  GCHK(hipMemcpy2D(mu, (size_t)g->P.d * 8, g->P.mu + (size_t)lattice * g->P.d * g->P.ldd,
                   (size_t)g->P.ldd * 8, (size_t)g->P.d * 8, g->P.d, hipMemcpyDeviceToHost));
  return FPHIP_OK;
}

extern "C" int fphip_gso_get_r(fphip_gso *g, int lattice, double *r)
{
  if (!g || !r || lattice < 0 || lattice >= g->P.batch)
    return FPHIP_ERROR;
  if (int rcg = session_guard(g, "fphip_gso_get_r"))
    return rcg;
  GCHK(hipMemcpy2D(r, (size_t)g->P.d * 8, g->P.r + (size_t)lattice * g->P.d * g->P.ldd,
                   (size_t)g->P.ldd * 8, (size_t)g->P.d * 8, g->P.d, hipMemcpyDeviceToHost));
  return FPHIP_OK;
}

extern "C" int fphip_gso_get_row_expo(fphip_gso *g, int lattice, int64_t *row_expo)
{
  if (!g || !row_expo || lattice < 0 || lattice >= g->P.batch)
    return FPHIP_ERROR;
  if (int rcg = session_guard(g, "fphip_gso_get_row_expo"))
    return rcg;
  GCHK(hipMemcpy(row_expo, g->P.rexp + (size_t)lattice * g->P.d, sizeof(long long) * g->P.d,
                 hipMemcpyDeviceToHost));
  return FPHIP_OK;
}

extern "C" double fphip_gso_last_kernel_ms(const fphip_gso *g) { return g ? g->last_ms : 0.0; }

// FETCH_SIZE calibration (not part of the product ABI; used by tests/perf/calib_fetch.py only)
namespace fphip
{
__global__ void gso_calib_kernel(const char *buf, size_t stride, int row_bytes, long long rows);
}
extern "C" int fphip_debug_stream(fphip_ctx *ctx, long long rows, int row_bytes, long long stride,
                                  double *ms_out)
{
  char *buf = nullptr;
  const size_t total = (size_t)rows * (size_t)stride + 4096;
  if (fphip_dev_alloc((void **)&buf, total, fphip_ctx_stream(ctx)) != hipSuccess)
    return FPHIP_ERROR;
  hipMemset(buf, 1, total);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipStream_t s = fphip_ctx_stream(ctx);
  hipEventRecord(e0, s);
  hipLaunchKernelGGL(gso_calib_kernel, dim3(fphip_ctx_num_cus(ctx) * 2), dim3(256), 4 * 8 * 2048, s, buf,
                     (size_t)stride, row_bytes, rows);
  hipEventRecord(e1, s);
  hipStreamSynchronize(s);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  if (ms_out)
    *ms_out = ms;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  fphip_dev_free(buf, fphip_ctx_stream(ctx));
  return FPHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// Batched Householder R-factor: MatHouseholder<Z_NR<long>, FP_NR<double>> refresh_R_bf() +
// update_R() (fplll/householder.h:532-536, 610-614)
// ---------------------------------------------------------------------------------------------
namespace fphip
{
template <int NQ> __global__ void hh_update_kernel(HhBatch P);
template <int NT> __global__ void hh_rows_kernel(HhBatch P);
__global__ void hh_size_reduce_kernel(HhBatch P, int k, int end, int start, int *reduced);
}

struct fphip_hh
{
  fphip_ctx *ctx;
  HhBatch P;
  hipEvent_t ev[2];
  float last_ms;
  double *Tbuf;  // blocked (MFMA) mode only: T of every block of 16 reflectors, [batch][ceil(d/16)][256]
  // extended-precision HLLL (hlll_x.hip): low planes of R and V, per-row scalars
  double *Rlo, *Vlo, *xsc;
  long long *xprevE;
  double *xThi = nullptr, *xTlo = nullptr;  // hlll_x: T of every block of 16 reflectors (blocked application)
  double *xRx = nullptr, *xVx = nullptr;    // hlll_x in quad-double: components 2 and 3 of R and V
};

#define HCHK(call)                     \
  do                                   \
  {                                    \
    hipError_t e_ = (call);            \
    if (e_ != hipSuccess)              \
      return gfail(h->ctx, #call, e_); \
  } while (0)

static int hh_allocate(fphip_hh *h);
extern "C" void fphip_hh_destroy(fphip_hh *h);

extern "C" int fphip_hh_create(fphip_ctx *ctx, int batch, int d, int n, int row_expo, fphip_hh **out)
{
  if (!ctx || !out)
    return FPHIP_ERROR;
  *out = nullptr;
  if (batch <= 0 || d <= 0 || n <= 0 || !fphip_ctx_stream(ctx))
  {
    snprintf(fphip_ctx_errbuf(ctx), 512, "fphip_hh_create: bad arguments or no device");
    return FPHIP_ERROR;
  }
  if (d > 256 || n > 256)
    return FPHIP_UNSUPPORTED;
  fphip_hh *h = new fphip_hh();
  memset(h, 0, sizeof *h);
  h->ctx        = ctx;
  h->P.batch    = batch;
  h->P.d        = d;
  h->P.n        = n;
  h->P.ldn      = (n + 15) / 16 * 16;
  if (h->P.ldn > 256)
    h->P.ldn = 256;
  h->P.row_expo = row_expo ? 1 : 0;
  const int rc  = hh_allocate(h);
  if (rc != FPHIP_OK)
  {
    fphip_hh_destroy(h);
    return rc;
  }
  *out = h;
  return FPHIP_OK;
}

static int hh_allocate(fphip_hh *h)
{
  const int batch = h->P.batch, d = h->P.d;
  const size_t B = (size_t)batch, ld = h->P.ldn, pad = 4096;
  HCHK(fphip_dev_alloc((void **)&h->P.b, B * d * ld * 8 + pad, fphip_ctx_stream(h->ctx)));
  HCHK(fphip_dev_alloc((void **)&h->P.V, B * d * ld * 8 + pad, fphip_ctx_stream(h->ctx)));
  HCHK(fphip_dev_alloc((void **)&h->P.R, B * d * ld * 8 + pad, fphip_ctx_stream(h->ctx)));
  HCHK(fphip_dev_alloc((void **)&h->P.sigma, B * d * 8, fphip_ctx_stream(h->ctx)));
  HCHK(fphip_dev_alloc((void **)&h->P.rexp, B * d * 8, fphip_ctx_stream(h->ctx)));
  HCHK(fphip_dev_alloc((void **)&h->P.status, B * sizeof(int), fphip_ctx_stream(h->ctx)));
  HCHK(hipMemsetAsync(h->P.b, 0, B * d * ld * 8 + pad, fphip_ctx_stream(h->ctx)));
  HCHK(hipMemsetAsync(h->P.V, 0, B * d * ld * 8 + pad, fphip_ctx_stream(h->ctx)));
  HCHK(hipMemsetAsync(h->P.R, 0, B * d * ld * 8 + pad, fphip_ctx_stream(h->ctx)));
  HCHK(hipStreamSynchronize(fphip_ctx_stream(h->ctx)));  // uploads use blocking copies on the null stream
  HCHK(hipEventCreate(&h->ev[0]));
  HCHK(hipEventCreate(&h->ev[1]));
  return FPHIP_OK;
}

extern "C" void fphip_hh_destroy(fphip_hh *h)
{
  if (!h)
    return;
  hipStreamSynchronize(fphip_ctx_stream(h->ctx));
  fphip_dev_free(h->P.b, fphip_ctx_stream(h->ctx));
  fphip_dev_free(h->P.V, fphip_ctx_stream(h->ctx));
  fphip_dev_free(h->P.R, fphip_ctx_stream(h->ctx));
  fphip_dev_free(h->P.sigma, fphip_ctx_stream(h->ctx));
  fphip_dev_free(h->P.rexp, fphip_ctx_stream(h->ctx));
  fphip_dev_free(h->P.status, fphip_ctx_stream(h->ctx));
  if (h->P.bf)
    fphip_dev_free(h->P.bf, fphip_ctx_stream(h->ctx));
  if (h->P.info)
    fphip_dev_free(h->P.info, fphip_ctx_stream(h->ctx));
  if (h->ev[0])
    hipEventDestroy(h->ev[0]);
  if (h->ev[1])
    hipEventDestroy(h->ev[1]);
  if (h->Tbuf)
    fphip_dev_free(h->Tbuf, fphip_ctx_stream(h->ctx));
  if (h->Rlo)
    fphip_dev_free(h->Rlo, fphip_ctx_stream(h->ctx));
  if (h->Vlo)
    fphip_dev_free(h->Vlo, fphip_ctx_stream(h->ctx));
  if (h->xsc)
    fphip_dev_free(h->xsc, fphip_ctx_stream(h->ctx));
  if (h->xprevE)
    fphip_dev_free(h->xprevE, fphip_ctx_stream(h->ctx));
  if (h->xThi)
    fphip_dev_free(h->xThi, fphip_ctx_stream(h->ctx));
  if (h->xTlo)
    fphip_dev_free(h->xTlo, fphip_ctx_stream(h->ctx));
  if (h->xRx)
    fphip_dev_free(h->xRx, fphip_ctx_stream(h->ctx));
  if (h->xVx)
    fphip_dev_free(h->xVx, fphip_ctx_stream(h->ctx));
  delete h;
}

extern "C" int fphip_hh_set_basis(fphip_hh *h, int first, int count, const int64_t *b)
{
  if (!h || !b || first < 0 || count <= 0 || first + count > h->P.batch)
    return FPHIP_ERROR;
  HCHK(hipMemcpy2D(h->P.b + (size_t)first * h->P.d * h->P.ldn, (size_t)h->P.ldn * 8, b,
                   (size_t)h->P.n * 8, (size_t)h->P.n * 8, (size_t)h->P.d * count,
                   hipMemcpyHostToDevice));
  HCHK(hipStreamSynchronize(nullptr));  // (see fphip_gso_set_basis: complete on the device before a kernel may run)
  return FPHIP_OK;
}

extern "C" int fphip_hh_broadcast_basis(fphip_hh *h, int src)
{
  if (!h || src < 0 || src >= h->P.batch)
    return FPHIP_ERROR;
  const size_t per = (size_t)h->P.d * h->P.ldn;
  for (int L = 0; L < h->P.batch; ++L)
    if (L != src)
      HCHK(hipMemcpyAsync(h->P.b + (size_t)L * per, h->P.b + (size_t)src * per, per * 8,
                          hipMemcpyDeviceToDevice, fphip_ctx_stream(h->ctx)));
  HCHK(hipStreamSynchronize(fphip_ctx_stream(h->ctx)));
  return FPHIP_OK;
}

// refresh_R_bf() + update_R() for every lattice; status[batch] = 1
extern "C" int fphip_hh_update_R(fphip_hh *h, int *status)
{
  FPHIP_RANGE("fphip_hh_update_R");
  if (!h)
    return FPHIP_ERROR;
  const int nq  = (h->P.n + 63) / 64;
  const int wpb = 4;
  const size_t lds = (size_t)wpb * FPHIP_GSO_RING * (size_t)((nq + 1) / 2) * 1024;
  int bpc          = (int)((160 * 1024) / lds);
  if (bpc * wpb > 32)
    bpc = 32 / wpb;
  int grid = (h->P.batch + wpb - 1) / wpb;
  if (grid > fphip_ctx_num_cus(h->ctx) * bpc)
    grid = fphip_ctx_num_cus(h->ctx) * bpc;
  hipStream_t s = fphip_ctx_stream(h->ctx);
  HCHK(hipEventRecord(h->ev[0], s));
  // rows up to 192 columns: the batch across the lanes (hh_rows.hip: a lane owns a row of one lattice and runs the
  // reference's scalar loops; the same bits); FPHIP_HH_ROWS=0 keeps the lane-per-column kernel (the A/B partner)
  const char *hr = getenv("FPHIP_HH_ROWS");
  if (h->P.n <= 192 && !(hr && hr[0] == '0'))
  {
    const int nt      = h->P.n <= 64 ? 4 : (h->P.n <= 128 ? 8 : 12);
    const size_t ldsr = ((size_t)3 * 16 * (16 * nt + 2) + 3 * 16) * sizeof(double);
    int gridr         = (h->P.batch + 15) / 16;
    if (gridr > fphip_ctx_num_cus(h->ctx))
      gridr = fphip_ctx_num_cus(h->ctx);
    switch (nt)
    {
    case 4: hipLaunchKernelGGL(hh_rows_kernel<4>, dim3(gridr), dim3(256), ldsr, s, h->P); break;
    case 8: hipLaunchKernelGGL(hh_rows_kernel<8>, dim3(gridr), dim3(256), ldsr, s, h->P); break;
    default: hipLaunchKernelGGL(hh_rows_kernel<12>, dim3(gridr), dim3(256), ldsr, s, h->P); break;
    }
  }
  else
  switch (nq)
  {
  case 1: hipLaunchKernelGGL(hh_update_kernel<1>, dim3(grid), dim3(wpb * 64), lds, s, h->P); break;
  case 2: hipLaunchKernelGGL(hh_update_kernel<2>, dim3(grid), dim3(wpb * 64), lds, s, h->P); break;
  case 3: hipLaunchKernelGGL(hh_update_kernel<3>, dim3(grid), dim3(wpb * 64), lds, s, h->P); break;
  default: hipLaunchKernelGGL(hh_update_kernel<4>, dim3(grid), dim3(wpb * 64), lds, s, h->P); break;
  }
  HCHK(hipGetLastError());
  HCHK(hipEventRecord(h->ev[1], s));
  HCHK(hipStreamSynchronize(s));
  HCHK(hipEventElapsedTime(&h->last_ms, h->ev[0], h->ev[1]));
  if (status)
    HCHK(hipMemcpy(status, h->P.status, sizeof(int) * h->P.batch, hipMemcpyDeviceToHost));
  return FPHIP_OK;
}

// MatHouseholder::size_reduce(kappa, size_reduction_end, size_reduction_start) on every lattice of the batch
// (householder.cpp:402-451; kernel in hh_rows.hip).  The state must be the one fphip_hh_update_R left (or any state
// in which rows < kappa of R are final and R(kappa, c < kappa) are the values update_R(kappa, false) gives).
extern "C" int fphip_hh_size_reduce(fphip_hh *h, int kappa, int size_reduction_end, int size_reduction_start,
                                    int *reduced, int *status)
{
  FPHIP_RANGE("fphip_hh_size_reduce");
  if (!h || !reduced)
    return FPHIP_ERROR;
  if (!(kappa > 0 && kappa < h->P.d) || size_reduction_start < 0 || size_reduction_end > kappa ||
      size_reduction_start > size_reduction_end)
  {
    snprintf(fphip_ctx_errbuf(h->ctx), 512,
             "fphip_hh_size_reduce: need 0 < kappa < d and 0 <= start <= end <= kappa (got %d, %d, %d)", kappa,
             size_reduction_end, size_reduction_start);
    return FPHIP_ERROR;
  }
  hipStream_t s = fphip_ctx_stream(h->ctx);
  int *dred     = nullptr;
  HCHK(fphip_dev_alloc((void **)&dred, sizeof(int) * (size_t)h->P.batch, s));
  HCHK(hipEventRecord(h->ev[0], s));
  hipLaunchKernelGGL(hh_size_reduce_kernel, dim3((h->P.batch + 3) / 4), dim3(256), 0, s, h->P, kappa,
                     size_reduction_end, size_reduction_start, dred);
  hipError_t le = hipGetLastError();
  HCHK(hipEventRecord(h->ev[1], s));
  hipError_t se = hipStreamSynchronize(s);
  if (le == hipSuccess && se == hipSuccess)
  {
    hipEventElapsedTime(&h->last_ms, h->ev[0], h->ev[1]);
    se = hipMemcpy(reduced, dred, sizeof(int) * h->P.batch, hipMemcpyDeviceToHost);
    if (se == hipSuccess && status)
      se = hipMemcpy(status, h->P.status, sizeof(int) * h->P.batch, hipMemcpyDeviceToHost);
  }
  fphip_dev_free(dred, s);
  HCHK(le);
  HCHK(se);
  return FPHIP_OK;
}

// The same R-factor in blocked compact-WY form on the MFMA matrix cores (hh_blocked.hip): the
// opt-in fast mode — sums in another order, hence other roundings than the reference's; R / mu / r
// agree with the exact mode to ~1e-13 relative (checked to 1e-9), row exponents and signs identical.
extern "C" int fphip_hh_update_R_blocked(fphip_hh *h, int *status)
{
  FPHIP_RANGE("fphip_hh_update_R_blocked");
  if (!h)
    return FPHIP_ERROR;
  const int nq   = (h->P.n + 63) / 64;
  const int nblk = (h->P.d + 15) / 16;
  if (!h->Tbuf)
    HCHK(fphip_dev_alloc((void **)&h->Tbuf, (size_t)h->P.batch * nblk * 256 * sizeof(double), fphip_ctx_stream(h->ctx)));
  const int ldx    = ((h->P.n + 31) & ~31) + 1;
  const size_t lds = (size_t)(16 * ldx + 256) * sizeof(double);
  int bpc          = (int)((160 * 1024) / lds);
  if (bpc > 16)
    bpc = 16;
  int grid = h->P.batch;
  if (grid > fphip_ctx_num_cus(h->ctx) * bpc)
    grid = fphip_ctx_num_cus(h->ctx) * bpc;
  hipStream_t s = fphip_ctx_stream(h->ctx);
  HCHK(hipEventRecord(h->ev[0], s));
  switch (nq)
  {
  case 1: hipLaunchKernelGGL(hh_blocked_kernel<1>, dim3(grid), dim3(64), lds, s, h->P, h->Tbuf); break;
  case 2: hipLaunchKernelGGL(hh_blocked_kernel<2>, dim3(grid), dim3(64), lds, s, h->P, h->Tbuf); break;
  case 3: hipLaunchKernelGGL(hh_blocked_kernel<3>, dim3(grid), dim3(64), lds, s, h->P, h->Tbuf); break;
  default: hipLaunchKernelGGL(hh_blocked_kernel<4>, dim3(grid), dim3(64), lds, s, h->P, h->Tbuf); break;
  }
  HCHK(hipGetLastError());
  HCHK(hipEventRecord(h->ev[1], s));
  HCHK(hipStreamSynchronize(s));
  HCHK(hipEventElapsedTime(&h->last_ms, h->ev[0], h->ev[1]));
  if (status)
    HCHK(hipMemcpy(status, h->P.status, sizeof(int) * h->P.batch, hipMemcpyDeviceToHost));
  return FPHIP_OK;
}

extern "C" int fphip_hh_get_basis(fphip_hh *h, int first, int count, int64_t *b)
{
  if (!h || !b || first < 0 || count <= 0 || first + count > h->P.batch)
    return FPHIP_ERROR;
  HCHK(hipMemcpy2D(b, (size_t)h->P.n * 8, h->P.b + (size_t)first * h->P.d * h->P.ldn,
                   (size_t)h->P.ldn * 8, (size_t)h->P.n * 8, (size_t)h->P.d * count,
                   hipMemcpyDeviceToHost));
  return FPHIP_OK;
}

// HLLLReduction<Z_NR<long>, FP_NR<double>>(m, delta, eta, theta, c, LLL_DEFAULT).hlll() on a fresh
// MatHouseholder of every lattice (hlll.cpp:26-169).  eta and c are accepted for interface parity:
// the default build of the reference uses neither (eR is delta*R(k,k), hlll.h:155-159; the size
// reduction stop rule uses the constant 0.1, hlll.cpp:297).  info (nullable) [batch][2]: swaps,
// loop iterations.
extern "C" int fphip_hh_hlll(fphip_hh *h, double delta, double eta, double theta, double c,
                             int *status, int *info)
{
  FPHIP_RANGE("fphip_hh_hlll");
  (void)eta;
  (void)c;
  if (!h)
    return FPHIP_ERROR;
  const size_t B = (size_t)h->P.batch, d = h->P.d, ld = h->P.ldn;
  if (!h->P.bf)
  {
    HCHK(fphip_dev_alloc((void **)&h->P.bf, B * d * ld * 8 + 4096, fphip_ctx_stream(h->ctx)));
    HCHK(fphip_dev_alloc((void **)&h->P.info, B * 2 * sizeof(int), fphip_ctx_stream(h->ctx)));
    HCHK(hipMemsetAsync(h->P.bf, 0, B * d * ld * 8 + 4096, fphip_ctx_stream(h->ctx)));
  }
  const int nq  = (h->P.n + 63) / 64;
  const int wpb = 4;
  const size_t lds = (size_t)wpb * FPHIP_RING_REDUCE * (size_t)((nq + 1) / 2) * 1024;
  int bpc          = (int)((160 * 1024) / lds);
  if (bpc * wpb > 32)
    bpc = 32 / wpb;
  int grid = (h->P.batch + wpb - 1) / wpb;
  if (grid > fphip_ctx_num_cus(h->ctx) * bpc)
    grid = fphip_ctx_num_cus(h->ctx) * bpc;
  const long long cap = 1LL << 40;
  hipStream_t s = fphip_ctx_stream(h->ctx);
  HCHK(hipEventRecord(h->ev[0], s));
  switch (nq)
  {
  case 1: hipLaunchKernelGGL(hlll_kernel<1>, dim3(grid), dim3(wpb * 64), lds, s, h->P, delta, theta, cap); break;
  case 2: hipLaunchKernelGGL(hlll_kernel<2>, dim3(grid), dim3(wpb * 64), lds, s, h->P, delta, theta, cap); break;
  case 3: hipLaunchKernelGGL(hlll_kernel<3>, dim3(grid), dim3(wpb * 64), lds, s, h->P, delta, theta, cap); break;
  default: hipLaunchKernelGGL(hlll_kernel<4>, dim3(grid), dim3(wpb * 64), lds, s, h->P, delta, theta, cap); break;
  }
  HCHK(hipGetLastError());
  HCHK(hipEventRecord(h->ev[1], s));
  HCHK(hipStreamSynchronize(s));
  HCHK(hipEventElapsedTime(&h->last_ms, h->ev[0], h->ev[1]));
  if (status)
    HCHK(hipMemcpy(status, h->P.status, sizeof(int) * B, hipMemcpyDeviceToHost));
  if (info)
    HCHK(hipMemcpy(info, h->P.info, sizeof(int) * 2 * B, hipMemcpyDeviceToHost));
  return FPHIP_OK;
}

// HLLL in a selectable floating-point type (hlll_x.hip): precision 106 = double-double, the device
// stand-in for the reference's FP_NR<dd_real> (BASELINE config 5 as stated); precision 53 = plain
// double with the same tree-sum kernel.  Same algorithm and status / info convention as
// fphip_hh_hlll; the sums run as wave-level trees (the exact-order double kernel is fphip_hh_hlll).
static int hh_hlll_ex(fphip_hh *h, double delta, double theta, int precision, const int *d_only_failed,
                      int *status, int *info)
{
  if (!h || (precision != 53 && precision != 106 && precision != 212))
    return FPHIP_ERROR;
  const size_t B = (size_t)h->P.batch, d = h->P.d, ld = h->P.ldn;
  const bool wide = precision >= 106;  // a low plane of R and V
  if (!h->P.bf)
  {
    HCHK(fphip_dev_alloc((void **)&h->P.bf, B * d * ld * 8 + 4096, fphip_ctx_stream(h->ctx)));
    HCHK(fphip_dev_alloc((void **)&h->P.info, B * 2 * sizeof(int), fphip_ctx_stream(h->ctx)));
    HCHK(hipMemsetAsync(h->P.bf, 0, B * d * ld * 8 + 4096, fphip_ctx_stream(h->ctx)));
  }
  if (!h->xsc)
  {
    HCHK(fphip_dev_alloc((void **)&h->xsc, B * 20 * d * sizeof(double), fphip_ctx_stream(h->ctx)));
    HCHK(fphip_dev_alloc((void **)&h->xprevE, B * d * sizeof(long long), fphip_ctx_stream(h->ctx)));
  }
  if (wide && !h->Rlo)
  {
    HCHK(fphip_dev_alloc((void **)&h->Rlo, B * d * ld * 8 + 4096, fphip_ctx_stream(h->ctx)));
    HCHK(fphip_dev_alloc((void **)&h->Vlo, B * d * ld * 8 + 4096, fphip_ctx_stream(h->ctx)));
  }
  hipStream_t s0 = fphip_ctx_stream(h->ctx);
  HCHK(hipMemsetAsync(h->xsc, 0, B * 20 * d * sizeof(double), s0));
  HCHK(hipMemsetAsync(h->xprevE, 0, B * d * sizeof(long long), s0));
  if (wide)
  {
    HCHK(hipMemsetAsync(h->Rlo, 0, B * d * ld * 8 + 4096, s0));
    HCHK(hipMemsetAsync(h->Vlo, 0, B * d * ld * 8 + 4096, s0));
  }
  if (precision == 212)
  {  // quad-double (the device stand-in for FP_NR<qd_real>, the ladder's third stage): two more planes each
    if (!h->xRx)
    {
      HCHK(fphip_dev_alloc((void **)&h->xRx, 2 * B * d * ld * 8 + 4096, s0));
      HCHK(fphip_dev_alloc((void **)&h->xVx, 2 * B * d * ld * 8 + 4096, s0));
    }
    HCHK(hipMemsetAsync(h->xRx, 0, 2 * B * d * ld * 8 + 4096, s0));
    HCHK(hipMemsetAsync(h->xVx, 0, 2 * B * d * ld * 8 + 4096, s0));
  }
  // the reflectors sixteen at a time (compact WY with a T per block, hlll_x.hip): opt-in, FPHIP_HLLL_BLOCKED=1.
  // Built for the lone-wave latency of config 5 and measured SLOWER there (n = 256, one lattice: 60.0 s against
  // 29.7 s one by one in double, 115.4 s against 56.5 s in double-double; same basis, 146 491 swaps): the chain of
  // dependent reductions is 16 times shorter, but every block waits for its 16 rows of V, its column of T and its
  // signs out of L2, where the one-by-one loop has the next row in flight behind each tree sum.
  const char *hb     = getenv("FPHIP_HLLL_BLOCKED");
  const bool blocked = hb && hb[0] == '1' && precision != 212;
  const size_t tdbl  = B * ((d + 15) / 16) * 256;
  if (blocked)
  {
    if (!h->xThi)
      HCHK(fphip_dev_alloc((void **)&h->xThi, tdbl * 8 + 4096, s0));
    if (precision == 106 && !h->xTlo)
      HCHK(fphip_dev_alloc((void **)&h->xTlo, tdbl * 8 + 4096, s0));
    HCHK(hipMemsetAsync(h->xThi, 0, tdbl * 8, s0));
    if (precision == 106)
      HCHK(hipMemsetAsync(h->xTlo, 0, tdbl * 8, s0));
  }
  HlllX X;
  X.Thi      = blocked ? h->xThi : nullptr;
  X.Tlo      = (blocked && precision == 106) ? h->xTlo : nullptr;
  X.Rlo      = wide ? h->Rlo : nullptr;
  X.Vlo      = wide ? h->Vlo : nullptr;
  X.Rx       = precision == 212 ? h->xRx : nullptr;
  X.Vx       = precision == 212 ? h->xVx : nullptr;
  X.sc       = h->xsc;
  X.prevE    = h->xprevE;
  X.delta    = delta;
  X.theta    = theta;
  X.iter_cap = 1LL << 40;
  X.only_failed = d_only_failed;
  const int nq = (h->P.n + 63) / 64;
  int grid     = h->P.batch;
  if (grid > fphip_ctx_num_cus(h->ctx) * 8)
    grid = fphip_ctx_num_cus(h->ctx) * 8;
  hipStream_t s = fphip_ctx_stream(h->ctx);
  HCHK(hipEventRecord(h->ev[0], s));
  if (precision == 212)
    switch (nq)
    {
    case 1: hipLaunchKernelGGL((hlll_x_kernel<1, QD>), dim3(grid), dim3(64), 0, s, h->P, X); break;
    case 2: hipLaunchKernelGGL((hlll_x_kernel<2, QD>), dim3(grid), dim3(64), 0, s, h->P, X); break;
    case 3: hipLaunchKernelGGL((hlll_x_kernel<3, QD>), dim3(grid), dim3(64), 0, s, h->P, X); break;
    default: hipLaunchKernelGGL((hlll_x_kernel<4, QD>), dim3(grid), dim3(64), 0, s, h->P, X); break;
    }
  else if (precision == 106)
    switch (nq)
    {
    case 1: hipLaunchKernelGGL((hlll_x_kernel<1, DD>), dim3(grid), dim3(64), 0, s, h->P, X); break;
    case 2: hipLaunchKernelGGL((hlll_x_kernel<2, DD>), dim3(grid), dim3(64), 0, s, h->P, X); break;
    case 3: hipLaunchKernelGGL((hlll_x_kernel<3, DD>), dim3(grid), dim3(64), 0, s, h->P, X); break;
    default: hipLaunchKernelGGL((hlll_x_kernel<4, DD>), dim3(grid), dim3(64), 0, s, h->P, X); break;
    }
  else
    switch (nq)
    {
    case 1: hipLaunchKernelGGL((hlll_x_kernel<1, double>), dim3(grid), dim3(64), 0, s, h->P, X); break;
    case 2: hipLaunchKernelGGL((hlll_x_kernel<2, double>), dim3(grid), dim3(64), 0, s, h->P, X); break;
    case 3: hipLaunchKernelGGL((hlll_x_kernel<3, double>), dim3(grid), dim3(64), 0, s, h->P, X); break;
    default: hipLaunchKernelGGL((hlll_x_kernel<4, double>), dim3(grid), dim3(64), 0, s, h->P, X); break;
    }
  HCHK(hipGetLastError());
  HCHK(hipEventRecord(h->ev[1], s));
  HCHK(hipStreamSynchronize(s));
  HCHK(hipEventElapsedTime(&h->last_ms, h->ev[0], h->ev[1]));
  if (status)
    HCHK(hipMemcpy(status, h->P.status, sizeof(int) * B, hipMemcpyDeviceToHost));
  if (info)
    HCHK(hipMemcpy(info, h->P.info, sizeof(int) * 2 * B, hipMemcpyDeviceToHost));
  return FPHIP_OK;
}

extern "C" int fphip_hh_hlll_ex(fphip_hh *h, double delta, double eta, double theta, double c, int precision,
                                int *status, int *info)
{
  FPHIP_RANGE("fphip_hh_hlll_ex");
  (void)eta;
  (void)c;
  return hh_hlll_ex(h, delta, theta, precision, nullptr, status, info);
}

// The precision ladder of the reference's wrapper (hlll_reduction with LM_WRAPPER, wrapper.cpp:
// 478-529: double, then the wider types, each stage continuing from the basis the previous one
// left) on the device: stage 1 = the exact-order double kernel for the whole batch; the lattices it
// gives up on with a precision alarm (RED_HLLL_SR_FAILURE -4, RED_HLLL_NORM_FAILURE -5) go on in
// double-double, and those it gives up on in quad-double (round 6).  stage[batch] (nullable) = 53, 106 or 212:
// where each lattice ended.  A lattice that fails at 212 bits keeps its status: the caller's MPFR stage (fplll's
// CPU path) is next.
extern "C" int fphip_hh_hlll_ladder(fphip_hh *h, double delta, double eta, double theta, double c, int *status,
                                    int *info, int *stage)
{
  FPHIP_RANGE("fphip_hh_hlll_ladder");
  if (!h)
    return FPHIP_ERROR;
  const size_t B = (size_t)h->P.batch;
  std::vector<int> st(B, 0), inf(2 * B, 0);
  int rc = fphip_hh_hlll(h, delta, eta, theta, c, st.data(), inf.data());
  if (rc != FPHIP_OK)
    return rc;
  float ms = h->last_ms;
  std::vector<int> stg(B, 53);
  bool any = false;
  // FPHIP_HLLL_LADDER_TEST=1 (tests only): treat the odd lattices as if stage 1 had raised an alarm, so
  // that the escalation path runs on inputs where plain doubles never fail (row exponents make the
  // double stage very robust: no long-sized lattice tried here trips it)
  const bool force = getenv("FPHIP_HLLL_LADDER_TEST") && atoi(getenv("FPHIP_HLLL_LADDER_TEST")) >= 1;
  if (force)
    for (size_t L = 1; L < B; L += 2)
      if (st[L] == 1)
        st[L] = -4;
  for (size_t L = 0; L < B; ++L)
    any |= (st[L] == -4 || st[L] == -5);
  if (any)
  {
    // status still sits on the device (P.status): 1 = done, anything else goes on in double-double
    std::vector<int> st2(B, 0), inf2(2 * B, 0), mask(B);
    for (size_t L = 0; L < B; ++L)
      mask[L] = (st[L] == -4 || st[L] == -5) ? 0 : 1;
    int *d_mask = nullptr;
    HCHK(fphip_dev_alloc((void **)&d_mask, B * sizeof(int), fphip_ctx_stream(h->ctx)));
    hipError_t e = hipMemcpy(d_mask, mask.data(), B * sizeof(int), hipMemcpyHostToDevice);
    rc           = (e == hipSuccess) ? hh_hlll_ex(h, delta, theta, 106, d_mask, st2.data(), inf2.data()) : FPHIP_ERROR;
    fphip_dev_free(d_mask, fphip_ctx_stream(h->ctx));
    if (rc != FPHIP_OK)
      return rc;
    ms += h->last_ms;
    for (size_t L = 0; L < B; ++L)
      if (!mask[L])
      {
        st[L]  = st2[L];
        stg[L] = 106;
        inf[2 * L] += inf2[2 * L];
        inf[2 * L + 1] += inf2[2 * L + 1];
      }
    // third stage (wrapper.cpp:630-710: FT_QD behind FT_DD): the lattices double-double gave up on, in quad-double
    // (FPHIP_HLLL_LADDER_TEST=2, tests only: every fourth lattice as if stage 2 had raised an alarm)
    const bool force3 = getenv("FPHIP_HLLL_LADDER_TEST") && atoi(getenv("FPHIP_HLLL_LADDER_TEST")) == 2;
    if (force3)
      for (size_t L = 3; L < B; L += 4)
        if (!mask[L] && st[L] == 1)
          st[L] = -4;
    bool any3 = false;
    for (size_t L = 0; L < B; ++L)
      any3 |= (!mask[L] && (st[L] == -4 || st[L] == -5));
    if (any3)
    {
      std::vector<int> st3(B, 0), inf3(2 * B, 0), mask3(B);
      for (size_t L = 0; L < B; ++L)
        mask3[L] = (!mask[L] && (st[L] == -4 || st[L] == -5)) ? 0 : 1;
      int *d_mask3 = nullptr;
      HCHK(fphip_dev_alloc((void **)&d_mask3, B * sizeof(int), fphip_ctx_stream(h->ctx)));
      hipError_t e3 = hipMemcpy(d_mask3, mask3.data(), B * sizeof(int), hipMemcpyHostToDevice);
      rc = (e3 == hipSuccess) ? hh_hlll_ex(h, delta, theta, 212, d_mask3, st3.data(), inf3.data()) : FPHIP_ERROR;
      fphip_dev_free(d_mask3, fphip_ctx_stream(h->ctx));
      if (rc != FPHIP_OK)
        return rc;
      ms += h->last_ms;
      for (size_t L = 0; L < B; ++L)
        if (!mask3[L])
        {
          st[L]  = st3[L];
          stg[L] = 212;
          inf[2 * L] += inf3[2 * L];
          inf[2 * L + 1] += inf3[2 * L + 1];
        }
    }
  }
  h->last_ms = ms;
  if (status)
    memcpy(status, st.data(), B * sizeof(int));
  if (info)
    memcpy(info, inf.data(), 2 * B * sizeof(int));
  if (stage)
    memcpy(stage, stg.data(), B * sizeof(int));
  return FPHIP_OK;
}

// low plane of R after fphip_hh_hlll_ex(precision 106): R(i,j) = hi + lo (hi through fphip_hh_get_R)
extern "C" int fphip_hh_get_R_lo(fphip_hh *h, int lattice, double *Rlo)
{
  if (!h || !Rlo || !h->Rlo || lattice < 0 || lattice >= h->P.batch)
    return FPHIP_ERROR;
  HCHK(hipMemcpy2D(Rlo, (size_t)h->P.n * 8, h->Rlo + (size_t)lattice * h->P.d * h->P.ldn,
                   (size_t)h->P.ldn * 8, (size_t)h->P.n * 8, h->P.d, hipMemcpyDeviceToHost));
  return FPHIP_OK;
}

// double-double arithmetic of the device (ftx.h), element-wise on host arrays — for its unit test
// against multiprecision.  op: 0 add, 1 sub, 2 mul, 3 div, 4 sqrt(a), 5 nint(a)
extern "C" int fphip_debug_dd_op(fphip_ctx *ctx, int op, int count, const double *ahi, const double *alo,
                                 const double *bhi, const double *blo, double *ohi, double *olo)
{
  if (!ctx || count <= 0)
    return FPHIP_ERROR;
  double *dv = nullptr;
  const size_t nb = (size_t)count * sizeof(double);
  if (fphip_dev_alloc((void **)&dv, 6 * nb, fphip_ctx_stream(ctx)) != hipSuccess)
    return FPHIP_ERROR;
  hipMemcpy(dv, ahi, nb, hipMemcpyHostToDevice);
  hipMemcpy(dv + count, alo, nb, hipMemcpyHostToDevice);
  hipMemcpy(dv + 2 * (size_t)count, bhi, nb, hipMemcpyHostToDevice);
  hipMemcpy(dv + 3 * (size_t)count, blo, nb, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(dd_op_kernel, dim3((count + 255) / 256), dim3(256), 0, fphip_ctx_stream(ctx), dv, dv + count,
                     dv + 2 * (size_t)count, dv + 3 * (size_t)count, dv + 4 * (size_t)count,
                     dv + 5 * (size_t)count, op, count);
  hipStreamSynchronize(fphip_ctx_stream(ctx));
  hipMemcpy(ohi, dv + 4 * (size_t)count, nb, hipMemcpyDeviceToHost);
  hipMemcpy(olo, dv + 5 * (size_t)count, nb, hipMemcpyDeviceToHost);
  fphip_dev_free(dv, fphip_ctx_stream(ctx));
  return FPHIP_OK;
}

// R as d×n row-major (only R(i, j<=i) is meaningful, as in the reference); exponents separately
extern "C" int fphip_hh_get_R(fphip_hh *h, int lattice, double *R)
{
  if (!h || !R || lattice < 0 || lattice >= h->P.batch)
    return FPHIP_ERROR;
  HCHK(hipMemcpy2D(R, (size_t)h->P.n * 8, h->P.R + (size_t)lattice * h->P.d * h->P.ldn,
                   (size_t)h->P.ldn * 8, (size_t)h->P.n * 8, h->P.d, hipMemcpyDeviceToHost));
  return FPHIP_OK;
}

extern "C" int fphip_hh_get_row_expo(fphip_hh *h, int lattice, int64_t *row_expo)
{
  if (!h || !row_expo || lattice < 0 || lattice >= h->P.batch)
    return FPHIP_ERROR;
  HCHK(hipMemcpy(row_expo, h->P.rexp + (size_t)lattice * h->P.d, 8 * (size_t)h->P.d,
                 hipMemcpyDeviceToHost));
  return FPHIP_OK;
}

extern "C" double fphip_hh_last_kernel_ms(const fphip_hh *h) { return h ? h->last_ms : 0.0; }
