// pruner_volume.hip — batched even-simplex volumes: the device kernel of the pruner
// and the host loop that computes the same doubles.
//
// V_k(y) (fplll: Pruner::relative_volume, pruner/pruner_simplex.h:6-46) is built by k rounds of
// "integrate the polynomial, evaluate it at y_i / y_k, make minus that value the new constant term",
// i = k-1 .. 0.  Round s (degree s -> s+1) touches every coefficient once, so the integration
// (c_j / (j+1) -> c_{j+1}) and the Horner evaluation (which consumes the NEW coefficients from the top
// down) are ONE descending pass here: read c_j, divide, store as c_{j+1}, multiply-then-add into the
// accumulator.  The operations each value goes through — and their order — are the reference's
// (eval_poly starts from 0 and ends by adding the zero constant term: both kept), so the volumes are
// its doubles on any IEEE machine without contraction (-ffp-contract=off; f64 division and the
// products are correctly rounded on gfx950).
//
// Device mapping: ONE LANE PER JOB (bound vector, k).  The lanes of a wave run the same round structure
// (round s has s + 1 steps whatever k is; a lane is finished after k rounds and idles), the host sorts
// the jobs by k so that a wave holds equal depths, and a lane's polynomial is a COLUMN of LDS
// (c_j at [j][lane]: 64 consecutive doubles per row — conflict-free, no cross-lane traffic at all).
// A gradient of a 60-dimensional block is ~7 400 jobs = 116 waves of at most 465 steps.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dev_mem.h"
#include "pruner_engine.h"

namespace fphip_pruner
{
// ---------------------------------------------------------------------------------------------
// host form
// ---------------------------------------------------------------------------------------------
double simplex_volume(const double *y, int k, double *c)
{
  c[0]             = 1.0;
  const double top = y[k - 1];
  for (int s = 0; s < k; ++s)
  {
    const double x = y[k - 1 - s] / top;
    double acc     = 0.0;
    for (int j = s; j >= 0; --j)
    {
      const double t = c[j] / (j + 1.0);
      c[j + 1]       = t;
      acc            = acc * x;
      acc            = acc + t;
    }
    acc  = acc * x;
    acc  = acc + 0.0;  // (the constant term of an integrated polynomial)
    c[0] = -1.0 * acc;
  }
  return c[0];  // signed and without the k! — see finish()
}

// the sign and the factorial (pruner_simplex.h:44-45) are applied on the host for both engines
extern const double *factorial_table();
static inline double finish(double raw, int k)
{
  const double res = raw * factorial_table()[k];
  return (k & 1) ? -res : res;
}

// ---------------------------------------------------------------------------------------------
// host form, SEVERAL jobs side by side.  One job is a chain: every step's multiply-then-add waits for the
// step before it (8 cycles on a core that could start two of each per cycle), and a candidate of the
// searches is m jobs of depths 1 .. m over the same bounds.  The jobs of a batch are therefore taken in
// groups of eight, deepest first, as the lanes of two 4-wide vectors: the same round structure as the
// device kernel below (round s has s + 1 steps whatever k is; a lane whose job is finished keeps going
// through the motions with x = 0 and does not touch its result), the same operations in the same order
// on every lane, so each lane's double is simplex_volume()'s.  The vectors are the compiler's generic
// ones: 256-bit registers where the CPU has AVX2 (checked at run time), pairs of SSE2 registers elsewhere.
//
// The division.  c_j / (j + 1) is what the divider of the core limits (one 4-wide division per 8 cycles).
// Where the CPU has FMA the quotient by the small integer d = j + 1 is computed WITHOUT the divider and
// still correctly rounded:   q0 = RN(c r),  r = RN(1 / d);   e = c - q0 d  (one FMA, exact);
//                            q  = RN(q0 + e r)               (one FMA)   ==  RN(c / d).
// Why this is the divider's result, for 1 <= d <= 256 and 2^-900 <= |c| <= 2^900 (u = the spacing of the
// doubles at c / d):  |q0 - c/d| <= 2^-52 |c/d| < 2u (1 + 2^-53), so e = d (c/d - q0) is a multiple of u / 2
// below 2^11 of them — exactly representable, the FMA does not round; c / d = q0 + e / d as real numbers and
// |e r - e / d| <= 2^-52 u.  c is a multiple of u (d >= 1), so c / d = (K + 1/2 + t) u with 2 d t an integer:
// c / d is either a double, or at least u / (2 d) >= u / 512 away from every midpoint between two doubles
// (t = 0 would need a 54-bit c) — moving it by 2^-52 u cannot change the way it rounds.  Zeros, infinities,
// NaNs and the subnormal neighbourhood are outside the range: every coefficient of a job descends from a
// constant term c_0 (or the initial 1) by at most k divisions by integers <= k, so with k <= 100 it is
// enough that every c_0 written lies in [2^-300, 2^300] (100! < 2^525); a group in which one does not is
// computed again with the divider.  (tests/test_pruner_properties_cpu.py compares with the one-chain loop
// bit for bit; every pruner fixture of the reference goes through this path in the CPU suite.)
// ---------------------------------------------------------------------------------------------
typedef double v4d __attribute__((vector_size(32)));
typedef long long v4l __attribute__((vector_size(32)));
typedef double v8d __attribute__((vector_size(64)));

struct LaneWork
{
  std::vector<v8d> c;  // the polynomials, [kmax + 2] rows of eight lanes (read as two v4d; 64-byte aligned)
  std::vector<v8d> x;  // the evaluation points of round s, [kmax] rows
};

struct DivisorTables
{
  double d[257], r[257];  // d and RN(1 / d), d = 0 .. 256 (entry 0 unused)
  DivisorTables()
  {
    d[0] = r[0] = 0.0;
    for (int i = 1; i <= 256; ++i)
    {
      d[i] = (double)i;
      r[i] = 1.0 / (double)i;
    }
  }
};
static const DivisorTables &divisor_tables()
{
  static const DivisorTables t;
  return t;
}
template <class VD, int LW> __attribute__((always_inline)) static inline VD splat(double v)
{
  VD r;
  for (int l = 0; l < LW; ++l)
    r[l] = v;
  return r;
}

// Eight lanes as NV vectors of LW lanes (VD / VL: the vector of doubles / of 64-bit integers); FMA: the
// quotient by the FMA correction above.  Returns false when a constant term left the range in which that
// quotient is proven (the caller repeats the group with FMA = false).
template <class VD, class VL, int LW, int NV, bool FMA>
__attribute__((always_inline)) static inline bool volumes_body(const double *const *y, const int *k, void *cbuf,
                                                               void *xbuf, double *raw)
{
  VD *c = (VD *)cbuf, *x = (VD *)xbuf;
  const int kmax    = k[0];  // (the caller sorts: lane 0 is the deepest)
  const DivisorTables &dt = divisor_tables();
  const VD zero     = {};
  const VL izero    = {};
  VL kv[NV];
  VD top[NV];
  for (int h = 0; h < NV; ++h)
    for (int l = 0; l < LW; ++l)
    {
      kv[h][l]  = k[LW * h + l];
      top[h][l] = y[LW * h + l][k[LW * h + l] - 1];
    }
  for (int s = 0; s < kmax; ++s)
    for (int h = 0; h < NV; ++h)
    {  // y / top of the live lanes (the one-chain loop's division), 0 / top on the finished ones
      VD num = zero;
      for (int l = 0; l < LW; ++l)
      {
        const int kl = k[LW * h + l];
        num[l]       = s < kl ? y[LW * h + l][kl - 1 - s] : 0.0;
      }
      x[NV * s + h] = num / top[h];
    }
  const VD one = zero + 1.0, mone = zero - 1.0, lo = zero + 0x1p-300, hi = zero + 0x1p300;
  const VL absmask = izero + 0x7fffffffffffffffLL;
  VL good          = izero - 1;
  for (int h = 0; h < NV; ++h)
    c[h] = one;
  for (int s = 0; s < kmax; ++s)
  {
    VD xs[NV], a[NV];
    for (int h = 0; h < NV; ++h)
    {
      xs[h] = x[NV * s + h];
      a[h]  = zero;
    }
    for (int j = s; j >= 0; --j)
    {
      const VD dv = splat<VD, LW>(dt.d[j + 1]);
      VD t[NV];
      if (FMA)
      {
        const VD rv = splat<VD, LW>(dt.r[j + 1]);
        for (int h = 0; h < NV; ++h)
        {
          const VD cj = c[NV * j + h];
          const VD q0 = cj * rv;
          const VD e  = __builtin_elementwise_fma(-q0, dv, cj);
          t[h]        = __builtin_elementwise_fma(e, rv, q0);
        }
      }
      else
      {
        for (int h = 0; h < NV; ++h)
          t[h] = c[NV * j + h] / dv;
      }
      for (int h = 0; h < NV; ++h)
      {
        c[NV * (j + 1) + h] = t[h];
        a[h]                = a[h] * xs[h];
      }
      for (int h = 0; h < NV; ++h)
        a[h] = a[h] + t[h];
    }
    const VL sv = izero + (long long)s;
    for (int h = 0; h < NV; ++h)
    {
      a[h] = a[h] * xs[h];
      a[h] = a[h] + zero;
      a[h] = mone * a[h];
      const VL on = sv < kv[h];
      if (FMA)
      {  // (a finished lane's a[h] is not written and does not count)
        const VD mag = (VD)((VL)a[h] & absmask);
        good &= ((mag >= lo) & (mag <= hi)) | ~on;
      }
      c[h] = (VD)(((VL)a[h] & on) | ((VL)c[h] & ~on));
    }
  }
  long long all = -1;
  for (int h = 0; h < NV; ++h)
    for (int l = 0; l < LW; ++l)
      raw[LW * h + l] = c[h][l];
  for (int l = 0; l < LW; ++l)
    all &= good[l];
  return !FMA || all != 0;
}
typedef bool (*volumes8_fn)(const double *const *, const int *, void *, void *, double *);
__attribute__((target("avx2,fma"))) static bool volumes8_fma(const double *const *y, const int *k, void *c, void *x,
                                                            double *raw)
{
  return volumes_body<v4d, v4l, 4, 2, true>(y, k, c, x, raw);
}
__attribute__((target("avx2"))) static bool volumes8_avx2(const double *const *y, const int *k, void *c, void *x,
                                                          double *raw)
{
  return volumes_body<v4d, v4l, 4, 2, false>(y, k, c, x, raw);
}
static bool volumes8_generic(const double *const *y, const int *k, void *c, void *x, double *raw)
{
  return volumes_body<v4d, v4l, 4, 2, false>(y, k, c, x, raw);
}

// FPHIP_PRUNER_HOST_MODE: 0 one job after the other; 1 eight lanes, the divider; 2 (default) eight lanes and
// the FMA quotient where the CPU has it (sixteen lanes were tried: no faster — with the divider out of the way
// a step of eight lanes is about as long as the chain)
static int host_mode()
{
  static const int mode = []
  {
    int v = getenv("FPHIP_PRUNER_HOST_MODE") ? atoi(getenv("FPHIP_PRUNER_HOST_MODE")) : 2;
    if (getenv("FPHIP_PRUNER_HOST_SCALAR") && atoi(getenv("FPHIP_PRUNER_HOST_SCALAR")) != 0)
      v = 0;
    return v;
  }();
  return mode;
}

// out[j] = the finished volume of jobs[j], for every job of the batch
static void host_volumes(const double *bounds, int m, const VolumeJob *jobs, int njobs, double *out, LaneWork &L,
                         std::vector<unsigned> &order, std::vector<double> &scratch)
{
  static const bool avx2 = __builtin_cpu_supports("avx2");
  static const bool fma  = avx2 && __builtin_cpu_supports("fma");
  static const volumes8_fn quick_fn = volumes8_fma;
  static const volumes8_fn exact_fn = avx2 ? volumes8_avx2 : volumes8_generic;
  const int mode         = host_mode();
  if ((int)scratch.size() < m + 2)
    scratch.resize(m + 2);
  if (njobs < 3 || mode == 0)
  {  // (one or two chains: nothing to put side by side)
    for (int j = 0; j < njobs; ++j)
      out[j] = finish(simplex_volume(bounds + (size_t)jobs[j].vec * m, jobs[j].k, scratch.data()), jobs[j].k);
    return;
  }
  // counting sort of the job indices by k, deepest first: the lanes of a group then finish together
  order.resize(njobs);
  {
    std::vector<int> head(m + 2, 0);
    for (int j = 0; j < njobs; ++j)
      ++head[m - jobs[j].k + 1];
    for (int k = 1; k <= m + 1; ++k)
      head[k] += head[k - 1];
    for (int j = 0; j < njobs; ++j)
      order[head[m - jobs[j].k]++] = (unsigned)j;
  }
  if (L.c.size() < (size_t)(m + 2))
  {
    L.c.resize((size_t)(m + 2));
    L.x.resize((size_t)(m + 2));
  }
  for (int g = 0; g < njobs;)
  {
    const int left = njobs - g;
    if (left < 3)
    {
      for (int l = 0; l < left; ++l)
      {
        const VolumeJob &J = jobs[order[g + l]];
        out[order[g + l]]  = finish(simplex_volume(bounds + (size_t)J.vec * m, J.k, scratch.data()), J.k);
      }
      break;
    }
    const bool quick = fma && mode >= 2 && jobs[order[g]].k <= 100;
    const int cnt    = std::min(8, left);
    const double *y[8];
    int k[8];
    double raw[8];
    for (int l = 0; l < 8; ++l)
    {  // (a short last group repeats its first job on the idle lanes)
      const VolumeJob &J = jobs[order[g + (l < cnt ? l : 0)]];
      y[l]               = bounds + (size_t)J.vec * m;
      k[l]               = J.k;
    }
    if (!(quick && quick_fn(y, k, L.c.data(), L.x.data(), raw)))
      exact_fn(y, k, L.c.data(), L.x.data(), raw);
    for (int l = 0; l < cnt; ++l)
      out[order[g + l]] = finish(raw[l], k[l]);
    g += cnt;
  }
}

namespace
{
class HostEngine : public VolumeEngine
{
public:
  bool run(const double *bounds, int nvec, int m, const VolumeJob *jobs, int njobs, double *out) override
  {
    (void)nvec;
    // (the one host engine serves every caller: per-thread work areas)
    static thread_local LaneWork L;
    static thread_local std::vector<unsigned> order;
    static thread_local std::vector<double> scratch;
    host_volumes(bounds, m, jobs, njobs, out, L, order, scratch);
    host_jobs += (unsigned long long)njobs;
    return true;
  }
  // 1: the host pays for every wasted candidate.  FPHIP_PRUNER_HOST_LOOKAHEAD (tests): a larger value makes
  // the searches take their batched / look-ahead paths on host arithmetic — the paths the device engine
  // takes — so that the CPU suite covers them
  int lookahead() const override
  {
    static const int la = getenv("FPHIP_PRUNER_HOST_LOOKAHEAD") ? atoi(getenv("FPHIP_PRUNER_HOST_LOOKAHEAD")) : 1;
    return la > 1 ? la : 1;
  }
};
}  // namespace

VolumeEngine *host_volume_engine()
{
  static HostEngine e;  // (stateless but for its job counter)
  return &e;
}

// ---------------------------------------------------------------------------------------------
// device form
// ---------------------------------------------------------------------------------------------
// jobs[] is sorted by k descending; job = vec | k << 20.  One wave per workgroup; LDS: (m + 1) rows of
// 64 doubles.
__global__ __launch_bounds__(64) void pruner_volume_kernel(const double *__restrict__ bounds, int m,
                                                           const unsigned *__restrict__ jobs, int njobs,
                                                           double *__restrict__ raw)
{
  extern __shared__ double lds[];
  const int lane  = threadIdx.x;
  const int idx   = blockIdx.x * 64 + lane;
  const bool live = idx < njobs;
  const unsigned w = live ? jobs[idx] : 0u;
  const int k      = (int)(w >> 20);
  const double *y  = bounds + (size_t)(w & 0xfffffu) * m;
  const int kmax   = __builtin_amdgcn_readfirstlane(k);  // lane 0 holds the deepest job of the wave
  double *c        = lds + lane;
  c[0]             = 1.0;
  const double top = live ? y[k - 1] : 1.0;
  for (int s = 0; s < kmax; ++s)
  {
    const bool on  = s < k;
    const double x = on ? y[k - 1 - s] / top : 0.0;
    double acc     = 0.0;
    for (int j = s; j >= 0; --j)
    {
      const double t  = c[j * 64] / (double)(j + 1);
      c[(j + 1) * 64] = t;
      acc             = acc * x;
      acc             = acc + t;
    }
    acc = acc * x;
    acc = acc + 0.0;
    if (on)
      c[0] = -1.0 * acc;
  }
  if (live)
    raw[idx] = c[0];
}

namespace
{
class DeviceEngine : public VolumeEngine
{
public:
  int device         = 0;
  hipStream_t stream = nullptr;
  char *pin          = nullptr;  // pinned staging: bounds | jobs | results
  size_t pin_bytes   = 0;
  char *dev          = nullptr;
  size_t dev_bytes   = 0;
  // batches below this many polynomial steps (sum of k (k + 1) / 2 over the jobs) are cheaper inline than
  // a launch with its two copies (~35 us, and a lone wave needs ~0.1 us per step of its longest job): one
  // candidate of a 36-dimensional block is 2 000 steps = 6 us of host arithmetic, the gradient batch of the
  // same block 140 000.  Measured (MI355X box, prune() of a 60-dimensional block): host loop 19.5 ms,
  // every batch on the device 11.0 ms (1 440 launches), with this threshold 5.9 ms (30 launches)
  long long min_steps = 16000;
  int lds_opt_in     = 0;
  char err[256]      = {0};
  std::vector<double> scratch;
  std::vector<unsigned> order;
  LaneWork lanes;

  ~DeviceEngine() override
  {
    if (stream)
    {
      (void)hipSetDevice(device);
      (void)hipStreamSynchronize(stream);
      fphip_dev_free(dev, stream);
      (void)hipStreamSynchronize(stream);
      (void)hipStreamDestroy(stream);
    }
    if (pin)
      fphip_pinned_put(pin);
  }
  const char *error() const override { return err; }
  int lookahead() const override { return 32; }

  bool fail(const char *what, hipError_t e)
  {
    snprintf(err, sizeof err, "pruner volume engine: %s: %s", what, hipGetErrorString(e));
    return false;
  }

  bool run(const double *bounds, int nvec, int m, const VolumeJob *jobs, int njobs, double *out) override
  {
    if (njobs <= 0)
      return true;
    long long steps = 0;
    for (int j = 0; j < njobs; ++j)
      steps += (long long)jobs[j].k * (jobs[j].k + 1) / 2;
    if (steps < min_steps || nvec >= (1 << 20) || m > 255)
    {
      host_volumes(bounds, m, jobs, njobs, out, lanes, order, scratch);
      host_jobs += (unsigned long long)njobs;
      return true;
    }
    // counting sort of the job indices by k, deepest first
    order.resize(njobs);
    {
      std::vector<int> head(m + 2, 0);
      for (int j = 0; j < njobs; ++j)
        ++head[m - jobs[j].k + 1];
      for (int k = 1; k <= m + 1; ++k)
        head[k] += head[k - 1];
      for (int j = 0; j < njobs; ++j)
        order[head[m - jobs[j].k]++] = (unsigned)j;
    }
    const size_t b_bytes = (size_t)nvec * m * sizeof(double);
    const size_t j_bytes = ((size_t)njobs * sizeof(unsigned) + 63) & ~(size_t)63;
    const size_t o_bytes = (size_t)njobs * sizeof(double);
    const size_t need    = b_bytes + j_bytes + o_bytes;
    hipError_t e         = hipSetDevice(device);
    if (e != hipSuccess)
      return fail("hipSetDevice", e);
    if (need > pin_bytes)
    {
      if (pin)
        fphip_pinned_put(pin);
      pin_bytes = std::max(need * 2, (size_t)1 << 20);
      pin       = (char *)fphip_pinned_get(pin_bytes);
      if (!pin)
      {
        pin_bytes = 0;
        snprintf(err, sizeof err, "pruner volume engine: no pinned memory");
        return false;
      }
    }
    if (need > dev_bytes)
    {
      fphip_dev_free(dev, stream);
      dev       = nullptr;
      dev_bytes = std::max(need * 2, (size_t)1 << 20);
      if ((e = fphip_dev_alloc((void **)&dev, dev_bytes, stream)) != hipSuccess)
      {
        dev_bytes = 0;
        return fail("device allocation", e);
      }
    }
    memcpy(pin, bounds, b_bytes);
    unsigned *pj = (unsigned *)(pin + b_bytes);
    for (int j = 0; j < njobs; ++j)
      pj[j] = (unsigned)jobs[order[j]].vec | ((unsigned)jobs[order[j]].k << 20);
    const size_t lds = (size_t)(m + 1) * 64 * sizeof(double);
    if (lds > 64 * 1024 && !lds_opt_in)
    {
      if ((e = hipFuncSetAttribute((const void *)pruner_volume_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024)) != hipSuccess)
        return fail("hipFuncSetAttribute", e);
      lds_opt_in = 1;
    }
    if ((e = hipMemcpyAsync(dev, pin, b_bytes + j_bytes, hipMemcpyHostToDevice, stream)) != hipSuccess)
      return fail("upload", e);
    double *d_raw = (double *)(dev + b_bytes + j_bytes);
    hipLaunchKernelGGL(pruner_volume_kernel, dim3((njobs + 63) / 64), dim3(64), lds, stream, (const double *)dev, m,
                       (const unsigned *)(dev + b_bytes), njobs, d_raw);
    if ((e = hipGetLastError()) != hipSuccess)
      return fail("launch", e);
    double *praw = (double *)(pin + b_bytes + j_bytes);
    if ((e = hipMemcpyAsync(praw, d_raw, o_bytes, hipMemcpyDeviceToHost, stream)) != hipSuccess)
      return fail("download", e);
    if ((e = hipStreamSynchronize(stream)) != hipSuccess)
      return fail("synchronize", e);
    for (int j = 0; j < njobs; ++j)
      out[order[j]] = finish(praw[j], jobs[order[j]].k);
    device_jobs += (unsigned long long)njobs;
    ++launches;
    return true;
  }
};
}  // namespace

VolumeEngine *create_device_volume_engine(int device, char *err, size_t errlen)
{
  DeviceEngine *e = new DeviceEngine;
  e->device       = device;
  hipError_t rc   = hipSetDevice(device);
  if (rc == hipSuccess)
  {
    // HIGH priority: a queue pool of its own — an engine's kernels must never queue behind the persistent
    // schedule kernel whose mailbox they are answering (enum_host.hip: "Streams and hardware queues")
    int least = 0, greatest = 0;
    rc = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (rc == hipSuccess)
      rc = hipStreamCreateWithPriority(&e->stream, hipStreamNonBlocking, greatest);
  }
  if (rc != hipSuccess)
  {
    if (err)
      snprintf(err, errlen, "pruner volume engine: %s", hipGetErrorString(rc));
    e->stream = nullptr;
    delete e;
    return nullptr;
  }
  if (const char *v = getenv("FPHIP_PRUNER_MIN_DEVICE_STEPS"))
    e->min_steps = atoll(v);
  return e;
}

void destroy_volume_engine(VolumeEngine *e)
{
  if (e && e != host_volume_engine())
    delete e;
}
}  // namespace fphip_pruner
