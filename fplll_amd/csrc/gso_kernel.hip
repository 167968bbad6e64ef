// gso_kernel.hip — batched floating-point Gram-Schmidt + size-reduction sweep for gfx950.
//
// Reference behaviour reproduced, bit for bit, for MatGSO<Z_NR<long>, FP_NR<double>> with
// GSO_ROW_EXPO (the BKZ fast path, fplll/bkz.cpp:816-829):
//   MatGSOInterface::update_gso_row   fplll/gso_interface.cpp:131-164
//   MatGSO::get_gram + dot_product    fplll/gso.h:314-331, fplll/nr/numvect.h:386-396
//   LLLReduction::babai               fplll/lll.cpp:166-224
//   LLLReduction::size_reduction      fplll/lll.h:107-122
//   MatGSO::row_addmul_we / update_bf fplll/gso.cpp:236-262, 24-48 ; row_op_end gso_interface.cpp:32-53
//
// Design (MI355X-first): the path is HBM-bound (≈0.25 flop/B), the working set of ONE lattice
// is cache-sized, and the algorithm is a long dependent chain (O(d^2) steps per sweep).  So the
// data-parallel axis is the BATCH of independent lattices: ONE WAVEFRONT OWNS ONE LATTICE, no
// LDS, no barriers, thousands of waves in flight hide the HBM latency of each other's chains.
// Inside a wave the 64 lanes are the vector axis of every inner loop, arranged so that each
// floating-point value is produced by exactly the reference's operation sequence:
//   * Gram row g(kappa,j): lane j walks the columns c = 0..n-1 in order (res = res + a*b, two
//     roundings) — bf is stored COLUMN-major so the 64 lanes read 512 contiguous bytes per c.
//   * GSO recurrence r(kappa,j) = g - sum_{k<j} mu(j,k) r(kappa,k): column-oriented forward
//     substitution — at step k the (now final) r(kappa,k) is broadcast with v_readlane and every
//     lane j>k subtracts mu(j,k)*r(kappa,k); each lane sees k = 0,1,2,… in the reference's
//     order.  mu is kept transposed as well (muT) so that step k reads one contiguous column.
//   * size-reduction sweep: lane k holds babai_mu[k]; j runs downwards, X_j is rounded from the
//     broadcast babai_mu[j] and lanes k<j subtract X_j*mu(j,k) (row j of mu: contiguous).
//   * integer AXPY b_kappa += sum_j lx_j b_j: lanes are columns; 64-bit wrapping arithmetic.
// d and n up to 64*NQ are handled with NQ registers per lane (template parameter).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (no FMA contraction).

#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include "gso_device.h"

namespace fphip
{

__device__ __forceinline__ double g_rl_f64(double v, int lane)
{
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ long long g_rl_i64(long long v, int lane)
{
  unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, lane);
  unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), lane);
  return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int wave_max_i32(int v)
{
  for (int off = 32; off > 0; off >>= 1)
    v = max(v, __shfl_xor(v, off));
  return __builtin_amdgcn_readfirstlane(v);
}
// FP_NR<double>::exponent(), nr_FP_d.inl:44 (glibc: ilogb(0) = INT_MIN)
__device__ __forceinline__ long long fexponent(double x)
{
  return (x == 0.0) ? ((long long)INT_MIN + 1) : ((long long)ilogb(x) + 1);
}

template <int NQ> struct Lattice
{
  int d, n, row_expo_on;
  long long *b;
  double *bfT, *mu, *muT, *r, *rdg;
  long long *rexp;
  int lane;
  double murow[NQ];  // mu(kappa, j) of the row last updated (lane j)
};

// ---------------------------------------------------------------------------------------------
// Software-pipelined "chain" loops.  Every hot loop of this kernel has the shape
//     for step s (in a fixed order):  v[q] = LOAD(s, lane);  state[q] = f(state[q], v[q], scalar_s)
// where the loads do not depend on the state but scalar_s does (it is read from a lane of the
// state with v_readlane).  The loads are issued U steps at a time, double-buffered (group g+1 is
// in flight while group g is consumed), so a wave keeps ~2·U·(active chunks) 512-byte loads in
// flight — that, times the waves per CU, is what hides HBM latency.  Steps are grouped inside a
// 64-chunk so that the chunk index of the scalar's register is a compile-time constant.
// ---------------------------------------------------------------------------------------------
#ifndef FPHIP_GSO_U
#define FPHIP_GSO_U 4
#endif

template <int NQ, int U> struct Grp
{
  double v[U][NQ];
};

// update_gso_row(kappa, last) recomputed from column 0 (identical values: every input is unchanged
// since the row was invalidated).  Returns false on a non-finite mu (RED_GSO_FAILURE).
template <int NQ> __device__ bool update_row(Lattice<NQ> &T, int kappa, int last)
{
  constexpr int U = FPHIP_GSO_U;
  const int d = T.d, n = T.n, lane = T.lane;
  const int qact = (last >> 6) + 1;  // chunks holding a lane j <= last
  double bk[NQ], acc[NQ], rd[NQ];
  unsigned jc[NQ];  // lane's row index per chunk, clamped into the slab for unconditional loads
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    const int c = lane + 64 * q;
    bk[q]       = (c < n) ? T.bfT[(size_t)c * d + kappa] : 0.0;
    acc[q]      = 0.0;
    rd[q]       = (c < kappa) ? T.rdg[c] : 1.0;
    jc[q]       = (unsigned)min(c, d - 1);
  }
  // ---- Gram row: g(kappa,j) = bf_kappa . bf_j, columns in ascending order (numvect.h:386-396)
#pragma unroll
  for (int cq = 0; cq < NQ; ++cq)
  {
    const int cbase = cq * 64;
    if (cbase >= n)
      break;
    const int cnt = min(64, n - cbase);
    auto load     = [&](Grp<NQ, U> &G, int g0)
    {
#pragma unroll
      for (int u = 0; u < U; ++u)
      {
        const int c = min(cbase + g0 + u, n - 1);
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          if (q < qact)
            G.v[u][q] = (T.bfT + (size_t)c * d)[jc[q]];
      }
    };
    auto proc = [&](const Grp<NQ, U> &G, int g0)
    {
#pragma unroll
      for (int u = 0; u < U; ++u)
      {
        if (g0 + u < cnt)
        {
          const double bkc = g_rl_f64(bk[cq], g0 + u);
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            if (q < qact)
            {
              const double p = bkc * G.v[u][q];
              acc[q]         = (cbase + g0 + u == 0) ? p : acc[q] + p;
            }
        }
      }
    };
    Grp<NQ, U> A, B;
    load(A, 0);
#pragma unroll 1
    for (int g0 = 0; g0 < cnt; g0 += 2 * U)
    {
      if (g0 + U < cnt)
        load(B, g0 + U);
      proc(A, g0);
      if (g0 + U < cnt)
      {
        if (g0 + 2 * U < cnt)
          load(A, g0 + 2 * U);
        proc(B, g0 + U);
      }
    }
  }
  // ---- recurrence, gso_interface.cpp:143-158, column-oriented
  bool ok = true;
#pragma unroll
  for (int kq = 0; kq < NQ; ++kq)
  {
    const int kbase = kq * 64;
    if (kbase > last)
      break;
    const int cnt = min(64, last - kbase + 1);
    auto load     = [&](Grp<NQ, U> &G, int g0)
    {
#pragma unroll
      for (int u = 0; u < U; ++u)
      {
        const int k = min(kbase + g0 + u, d - 1);
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          if (q < qact && q >= kq)  // rows j <= k hold zeros: chunks below kq are never needed
            G.v[u][q] = (T.muT + (size_t)k * d)[jc[q]];
      }
    };
    auto proc = [&](const Grp<NQ, U> &G, int g0)
    {
#pragma unroll
      for (int u = 0; u < U; ++u)
      {
        if (g0 + u < cnt)
        {
          const int k     = kbase + g0 + u;
          const double rk = g_rl_f64(acc[kq], g0 + u);  // r(kappa,k) is final
          double muk      = 0.0;
          if (k < kappa)
          {
            muk = rk / g_rl_f64(rd[kq], g0 + u);  // mu(kappa,k) = r(kappa,k) / r(k,k)
            if (!isfinite(muk))
              ok = false;
          }
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            if (q < qact && q >= kq)
            {
              const int j = lane + 64 * q;
              if (j > k && j <= last)
              {
                const double m = (j == kappa) ? muk : G.v[u][q];
                acc[q]         = acc[q] - m * rk;
              }
            }
        }
      }
    };
    Grp<NQ, U> A, B;
    load(A, 0);
#pragma unroll 1
    for (int g0 = 0; g0 < cnt; g0 += 2 * U)
    {
      if (g0 + U < cnt)
        load(B, g0 + U);
      proc(A, g0);
      if (g0 + U < cnt)
      {
        if (g0 + 2 * U < cnt)
          load(A, g0 + 2 * U);
        proc(B, g0 + U);
      }
    }
  }
  // ---- store the row
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    const int j = lane + 64 * q;
    T.murow[q]  = 0.0;
    if (j <= last)
    {
      T.r[(size_t)kappa * d + j] = acc[q];
      if (j < kappa)
      {
        const double m               = acc[q] / rd[q];
        T.murow[q]                   = m;
        T.mu[(size_t)kappa * d + j]  = m;
        T.muT[(size_t)j * d + kappa] = m;
      }
      else
      {
        T.rdg[kappa] = acc[q];
      }
    }
  }
  return __all(ok);
}

// LLLReduction::babai(kappa, kappa, 0).  1 ok, 0 GSO failure, -1 babai failure, -2 multiplier.
template <int NQ> __device__ int babai(Lattice<NQ> &T, int kappa, double eta)
{
  constexpr int U = FPHIP_GSO_U;
  const int d = T.d, n = T.n, lane = T.lane;
  const int qact = ((kappa - 1) >> 6) + 1;  // chunks holding a row index < kappa
  const int nq_c = ((n - 1) >> 6) + 1;      // chunks holding a column index < n
  long long max_expo = LLONG_MAX;
  unsigned jc[NQ], cc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    jc[q] = (unsigned)min(lane + 64 * q, d - 1);
    cc[q] = (unsigned)min(lane + 64 * q, n - 1);
  }
  for (int iter = 0;; ++iter)
  {
    if (!update_row<NQ>(T, kappa, kappa - 1))
      return 0;
    const long long rexpk = T.rexp[kappa];
    int e[NQ];
    bool need = false;
    int mexp  = INT_MIN;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      const int j = lane + 64 * q;
      e[q]        = 0;
      if (j < kappa)
      {
        e[q]           = (int)(rexpk - T.rexp[j]);
        const double f = fabs(ldexp(T.murow[q], e[q]));  // get_mu, gso_interface.h:694-702
        need |= (f > eta);
        const long long ex = (long long)e[q] + fexponent(T.murow[q]);
        mexp               = max(mexp, (int)max(ex, (long long)INT_MIN + 2));
      }
    }
    if (!__any(need))
      break;
    if (iter >= 2)
    {  // lll.cpp:187-195
      const long long new_max = (long long)wave_max_i32(mexp);
      if (new_max > max_expo - 5)
        return -1;
      max_expo = new_max;
    }
    double bm[NQ];
    long long xl[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      bm[q] = T.murow[q];
      xl[q] = 0;
    }
    bool too_big = false;
    // ---- lll.cpp:202-220, j = kappa-1 … 0 (descending), lane k owns babai_mu[k]
#pragma unroll
    for (int jq = NQ - 1; jq >= 0; --jq)
    {
      const int jbase = jq * 64;
      if (jbase >= kappa)
        continue;
      const int top = min(63, kappa - 1 - jbase);  // first (highest) jj of this chunk
      const int cnt = top + 1;                     // steps: jj = top, top-1, …, 0
      auto load     = [&](Grp<NQ, U> &G, int g0)
      {
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
          const int j = jbase + max(top - (g0 + u), 0);
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            if (q <= jq)  // mu(j,k) = 0 for k >= j
              G.v[u][q] = (T.mu + (size_t)j * d)[jc[q]];
        }
      };
      auto proc = [&](const Grp<NQ, U> &G, int g0)
      {
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
          if (g0 + u < cnt)
          {
            const int jj     = top - (g0 + u);
            const int j      = jbase + jj;
            const double bmj = g_rl_f64(bm[jq], jj);
            const int ej     = __builtin_amdgcn_readlane(e[jq], jj);
            double X;  // rnd_we, nr_FP_d.inl:226-233
            if (fexponent(bmj) + ej >= 53)
              X = bmj;
            else
              X = ldexp(rint(ldexp(bmj, ej)), -ej);
            if (X != 0.0)
            {
              {  // row_addmul_we(kappa, j, -X, ej): get_si_exp_we, nr_FP_d.inl:46-53
                const long long ex = fexponent(-X) + ej - 63;
                if (ex > 0)
                  too_big = true;
                const long long lx = (long long)ldexp(-X, ej);
                xl[jq]             = (lane == jj) ? lx : xl[jq];
              }
#pragma unroll
              for (int q = 0; q < NQ; ++q)
                if (q <= jq)
                {
                  const int k = lane + 64 * q;
                  if (k < j)
                  {
                    const double t = X * G.v[u][q];
                    bm[q]          = bm[q] - t;
                  }
                }
            }
          }
        }
      };
      Grp<NQ, U> A, B;
      load(A, 0);
  #pragma unroll 1
    for (int g0 = 0; g0 < cnt; g0 += 2 * U)
      {
        if (g0 + U < cnt)
          load(B, g0 + U);
        proc(A, g0);
        if (g0 + U < cnt)
        {
          if (g0 + 2 * U < cnt)
            load(A, g0 + 2 * U);
          proc(B, g0 + U);
        }
      }
    }
    if (too_big)
      return -2;
    // ---- integer AXPY on row kappa (row_add / row_sub / row_addmul_si, gso.cpp:84-158)
    long long bv[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      const int c = lane + 64 * q;
      bv[q]       = (c < n) ? T.b[(size_t)kappa * n + c] : 0;
    }
#pragma unroll
    for (int jq = NQ - 1; jq >= 0; --jq)
    {
      const int jbase = jq * 64;
      if (jbase >= kappa)
        continue;
      const int top = min(63, kappa - 1 - jbase);
      const int cnt = top + 1;
      struct GI
      {
        long long v[U][NQ];
      };
      auto load = [&](GI &G, int g0)
      {
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
          const int j = jbase + max(top - (g0 + u), 0);
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            if (q < nq_c)
              G.v[u][q] = (T.b + (size_t)j * n)[cc[q]];
        }
      };
      auto proc = [&](const GI &G, int g0)
      {
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
          if (g0 + u < cnt)
          {
            const int jj       = top - (g0 + u);
            const long long lx = g_rl_i64(xl[jq], jj);
            if (lx != 0)
            {
#pragma unroll
              for (int q = 0; q < NQ; ++q)
                if (q < nq_c)
                  bv[q] = (long long)((unsigned long long)bv[q] +
                                      (unsigned long long)G.v[u][q] * (unsigned long long)lx);
            }
          }
        }
      };
      GI A, B;
      load(A, 0);
  #pragma unroll 1
    for (int g0 = 0; g0 < cnt; g0 += 2 * U)
      {
        if (g0 + U < cnt)
          load(B, g0 + U);
        proc(A, g0);
        if (g0 + U < cnt)
        {
          if (g0 + 2 * U < cnt)
            load(A, g0 + 2 * U);
          proc(B, g0 + U);
        }
      }
    }
    // ---- row_op_end: update_bf(kappa), gso.cpp:24-48
    int ce[NQ];
    double cm[NQ];
    int emax = INT_MIN;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      const int c = lane + 64 * q;
      ce[q]       = INT_MIN;
      cm[q]       = 0.0;
      if (c < n)
      {
        T.b[(size_t)kappa * n + c] = bv[q];
        if (T.row_expo_on)
        {
          int ex;
          cm[q] = frexp((double)bv[q], &ex);
          ce[q] = ex;
          emax  = max(emax, ex);
        }
        else
        {
          cm[q] = (double)bv[q];
          ce[q] = 0;
          emax  = 0;
        }
      }
    }
    emax = wave_max_i32(emax);
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      const int c = lane + 64 * q;
      if (c < n)
        T.bfT[(size_t)c * d + kappa] = T.row_expo_on ? ldexp(cm[q], ce[q] - emax) : cm[q];
    }
    if (lane == 0)
      T.rexp[kappa] = T.row_expo_on ? (long long)emax : 0;
    // later reads of b / bfT / rexp in this wave must see these stores
    __threadfence_block();
  }
  return 1;
}

// mode 0: update_gso() (every row, no size reduction); mode 1: size_reduction(kmin,kend);
// mode 2: (re)build bfT / row_expo from b for every row (after a basis upload)
template <int NQ>
__global__ void __launch_bounds__(256)
    gso_sweep_kernel(GsoBatch P, int kmin, int kend, double eta, int mode)
{
  const int lane = threadIdx.x & 63;
  const int wpb  = blockDim.x >> 6;
  const int wave = threadIdx.x >> 6;
  for (int L = blockIdx.x * wpb + wave; L < P.batch; L += gridDim.x * wpb)
  {
    Lattice<NQ> T;
    T.d           = P.d;
    T.n           = P.n;
    T.row_expo_on = P.row_expo;
    T.lane        = lane;
    T.b           = P.b + (size_t)L * P.d * P.n;
    T.bfT         = P.bfT + (size_t)L * P.n * P.d;
    T.mu          = P.mu + (size_t)L * P.d * P.d;
    T.muT         = P.muT + (size_t)L * P.d * P.d;
    T.r           = P.r + (size_t)L * P.d * P.d;
    T.rdg         = P.rdg + (size_t)L * P.d;
    T.rexp        = P.rexp + (size_t)L * P.d;
    int status    = 1;
    if (mode == 2)
    {
      for (int i = 0; i < P.d; ++i)
      {
        int ce[NQ];
        double cm[NQ];
        int emax = INT_MIN;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
        {
          const int c = lane + 64 * q;
          ce[q]       = INT_MIN;
          cm[q]       = 0.0;
          if (c < P.n)
          {
            const long long v = T.b[(size_t)i * P.n + c];
            if (P.row_expo)
            {
              int ex;
              cm[q] = frexp((double)v, &ex);
              ce[q] = ex;
              emax  = max(emax, ex);
            }
            else
            {
              cm[q] = (double)v;
              ce[q] = 0;
              emax  = 0;
            }
          }
        }
        emax = wave_max_i32(emax);
#pragma unroll
        for (int q = 0; q < NQ; ++q)
        {
          const int c = lane + 64 * q;
          if (c < P.n)
            T.bfT[(size_t)c * P.d + i] = P.row_expo ? ldexp(cm[q], ce[q] - emax) : cm[q];
        }
        if (lane == 0)
          T.rexp[i] = P.row_expo ? (long long)emax : 0;
      }
    }
    else
    {
      for (int kappa = kmin; kappa < kend; ++kappa)
      {
        if (mode == 1 && kappa > 0)
        {
          const int rc = babai<NQ>(T, kappa, eta);
          if (rc != 1)
          {
            status = rc;
            break;
          }
        }
        if (!update_row<NQ>(T, kappa, kappa))
        {
          status = 0;
          break;
        }
        __threadfence_block();
      }
    }
    if (lane == 0)
      P.status[L] = status;
  }
}

template __global__ void gso_sweep_kernel<1>(GsoBatch, int, int, double, int);
template __global__ void gso_sweep_kernel<2>(GsoBatch, int, int, double, int);
template __global__ void gso_sweep_kernel<3>(GsoBatch, int, int, double, int);
template __global__ void gso_sweep_kernel<4>(GsoBatch, int, int, double, int);

}  // namespace fphip
