// lll_wave.h — the LLL machinery of one lattice owned by one wavefront, shared by lll_kernel.hip and
// bkz_kernel.hip (device code only; the design notes are in lll_kernel.hip).
#ifndef FPHIP_LLL_WAVE_H
#define FPHIP_LLL_WAVE_H
#include <type_traits>

#include "gso_wave.h"
#include "lll_stream.h"

namespace fphip
{

// The streaming machinery of the slot-mode kernels: the block streams of lll_stream.h, or (build with
// -DFPHIP_LLL_STREAM=0: the A/B and fallback build) the first generation's ring of single rows.
#ifndef FPHIP_LLL_STREAM
#define FPHIP_LLL_STREAM 1
#endif
// a Gram row with at most this many unknown entries is completed entry by entry (update_row_cached)
#ifndef FPHIP_LLL_GRAM_SINGLES
#define FPHIP_LLL_GRAM_SINGLES 4
#endif
#if FPHIP_LLL_STREAM
template <int NQ> struct ReduceRing : LStream<NQ>
{
};
#else
template <int NQ> struct ReduceRing : Ring<NQ, (NQ + 1) / 2, FPHIP_RING_REDUCE>
{
  using Base = Ring<NQ, (NQ + 1) / 2, FPHIP_RING_REDUCE>;
  static constexpr int BYTES = Base::R * Base::SLOT;
  __device__ __forceinline__ void init(int wave, int lane_)
  {
    this->base  = (unsigned)(wave * BYTES);
    this->lane  = lane_;
    this->head  = 0;
    this->tail  = 0;
    this->ahead = 0;
  }
};
#endif
static_assert(ReduceRing<1>::BYTES == 16384 && ReduceRing<2>::BYTES == 16384 || !FPHIP_LLL_STREAM, "fphip_reduce_ring_bytes");
static_assert(ReduceRing<3>::BYTES == 15360 && ReduceRing<4>::BYTES == 16384 || !FPHIP_LLL_STREAM, "fphip_reduce_ring_bytes");

struct LllCtx
{
  double *gf;  // [d][ldd] symmetric Gram cache indexed by slots, NaN = unknown
  int *vc;     // [d] valid columns of the row in each slot
};

// every row of lattice L carries the 2^24 flag the host's re-float pass (mode 2 of the sweep kernel) has just
// written, and the narrow paths are not switched off (FPHIP_GSO_NARROW=0): the slot-mode kernels may then
// stream the float mirror of bf in their Gram passes (Lattice::f32ok)
template <int NQ> __device__ __forceinline__ int all_rows_narrow(const GsoBatch &P, size_t L, int lane)
{
  const int *fl = P.narrow + L * (size_t)P.d;
  bool ok       = P.use_narrow != 0;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    const int i = lane + 64 * q;
    if (i < P.d)
      ok &= (fl[i] != 0);
  }
  return __all(ok) ? 1 : 0;
}

__device__ __forceinline__ double make_nan() { return __longlong_as_double(0x7ff8000000000000LL); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <int NQ> __device__ __forceinline__ double lane_get(const double (&v)[NQ], int idx)
{
  double r = 0.0;
  dispatch_chunk<NQ>(idx, [&](auto q, int ii) { r = g_rl_f64(v[decltype(q)::value], ii); });
  return r;
}

// Z_NR<long>::exponent(), nr/nr_Z_l.inl:30-48 (MAX_LONG_FAST = 2^53 on LP64, defs.h:134)
__device__ __forceinline__ int zexponent(long long v)
{
  int e;
  const double f = frexp((double)v, &e);
  if ((double)v > 0x1p53 && fabs(f) == 0.5)
  {
    const unsigned long long y = (unsigned long long)(v < 0 ? -v : v);
    return 64 - __clzll((long long)y);
  }
  return e;
}

// update_gso_row(kappa, last) from the first invalid column of the row (gso_interface.cpp:131-164).
// Leaves mu(kappa, .) / r(kappa, .) of columns <= last in T.murow / T.rrow (lane = column).
template <int NQ, int IPS, int RR>
__device__ __forceinline__ bool update_row_cached(Lattice<NQ> &T, LllCtx &C, const SlotMap<NQ> &M,
                                  Ring<NQ, IPS, RR> &ring, int kappa, int last)
{
  const int n = T.n, lane = T.lane, ldd = T.ldd;
  const int sk    = M.phys(kappa);
  const int start = uni(C.vc[sk]);
  double *rrowp   = T.r + (size_t)sk * ldd;
  double *murowp  = T.mu + (size_t)sk * ldd;
  double *gfrow   = C.gf + (size_t)sk * ldd;
  if (start > last)
  {  // nothing to compute: bring the row into registers
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      const int j = lane + 64 * q;
      T.rrow[q]   = (j < start && j <= kappa) ? rrowp[j] : 0.0;
      T.murow[q]  = (j < start && j < kappa) ? murowp[j] : 0.0;
    }
    return true;
  }
  double acc[NQ], rd[NQ], mold[NQ];
  unsigned off[NQ];
  bool miss = false;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    const int j = lane + 64 * q;
    acc[q]      = 0.0;
    rd[q]       = 1.0;
    mold[q]     = 0.0;
    off[q]      = (unsigned)M.sl[q] * 8u;
    if (j < start)
    {
      acc[q]  = rrowp[j];  // final r(kappa,j)
      mold[q] = (j < kappa) ? murowp[j] : 0.0;
    }
    else if (j <= last)
    {
      acc[q] = gfrow[M.sl[q]];  // cached g(kappa,j)
      miss |= (acc[q] != acc[q]);
    }
    if (j < kappa && j <= last)
      rd[q] = T.rdg[M.sl[q]];
  }
  // rows are gathered by slot: only slots up to the largest one among positions <= last are needed
  // (the slots of positions 0..kappa_max are a permutation of 0..kappa_max: moves stay inside the
  // processed prefix), so the DMA window of a streamed row ends there
  int hi_slot = 0;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
    if (lane + 64 * q <= last)
      hi_slot = max(hi_slot, M.sl[q]);
  const int row_bytes = min((wave_max_i32(hi_slot) + 1) * 8, ldd * 8);
  if (__any(miss))
  {
    // ---- Gram row: g(kappa,j) = bf_kappa . bf_j, columns ascending (numvect.h:386-396)
    double bk[NQ], g[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      const int c = lane + 64 * q;
      bk[q]       = (c < n) ? T.bfT[(size_t)c * ldd + sk] : 0.0;
      g[q]        = 0.0;
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      settle(bk[q]);
      settle(acc[q]);
      settle(rd[q]);
      settle(mold[q]);
    }
    auto gram_body = [&](int c, const double(&v)[NQ])
    {
      const double bkc = lane_get<NQ>(bk, c);
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const double p = bkc * v[q];
        g[q]           = (c == 0) ? p : g[q] + p;
      }
    };
    ring.reset();
    if (T.f32ok)
    {  // every row is below 2^24: the float mirror holds the same numbers in half the bytes
      unsigned off4[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        off4[q] = off[q] >> 1;
      auto gram_row4 = [&](int c) { return RowDesc{T.bfT32 + (size_t)c * ldd, 0, row_bytes >> 1}; };
      ring.run_with(n, gram_row4, gram_body, 0, gram_row4, typename Ring<NQ, IPS, RR>::GatherF32Fetch{off4});
    }
    else
    {
      auto gram_row = [&](int c) { return RowDesc{T.bfT + (size_t)c * ldd, 0, row_bytes}; };
      ring.run_with(n, gram_row, gram_body, 0, gram_row, typename Ring<NQ, IPS, RR>::GatherFetch{off});
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      const int j = lane + 64 * q;
      if (j >= start && j <= last && acc[q] != acc[q])
      {
        acc[q]                                = g[q];
        gfrow[M.sl[q]]                        = g[q];
        C.gf[(size_t)M.sl[q] * ldd + sk]      = g[q];
      }
    }
  }
  if (start == 0 || last - start >= 3)
  {
    // ---- column-oriented recurrence over k = 0..last-1 (gso_interface.cpp:143-158): lanes
    //      j >= max(start, k+1) subtract mu(j,k) r(kappa,k); columns of muT are gathered by slot
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      settle(acc[q]);
      settle(rd[q]);
      settle(mold[q]);
    }
    auto rec_row = [&](int k) { return RowDesc{T.muT + (size_t)k * ldd, 0, row_bytes}; };
    ring.reset();
    ring.run_with(
        last, rec_row,
        [&](int k, const double(&v)[NQ])
        {
          dispatch_chunk<NQ>(k,
                             [&](auto kq_, int kk)
                             {
                               constexpr int kq = decltype(kq_)::value;
                               const double rk  = g_rl_f64(acc[kq], kk);  // r(kappa,k) is final
                               double muk       = 0.0;
                               if (last == kappa)
                                 muk = rk / g_rl_f64(rd[kq], kk);  // mu(kappa,k)
#pragma unroll
                               for (int q = kq; q < NQ; ++q)
                               {
                                 const int j   = lane + 64 * q;
                                 const bool on = j <= last && j >= start && (q > kq || lane > kk);
                                 if (on)
                                 {
                                   const double m = (j == kappa) ? muk : v[q];
                                   acc[q]         = acc[q] - m * rk;
                                 }
                               }
                             });
        },
        0, rec_row, typename Ring<NQ, IPS, RR>::GatherFetch{off});
  }
  else
  {
    // ---- a few new columns: row-oriented, the reference's own loop order.  r(kappa,j) =
    //      g(kappa,j) - sum_{k<j} mu(j,k) r(kappa,k), k ascending: products are formed in parallel
    //      (lane k), the subtraction chain runs over v_readlane
    for (int j = start; j <= last; ++j)
    {
      const double *mj = T.mu + (size_t)M.phys(j) * ldd;
      double p[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int k = lane + 64 * q;
        double mm   = 0.0;
        if (k < j)
          mm = (j == kappa) ? acc[q] / rd[q] : mj[k];
        p[q] = mm * acc[q];
      }
      double s = lane_get<NQ>(acc, j);
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int hi = min(j - 64 * q, 64);
        for (int kk = 0; kk < hi; ++kk)
          s = s - g_rl_f64(p[q], kk);
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        acc[q] = (lane + 64 * q == j) ? s : acc[q];
    }
  }
  // ---- store the new columns
  bool ok = true;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    const int j = lane + 64 * q;
    T.murow[q]  = 0.0;
    T.rrow[q]   = (j <= last) ? acc[q] : 0.0;
    if (j <= last)
    {
      if (j < start)
      {
        T.murow[q] = mold[q];
      }
      else if (j < kappa)
      {
        const double m = acc[q] / rd[q];  // mu(kappa,j) = r(kappa,j) / r(j,j)
        if (!isfinite(m))
          ok = false;
        T.murow[q]                  = m;
        rrowp[j]                    = acc[q];
        murowp[j]                   = m;
        T.muT[(size_t)j * ldd + sk] = m;
      }
      else
      {  // j == kappa
        rrowp[j]  = acc[q];
        T.rdg[sk] = acc[q];
      }
    }
  }
  if (lane == 0)
    C.vc[sk] = last + 1;
  return __all(ok);
}

// ---- the same on the block streams of lll_stream.h (round 5) ------------------------------------------
template <int NQ>
__device__ __forceinline__ bool update_row_cached(Lattice<NQ> &T, LllCtx &C, const SlotMap<NQ> &M,
                                                  LStream<NQ> &S, int kappa, int last)
{
  const int n = T.n, lane = T.lane, ldd = T.ldd, ldn = T.ldn;
  const int sk    = M.phys(kappa);
  const int start = uni(C.vc[sk]);
  double *rrowp   = T.r + (size_t)sk * ldd;
  double *murowp  = T.mu + (size_t)sk * ldd;
  double *gfrow   = C.gf + (size_t)sk * ldd;
  if (start > last)
  {  // nothing to compute: bring the row into registers
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      const int j = lane + 64 * q;
      T.rrow[q]   = (j < start && j <= kappa) ? rrowp[j] : 0.0;
      T.murow[q]  = (j < start && j < kappa) ? murowp[j] : 0.0;
    }
    return true;
  }
  double acc[NQ], rd[NQ], mold[NQ];
  bool miss = false;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    const int j = lane + 64 * q;
    acc[q]      = 0.0;
    rd[q]       = 1.0;
    mold[q]     = 0.0;
    if (j < start)
    {
      acc[q]  = rrowp[j];  // final r(kappa,j)
      mold[q] = (j < kappa) ? murowp[j] : 0.0;
    }
    else if (j <= last)
    {
      acc[q] = gfrow[M.sl[q]];  // cached g(kappa,j)
      miss |= (acc[q] != acc[q]);
    }
    if (j < kappa && j <= last)
      rd[q] = T.rdg[M.sl[q]];
  }
  // rows are gathered by slot: only slots up to the largest one among positions <= last are needed
  int hi_slot = 0;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
    if (lane + 64 * q <= last)
      hi_slot = max(hi_slot, M.sl[q]);
  hi_slot             = wave_max_i32(hi_slot);
  const int row_bytes = min((hi_slot + 1) * 8, ldd * 8);
  int nmiss           = 0;
  unsigned long long missb[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    missb[q] = __ballot(lane + 64 * q >= start && lane + 64 * q <= last && acc[q] != acc[q]);
    nmiss += __builtin_popcountll(missb[q]);
  }
  if (nmiss > 0)
  {
    // bf(kappa, c) for lane c from the contiguous integer row: bf = b 2^-row_expo exactly (update_bf,
    // gso.cpp:24-48: mantissa and exponent of every entry, renormalised to the row's largest exponent) —
    // the doubles store_row_and_refloat wrote into column sk of bfT, from 8 cache lines instead of n
    double bk[NQ];
    {
      const int ek = T.row_expo_on ? (int)T.rexp[sk] : 0;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int c = lane + 64 * q;
        bk[q]       = (c < n) ? ldexp((double)T.b[(size_t)sk * ldn + c], -ek) : 0.0;
      }
    }
    if (nmiss <= FPHIP_LLL_GRAM_SINGLES)
    {
      // ---- a few entries only (the vectors that changed since this row was last here): each one its own
      //      ordered dot product, columns ascending (numvect.h:386-396), from the two integer rows
      RowCursor<NQ> cur;
      const unsigned long long pt0 = LStream<NQ>::now();
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        cur.m[q] = missb[q];
      for (int t = 0; t < nmiss; ++t)
      {
        const int j  = cur.next();
        const int sj = M.phys(j);
        const int ej = T.row_expo_on ? (int)T.rexp[sj] : 0;
        double pr[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
        {
          const int c    = lane + 64 * q;
          const double w = (c < n) ? ldexp((double)T.b[(size_t)sj * ldn + c], -ej) : 0.0;
          pr[q]          = bk[q] * w;
        }
        const double gj = seq_sum<NQ>(pr, 0, n);
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          if (lane + 64 * q == j)
          {
            acc[q]                     = gj;
            gfrow[sj]                  = gj;
            C.gf[(size_t)sj * ldd + sk] = gj;
          }
      }
      S.prof_add(LS_SINGLE, (unsigned long long)nmiss, 0, LStream<NQ>::now() - pt0);
    }
    else
    {
    // ---- Gram row: g(kappa,j) = bf_kappa . bf_j, columns ascending (numvect.h:386-396).  The pass also
    //      yields g(kappa,kappa) in the lane of kappa (the Lovasz test asks for it next, lll.cpp:110): the
    //      window and the chunks are extended to position kappa when that entry is unknown
    const double gkk_old = gfrow[sk];
    const bool want_diag = uni((gkk_old != gkk_old) ? 1 : 0) != 0 && last == kappa - 1;
    const int glast      = want_diag ? kappa : last;
    const int ghi        = want_diag ? max(hi_slot, sk) : hi_slot;
    const int grow_bytes = min((ghi + 1) * 8, ldd * 8);
    const int gqact      = (glast >> 6) + 1;
    double g[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      g[q] = -0.0;  // -0.0 + p == p for every p: the first product starts the sum
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      settle(bk[q]);
      settle(acc[q]);
      settle(rd[q]);
      settle(mold[q]);
    }
    if (T.f32ok)
    {  // every row is below 2^24: the float mirror holds the same numbers in half the bytes
      GramPh<NQ, true> ph{(const char *)T.bfT32, (long)ldd * 4, ls_make_win(grow_bytes >> 1), n, 0, S.lane16, {}, g, bk, gqact};
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        ph.off[q] = (unsigned)M.sl[q] * 4u;
      ls_run<NQ>(S, ph, n);
    }
    else
    {
      GramPh<NQ, false> ph{(const char *)T.bfT, (long)ldd * 8, ls_make_win(grow_bytes), n, 0, S.lane16, {}, g, bk, gqact};
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        ph.off[q] = (unsigned)M.sl[q] * 8u;
      ls_run<NQ>(S, ph, n);
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      const int j = lane + 64 * q;
      if (j >= start && j <= last && acc[q] != acc[q])
      {
        acc[q]                           = g[q];
        gfrow[M.sl[q]]                   = g[q];
        C.gf[(size_t)M.sl[q] * ldd + sk] = g[q];
      }
      if (want_diag && j == kappa)
        gfrow[sk] = g[q];
    }
    }
  }
  const int qact = (last >> 6) + 1;
  (void)qact;
  if (start == 0 || last - start >= 3)
  {
    // ---- column-oriented recurrence over k = 0..last-1 (gso_interface.cpp:143-158)
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      settle(acc[q]);
      settle(rd[q]);
      settle(mold[q]);
    }
    RecPh<NQ> ph{(const char *)T.muT, (long)ldd * 8, ls_make_win(row_bytes), last, 0, S.lane16, lane, {}, acc, rd, {}, last == kappa, kappa};
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      const int j = lane + 64 * q;
      ph.off[q]   = (unsigned)M.sl[q] * 8u;
      ph.bmask[q] = __ballot(j <= last && j >= start);
    }
    ls_run<NQ>(S, ph, last);
  }
  else
  {
    // ---- a few new columns: row-oriented, the reference's own loop order
    for (int j = start; j <= last; ++j)
    {
      const double *mj = T.mu + (size_t)M.phys(j) * ldd;
      double p[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int k = lane + 64 * q;
        double mm   = 0.0;
        if (k < j)
          mm = (j == kappa) ? acc[q] / rd[q] : mj[k];
        p[q] = mm * acc[q];
      }
      double s = lane_get<NQ>(acc, j);
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int hi = min(j - 64 * q, 64);
        for (int kk = 0; kk < hi; ++kk)
          s = s - g_rl_f64(p[q], kk);
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        acc[q] = (lane + 64 * q == j) ? s : acc[q];
    }
  }
  // ---- store the new columns
  bool ok = true;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    const int j = lane + 64 * q;
    T.murow[q]  = 0.0;
    T.rrow[q]   = (j <= last) ? acc[q] : 0.0;
    if (j <= last)
    {
      if (j < start)
      {
        T.murow[q] = mold[q];
      }
      else if (j < kappa)
      {
        const double m = acc[q] / rd[q];  // mu(kappa,j) = r(kappa,j) / r(j,j)
        if (!isfinite(m))
          ok = false;
        T.murow[q]                  = m;
        rrowp[j]                    = acc[q];
        murowp[j]                   = m;
        T.muT[(size_t)j * ldd + sk] = m;
      }
      else
      {  // j == kappa
        rrowp[j]  = acc[q];
        T.rdg[sk] = acc[q];
      }
    }
  }
  if (lane == 0)
    C.vc[sk] = last + 1;
  return __all(ok);
}

// LLLReduction::babai(kappa, kappa, size_reduction_start), lll.cpp:166-224, on the block streams.
// 1 ok, 0 GSO failure, -1 babai failure, -2 multiplier beyond 63 bits.
// sr_end = size_reduction_end (-1: kappa): row kappa is reduced against the rows [sr_start, sr_end) only — early
// reduction (lll.h:125-140) calls it for the rows behind kappa with sr_end = kappa.
template <int NQ, class Upd, class After>
__device__ __forceinline__ int babai_impl(Lattice<NQ> &T, LStream<NQ> &S, int kappa, double eta,
                                          const SlotMap<NQ> &map, Upd upd, After after, int sr_start = 0,
                                          int sr_end = -1)
{
  const int pk = map.phys(kappa);  // physical slot of row kappa
  const int n = T.n, lane = T.lane, ldd = T.ldd, ldn = T.ldn;
  const int send = sr_end < 0 ? kappa : sr_end;
  long long max_expo = LLONG_MAX;
  for (int iter = 0;; ++iter)
  {
    if (!upd(kappa, send - 1))
      return 0;
    const long long rexpk = T.rexp[pk];
    int e[NQ];
    bool need = false;
    int mexp  = INT_MIN;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      const int j = lane + 64 * q;
      e[q]        = 0;
      if (j < send)
      {
        e[q]           = (int)(rexpk - T.rexp[map.sl[q]]);
        const double f = fabs(ldexp(T.murow[q], e[q]));  // get_mu, gso_interface.h:694-702
        need |= (j >= sr_start) && (f > eta);
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
    double bm[NQ], xs[NQ];
    unsigned long long nz[NQ];
    long long bv[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      const int c = lane + 64 * q;
      bm[q]       = T.murow[q];
      xs[q]       = 0.0;
      nz[q]       = 0;
      bv[q]       = (c < n) ? T.b[(size_t)pk * ldn + c] : 0;
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      settle(e[q]);
      settle(bv[q]);
      settle(bm[q]);
    }
    // ---- lll.cpp:202-214: lane k owns babai_mu[k]; rows j = kappa-1 .. sr_start, descending
    {
      constexpr int U = LStream<NQ>::U;
      const int jtop  = (send - 1) | (U - 1);
      SweepPh<NQ> ph{(const char *)T.mu, (long)ldd * 8, jtop, send, sr_start, 0, S.lane16, lane, map, bm, xs, e, nz, {}};
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        ph.srmask[q] = __ballot(lane + 64 * q >= sr_start);
      ls_run<NQ>(S, ph, jtop - sr_start + 1);
    }
    // ---- the multipliers: row_addmul_we(kappa, j, -X, e_j) -> get_si_exp_we, nr_FP_d.inl:46-53
    long long lxv[NQ];
    bool too_big = false, big32 = false;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      lxv[q] = 0;
      if (xs[q] != 0.0)
      {
        if (fexponent(-xs[q]) + e[q] - 63 > 0)
          too_big = true;
        lxv[q] = (long long)ldexp(-xs[q], e[q]);
        big32 |= (lxv[q] != (long long)(int)lxv[q]);
      }
    }
    if (__any(too_big))
      return -2;  // nothing has been stored yet: the basis is unchanged
    // ---- integer row operation on row kappa (row_add / row_sub / row_addmul_si, gso.cpp:84-158)
    {
      int rows = 0;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        rows += __builtin_popcountll(nz[q]);
      if (T.f32ok && !__any(big32))
      {
        AxpyPh<NQ, true> ph{(const char *)T.b, (long)ldn * 8, ls_make_win(n * 8), S.lane16, lane, map, bv, lxv, {}, {}};
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          ph.ic.m[q] = ph.cc.m[q] = nz[q];
        ls_run<NQ>(S, ph, rows);
      }
      else
      {
        AxpyPh<NQ, false> ph{(const char *)T.b, (long)ldn * 8, ls_make_win(n * 8), S.lane16, lane, map, bv, lxv, {}, {}};
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          ph.ic.m[q] = ph.cc.m[q] = nz[q];
        ls_run<NQ>(S, ph, rows);
      }
    }
    // ---- row_op_end: update_bf(kappa), gso.cpp:24-48
    store_row_and_refloat<NQ, false>(T, pk, bv);
    if (T.u != nullptr)
    {  // enable_transform (gso.cpp:88-91,111-114,134-137,...): the same operation on the rows of u
      const int d = T.d;
      long long uv[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int c = lane + 64 * q;
        uv[q]       = (c < d) ? T.u[(size_t)pk * ldd + c] : 0;
      }
      int rows = 0;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        rows += __builtin_popcountll(nz[q]);
      AxpyPh<NQ, false> ph{(const char *)T.u, (long)ldd * 8, ls_make_win(d * 8), S.lane16, lane, map, uv, lxv, {}, {}};
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        ph.ic.m[q] = ph.cc.m[q] = nz[q];
      ls_run<NQ>(S, ph, rows);
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int c = lane + 64 * q;
        if (c < d)
          T.u[(size_t)pk * ldd + c] = uv[q];
      }
    }
    after(kappa);
    // later reads of b / bfT / rexp in this wave must see these stores
    __threadfence_block();
  }
  return 1;
}

// row_op_end(kappa, kappa+1) after b_kappa changed (gso_interface.cpp:32-53): the vector's Gram row
// and column, its own GSO row, and columns >= kappa of every later row become invalid
template <int NQ>
__device__ __forceinline__ void after_rowop(Lattice<NQ> &T, LllCtx &C, const SlotMap<NQ> &M, int kappa)
{
  const int lane = T.lane, ldd = T.ldd, d = T.d;
  const int sk     = M.phys(kappa);
  const double nan = make_nan();
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    const int t = lane + 64 * q;
    if (t < ldd)
      C.gf[(size_t)sk * ldd + t] = nan;
    if (t < d)
      C.gf[(size_t)t * ldd + sk] = nan;
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    const int p = lane + 64 * q;
    if (p == kappa)
      C.vc[sk] = 0;
    else if (p > kappa && p < d)
    {
      const int s = M.sl[q];
      if (C.vc[s] > kappa)
        C.vc[s] = kappa;
    }
  }
}

// rows at positions >= from keep only their columns < from (invalidate_gso_row(i, from) for all i)
template <int NQ> __device__ __forceinline__ void clamp_valid(Lattice<NQ> &T, LllCtx &C, const SlotMap<NQ> &M, int from)
{
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    const int p = T.lane + 64 * q;
    if (p >= from && p < T.d)
    {
      const int s = M.sl[q];
      if (C.vc[s] > from)
        C.vc[s] = from;
    }
  }
}

// move_row(kold, knew), knew < kold: the row at kold goes to knew, rows knew..kold-1 shift up by one
template <int NQ> __device__ __forceinline__ void rotate_right(SlotMap<NQ> &M, int knew, int kold, int lane)
{
  const int skold = M.phys(kold);
  int up[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    up[q] = __shfl_up(M.sl[q], 1);
    if (q > 0)
    {
      const int carry = __builtin_amdgcn_readlane(M.sl[q > 0 ? q - 1 : 0], 63);
      up[q]           = (lane == 0) ? carry : up[q];
    }
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    const int p = lane + 64 * q;
    if (p == knew)
      M.sl[q] = skold;
    else if (p > knew && p <= kold)
      M.sl[q] = up[q];
  }
}

// move_row(a, b), a < b: the row at a goes to b, rows a+1..b shift down by one
template <int NQ> __device__ __forceinline__ void rotate_left(SlotMap<NQ> &M, int a, int b, int lane)
{
  const int sa = M.phys(a);
  int dn[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    dn[q] = __shfl_down(M.sl[q], 1);
    if (q + 1 < NQ)
    {
      const int carry = __builtin_amdgcn_readlane(M.sl[q + 1 < NQ ? q + 1 : q], 0);
      dn[q]           = (lane == 63) ? carry : dn[q];
    }
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    const int p = lane + 64 * q;
    if (p == b)
      M.sl[q] = sa;
    else if (p >= a && p < b)
      M.sl[q] = dn[q];
  }
}

// A fresh MatGSO: nothing known (gso.h:33 ctor + size_increased); rows sit in their own slots.
template <int NQ>
__device__ __forceinline__ void lll_init_state(Lattice<NQ> &T, LllCtx &C, SlotMap<NQ> &M)
{
  const int lane = T.lane, d = T.d, ldd = T.ldd;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
    M.sl[q] = lane + 64 * q;
  const double nan = make_nan();
  for (int i = 0; i < d; ++i)
  {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      if (lane + 64 * q < ldd)
        C.gf[(size_t)i * ldd + lane + 64 * q] = nan;
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q)
    if (lane + 64 * q < d)
      C.vc[lane + 64 * q] = 0;
  __threadfence_block();
}

// LLLReduction::lll(kmin, kstart, kend, 0), lll.cpp:44-164, on the cached state (T, C, M).
// status: 1 RED_SUCCESS, 0 RED_GSO_FAILURE, -1 RED_BABAI_FAILURE, -2 multiplier beyond 63 bits,
//         -3 RED_LLL_FAILURE (iteration limit, lll.cpp:159-160)
template <int NQ, class RingT>
__device__ __forceinline__ int lll_run(Lattice<NQ> &T, LllCtx &C, SlotMap<NQ> &M, RingT &ring,
                                       int kmin, int kstart, int kend, double delta, double eta,
                                       double logdelta, int &final_kappa, int &nswaps, int &zeros,
                                       long long &iter, int &vp, bool siegel = false, bool early = false,
                                       int *early_red = nullptr)
{
  // early (LLL_EARLY_RED, lll.cpp:35,84-99, lll.h:125-140): early reduction on; *early_red is then the LLLReduction
  // object's last_early_red, in and out (always the address of a caller's local: a pointer that may be null keeps
  // the variable in private memory, and what is loaded from there is a vector value)
  // siegel (LLL_SIEGEL, lll.cpp:38-40,122,134): `delta` is then the caller's swap_threshold = delta - eta^2 and
  // the two tests compare with lovasz_tests[kappa] instead of [kappa - 1]; logdelta stays log(delta)
  // vp ("verified prefix"): rows 0..vp-1 are known to be a fixed point of this loop — babai is a
  // no-op on each, Lovasz holds between neighbours, r(k,k) is set — and nothing they depend on has
  // changed since.  The reference would walk them again without effect (lll.cpp:82-155 with
  // kappa_start = 0, as BKZ calls it for every block); the walk resumes behind them instead and
  // the skipped iterations are added to the iteration count.  Any change at row position p lowers
  // vp to p.  A caller that keeps no state across calls passes vp = 0.
  const int lane = T.lane, d = T.d, n = T.n, ldd = T.ldd, ldn = T.ldn;
  // ---- iteration limit, lll.cpp:79-80: Matrix::get_max_exp over b.  With row exponents every
  //      row's maximum is its row_expo (MatGSO::update_bf), except for |entries| >= 2^53 where
  //      Z_NR<long>::exponent() can be one less than frexp's: scan the matrix only then
  int mexp = 0;
  {
    int rmax = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      if (lane + 64 * q < d)
        rmax = max(rmax, (int)T.rexp[lane + 64 * q]);
    mexp = wave_max_i32(rmax);
  }
  if (!T.row_expo_on || mexp >= 53)
  {
    mexp = 0;
    for (int i = 0; i < d; ++i)
    {
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int c = lane + 64 * q;
        if (c < n)
          mexp = max(mexp, zexponent(T.b[(size_t)i * ldn + c]));
      }
    }
    mexp = wave_max_i32(mexp);
  }
  const int dd = kend - kmin;
  const long long max_iter =
      (long long)((double)dd - (double)(2 * dd * (dd + 1)) * ((double)(mexp + 3) / logdelta));
  __threadfence_block();

  auto upd = [&](int k, int last) { return update_row_cached(T, C, M, ring, k, last); };
  auto after = [&](int k)
  {
    after_rowop<NQ>(T, C, M, k);
    vp = min(vp, k);
  };
  auto row_is_zero = [&](int s)
  {
    bool nz = false;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      const int c = lane + 64 * q;
      if (c < n)
        nz |= (T.b[(size_t)s * ldn + c] != 0);
    }
    return !__any(nz);
  };

  int status = 1;
  final_kappa = 0;
  zeros       = 0;
  nswaps      = 0;
  iter        = 0;
  bool ok     = true;
  // zero rows go to the end, lll.cpp:68-71
  for (; zeros < dd && row_is_zero(M.phys(0)); ++zeros)
  {
    rotate_left<NQ>(M, kmin, kend - 1 - zeros, lane);
    vp = min(vp, kmin);
  }
  // rows <= kstart are verified: nothing to do (not with early reduction: the skipped iterations would skip the
  // kappa values that trigger it)
  const bool resume = kmin == 0 && vp > kstart + 1 && !early;
  if (zeros < dd && !resume)
  {
    // the reference expects rows below kappa_start to be valid already; on a fresh GSO that is
    // update_gso_row(i) for each of them (a no-op for rows whose cache is valid)
    for (int i = 0; i < kstart && ok; ++i)
    {
      ok = upd(i, i);
      __threadfence_block();
    }
    if (!ok)
      status = 0;
    if (ok && kstart > 0)
    {
      const int rc = babai_impl(T, ring, kstart, eta, M, upd, after);
      if (rc != 1)
      {
        status = rc;
        ok     = false;
      }
    }
    if (ok && !upd(kstart, kstart))
    {
      status = 0;
      ok     = false;
    }
    if (!ok)
      final_kappa = kstart;
    __threadfence_block();
  }
  int kappa = kstart + 1;
  if (resume && ok)
  {
    kappa = max(kappa, min(vp, kend - zeros));
    iter  = kappa - (kstart + 1);  // the no-op iterations the reference spends on rows < kappa
  }
  int kappa_max = 0;
  for (; ok && iter < max_iter && kappa < kend - zeros; ++iter)
  {
    // ---- early reduction, lll.cpp:84-99 / lll.h:125-140: when kappa reaches a new maximum that is a power of two,
    //      every row from kappa on is size-reduced against the rows below kappa (babai(i, kappa)) before the lazy
    //      size reduction of row kappa itself.  (The reference locks n_known_cols meanwhile and forgets the rows it
    //      discovered for this; what it recomputes for them later are the same numbers from the same vectors — the
    //      cache keeps them.)  One call site of babai_impl serves both: er_i walks the rows of the early pass.
    int er_i = -1;
    if constexpr (std::is_base_of<LStream<NQ>, RingT>::value)
    {
      if (early && kappa > kappa_max)
      {
        kappa_max = kappa;
        if ((kappa & (kappa - 1)) == 0 && kappa > uni(*early_red))
          er_i = kappa;
      }
    }
    int rc;
    for (;;)
    {
      // ---- lazy size reduction, lll.cpp:103-108
      if constexpr (std::is_base_of<LStream<NQ>, RingT>::value)
        rc = babai_impl(T, ring, er_i >= 0 ? er_i : kappa, eta, M, upd, after, 0, er_i >= 0 ? kappa : -1);
      else
        rc = babai_impl(T, ring, kappa, eta, M, upd, after);
      if (er_i < 0 || rc != 1)
        break;
      if (++er_i >= d)
      {
        er_i       = -1;
        *early_red = kappa;
      }
    }
    if (rc != 1)
    {
      status      = rc;
      final_kappa = kappa;
      ok          = false;
      break;
    }
    const int sk = M.phys(kappa);
    // ---- Lovasz prefix values, lll.cpp:110-116: lt[0] = g(kappa,kappa),
    //      lt[i] = lt[i-1] - mu(kappa,i-1) r(kappa,i-1); lane i keeps lt[i] for i < kappa
    double lt0 = C.gf[(size_t)sk * ldd + sk];
    if (lt0 != lt0)
    {
      double p[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int c    = lane + 64 * q;
        const double a = (c < n) ? T.bfT[(size_t)c * ldd + sk] : 0.0;
        p[q]           = a * a;
      }
      double s = 0.0;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int hi = min(n - 64 * q, 64);
        for (int cc = 0; cc < hi; ++cc)
        {
          const double v = g_rl_f64(p[q], cc);
          s              = (q == 0 && cc == 0) ? v : s + v;
        }
      }
      lt0 = s;
      if (lane == 0)
        C.gf[(size_t)sk * ldd + sk] = s;
    }
    double prod[NQ], ltv[NQ], f[NQ];
    const long long rexpk = T.rexp[sk];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      const int t = lane + 64 * q;
      prod[q]     = T.murow[q] * T.rrow[q];
      ltv[q]      = 0.0;
      f[q]        = 0.0;
      if (t < kappa)
      {
        // delta * r(t,t) * 2^(2 (row_expo[t] - row_expo[kappa])), lll.cpp:117-121,131-135
        const int s = M.sl[q];
        double x    = T.rdg[s] * delta;
        if (T.row_expo_on)
        {
          long long e2 = 2 * (T.rexp[s] - rexpk);
          e2           = e2 > 100000 ? 100000 : (e2 < -100000 ? -100000 : e2);
          x            = ldexp(x, (int)e2);
        }
        f[q] = x;
      }
    }
    double g = lt0;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      const int hi = min(kappa - 64 * q, 64);
      for (int kk = 0; kk < hi; ++kk)
      {
        ltv[q] = (lane == kk) ? g : ltv[q];
        g      = g - g_rl_f64(prod[q], kk);
      }
    }
    // g = lt[kappa]
    double ltc[NQ];  // lane t: the value row t's threshold is compared with: lt[t], or lt[t + 1] with Siegel
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      ltc[q] = ltv[q];
    if (siegel)
    {
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        double dn = __shfl_down(ltv[q], 1);
        if (q + 1 < NQ)
        {
          const double carry = g_rl_f64(ltv[q + 1 < NQ ? q + 1 : q], 0);
          dn                 = (lane == 63) ? carry : dn;
        }
        ltc[q] = (lane + 64 * q == kappa - 1) ? g : dn;
      }
    }
    bool swp = false;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      swp |= (lane + 64 * q == kappa - 1) && (f[q] > ltc[q]);
    double ltk = g;
    if (__any(swp))
    {
      ++nswaps;
      const int old_k = kappa;
      // insertion index: largest kappa' in (kmin, old_k) with delta r(kappa'-1) < lt[kappa'-1]
      int knew = kmin;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int t      = lane + 64 * q;
        const bool cand  = t >= kmin && t <= old_k - 2 && f[q] < ltc[q];
        const uint64_t m = __ballot(cand);
        if (m)
          knew = max(knew, 64 * q + (63 - __clzll((long long)m)) + 1);
      }
      knew = uni(knew);
      ltk  = lane_get<NQ>(ltv, knew);
      if (ltk > 0.0)
      {
        rotate_right<NQ>(M, knew, old_k, lane);
        clamp_valid<NQ>(T, C, M, knew);
        kappa = knew;
        vp    = min(vp, knew);
      }
      else
      {  // linearly dependent row: to the end, lll.cpp:144-150
        ++zeros;
        rotate_left<NQ>(M, old_k, kend - zeros, lane);
        clamp_valid<NQ>(T, C, M, old_k);
        kappa = old_k;
        vp    = min(vp, old_k);
        __threadfence_block();
        continue;
      }
    }
    // ---- set_r(kappa, kappa, lt[kappa]), lll.cpp:153
    {
      const int s = M.phys(kappa);
      if (lane == 0)
      {
        T.r[(size_t)s * ldd + kappa] = ltk;
        T.rdg[s]                     = ltk;
        if (C.vc[s] == kappa)
          C.vc[s] = kappa + 1;
      }
    }
    // rows 0..kappa are now a fixed point of this loop (see vp above); an insertion at kmin > 0
    // has not been tested against row kmin-1 and does not extend the prefix
    if (vp >= kappa && (kappa > kmin || kmin == 0))
      vp = max(vp, kappa + 1);
    ++kappa;
    __threadfence_block();
  }
  if (ok)
    status = (kappa < kend - zeros) ? -3 : 1;
  return status;
}

// ---------------------------------------------------------------------------------------------------
// Out-of-line entry points for kernels that reach the LLL machinery from several places (bkz_kernel.hip,
// bkzs_kernel.hip).  Everything above is force-inlined, and with the block streams an inlined copy of
// update_row_cached / babai_impl is a few thousand instructions of unrolled loops: three lll() sites, a size
// reduction and two GSO updates per kernel, times eight kernels, took the compile of bkzs_kernel.hip beyond
// a quarter of an hour.  Here the state goes through ONE block of private memory per call (copied in, copied
// back: some fifty words per lane, against the thousands of iterations of an lll() or the d rows of a size
// reduction), and each entry point exists once per chunk count.  (The first generation's ring must not see
// scratch traffic between its counted waits: that build keeps everything inline.)
// ---------------------------------------------------------------------------------------------------
template <int NQ, class RingT> struct LllFrame
{
  Lattice<NQ> T;
  LllCtx C;
  SlotMap<NQ> M;
  RingT ring;
  int vp, final_kappa, nswaps, zeros;
  long long iter;
};

#if FPHIP_LLL_STREAM
#define FPHIP_LLL_OOL __noinline__
#else
#define FPHIP_LLL_OOL __forceinline__
#endif

// What comes out of private memory (and every argument of an out-of-line function) is a vector value for the
// compiler; the wave-uniform members go back to scalar registers here, so that the loops branch on the scalar unit
// and the lane masks are scalar pairs, as in the inlined form.
template <class P> __device__ __forceinline__ P *uni_ptr(P *p)
{
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo          = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi          = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return (P *)(((unsigned long long)hi << 32) | lo);
}
template <int NQ, class RingT> __device__ __forceinline__ void uniformize(LllFrame<NQ, RingT> &f)
{
#if FPHIP_LLL_STREAM
  f.T.d           = uni(f.T.d);
  f.T.n           = uni(f.T.n);
  f.T.ldd         = uni(f.T.ldd);
  f.T.ldn         = uni(f.T.ldn);
  f.T.row_expo_on = uni(f.T.row_expo_on);
  f.T.b           = uni_ptr(f.T.b);
  f.T.u           = nullptr;  // (the BKZ kernels do not track the transformation matrix)
  f.T.bfT         = uni_ptr(f.T.bfT);
  f.T.mu          = uni_ptr(f.T.mu);
  f.T.muT         = uni_ptr(f.T.muT);
  f.T.r           = uni_ptr(f.T.r);
  f.T.rdg         = uni_ptr(f.T.rdg);
  f.T.rexp        = uni_ptr(f.T.rexp);
  f.T.bfT32       = uni_ptr(f.T.bfT32);
  f.T.b32         = uni_ptr(f.T.b32);
  f.T.narrow_flag = uni_ptr(f.T.narrow_flag);
  f.T.np          = uni(f.T.np);
  f.T.f32ok       = uni(f.T.f32ok);
  f.C.gf          = uni_ptr(f.C.gf);
  f.C.vc          = uni_ptr(f.C.vc);
  f.vp            = uni(f.vp);
#endif
}
__device__ __forceinline__ double uni_f64(double v)
{
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                          __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

template <int NQ, class RingT>
__device__ FPHIP_LLL_OOL int lll_run_ool(LllFrame<NQ, RingT> *fp, int kmin, int kstart, int kend, double delta,
                                         double eta, double logdelta)
{
  LllFrame<NQ, RingT> f = *fp;
  uniformize(f);
  const int rc = lll_run(f.T, f.C, f.M, f.ring, uni(kmin), uni(kstart), uni(kend), uni_f64(delta), uni_f64(eta),
                         uni_f64(logdelta), f.final_kappa, f.nswaps, f.zeros, f.iter, f.vp);
  *fp = f;
  return rc;
}
template <int NQ, class RingT>
__device__ __forceinline__ int lll_run_call(Lattice<NQ> &T, LllCtx &C, SlotMap<NQ> &M, RingT &ring, int kmin,
                                            int kstart, int kend, double delta, double eta, double logdelta,
                                            int &final_kappa, int &nswaps, int &zeros, long long &iter, int &vp)
{
  LllFrame<NQ, RingT> f{T, C, M, ring, vp, 0, 0, 0, 0};
  const int rc = lll_run_ool<NQ, RingT>(&f, kmin, kstart, kend, delta, eta, logdelta);
  T           = f.T;
  M           = f.M;
  ring        = f.ring;
  vp          = f.vp;
  final_kappa = f.final_kappa;
  nswaps      = f.nswaps;
  zeros       = f.zeros;
  iter        = f.iter;
  return rc;
}

// lll_obj.size_reduction(kfrom, kend, sr_start), lll.h:107-122, on the cached state: babai(k) then
// update_gso_row(k, k) for every row.  1, or the failing status (0 GSO, -1 babai, -2 multiplier).
template <int NQ, class RingT>
__device__ __forceinline__ int size_reduce_rows(Lattice<NQ> &T, LllCtx &C, SlotMap<NQ> &M, RingT &ring, int kfrom,
                                                int kend, double eta, int sr_start, int &vp)
{
  auto upd   = [&](int k, int last) { return update_row_cached(T, C, M, ring, k, last); };
  auto after = [&](int k)
  {
    after_rowop<NQ>(T, C, M, k);
    vp = min(vp, k);
  };
  for (int k = kfrom; k < kend; ++k)
  {
    if (k > 0)
    {
      const int rc = babai_impl(T, ring, k, eta, M, upd, after, sr_start);
      if (rc != 1)
        return rc;
    }
    if (!upd(k, k))
      return 0;
    __threadfence_block();
  }
  return 1;
}
template <int NQ, class RingT>
__device__ FPHIP_LLL_OOL int size_reduce_ool(LllFrame<NQ, RingT> *fp, int kfrom, int kend, double eta, int sr_start)
{
  LllFrame<NQ, RingT> f = *fp;
  uniformize(f);
  const int rc = size_reduce_rows(f.T, f.C, f.M, f.ring, uni(kfrom), uni(kend), uni_f64(eta), uni(sr_start), f.vp);
  *fp                   = f;
  return rc;
}
template <int NQ, class RingT>
__device__ __forceinline__ int size_reduce_call(Lattice<NQ> &T, LllCtx &C, SlotMap<NQ> &M, RingT &ring, int kfrom,
                                                int kend, double eta, int sr_start, int &vp)
{
  if (kfrom >= kend)
    return 1;
  LllFrame<NQ, RingT> f{T, C, M, ring, vp, 0, 0, 0, 0};
  const int rc = size_reduce_ool<NQ, RingT>(&f, kfrom, kend, eta, sr_start);
  T            = f.T;
  ring         = f.ring;
  vp           = f.vp;
  return rc;
}

// update_gso_row(k, k) for the rows of [k0, k1) whose cached row is shorter than that.  false: GSO failure.
template <int NQ, class RingT>
__device__ FPHIP_LLL_OOL int update_rows_ool(LllFrame<NQ, RingT> *fp, int k0, int k1)
{
  LllFrame<NQ, RingT> f = *fp;
  uniformize(f);
  k0     = uni(k0);
  k1     = uni(k1);
  int ok = 1;
  for (int k = k0; k < k1 && ok; ++k)
    if (uni(f.C.vc[f.M.phys(k)]) <= k)
    {
      if (!update_row_cached(f.T, f.C, f.M, f.ring, k, k))
        ok = 0;
      __threadfence_block();
    }
  *fp = f;
  return ok;
}
template <int NQ, class RingT>
__device__ __forceinline__ bool update_rows_call(Lattice<NQ> &T, LllCtx &C, SlotMap<NQ> &M, RingT &ring, int k0, int k1)
{
  LllFrame<NQ, RingT> f{T, C, M, ring, 0, 0, 0, 0, 0};
  const int ok = update_rows_ool<NQ, RingT>(&f, k0, k1);
  T            = f.T;
  ring         = f.ring;
  return ok != 0;
}

// the rows in position order (b2), after which the host rebuilds the identity-layout GSO
template <int NQ>
__device__ __forceinline__ void lll_write_ordered(const Lattice<NQ> &T, const SlotMap<NQ> &M, long long *bo)
{
  const int lane = T.lane, d = T.d, ldn = T.ldn;
  for (int p = 0; p < d; ++p)
  {
    const int s = M.phys(p);
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      const int c = lane + 64 * q;
      if (c < ldn)
        bo[(size_t)p * ldn + c] = T.b[(size_t)s * ldn + c];
    }
  }
}

}  // namespace fphip
#endif
