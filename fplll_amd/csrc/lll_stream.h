// lll_stream.h — the streaming loops of the slot-mode reduction kernels (lll_kernel.hip, bkz_kernel.hip,
// bkzs_kernel.hip), second generation (round 5).  Device code only.
//
// The four hot loops of an LLL iteration — the Gram row of b_kappa (numvect.h:386-396 through
// gso.h:314-331), the column recurrence of update_gso_row (gso_interface.cpp:143-158), the mu sweep of
// babai (lll.cpp:202-214) and its integer row operation (gso.cpp:84-158) — all have the shape
//     for step s:  v = ROW_s;  state = f(state, v, scalar_s)
// with ROW_s a contiguous row in HBM whose address does not depend on the state.
//
// The first generation (Ring in gso_wave.h) streamed the rows through an LDS ring filled by global_load_lds
// and spent 57 issued instructions and 500 cycles per row on a lone wave (profiles/r04_lll_kernel_pmc_summary.txt:
// 13 600 instructions per LLL iteration).  Rebuilding that ring in blocks of four rows with double-buffered LDS
// reads, scalar lane masks and precomputed windows (this file's first form this round, commit 98065d4) passed every
// parity test and took the batched LLL from 171 to 245 lattices/s, but its per-phase timers
// (profiles/r05_lll_phase_timers_lds_dma_*.log) still showed 88-205 ns per streamed row: one global_load_lds
// instruction stalls the issuing wave for ~60 cycles (MI355X_MICROARCH.md, "LDS-DMA piece issue cost"), and the
// 16 KiB of ring a wave can have hold 8-12 rows for the 400-900 cycles a row is under way.
//
// The rows therefore go through REGISTERS now:
//   * one raw buffer load per row and 64-lane chunk — descriptor base, scalar row offset, per-lane element
//     offset (the slot gather is the lane's own offset), no address arithmetic on the vector unit;
//   * D rows in flight (16 up to 128 columns, 8 above) in a rotating window of registers: the loop body is D
//     rows unrolled, every row = use the window entry, then refill it with the row D steps ahead; the waits are
//     the compiler's own vmcnt bookkeeping (the loads are ordinary tracked instructions);
//   * loops are chunk-major: the register that holds the step's scalar is a compile-time constant of the
//     unrolled group (Gram, recurrence) or chosen by one branch per row (sweep, row operation);
//   * lane predicates of the recurrence and the sweep are 64-bit scalar masks applied with v_cndmask_b32_e64;
//   * the integer row operation streams only the rows whose multiplier is not zero, and on lattices below
//     2^24 with 32-bit multipliers it is one v_mad_i64_i32 per chunk on the low words;
//   * the sums start from -0.0 (x + -0.0 == x for every x) instead of selecting the first product.
// No LDS at all.  The arithmetic — every product, sum, quotient and rounding, and their order per output
// element — is the first generation's, which is the reference's.
#ifndef FPHIP_LLL_STREAM_H
#define FPHIP_LLL_STREAM_H

#include <type_traits>

#include "gso_wave.h"

namespace fphip
{

// m ? a : b per lane, the mask in a scalar register pair
__device__ __forceinline__ int ls_sel_i32(unsigned long long m, int a, int b)
{
  int r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
  return r;
}
__device__ __forceinline__ double ls_sel_f64(unsigned long long m, double a, double b)
{
  return __hiloint2double(ls_sel_i32(m, __double2hiint(a), __double2hiint(b)),
                          ls_sel_i32(m, __double2loint(a), __double2loint(b)));
}
extern "C" __device__ int fphip_ls_llvm_writelane(int, int, int) __asm("llvm.amdgcn.writelane.i32");
// the wave-uniform val into lane `lane` of old
__device__ __forceinline__ double ls_wl_f64(double val, int lane, double old)
{
  const int l  = __builtin_amdgcn_readfirstlane(lane);
  const int lo = fphip_ls_llvm_writelane(__builtin_amdgcn_readfirstlane(__double2loint(val)), l, __double2loint(old));
  const int hi = fphip_ls_llvm_writelane(__builtin_amdgcn_readfirstlane(__double2hiint(val)), l, __double2hiint(old));
  return __hiloint2double(hi, lo);
}

// an array of `bytes` bytes as a buffer resource (untyped 32-bit data, range-checked: reads behind the end give 0)
typedef unsigned ls_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ls_rsrc(const void *base, unsigned bytes)
{
  return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ unsigned ls_ld32(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
  return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ ls_v2u ls_ld64(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
  return __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ double ls_f64(ls_v2u v) { return __hiloint2double((int)v.y, (int)v.x); }
__device__ __forceinline__ long long ls_i64(ls_v2u v)
{
  return (long long)(((unsigned long long)v.y << 32) | (unsigned long long)v.x);
}

// -DFPHIP_LLL_PROF=1 (tests/perf/build_lll_variants.sh, libPROF.so): every phase adds its count, rows, start-up
// time (first group computed) and total time, in ticks of the 100 MHz real-time counter, to pf[4 * KIND ..]
#ifndef FPHIP_LLL_PROF
#define FPHIP_LLL_PROF 0
#endif
enum
{
  LS_GRAM   = 0,
  LS_REC    = 1,
  LS_SWEEP  = 2,
  LS_AXPY   = 3,
  LS_SINGLE = 4,  // Gram entries completed one by one (update_row_cached)
  LS_KINDS  = 5
};

template <int NQ> struct LStream
{
  static constexpr int D     = (NQ <= 2) ? 16 : 8;  // rows in flight
  static constexpr int BYTES = 0;                   // no LDS
  int lane;
#if FPHIP_LLL_PROF
  unsigned long long pf[4 * LS_KINDS];
#endif
  __device__ __forceinline__ void init(int, int lane_)
  {
    lane = lane_;
#if FPHIP_LLL_PROF
    for (int i = 0; i < 4 * LS_KINDS; ++i)
      pf[i] = 0;
#endif
  }
  __device__ __forceinline__ void prof_add(int kind, unsigned long long rows, unsigned long long t_first,
                                           unsigned long long t_all)
  {
#if FPHIP_LLL_PROF
    pf[4 * kind + 0] += 1;
    pf[4 * kind + 1] += rows;
    pf[4 * kind + 2] += t_first;
    pf[4 * kind + 3] += t_all;
#endif
  }
  static __device__ __forceinline__ unsigned long long now()
  {
#if FPHIP_LLL_PROF
    return __builtin_amdgcn_s_memrealtime();
#else
    return 0;
#endif
  }
};

// One phase: nrows rows through the register window.  Ph provides
//   struct Row                      this lane's words of one row
//   fetch(Row &, s)                 issue the loads of row s (any s >= 0: behind the end a valid row is read again)
//   group(Row (&)[D], s0)           rows s0 .. s0+D-1: for each, the arithmetic (rows < nrows), then fetch(s + D)
template <int NQ, class Ph> __device__ __forceinline__ void ls_run(LStream<NQ> &S, Ph &ph, int nrows)
{
  constexpr int D = LStream<NQ>::D;
  if (nrows <= 0)
    return;
  const unsigned long long pt0 = LStream<NQ>::now();
  unsigned long long pt1       = pt0;
  typename Ph::Row win[D];
#pragma unroll
  for (int u = 0; u < D; ++u)
    ph.fetch(win[u], u);
#pragma unroll 1
  for (int s0 = 0; s0 < nrows; s0 += D)
  {
    ph.group(win, s0);
#if FPHIP_LLL_PROF
    if (s0 == 0)
      pt1 = LStream<NQ>::now();
#endif
  }
  S.prof_add(Ph::KIND, (unsigned long long)nrows, pt1 - pt0, LStream<NQ>::now() - pt0);
}

// ---------------------------------------------------------------------------------------------------
// Gram row: lane (q, l) = row position l + 64 q accumulates g(kappa, position) over the columns c
// ascending; row c of the column-major mirror is gathered by slot.
// ---------------------------------------------------------------------------------------------------
template <int NQ, bool F32> struct GramPh
{
  using L = LStream<NQ>;
  static constexpr int KIND = LS_GRAM;
  __amdgpu_buffer_rsrc_t rs;  // bfT32 or bfT of the lattice
  unsigned stride;            // bytes per row
  int nrows;
  unsigned off[NQ];  // byte offset of this lane's element in a row (slot * 4 or slot * 8)
  double (&g)[NQ];
  const double (&bk)[NQ];
  int qact;  // chunks 0 .. qact-1 hold a position that is wanted
  struct Row
  {
    typename std::conditional<F32, unsigned, ls_v2u>::type x[NQ];
  };
  __device__ __forceinline__ void fetch(Row &R, int s) const
  {
    const unsigned so = (unsigned)(s < nrows ? s : nrows - 1) * stride;  // behind the end: the last row again
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      if (q < qact)
      {
        if constexpr (F32)
          R.x[q] = ls_ld32(rs, off[q], so);
        else
          R.x[q] = ls_ld64(rs, off[q], so);
      }
  }
  __device__ __forceinline__ double val(const Row &R, int q) const
  {
    if constexpr (F32)
      return (double)__uint_as_float(R.x[q]);
    else
      return ls_f64(R.x[q]);
  }
  __device__ __forceinline__ void group(Row (&W)[L::D], int s0)
  {
    dispatch_chunk<NQ>(s0,
                       [&](auto cq_, int cc0)
                       {
                         constexpr int cq = decltype(cq_)::value;
#pragma unroll
                         for (int u = 0; u < L::D; ++u)
                         {
                           if (s0 + u < nrows)
                           {
                             const double bkc = g_rl_f64(bk[cq], cc0 + u);
#pragma unroll
                             for (int q = 0; q < NQ; ++q)
                               if (q < qact)
                               {
                                 const double pr = bkc * val(W[u], q);
                                 g[q]            = g[q] + pr;
                               }
                           }
                           fetch(W[u], s0 + u + L::D);
                         }
                       });
  }
};

// ---------------------------------------------------------------------------------------------------
// Column recurrence of update_gso_row: step k subtracts mu(position, k) r(kappa, k) from the lanes
// position > k inside [start, last]; row k of muT is gathered by slot.  diag: last == kappa, the lane of
// kappa itself takes mu(kappa, k) = r(kappa, k) / r(k, k).
// ---------------------------------------------------------------------------------------------------
template <int NQ> struct RecPh
{
  using L = LStream<NQ>;
  static constexpr int KIND = LS_REC;
  __amdgpu_buffer_rsrc_t rs;  // muT of the lattice
  unsigned stride;
  int nrows;
  int lane;
  unsigned off[NQ];
  double (&acc)[NQ];
  const double (&rd)[NQ];
  unsigned long long bmask[NQ];  // lanes whose position lies in [start, last]
  bool diag;
  int kappa;
  struct Row
  {
    ls_v2u m[NQ];
  };
  __device__ __forceinline__ void fetch(Row &R, int s) const
  {
    const unsigned so = (unsigned)(s < nrows ? s : nrows - 1) * stride;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      if (bmask[q] != 0)
        R.m[q] = ls_ld64(rs, off[q], so);
  }
  __device__ __forceinline__ void group(Row (&W)[L::D], int s0)
  {
    dispatch_chunk<NQ>(s0,
                       [&](auto kq_, int kk0)
                       {
                         constexpr int kq = decltype(kq_)::value;
#pragma unroll
                         for (int u = 0; u < L::D; ++u)
                         {
                           if (s0 + u < nrows)
                           {
                             const int kk    = kk0 + u;
                             const double rk = g_rl_f64(acc[kq], kk);  // r(kappa,k) is final
                             double muk      = 0.0;
                             if (diag)
                               muk = rk / g_rl_f64(rd[kq], kk);  // mu(kappa,k)
#pragma unroll
                             for (int q = kq; q < NQ; ++q)
                               if (bmask[q] != 0)
                               {
                                 const unsigned long long mk =
                                     (q == kq) ? (bmask[q] & ((~1ull) << kk)) : bmask[q];  // positions > k
                                 double m = ls_f64(W[u].m[q]);
                                 if (diag)
                                   m = (lane + 64 * q == kappa) ? muk : m;
                                 const double t  = m * rk;
                                 const double uu = acc[q] - t;
                                 acc[q]          = ls_sel_f64(mk, uu, acc[q]);
                               }
                           }
                           fetch(W[u], s0 + u + L::D);
                         }
                       });
  }
};

// rnd_we, nr/nr_FP_d.inl:226-233.  frexp's exponent replaces FP_NR<double>::exponent() = ilogb + 1 here:
// they differ for 0, inf and NaN only, and for those both branches of rnd_we return b itself.
__device__ __forceinline__ double ls_rnd_we(double b, int e)
{
  if ((long long)__builtin_amdgcn_frexp_exp(b) + (long long)e >= 53)
    return b;
  return ldexp(rint(ldexp(b, e)), -e);
}

// ---------------------------------------------------------------------------------------------------
// babai's sweep, lll.cpp:202-214: rows j = kappa-1 .. sr_start of mu, descending (row s is j = kappa-1-s);
// lane k owns babai_mu[k].
// ---------------------------------------------------------------------------------------------------
template <int NQ> struct SweepPh
{
  using L = LStream<NQ>;
  static constexpr int KIND = LS_SWEEP;
  __amdgpu_buffer_rsrc_t rs;  // mu of the lattice
  unsigned stride;            // ldd * 8
  int kappa, sr_start;
  int lane;
  const SlotMap<NQ> &M;
  double (&bm)[NQ];
  double (&xs)[NQ];              // the multipliers X_j (lane j), 0 where none
  const int (&e)[NQ];
  unsigned long long (&nz)[NQ];  // rows of each chunk with X_j != 0
  unsigned long long srmask[NQ];  // lanes k >= sr_start
  struct Row
  {
    ls_v2u m[NQ];
  };
  __device__ __forceinline__ void fetch(Row &R, int s) const
  {
    int j = kappa - 1 - s;
    j     = j < 0 ? 0 : j;  // behind the last row: row 0 again, never used
    const unsigned so = (unsigned)M.phys(j) * stride;
    const int qj      = j >> 6;  // mu(j, k) is wanted for k < j only
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      if (q <= qj)
        R.m[q] = ls_ld64(rs, (unsigned)(lane + 64 * q) * 8u, so);
  }
  __device__ __forceinline__ void group(Row (&W)[L::D], int s0)
  {
#pragma unroll
    for (int u = 0; u < L::D; ++u)
    {
      const int j = kappa - 1 - (s0 + u);
      if (j >= sr_start)
        dispatch_chunk<NQ>(j,
                           [&](auto jq_, int jj)
                           {
                             constexpr int jq = decltype(jq_)::value;
                             const double bmj = g_rl_f64(bm[jq], jj);
                             const int ej     = __builtin_amdgcn_readlane(e[jq], jj);
                             const double X   = ls_rnd_we(bmj, ej);
                             if (X != 0.0)
                             {
                               nz[jq] |= 1ull << jj;
                               xs[jq] = ls_wl_f64(X, jj, xs[jq]);
#pragma unroll
                               for (int q = 0; q <= jq; ++q)
                               {
                                 // chunks below jq hold only k < j; the chunk of j itself needs the test
                                 const unsigned long long mk =
                                     (q == jq) ? (srmask[q] & ((1ull << jj) - 1)) : srmask[q];
                                 const double t  = X * ls_f64(W[u].m[q]);
                                 const double uu = bm[q] - t;
                                 bm[q]           = ls_sel_f64(mk, uu, bm[q]);
                               }
                             }
                           });
      fetch(W[u], s0 + u + L::D);
    }
  }
};

// ---------------------------------------------------------------------------------------------------
// The integer row operation b_kappa += sum_j x_j b_j (row_addmul_si and friends, gso.cpp:84-158; the
// sum is in wrapping 64-bit arithmetic, so its order is free): only rows with x_j != 0 are streamed,
// highest j first.  SMALL: every entry of the rows and every multiplier fits 32 bits.
// ---------------------------------------------------------------------------------------------------
template <int NQ> struct RowCursor
{
  unsigned long long m[NQ];
  __device__ __forceinline__ int next()
  {
    int j = -1;
#pragma unroll
    for (int q = NQ - 1; q >= 0; --q)
      if (j < 0 && m[q] != 0)
      {
        const int jj = 63 - __builtin_clzll(m[q]);
        m[q] &= ~(1ull << jj);
        j = 64 * q + jj;
      }
    return j;
  }
};

template <int NQ, bool SMALL> struct AxpyPh
{
  using L = LStream<NQ>;
  static constexpr int KIND = LS_AXPY;
  __amdgpu_buffer_rsrc_t rs;  // b of the lattice
  unsigned stride;            // ldn * 8
  int lane;
  const SlotMap<NQ> &M;
  long long (&bv)[NQ];
  const long long (&lxv)[NQ];  // multiplier of row j in lane j
  RowCursor<NQ> ic, cc;        // rows still to request / to apply
  struct Row
  {
    typename std::conditional<SMALL, unsigned, ls_v2u>::type w[NQ];
  };
  __device__ __forceinline__ void fetch(Row &R, int)
  {
    int j = ic.next();
    j     = j < 0 ? 0 : j;  // behind the last row: row 0 again, never used
    const unsigned so = (unsigned)M.phys(j) * stride;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      if constexpr (SMALL)
        R.w[q] = ls_ld32(rs, (unsigned)(lane + 64 * q) * 8u, so);  // (the low words)
      else
        R.w[q] = ls_ld64(rs, (unsigned)(lane + 64 * q) * 8u, so);
    }
  }
  __device__ __forceinline__ void group(Row (&W)[L::D], int s0)
  {
#pragma unroll
    for (int u = 0; u < L::D; ++u)
    {
      const int j = cc.next();
      if (j >= 0)
      {
        long long lx = 0;
        dispatch_chunk<NQ>(j, [&](auto jq_, int jj) { lx = g_rl_i64(lxv[decltype(jq_)::value], jj); });
        if constexpr (SMALL)
        {
          const int sx = (int)lx;
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            bv[q] = bv[q] + (long long)(int)W[u].w[q] * (long long)sx;
        }
        else
        {
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            bv[q] = (long long)((unsigned long long)bv[q] +
                                (unsigned long long)ls_i64(W[u].w[q]) * (unsigned long long)lx);
        }
      }
      fetch(W[u], s0 + u + L::D);
    }
  }
};

}  // namespace fphip
#endif
