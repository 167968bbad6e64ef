// lll_stream.h — the streaming loops of the slot-mode reduction kernels (lll_kernel.hip, bkz_kernel.hip,
// bkzs_kernel.hip), second generation (round 5).  Device code only.
//
// The four hot loops of an LLL iteration — the Gram row of b_kappa (numvect.h:386-396 through
// gso.h:314-331), the column recurrence of update_gso_row (gso_interface.cpp:143-158), the mu sweep of
// babai (lll.cpp:202-214) and its integer row operation (gso.cpp:84-158) — all have the shape
//     for step s:  v = ROW_s;  state = f(state, v, scalar_s)
// with ROW_s a contiguous row in HBM.  The first generation (Ring in gso_wave.h) spent 57 issued
// instructions per streamed row, most of them scalar bookkeeping (ring indices modulo R, a window test
// and an m0 save / restore per DMA instruction, a branch on the chunk that owns the step's scalar, a
// pipeline-state switch), and exposed the LDS latency of every row: a lone wave needed 500 cycles per row
// (profiles/r04_lll_kernel_pmc_summary.txt: 13 600 instructions per LLL iteration).  Here
//   * rows move in BLOCKS of U (4 for up to 128 columns, 2 above): one counted s_waitcnt, one chunk
//     dispatch, one ring-index update and U x NQ ds_read with compile-time offsets per block;
//   * the LDS reads of block b+1 are issued BEFORE the arithmetic of block b (two register sets), so a
//     lone wave no longer waits for LDS once per row;
//   * a DMA instruction is s_mov m0 / s_mov exec / global_load_lds / s_mov exec — the window of a row is a
//     lane mask, m0 is declared clobbered instead of saved;
//   * lane predicates of the recurrence and the sweep are 64-bit scalar masks applied with
//     v_cndmask_b32_e64; the per-step part of a mask is one s_lshl_b64 + s_and_b64;
//   * the integer row operation streams only the rows whose multiplier is not zero, and on lattices below
//     2^24 with 32-bit multipliers it is one v_mad_i64_i32 per chunk;
//   * the sums start from -0.0 (x + -0.0 == x for every x) instead of selecting the first product.
// The arithmetic — every product, sum, quotient and rounding, and their order per output element — is the
// first generation's, which is the reference's.
//
// Tried and measured against this (commit 7c6a3d6, profiles/r05_lll_phase_timers_register_streams_*.log): the same
// phases through REGISTERS — one raw buffer load per row and chunk (the slot gather as the lane's own offset),
// 16 rows in flight in a rotating register window, no LDS.  Parity green, but 129 ns per Gram row instead of 88,
// 137 LLL/s at batch 2048 instead of 245, 6.45 s for a lone lattice instead of 5.55: a gathered dword load per
// 64 lanes costs the memory pipeline more than a sixteenth of a 1-KiB LDS-DMA row, and 256 registers per wave leave
// the same two waves per SIMD.  The contiguous phases (sweep, row operation) were 20 % faster per row that way and
// 2.5 us slower to start; not kept.
//
// Also tried (call r5j): a ring of 16 blocks (64 KiB, one wave per workgroup) for launches of a few lattices —
// no gain (a lone 120-dimensional LLL 5.72 s against 5.48 s, 83 ns per Gram row either way, the prologue's 64 DMA
// instructions cost 1.6 us): a lone wave is bound by ISSUING its rows (an LDS-DMA instruction ~60-100 cycles plus
// ~24 other instructions per row), not by how many it has in flight.
#ifndef FPHIP_LLL_STREAM_H
#define FPHIP_LLL_STREAM_H

#include <type_traits>

#include "gso_wave.h"

namespace fphip
{

// every kernel that includes this header runs with dynamic LDS only: the rings start at LDS address 0
extern __shared__ __attribute__((aligned(16))) char fphip_lds[];

__device__ __forceinline__ unsigned long long ls_uni64(unsigned long long v)
{
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
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

// -DFPHIP_LLL_PROF=1 (tests/perf/build_lll_variants.sh, libPROF.so): every phase adds its count, rows, start-up
// time (first block landed) and total time, in ticks of the 100 MHz real-time counter, to pf[4 * KIND ..]
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
  static constexpr int U    = (NQ <= 2) ? 4 : 2;                       // rows per block
  static constexpr int ROWB = 512 * NQ;                                // bytes of a ring row (64 NQ doubles)
  static constexpr int NB   = (NQ == 1) ? 8 : (NQ == 3 ? 5 : 4);       // blocks in the ring
  static constexpr int BLKB = U * ROWB;
  static constexpr int BYTES = NB * BLKB;                              // per wave: 16 / 16 / 15 / 16 KiB
  unsigned base;    // LDS byte address of this wave's ring
  int lane;
  unsigned lane16;
#if FPHIP_LLL_PROF
  unsigned long long pf[4 * LS_KINDS];
#endif
  __device__ __forceinline__ void init(int wave, int lane_)
  {
    base   = (unsigned)(wave * BYTES);
    lane   = lane_;
    lane16 = (unsigned)lane_ * 16u;
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

// bytes [0, len) of the row at p (wave-uniform, 16-byte aligned) to LDS address dst (wave-uniform) in IPR
// instructions of 1 KiB; len <= 0 (and every instruction whose span lies behind len): lane 0 alone fetches
// the row's first 16 bytes again — the count of instructions in flight per row is always IPR.
template <int IPR>
__device__ __forceinline__ void ls_dma_row(const char *p_, int len, unsigned dst_, unsigned lane16)
{
  const char *p      = (const char *)ls_uni64((unsigned long long)p_);
  const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)dst_);
  const int nl       = __builtin_amdgcn_readfirstlane((len + 15) >> 4);  // 16-byte lanes
  const unsigned long long mA = nl >= 64 ? ~0ull : (nl > 0 ? ((1ull << nl) - 1) : 1ull);
  if constexpr (IPR == 1)
  {
    asm volatile("s_mov_b32 m0, %1\n\t"
                 "s_mov_b64 exec, %2\n\t"
                 "global_load_lds_dwordx4 %3, %0\n\t"
                 "s_mov_b64 exec, -1"
                 :
                 : "s"(p), "s"(dst), "s"(mA), "v"(lane16)
                 : "memory", "m0");
  }
  else
  {
    const bool two               = nl > 64;
    const unsigned long long mB  = two ? (nl >= 128 ? ~0ull : ((1ull << (nl - 64)) - 1)) : 1ull;
    const char *pB               = two ? p + 1024 : p;
    const unsigned dstB          = two ? dst + 1024 : dst;
    asm volatile("s_mov_b32 m0, %2\n\t"
                 "s_mov_b64 exec, %4\n\t"
                 "global_load_lds_dwordx4 %6, %0\n\t"
                 "s_mov_b32 m0, %3\n\t"
                 "s_mov_b64 exec, %5\n\t"
                 "global_load_lds_dwordx4 %6, %1\n\t"
                 "s_mov_b64 exec, -1"
                 :
                 : "s"(p), "s"(pB), "s"(dst), "s"(dstB), "s"(mA), "s"(mB), "v"(lane16)
                 : "memory", "m0");
  }
}

// the same with the lane masks of a window that many rows share, computed once
struct LsWin
{
  unsigned long long mA, mB;
  unsigned offB;  // 1024 when the second instruction carries data, else 0 (lane 0 repeats the first 16 bytes)
};
__device__ __forceinline__ LsWin ls_make_win(int len)
{
  const int nl = __builtin_amdgcn_readfirstlane((len + 15) >> 4);
  LsWin w;
  w.mA           = nl >= 64 ? ~0ull : (nl > 0 ? ((1ull << nl) - 1) : 1ull);
  const bool two = nl > 64;
  w.mB           = two ? (nl >= 128 ? ~0ull : ((1ull << (nl - 64)) - 1)) : 1ull;
  w.offB         = two ? 1024u : 0u;
  return w;
}
template <int IPR>
__device__ __forceinline__ void ls_dma_win(const char *p_, const LsWin &w, unsigned dst_, unsigned lane16)
{
  const char *p      = (const char *)ls_uni64((unsigned long long)p_);
  const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)dst_);
  if constexpr (IPR == 1)
  {
    asm volatile("s_mov_b32 m0, %1\n\t"
                 "s_mov_b64 exec, %2\n\t"
                 "global_load_lds_dwordx4 %3, %0\n\t"
                 "s_mov_b64 exec, -1"
                 :
                 : "s"(p), "s"(dst), "s"(w.mA), "v"(lane16)
                 : "memory", "m0");
  }
  else
  {
    const char *pB      = p + w.offB;
    const unsigned dstB = dst + w.offB;
    asm volatile("s_mov_b32 m0, %2\n\t"
                 "s_mov_b64 exec, %4\n\t"
                 "global_load_lds_dwordx4 %6, %0\n\t"
                 "s_mov_b32 m0, %3\n\t"
                 "s_mov_b64 exec, %5\n\t"
                 "global_load_lds_dwordx4 %6, %1\n\t"
                 "s_mov_b64 exec, -1"
                 :
                 : "s"(p), "s"(pB), "s"(dst), "s"(dstB), "s"(w.mA), "s"(w.mB), "v"(lane16)
                 : "memory", "m0");
  }
}

template <int K> __device__ __forceinline__ void ls_wait()
{
  static_assert(K >= 0 && K <= 63, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory");
}
// at most `blocks` blocks of IPB instructions may still be in flight
template <int IPB, int MAXB> __device__ __forceinline__ void ls_wait_dyn(int blocks)
{
  if constexpr (MAXB <= 0)
    ls_wait<0>();
  else
  {
    if (blocks >= MAXB)
      ls_wait<(MAXB * IPB <= 63 ? MAXB * IPB : 63)>();
    else
      ls_wait_dyn<IPB, MAXB - 1>(blocks);
  }
}

// One phase: nrows rows through the ring.  Ph provides
//   IPR                               DMA instructions per row
//   issue<FULL>(dst)                  request the next row (or a dummy behind the last) into LDS address dst;
//                                     FULL: the row is known to exist
//   struct Regs; load(Regs &, addr)   this lane's words of the U rows of the block at LDS address addr
//   compute(const Regs &, s0)         the arithmetic of rows s0 .. s0+U-1 (those below nrows)
template <int NQ, class Ph> __device__ __forceinline__ void ls_run(LStream<NQ> &S, Ph &ph, int nrows)
{
  using L           = LStream<NQ>;
  constexpr int U   = L::U;
  constexpr int NB  = L::NB;
  constexpr int IPB = U * Ph::IPR;
  static_assert((NB - 1) * IPB <= 63, "the ring holds more instructions than vmcnt can count");
  if (nrows <= 0)
    return;
  const int nblk = (nrows + U - 1) / U;
  // everything older (stores of the previous phase, ordinary loads) is retired first: the counted waits
  // below then only ever see this phase's DMA instructions
  const unsigned long long pt0 = L::now();
  ls_wait<0>();
  unsigned hoff = 0;  // ring offset of the block slot to fill next
  int rreq      = 0;  // rows requested so far
  auto issue_block = [&]()
  {
    if (rreq + U <= nrows)
    {  // every row of the block exists
#pragma unroll
      for (int u = 0; u < U; ++u)
        ph.template issue<true>(S.base + hoff + (unsigned)(u * L::ROWB));
    }
    else
    {
#pragma unroll
      for (int u = 0; u < U; ++u)
        ph.template issue<false>(S.base + hoff + (unsigned)(u * L::ROWB));
    }
    rreq += U;
    hoff = (hoff + L::BLKB == (unsigned)L::BYTES) ? 0u : hoff + L::BLKB;
  };
  const int npro = nblk < NB ? nblk : NB;
#pragma unroll 1
  for (int i = 0; i < npro; ++i)
    issue_block();
  ls_wait_dyn<IPB, NB - 1>(npro - 1);  // block 0 has landed
  const unsigned long long pt1 = L::now();
  typename Ph::Regs A, B;
  unsigned toff = 0;
  ph.load(A, S.base + toff);
  toff = (toff + L::BLKB == (unsigned)L::BYTES) ? 0u : toff + L::BLKB;
#pragma unroll 1
  for (int b = 0; b < nblk; ++b)
  {
    const int rest = nblk - b - 2;  // blocks behind block b+1 that exist
    if (rest >= 0)
    {
      // blocks 0 .. min(b + NB, nblk) - 1 have been requested: all but min(NB - 2, rest) of them must be here
      if (rest >= NB - 2)
        ls_wait<(NB - 2) * IPB>();
      else
        ls_wait_dyn<IPB, NB - 2>(rest);
      ph.load(B, S.base + toff);
      toff = (toff + L::BLKB == (unsigned)L::BYTES) ? 0u : toff + L::BLKB;
    }
    ph.compute(A, b * U);
    if (b + NB < nblk)
    {
      // the slot of block b is free once its reads have returned (they were issued before those of B)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      issue_block();
    }
    A = B;
  }
  S.prof_add(Ph::KIND, (unsigned long long)nrows, pt1 - pt0, L::now() - pt0);
}

// ---------------------------------------------------------------------------------------------------
// Gram row: lane (q, l) = row position l + 64 q accumulates g(kappa, position) over the columns c
// ascending; row c of the column-major mirror is gathered by slot.
// ---------------------------------------------------------------------------------------------------
template <int NQ, bool F32> struct GramPh
{
  using L = LStream<NQ>;
  static constexpr int KIND = LS_GRAM;
  static constexpr int IPR = F32 ? 1 : (NQ + 1) / 2;
  const char *p;
  long stride;
  LsWin win;  // bytes [0, len) of every row
  int nrows, srq;
  unsigned lane16;
  unsigned off[NQ];  // byte offset of this lane's element in a row (slot * 4 or slot * 8)
  double (&g)[NQ];
  const double (&bk)[NQ];
  int qact;  // chunks 0 .. qact-1 hold a position <= last
  struct Regs
  {
    typename std::conditional<F32, float, double>::type x[L::U][NQ];
  };
  template <bool FULL> __device__ __forceinline__ void issue(unsigned dst)
  {
    if (FULL || srq < nrows)
    {
      ls_dma_win<IPR>(p, win, dst, lane16);
      p += stride;
    }
    else
      ls_dma_row<IPR>(p - stride, 0, dst, lane16);  // (a dummy: the last row's first bytes)
    ++srq;
  }
  __device__ __forceinline__ void load(Regs &R, unsigned a) const
  {
    using E = typename std::conditional<F32, float, double>::type;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int u = 0; u < L::U; ++u)
        R.x[u][q] = *(const E *)(fphip_lds + (a + off[q] + (unsigned)(u * L::ROWB)));
  }
  __device__ __forceinline__ void compute(const Regs &R, int s0)
  {
    const int nv = nrows - s0;  // >= 1
    dispatch_chunk<NQ>(s0,
                       [&](auto cq_, int cc0)
                       {
                         constexpr int cq = decltype(cq_)::value;
                         double bkc[L::U];
#pragma unroll
                         for (int u = 0; u < L::U; ++u)
                           bkc[u] = g_rl_f64(bk[cq], cc0 + u);
#pragma unroll
                         for (int q = 0; q < NQ; ++q)
                           if (q < qact)
                           {
                             if (nv >= L::U)
                             {
#pragma unroll
                               for (int u = 0; u < L::U; ++u)
                               {
                                 const double pr = bkc[u] * (double)R.x[u][q];
                                 g[q]            = g[q] + pr;
                               }
                             }
                             else
                             {
#pragma unroll
                               for (int u = 0; u < L::U; ++u)
                                 if (u < nv)
                                 {
                                   const double pr = bkc[u] * (double)R.x[u][q];
                                   g[q]            = g[q] + pr;
                                 }
                             }
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
  static constexpr int IPR = (NQ + 1) / 2;
  const char *p;
  long stride;
  LsWin win;  // bytes [0, len) of every row
  int nrows, srq;
  unsigned lane16;
  int lane;
  unsigned off[NQ];
  double (&acc)[NQ];
  const double (&rd)[NQ];
  unsigned long long bmask[NQ];  // lanes whose position lies in [start, last]
  bool diag;
  int kappa;
  struct Regs
  {
    double m[L::U][NQ];
  };
  template <bool FULL> __device__ __forceinline__ void issue(unsigned dst)
  {
    if (FULL || srq < nrows)
    {
      ls_dma_win<IPR>(p, win, dst, lane16);
      p += stride;
    }
    else
      ls_dma_row<IPR>(p - stride, 0, dst, lane16);
    ++srq;
  }
  __device__ __forceinline__ void load(Regs &R, unsigned a) const
  {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int u = 0; u < L::U; ++u)
        R.m[u][q] = *(const double *)(fphip_lds + (a + off[q] + (unsigned)(u * L::ROWB)));
  }
  __device__ __forceinline__ void compute(const Regs &R, int s0)
  {
    dispatch_chunk<NQ>(s0,
                       [&](auto kq_, int kk0)
                       {
                         constexpr int kq = decltype(kq_)::value;
#pragma unroll
                         for (int u = 0; u < L::U; ++u)
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
                                 double m = R.m[u][q];
                                 if (diag)
                                   m = (lane + 64 * q == kappa) ? muk : m;
                                 const double t  = m * rk;
                                 const double uu = acc[q] - t;
                                 acc[q]          = ls_sel_f64(mk, uu, acc[q]);
                               }
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
// babai's sweep, lll.cpp:202-214: rows j = kappa-1 .. sr_start of mu, descending; lane k owns
// babai_mu[k].  The stream is aligned to blocks of U rows: row s is j = jtop - s with
// jtop = (kappa - 1) | (U - 1), so that a block never straddles a chunk of 64.
// ---------------------------------------------------------------------------------------------------
template <int NQ> struct SweepPh
{
  using L = LStream<NQ>;
  static constexpr int KIND = LS_SWEEP;
  static constexpr int IPR = (NQ + 1) / 2;
  const char *mu;  // T.mu
  long stride;     // ldd * 8
  int jtop, kappa, sr_start, srq;
  unsigned lane16;
  int lane;
  const SlotMap<NQ> &M;
  double (&bm)[NQ];
  double (&xs)[NQ];             // the multipliers X_j (lane j), 0 where none
  const int (&e)[NQ];
  unsigned long long (&nz)[NQ];  // rows of each chunk with X_j != 0
  unsigned long long srmask[NQ];  // lanes k >= sr_start
  struct Regs
  {
    double m[L::U][NQ];
  };
  template <bool FULL> __device__ __forceinline__ void issue(unsigned dst)
  {
    const int j = jtop - srq;
    ++srq;
    if (j < kappa && j > sr_start)
    {  // mu(j, 0 .. j-1); row sr_start itself has nothing to its left that is reduced
      const int slot = M.phys(j);
      ls_dma_row<IPR>(mu + (long)slot * stride, j * 8, dst, lane16);
    }
    else
      ls_dma_row<IPR>(mu, 0, dst, lane16);
  }
  __device__ __forceinline__ void load(Regs &R, unsigned a) const
  {
    const unsigned la = a + (unsigned)lane * 8u;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int u = 0; u < L::U; ++u)
        R.m[u][q] = *(const double *)(fphip_lds + (la + (unsigned)(q * 512 + u * L::ROWB)));
  }
  __device__ __forceinline__ void compute(const Regs &R, int s0)
  {
    const int jhi = jtop - s0;
    dispatch_chunk<NQ>(jhi,
                       [&](auto jq_, int jjhi)
                       {
                         constexpr int jq = decltype(jq_)::value;
#pragma unroll
                         for (int u = 0; u < L::U; ++u)
                         {
                           const int j = jhi - u;
                           if (j < kappa && j >= sr_start)
                           {
                             const int jj     = jjhi - u;
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
                                 const double t  = X * R.m[u][q];
                                 const double uu = bm[q] - t;
                                 bm[q]           = ls_sel_f64(mk, uu, bm[q]);
                               }
                             }
                           }
                         }
                       });
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
  static constexpr int IPR = (NQ + 1) / 2;
  const char *b;  // T.b
  long stride;    // ldn * 8
  LsWin win;      // bytes [0, n * 8) of every row
  unsigned lane16;
  int lane;
  const SlotMap<NQ> &M;
  long long (&bv)[NQ];
  const long long (&lxv)[NQ];  // multiplier of row j in lane j
  RowCursor<NQ> ic, cc;        // rows still to request / to apply
  struct Regs
  {
    typename std::conditional<SMALL, int, long long>::type w[L::U][NQ];
  };
  template <bool FULL> __device__ __forceinline__ void issue(unsigned dst)
  {
    const int j = ic.next();
    if (FULL || j >= 0)
    {
      const int slot = M.phys(j);
      ls_dma_win<IPR>(b + (long)slot * stride, win, dst, lane16);
    }
    else
      ls_dma_row<IPR>(b, 0, dst, lane16);
  }
  __device__ __forceinline__ void load(Regs &R, unsigned a) const
  {
    using E           = typename std::conditional<SMALL, int, long long>::type;
    const unsigned la = a + (unsigned)lane * 8u;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int u = 0; u < L::U; ++u)
        R.w[u][q] = *(const E *)(fphip_lds + (la + (unsigned)(q * 512 + u * L::ROWB)));
  }
  __device__ __forceinline__ void compute(const Regs &R, int)
  {
#pragma unroll
    for (int u = 0; u < L::U; ++u)
    {
      const int j = cc.next();
      if (j >= 0)
      {
        long long lx = 0;
        dispatch_chunk<NQ>(j, [&](auto jq_, int jj) { lx = g_rl_i64(lxv[decltype(jq_)::value], jj); });
        if constexpr (SMALL)
        {
          const int s = (int)lx;
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            bv[q] = bv[q] + (long long)R.w[u][q] * (long long)s;
        }
        else
        {
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            bv[q] = (long long)((unsigned long long)bv[q] + (unsigned long long)R.w[u][q] * (unsigned long long)lx);
        }
      }
    }
  }
};

}  // namespace fphip
#endif
