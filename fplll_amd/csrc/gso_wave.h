// gso_wave.h — wave-level device building blocks shared by gso_kernel.hip and lll_kernel.hip:
// lane helpers, the LDS-DMA ring, update_gso_row / babai for one lattice owned by one wavefront.
// (Device code only; see gso_kernel.hip for the design notes and the reference citations.)
#ifndef FPHIP_GSO_WAVE_H
#define FPHIP_GSO_WAVE_H

#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#include <utility>

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
  int d, n, ldd, ldn, row_expo_on;
  long long *b;
  long long *u = nullptr;  // [d][ldd] transformation rows (LLL kernel, optional): same slots as b
  double *bfT, *mu, *muT, *r, *rdg;
  long long *rexp;
  // narrow mirrors (sweep kernels only; nullptr elsewhere): while every entry of the lattice is
  // below 2^24 in magnitude, bf fits a float exactly and b an int32 — the Gram pass and the
  // integer AXPY then stream 4-byte rows (half the HBM bytes) and widen in registers, bit-exactly
  float *bfT32;      // [n][ldd]
  int *b32;          // [d][ldn]
  int *narrow_flag;  // [d] per row: every entry of the row is below 2^24 in magnitude
  int np;            // narrow prefix: rows 0..np-1 are all narrow (wave-uniform)
  // slot-mode kernels (LLL / BKZ: rows never move, a slot table maps positions): 1 while EVERY row of the
  // lattice is below 2^24 — the Gram passes then stream the float mirror (half the bytes of bfT, exact);
  // kept by store_row_and_refloat, cleared for good by the first wide row.  0 in the other kernels.
  int f32ok;
  int wide_ring;     // sweep kernel: rows that are not narrow go through the LDS-DMA ring as 8-byte rows (gso_sweep2.hip)
  int lane;
  double murow[NQ];  // mu(kappa, j) of the row last updated (lane j)
  double rrow[NQ];   // r(kappa, j)  of the row last updated (lane j)
  double bfk[NQ];    // bf(kappa, c) of the row last updated (lane c)
};

// Row position -> physical row slot.  The sweep kernels keep rows in place (identity); the LLL kernel
// moves rows by rotating a lane-resident slot table (MatGSO::move_row rotates row POINTERS,
// fplll/gso.cpp:289-366, nr/matrix.h rotate_left/right).
struct IdentityMap
{
  static constexpr bool kNarrowMirrors = true;  // the sweep kernels keep the 4-byte mirrors
  __device__ __forceinline__ int phys(int j) const { return j; }
  template <int Q> __device__ __forceinline__ int lane_phys(int lane) const { return lane + 64 * Q; }
};
template <int NQ> struct SlotMap
{
  static constexpr bool kNarrowMirrors = false;
  int sl[NQ];  // lane j of chunk q holds the slot of row position j + 64 q
  __device__ __forceinline__ int phys(int j) const
  {
    int r = 0;
    if constexpr (NQ == 1)
      r = __builtin_amdgcn_readlane(sl[0], j);
    else
    {
      const int q = j >> 6, jj = j & 63;
#pragma unroll
      for (int t = 0; t < NQ; ++t)
        if (q == t)
          r = __builtin_amdgcn_readlane(sl[t], jj);
    }
    return r;
  }
  template <int Q> __device__ __forceinline__ int lane_phys(int) const { return sl[Q]; }
};

template <int NQ, class Map, int Q = 0>
__device__ __forceinline__ void fill_lane_phys(const Map &m, int lane, int (&out)[NQ])
{
  if constexpr (Q < NQ)
  {
    out[Q] = m.template lane_phys<Q>(lane);
    fill_lane_phys<NQ, Map, Q + 1>(m, lane, out);
  }
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA ring ("chain" loops).  Every hot loop of this kernel has the shape
//     for step s (in a fixed order):  v[q] = ROW_s[lane + 64 q];  state[q] = f(state[q], v[q], scalar_s)
// where ROW_s is a contiguous row in HBM whose address does not depend on the state, while scalar_s
// does (it is read from a lane of the state with v_readlane).  The rows are streamed into a per-wave
// ring in LDS with global_load_lds_dwordx4 (16 B per lane, 1 KiB per instruction, no VGPRs), R-1
// steps ahead of the consumer; the consumer waits with a COUNTED s_waitcnt vmcnt(pending·IPS) — loads
// retire in order — and reads its 8 bytes with ds_read_b64.  A wave therefore keeps up to
// (R-1)·IPS KiB of HBM reads in flight; that, times the waves per CU, is what hides HBM latency
// (a register-buffered version of the same loops reached 24 % of the HBM roofline; the chain
// itself is a handful of VALU ops per step).
//   * hipcc neither counts nor orders these instructions (cdna_hip_programming.md §5.7): the ring
//     code owns every vmcnt wait, and NO other vector-memory instruction (including scratch spills)
//     may be issued between ring_begin() and ring_end() — ScratchSize must stay 0 for this kernel.
//   * rows start 16-byte aligned: leading dimensions are padded to even (ldd, ldn).
// ---------------------------------------------------------------------------------------------
#ifndef FPHIP_GSO_RING
#define FPHIP_GSO_RING 6
#endif

__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst)
{
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

// Make the compiler finish (wait for) an ordinary load NOW: an empty asm that reads and writes the
// value.  hipcc places its own s_waitcnt vmcnt(0) at the FIRST USE of a loaded value; if that first
// use sits inside a ring loop the wait would drain the DMA pipe on every iteration.
__device__ __forceinline__ void settle(double &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void settle(long long &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void settle(int &x) { asm volatile("" : "+v"(x)); }

template <int N> __device__ __forceinline__ void wait_vmcnt()
{
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct RowDesc
{
  const void *ptr;  // wave-uniform, 16-byte aligned start of the row
  int lo, hi;       // bytes [lo, hi) of the row are needed
};

template <int Q> struct IC
{
  static constexpr int value = Q;
};

// f(IC<q>, idx - 64 q) for the (wave-uniform) chunk q = idx >> 6: gives loop bodies a COMPILE-TIME
// register index for the chunk that owns step idx.
template <int NQ, class F> __device__ __forceinline__ void dispatch_chunk(int idx, F f)
{
  if constexpr (NQ == 1)
  {
    f(IC<0>{}, idx);
  }
  else
  {
    const int q = idx >> 6;
    if (q == 0)
      f(IC<0>{}, idx);
    else if (q == 1)
      f(IC<1>{}, idx - 64);
    else if constexpr (NQ >= 3)
    {
      if (q == 2)
        f(IC<2>{}, idx - 128);
      else if constexpr (NQ >= 4)
        f(IC<3>{}, idx - 192);
    }
  }
}

// RR = ring depth: FPHIP_GSO_RING for the HBM-bound sweep kernels (more waves per CU matter more
// than depth), FPHIP_RING_REDUCE for the reduction drivers (LLL / BKZ / HLLL), which run few waves
// per CU for a long time and are bound by the latency of each streamed row.
template <int NQ, int IPS, int RR = FPHIP_GSO_RING> struct Ring
{
  static constexpr int R     = RR;
  static constexpr int SLOT  = IPS * 1024;
  static constexpr int AHEAD = (R - 1 < 7 ? R - 1 : 7);  // rows kept in flight behind the consumer
  unsigned base;  // LDS byte address of this wave's ring (wave-uniform)
  int lane;
  int head, tail;
  int ahead;  // rows issued and not yet consumed

  // Start a new stream: nothing of ours is in flight; also retires every older vector-memory
  // operation (stores of the previous phase) so that the counted waits below only see ring loads.
  __device__ __forceinline__ void reset()
  {
    wait_vmcnt<0>();
    head = tail = 0;
    ahead       = 0;
  }

  // stream bytes [lo, hi) of one row.  Every one of the IPS instructions is issued, so that the
  // vmcnt arithmetic of the consumer holds; lanes outside the window move no data.  An instruction
  // whose whole 1 KiB span lies outside the window is issued by lane 0 alone, re-reading the first
  // 16 bytes of the window (a line the other instruction fetches anyway — no extra HBM traffic)
  // into its own, unused, part of the slot.
  // NI = number of 1 KiB instructions issued for the row (IPS for 8-byte rows, IPS32 for the
  // 4-byte mirrors; rows prefetched for the NEXT phase always get IPS, see run_with)
  template <int NI = IPS> __device__ __forceinline__ void issue(const RowDesc &r)
  {
    const char *g      = (const char *)r.ptr + lane * 16;
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(base + head * SLOT));
#pragma unroll
    for (int i = 0; i < NI; ++i)
    {
      const int off  = lane * 16 + i * 1024;
      const bool any = r.hi > r.lo && r.hi > i * 1024 && r.lo < (i + 1) * 1024;  // wave-uniform
      if (any)
      {
        if (off + 16 > r.lo && off < r.hi)
          glds16(g + i * 1024, dst + i * 1024);
      }
      else if (lane == 0)
      {
        glds16((const char *)r.ptr + (r.lo & ~15), dst + i * 1024);
      }
    }
    head = (head + 1 == R) ? 0 : head + 1;
    ++ahead;
  }

  template <int W> __device__ __forceinline__ void read(double (&v)[NQ], unsigned addr)
  {
    // wait until at most P newer rows are outstanding, then fetch this lane's elements
    if constexpr (NQ == 1)
      asm volatile("s_waitcnt vmcnt(%2)\n\tds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(v[0])
                   : "v"(addr), "n"(W)
                   : "memory");
    else if constexpr (NQ == 2)
      asm volatile("s_waitcnt vmcnt(%3)\n\tds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:512\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(v[0]), "=&v"(v[1])
                   : "v"(addr), "n"(W)
                   : "memory");
    else if constexpr (NQ == 3)
      asm volatile("s_waitcnt vmcnt(%4)\n\tds_read_b64 %0, %3\n\tds_read_b64 %1, %3 offset:512\n\t"
                   "ds_read_b64 %2, %3 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2])
                   : "v"(addr), "n"(W)
                   : "memory");
    else
      asm volatile("s_waitcnt vmcnt(%5)\n\tds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:512\n\t"
                   "ds_read_b64 %2, %4 offset:1024\n\tds_read_b64 %3, %4 offset:1536\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                   : "v"(addr), "n"(W)
                   : "memory");
  }

  // 4-byte elements (narrow rows): this lane's element of each chunk, raw bits
  template <int W> __device__ __forceinline__ void read32(unsigned (&w)[NQ], unsigned addr)
  {
    if constexpr (NQ == 1)
      asm volatile("s_waitcnt vmcnt(%2)\n\tds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(w[0])
                   : "v"(addr), "n"(W)
                   : "memory");
    else if constexpr (NQ == 2)
      asm volatile("s_waitcnt vmcnt(%3)\n\tds_read_b32 %0, %2\n\tds_read_b32 %1, %2 offset:256\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(w[0]), "=&v"(w[1])
                   : "v"(addr), "n"(W)
                   : "memory");
    else if constexpr (NQ == 3)
      asm volatile("s_waitcnt vmcnt(%4)\n\tds_read_b32 %0, %3\n\tds_read_b32 %1, %3 offset:256\n\t"
                   "ds_read_b32 %2, %3 offset:512\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2])
                   : "v"(addr), "n"(W)
                   : "memory");
    else
      asm volatile("s_waitcnt vmcnt(%5)\n\tds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:256\n\t"
                   "ds_read_b32 %2, %4 offset:512\n\tds_read_b32 %3, %4 offset:768\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3])
                   : "v"(addr), "n"(W)
                   : "memory");
  }

  // gather variant of read(): chunk q of this lane takes the element at byte offset a[q] of the slot
  template <int W> __device__ __forceinline__ void readg(double (&v)[NQ], const unsigned (&a)[NQ])
  {
    if constexpr (NQ == 1)
      asm volatile("s_waitcnt vmcnt(%2)\n\tds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(v[0])
                   : "v"(a[0]), "n"(W)
                   : "memory");
    else if constexpr (NQ == 2)
      asm volatile("s_waitcnt vmcnt(%4)\n\tds_read_b64 %0, %2\n\tds_read_b64 %1, %3\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(v[0]), "=&v"(v[1])
                   : "v"(a[0]), "v"(a[1]), "n"(W)
                   : "memory");
    else if constexpr (NQ == 3)
      asm volatile("s_waitcnt vmcnt(%6)\n\tds_read_b64 %0, %3\n\tds_read_b64 %1, %4\n\t"
                   "ds_read_b64 %2, %5\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2])
                   : "v"(a[0]), "v"(a[1]), "v"(a[2]), "n"(W)
                   : "memory");
    else
      asm volatile("s_waitcnt vmcnt(%8)\n\tds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\t"
                   "ds_read_b64 %2, %6\n\tds_read_b64 %3, %7\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                   : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "n"(W)
                   : "memory");
  }

  // gather variant of read32()
  template <int W> __device__ __forceinline__ void readg32(unsigned (&w)[NQ], const unsigned (&a)[NQ])
  {
    if constexpr (NQ == 1)
      asm volatile("s_waitcnt vmcnt(%2)\n\tds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(w[0])
                   : "v"(a[0]), "n"(W)
                   : "memory");
    else if constexpr (NQ == 2)
      asm volatile("s_waitcnt vmcnt(%4)\n\tds_read_b32 %0, %2\n\tds_read_b32 %1, %3\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(w[0]), "=&v"(w[1])
                   : "v"(a[0]), "v"(a[1]), "n"(W)
                   : "memory");
    else if constexpr (NQ == 3)
      asm volatile("s_waitcnt vmcnt(%6)\n\tds_read_b32 %0, %3\n\tds_read_b32 %1, %4\n\t"
                   "ds_read_b32 %2, %5\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2])
                   : "v"(a[0]), "v"(a[1]), "v"(a[2]), "n"(W)
                   : "memory");
    else
      asm volatile("s_waitcnt vmcnt(%8)\n\tds_read_b32 %0, %4\n\tds_read_b32 %1, %5\n\t"
                   "ds_read_b32 %2, %6\n\tds_read_b32 %3, %7\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3])
                   : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "n"(W)
                   : "memory");
  }

  // One phase of `cnt` steps.  Rows [0, ahead) of it may already be in flight (prefetched by the
  // previous phase).  While consuming, keep the pipe full: first with this phase's remaining rows,
  // then with the first rows of the NEXT phase (ncnt rows, nrow(s)) — phases are chained without
  // draining the pipe whenever no other vector-memory instruction separates them.
  //
  // The loop is split by pipeline state so that the steady state is branch-free: while a row can
  // be issued for every row consumed, exactly AHEAD newer rows are behind the one being read and
  // the wait is the compile-time s_waitcnt vmcnt(AHEAD*IPS); only the last <= AHEAD steps of an
  // unchained phase take the dynamic wait.
  template <class RowA, class BodyF, class RowB>
  __device__ __forceinline__ void run(int cnt, RowA row, BodyF body, int ncnt, RowB nrow)
  {
    run_with(cnt, row, body, ncnt, nrow, PlainFetch{});
  }
  static constexpr int IPS32 = (NQ + 3) / 4;  // 1 KiB instructions per 4-byte row (256 elements each)
  struct PlainFetch
  {
    static constexpr int IPR = IPS;
  };
  struct GatherFetch
  {
    static constexpr int IPR = IPS;
    const unsigned (&off)[NQ];
  };
  struct F32Fetch  // rows of float, widened to double (exact)
  {
    static constexpr int IPR = IPS32;
  };
  struct GatherF32Fetch  // rows of float read back by slot (byte offsets slot * 4), widened to double
  {
    static constexpr int IPR = IPS32;
    const unsigned (&off)[NQ];
  };
  struct I32Fetch  // rows of int32, delivered as the bit pattern of the sign-extended int64
  {
    static constexpr int IPR = IPS32;
  };
  template <int P> __device__ __forceinline__ void consume(double (&v)[NQ], const F32Fetch &)
  {
    const unsigned addr = base + tail * SLOT + lane * 4;
    tail                = (tail + 1 == R) ? 0 : tail + 1;
    --ahead;
    unsigned w[NQ];
    read32<P * IPS32>(w, addr);
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      v[q] = (double)__uint_as_float(w[q]);
  }
  template <int P> __device__ __forceinline__ void consume(double (&v)[NQ], const GatherF32Fetch &g)
  {
    const unsigned sb = base + tail * SLOT;
    tail              = (tail + 1 == R) ? 0 : tail + 1;
    --ahead;
    unsigned w[NQ], a[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      a[q] = sb + g.off[q];
    readg32<P * IPS32>(w, a);
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      v[q] = (double)__uint_as_float(w[q]);
  }
  template <int P> __device__ __forceinline__ void consume(double (&v)[NQ], const I32Fetch &)
  {
    const unsigned addr = base + tail * SLOT + lane * 4;
    tail                = (tail + 1 == R) ? 0 : tail + 1;
    --ahead;
    unsigned w[NQ];
    read32<P * IPS32>(w, addr);
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      v[q] = __longlong_as_double((long long)(int)w[q]);
  }
  template <int P> __device__ __forceinline__ void consume(double (&v)[NQ], const PlainFetch &)
  {
    const unsigned addr = base + tail * SLOT + lane * 8;
    tail                = (tail + 1 == R) ? 0 : tail + 1;
    --ahead;
    read<P * IPS>(v, addr);
  }
  template <int P> __device__ __forceinline__ void consume(double (&v)[NQ], const GatherFetch &g)
  {
    unsigned a[NQ];
    const unsigned sb = base + tail * SLOT;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      a[q] = sb + g.off[q];
    tail = (tail + 1 == R) ? 0 : tail + 1;
    --ahead;
    readg<P * IPS>(v, a);
  }
  // dynamic number of newer rows in flight (the drain of a phase)
  template <class FetchP> __device__ __forceinline__ void consume_dyn(double (&v)[NQ], const FetchP &fp)
  {
    switch (ahead - 1)
    {
    case 0: consume<0>(v, fp); break;
    case 1: consume<1>(v, fp); break;
    case 2: consume<2>(v, fp); break;
    case 3: consume<3>(v, fp); break;
    case 4: consume<4>(v, fp); break;
    case 5: consume<5>(v, fp); break;
    case 6: consume<6>(v, fp); break;
    default: consume<7>(v, fp); break;
    }
  }
  template <class RowA, class BodyF, class RowB, class FetchP>
  __device__ __forceinline__ void run_with(int cnt, RowA row, BodyF body, int ncnt, RowB nrow,
                                           const FetchP &fp)
  {
    // The waits of this phase count FetchP::IPR instructions per newer row.  That is safe as long
    // as every row in flight has AT LEAST that many: this phase's own rows are issued with exactly
    // IPR, rows prefetched for the next phase always with the full IPS (>= any IPR).
    constexpr int NI = FetchP::IPR;
    int issued  = ahead;  // rows of this phase issued so far
    int nissued = 0;      // rows of the next phase issued so far
    // fill the pipe
    while (ahead <= AHEAD)
    {
      if (issued < cnt)
        issue<NI>(row(issued++));
      else if (nissued < ncnt)
        issue<IPS>(nrow(nissued++));
      else
        break;
    }
    int s = 0;
    if (ahead == AHEAD + 1)
    {
      // steady state, refilled from this phase
#pragma unroll 1
      for (; s < cnt && issued < cnt; ++s)
      {
        double v[NQ];
        consume<AHEAD>(v, fp);
        issue<NI>(row(issued++));
        body(s, v);
      }
      // steady state, refilled from the next phase
#pragma unroll 1
      for (; s < cnt && nissued < ncnt; ++s)
      {
        double v[NQ];
        consume<AHEAD>(v, fp);
        issue<IPS>(nrow(nissued++));
        body(s, v);
      }
    }
    // drain (nothing left to issue, or the phase is shorter than the pipe)
#pragma unroll 1
    for (; s < cnt; ++s)
    {
      double v[NQ];
      consume_dyn(v, fp);
      body(s, v);
    }
  }
  template <class RowA, class BodyF> __device__ __forceinline__ void run(int cnt, RowA row, BodyF body)
  {
    run(cnt, row, body, 0, row);
  }
};

// update_gso_row(kappa, last) recomputed from column 0 (identical values: every input is unchanged
// since the row was invalidated).  Returns false on a non-finite mu (RED_GSO_FAILURE).
template <int NQ, int IPS, int RR>
__device__ bool update_row(Lattice<NQ> &T, Ring<NQ, IPS, RR> &ring, int kappa, int last)
{
  const int n = T.n, lane = T.lane, ldd = T.ldd;
  const int qact = (last >> 6) + 1;  // chunks holding a lane j <= last
  double bk[NQ], acc[NQ], rd[NQ];
  bool inrow[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    const int c = lane + 64 * q;
    bk[q]       = (c < n) ? T.bfT[(size_t)c * ldd + kappa] : 0.0;
    acc[q]      = 0.0;
    rd[q]       = (c < kappa) ? T.rdg[c] : 1.0;
    inrow[q]    = c <= last;
  }
  const int need_bytes = (last + 1) * 8;  // lanes j <= last of a row
  const bool narrow    = T.np > last;     // rows 0..last are read: wave-uniform
  auto gram_row        = [&](int c)
  {
    return narrow ? RowDesc{T.bfT32 + (size_t)c * ldd, 0, (last + 1) * 4}
                  : RowDesc{T.bfT + (size_t)c * ldd, 0, need_bytes};
  };
  auto rec_row  = [&](int k) { return RowDesc{T.muT + (size_t)k * ldd, (k + 1) * 8, need_bytes}; };
  bool ok       = true;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    settle(bk[q]);
    settle(rd[q]);
  }
  ring.reset();
  // ---- Gram row: g(kappa,j) = bf_kappa . bf_j, columns in ascending order (numvect.h:386-396);
  //      the recurrence's first mu columns are prefetched behind it
  auto gram_body = [&](int c, const double(&v)[NQ])
  {
    dispatch_chunk<NQ>(c,
                       [&](auto cq, int cc)
                       {
                         const double bkc = g_rl_f64(bk[decltype(cq)::value], cc);
#pragma unroll
                         for (int q = 0; q < NQ; ++q)
                           if (q < qact)
                           {
                             const double p = bkc * v[q];
                             acc[q]         = (c == 0) ? p : acc[q] + p;
                           }
                       });
  };
  if (narrow)
    ring.run_with(n, gram_row, gram_body, last + 1, rec_row, typename Ring<NQ, IPS, RR>::F32Fetch{});
  else
    ring.run(n, gram_row, gram_body, last + 1, rec_row);
  // ---- recurrence, gso_interface.cpp:143-158, column-oriented
  ring.run(last + 1, rec_row,
           [&](int k, const double(&v)[NQ])
           {
             dispatch_chunk<NQ>(
                 k,
                 [&](auto kq_, int kk)
                 {
                   constexpr int kq = decltype(kq_)::value;
                   const double rk  = g_rl_f64(acc[kq], kk);  // r(kappa,k) is final
                   double muk       = 0.0;
                   if (last == kappa && k < kappa)  // only the diagonal lane j == kappa needs it
                     muk = rk / g_rl_f64(rd[kq], kk);  // mu(kappa,k) = r(kappa,k) / r(k,k)
#pragma unroll
                   for (int q = kq; q < NQ; ++q)
                     if (q < qact)
                     {
                       const int j = lane + 64 * q;
                       // chunks above kq hold only rows j > k; the diagonal chunk needs the test
                       const bool on = inrow[q] && (q > kq || lane > kk);
                       if (on)
                       {
                         const double m = (j == kappa) ? muk : v[q];
                         acc[q]         = acc[q] - m * rk;
                       }
                     }
                 });
           });
  // ---- store the row
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    const int j = lane + 64 * q;
    T.murow[q]  = 0.0;
    T.rrow[q]   = acc[q];
    T.bfk[q]    = bk[q];
    if (j <= last)
    {
      T.r[(size_t)kappa * ldd + j] = acc[q];
      if (j < kappa)
      {
        const double m = acc[q] / rd[q];  // mu(kappa,j) = r(kappa,j) / r(j,j), gso_interface.cpp:154
        if (!isfinite(m))
          ok = false;
        T.murow[q]                     = m;
        T.mu[(size_t)kappa * ldd + j]  = m;
        T.muT[(size_t)j * ldd + kappa] = m;
      }
      else
      {
        T.rdg[kappa] = acc[q];
      }
    }
  }
  return __all(ok);
}

// update_gso_row(kappa, kappa) right after update_gso_row(kappa, kappa-1): gso_valid_cols = kappa,
// so the reference computes ONLY column j = kappa (gso_interface.cpp:141-152):
//   g(kappa,kappa) = bf_kappa . bf_kappa (columns ascending), r(kappa,kappa) = g - sum_k mu(kappa,k) r(kappa,k)
// (k ascending).  Both operands are still in registers (lane-distributed), so this is two short
// v_readlane chains and no memory traffic.
template <int NQ> __device__ void finish_diag(Lattice<NQ> &T, int kappa)
{
  const int n = T.n, ldd = T.ldd;
  double g = 0.0;
  for (int c = 0; c < n; ++c)
  {
    double a = 0.0;
    dispatch_chunk<NQ>(c, [&](auto cq, int cc) { a = g_rl_f64(T.bfk[decltype(cq)::value], cc); });
    const double p = a * a;
    g              = (c == 0) ? p : g + p;
  }
  for (int k = 0; k < kappa; ++k)
  {
    double m = 0.0, r = 0.0;
    dispatch_chunk<NQ>(k,
                       [&](auto kq, int kk)
                       {
                         m = g_rl_f64(T.murow[decltype(kq)::value], kk);
                         r = g_rl_f64(T.rrow[decltype(kq)::value], kk);
                       });
    g = g - m * r;
  }
  if (T.lane == 0)
  {
    T.r[(size_t)kappa * ldd + kappa] = g;
    T.rdg[kappa]                     = g;
  }
}

// Store the integer row held in registers into slot pk and re-float it: MatGSO::update_bf,
// gso.cpp:24-48 (mantissa/exponent per entry, renormalised to the row maximum).
template <int NQ, bool MIRRORS, bool OUT>
__device__ __forceinline__ void store_row_and_refloat_impl(Lattice<NQ> &T, int pk, const long long (&bv)[NQ],
                                                           double (&fout)[NQ], long long &eout)
{
  const int n = T.n, lane = T.lane, ldd = T.ldd, ldn = T.ldn;
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
      T.b[(size_t)pk * ldn + c] = bv[q];
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
  bool wide = false;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    const int c = lane + 64 * q;
    if (c < n)
    {
      const double f              = T.row_expo_on ? ldexp(cm[q], ce[q] - emax) : cm[q];
      T.bfT[(size_t)c * ldd + pk] = f;
      if constexpr (OUT)
        fout[q] = f;
      if constexpr (MIRRORS)
      {  // narrow mirrors (exact while |entry| < 2^24)
        T.bfT32[(size_t)c * ldd + pk] = (float)f;
        T.b32[(size_t)pk * ldn + c]   = (int)bv[q];
        wide |= (bv[q] >= (1ll << 24) || bv[q] <= -(1ll << 24));
      }
      else if (T.f32ok)
      {  // slot-mode kernels: the float mirror of bf only (their Gram passes stream it)
        T.bfT32[(size_t)c * ldd + pk] = (float)f;
        wide |= (bv[q] >= (1ll << 24) || bv[q] <= -(1ll << 24));
      }
    }
  }
  if constexpr (!MIRRORS)
  {
    if (T.f32ok && __any(wide))
      T.f32ok = 0;  // a row left the exact range of the float mirror: the 8-byte rows from now on
  }
  if constexpr (MIRRORS)
  {  // keep the row's flag and the narrow prefix (identity layout: slot pk = position pk)
    const bool w = __any(wide);
    if (lane == 0)
      T.narrow_flag[pk] = w ? 0 : 1;
    if (w)
      T.np = min(T.np, pk);
    else if (pk == T.np)
    {
      int p = pk + 1;
      while (p < T.d && __builtin_amdgcn_readfirstlane(T.narrow_flag[p]) != 0)
        ++p;
      T.np = p;
    }
  }
  if (lane == 0)
    T.rexp[pk] = T.row_expo_on ? (long long)emax : 0;
  if constexpr (OUT)
    eout = T.row_expo_on ? (long long)emax : 0;
}
template <int NQ, bool MIRRORS = false>
__device__ __forceinline__ void store_row_and_refloat(Lattice<NQ> &T, int pk, const long long (&bv)[NQ])
{
  double f[NQ];
  long long e;
  store_row_and_refloat_impl<NQ, MIRRORS, false>(T, pk, bv, f, e);
}
// … and hand back the re-floated row (lane c = column c) and its exponent
template <int NQ, bool MIRRORS = false>
__device__ __forceinline__ void store_row_and_refloat(Lattice<NQ> &T, int pk, const long long (&bv)[NQ],
                                                      double (&fout)[NQ], long long &eout)
{
  store_row_and_refloat_impl<NQ, MIRRORS, true>(T, pk, bv, fout, eout);
}

// LLLReduction::babai(kappa, kappa, 0).  1 ok, 0 GSO failure, -1 babai failure, -2 multiplier.
// `upd(kappa, last)` brings row kappa of the GSO up to column `last` (and leaves mu/r of the row in
// T.murow / T.rrow); `after(kappa)` runs after b_kappa changed (row_op_end's invalidations).
template <int NQ, int IPS, int RR, class Map, class Upd, class After>
__device__ __forceinline__ int babai_impl(Lattice<NQ> &T, Ring<NQ, IPS, RR> &ring, int kappa, double eta,
                                          const Map &map, Upd upd, After after, int sr_start = 0)
{
  // sr_start = size_reduction_start (lll.cpp:167): only columns j in [sr_start, kappa) are reduced
  const int pk = map.phys(kappa);  // physical slot of row kappa
  const int n = T.n, lane = T.lane, ldd = T.ldd, ldn = T.ldn;
  long long max_expo = LLONG_MAX;
  for (int iter = 0;; ++iter)
  {
    if (!upd(kappa, kappa - 1))
      return 0;
    const long long rexpk = T.rexp[pk];
    int e[NQ], lphys[NQ];
    bool need = false;
    int mexp  = INT_MIN;
    fill_lane_phys<NQ>(map, lane, lphys);
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      const int j = lane + 64 * q;
      e[q]        = 0;
      if (j < kappa)
      {
        e[q]           = (int)(rexpk - T.rexp[lphys[q]]);
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
    double bm[NQ];
    long long xl[NQ], bv[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      const int c = lane + 64 * q;
      bm[q]       = T.murow[q];
      xl[q]       = 0;
      bv[q]       = (c < n) ? T.b[(size_t)pk * ldn + c] : 0;
    }
    bool too_big = false;
    // step s of both loops handles row j = kappa-1-s (descending, lll.cpp:202)
    auto mu_row = [&](int s)
    {
      const int j = kappa - 1 - s;
      return RowDesc{T.mu + (size_t)map.phys(j) * ldd, 0, j * 8};  // mu(j,k) is needed for k < j
    };
    const bool narrow = T.np >= kappa;  // rows below kappa are read: wave-uniform
    auto b_row        = [&](int s)
    {
      const size_t ro = (size_t)map.phys(kappa - 1 - s) * ldn;
      return narrow ? RowDesc{T.b32 + ro, 0, n * 4} : RowDesc{T.b + ro, 0, n * 8};
    };
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      settle(e[q]);
      settle(bv[q]);
      settle(bm[q]);
    }
    ring.reset();
    // ---- lll.cpp:202-220: lane k owns babai_mu[k]; the basis rows of the integer AXPY are
    //      prefetched behind the sweep
    const int nsteps = kappa - sr_start;  // rows j = kappa-1 … sr_start
    ring.run(
        nsteps, mu_row,
        [&](int s, const double(&v)[NQ])
        {
          const int j = kappa - 1 - s;
          dispatch_chunk<NQ>(
              j,
              [&](auto jq_, int jj)
              {
                constexpr int jq = decltype(jq_)::value;
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
                  for (int q = 0; q <= jq; ++q)
                  {
                    // chunks below jq hold only k < j; the chunk of j itself needs the test
                    if ((q < jq || lane < jj) && lane + 64 * q >= sr_start)
                    {
                      const double t = X * v[q];
                      bm[q]          = bm[q] - t;
                    }
                  }
                }
              });
        },
        nsteps, b_row);
    // ---- integer AXPY on row kappa (row_add / row_sub / row_addmul_si, gso.cpp:84-158)
    auto axpy_body = [&](int s, const double(&v)[NQ])
    {
      const int j = kappa - 1 - s;
      dispatch_chunk<NQ>(j,
                         [&](auto jq_, int jj)
                         {
                           const long long lx = g_rl_i64(xl[decltype(jq_)::value], jj);
                           if (lx != 0)
                           {
#pragma unroll
                             for (int q = 0; q < NQ; ++q)
                               bv[q] = (long long)((unsigned long long)bv[q] +
                                                   (unsigned long long)__double_as_longlong(v[q]) *
                                                       (unsigned long long)lx);
                           }
                         });
    };
    if (narrow)
      ring.run_with(nsteps, b_row, axpy_body, 0, b_row, typename Ring<NQ, IPS, RR>::I32Fetch{});
    else
      ring.run(nsteps, b_row, axpy_body);
    if (too_big)
      return -2;  // nothing has been stored yet: the basis is unchanged
    // ---- row_op_end: update_bf(kappa), gso.cpp:24-48
    store_row_and_refloat<NQ, Map::kNarrowMirrors>(T, pk, bv);
    after(kappa);
    // later reads of b / bfT / rexp in this wave must see these stores
    __threadfence_block();
  }
  return 1;
}

template <int NQ, int IPS, int RR>
__device__ int babai(Lattice<NQ> &T, Ring<NQ, IPS, RR> &ring, int kappa, double eta)
{
  return babai_impl(
      T, ring, kappa, eta, IdentityMap{},
      [&](int k, int last) { return update_row(T, ring, k, last); }, [](int) {});
}

// sum of p[c] for c in [from, to), ascending, starting from `init` if has_init, else from p[from]
template <int NQ>
__device__ __forceinline__ double seq_sum(const double (&p)[NQ], int from, int to)
{
  double s   = 0.0;
  bool first = true;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    const int lo = max(from, 64 * q) - 64 * q;
    const int hi = min(to, 64 * q + 64) - 64 * q;
    for (int cc = lo; cc < hi; ++cc)
    {
      const double v = g_rl_f64(p[q], cc);
      s              = first ? v : s + v;
      first          = false;
    }
  }
  return s;
}

}  // namespace fphip
#endif

