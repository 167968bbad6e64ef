// gso_sweep2.hip — batched floating-point Gram-Schmidt + size-reduction sweep for gfx950, second
// generation of the kernel (the first one, gso_sweep_kernel in gso_kernel.hip, keeps serving as the
// in-tree cross-check: FPHIP_GSO_SWEEP=1 selects it).
//
// Reference behaviour reproduced, bit for bit, for MatGSO<Z_NR<long>, FP_NR<double>> with
// GSO_ROW_EXPO (the BKZ fast path, fplll/bkz.cpp:816-829):
//   MatGSOInterface::update_gso_row   fplll/gso_interface.cpp:131-164
//   MatGSO::get_gram + dot_product    fplll/gso.h:314-331, fplll/nr/numvect.h:386-396
//   LLLReduction::babai               fplll/lll.cpp:166-224
//   LLLReduction::size_reduction      fplll/lll.h:107-122
//   MatGSO::row_addmul_we / update_bf fplll/gso.cpp:236-262, 24-48 ; row_op_end gso_interface.cpp:32-53
//
// One wavefront owns one lattice (see gso_kernel.hip for why).  What is new here is the shape of the
// inner loops — the first kernel spent ~60 issued instructions and half a dozen branches per
// streamed row and was bound by that, not by HBM:
//   * EVERYTHING STREAMED IS A PAIR OF LDS-DMA INSTRUCTIONS INTO ONE PAIR SLOT of the ring (2 entries of
//     NQ*256 B): two rows of the float / int16 mirror of bf (Gram pass), two rows of the int32 / int16
//     mirror of b (integer AXPY), or ONE row of mu as doubles (64*NQ doubles = the slot; lanes 0..63 of the
//     first instruction fetch the first 1024 bytes, the second the rest).  A lane fetches 16 bytes, so a
//     window of a row is a lane mask set with one s_mov exec inside the issue sequence.
//   * EVERY mu WINDOW STARTS ON ITS ROW'S FIRST BYTE (round 4; the rows are 128-byte aligned): the
//     size-reduction sweep reads mu(j, 0..j) straight from the row-major mu array; the GSO recurrence reads
//     column k of mu from an ANCHORED transposed copy muA[k][p] = mu(k + 1 + p, k) — the entries below the
//     diagonal pushed to the front of the row — so the window (k, kappa) is bytes [0, 8 (kappa - 1 - k)).
//     Round 2-3 kept mu as two 4-byte planes with the recurrence window starting at element k + 1: two
//     partial 128-byte lines more per streamed row, 10 % of the kernel's HBM traffic (tests/perf README).
//   * THE RING MOVES IN PAIRS of entries: a step consumes one pair (two Gram rows, two AXPY rows,
//     or the two planes of one mu row) and issues one pair.  INFL entries are ALWAYS in flight —
//     when a chain of phases has no more rows to request the issue side sends one-lane dummies —
//     so every wait is the compile-time s_waitcnt vmcnt(INFL-2) and the loops have no pipeline
//     states.  (Newer loads never complete before older ones, so anything else outstanding only
//     makes that wait stricter.)
//   * loops are CHUNK-MAJOR: the lane that owns the scalar of a step (bf(kappa,c), r(kappa,k),
//     babai_mu[j], the multiplier of row j) is read with v_readlane from a register whose index is
//     a compile-time constant of the loop.
//   * rows whose multiplier is zero are neither fetched nor multiplied in the integer AXPY;
//     r(kappa,kappa) is finished from registers with one subtraction chain; the row's mu / r are
//     stored once, when babai has confirmed it.
// Rows that are not "narrow" (an entry of magnitude >= 2^24, so that the 4-byte mirrors are not
// exact) take the plain-load paths gram_wide / axpy_wide on the 8-byte arrays: same arithmetic.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (no FMA contraction).

#include "gso_wave.h"
#include "gso_sweep2.h"

namespace fphip
{
namespace s2
{

extern __shared__ __attribute__((aligned(16))) unsigned s2_smem[];

enum
{
  K_GRAM  = 0,
  K_REC   = 1,
  K_SWEEP = 2,
  K_AXPY  = 3,
  K_DUMMY = 4,
  K_GRAM8 = 5,  // round 5: the 8-byte arrays through the ring too — ONE row of bfT (doubles) per pair slot
  K_AXPY8 = 6   // … ONE row of b (int64) per pair slot
};

// All members are wave-uniform (SGPRs) except lane4 / lane16.
template <int NQ> struct Stream
{
  using C = Cfg<NQ>;
  unsigned base;            // LDS byte address of this wave's ring
  unsigned hoff, toff;      // byte offsets of the pair slot to fill next / to read next
  unsigned lane4, lane16;   // per lane
  const char *dummy;        // a valid 16-byte aligned address of this lattice (dummy requests)
  // ---- the chain being requested: current segment, then `nkind`, then dummies
  int kind, left, nkind, nleft;
  // Gram: rows c, c+1 of the float mirror (stride gstride bytes), lanes of gmask
  const char *gp;
  long gstride;
  unsigned long long gmask;
  // recurrence: row k of the anchored transpose muA, positions [0, rlast - k) = mu(k+1 .. rlast, k)
  const char *rp;
  long rstride;
  int rk, rlast;
  // size-reduction sweep: row j of mu, elements [0, j), j descending
  const char *sp;
  long sstride;
  int sj;
  // integer AXPY: rows of the int32 mirror whose multiplier is not zero, descending
  const char *ap;
  long astride;
  unsigned long long amask;
  unsigned long long ab0, ab1, ab2, ab3;  // chunk aq, aq-1, … (shifted down as they are exhausted)
  int aq;

  // a wave-uniform 64-bit value the compiler may have parked in VGPRs: back into SGPRs (free when it
  // already is scalar) — the "s" constraint of an inline asm does not do that for 64-bit operands
  static __device__ __forceinline__ unsigned long long uni64(unsigned long long v)
  {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
  }
  __device__ __forceinline__ void dma_pair(const char *p0, const char *p1, unsigned long long mask_)
  {
    const unsigned dst            = base + hoff;
    const unsigned long long mask = uni64(mask_);
    // s_mov exec doubles as the wait state the LDS-DMA needs after the M0 write
    asm volatile("s_mov_b32 m0, %2\n\t"
                 "s_mov_b64 exec, %3\n\t"
                 "global_load_lds_dwordx4 %4, %0\n\t"
                 "s_add_u32 m0, m0, %5\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %4, %1\n\t"
                 "s_mov_b64 exec, -1"
                 :
                 : "s"(p0), "s"(p1), "s"(dst), "s"(mask), "v"(lane16), "n"(C::ESZ)
                 : "memory", "m0", "scc");
    hoff = (hoff + 2 * C::ESZ == (unsigned)C::RING) ? 0u : hoff + 2 * C::ESZ;
  }
  __device__ __forceinline__ void issue_gram()
  {
    dma_pair(gp, gp + gstride, gmask);
    gp += 2 * gstride;
  }
  // one row of doubles, elements [0, len) from the row's first byte (len >= 1), into one pair slot:
  // lanes 0..63 of the first instruction cover the first 1024 bytes, the second instruction the rest —
  // or, when there is no rest, lane 0 of the first again (same bytes to the same place: the count of
  // instructions in flight stays two per pair)
  __device__ __forceinline__ void dma_row8(const char *p, int len)
  {
    const int nl                   = (len + 1) >> 1;  // 16-byte lanes
    const unsigned dst             = base + hoff;
    const bool two                 = nl > 64;
    const unsigned long long maskA = uni64(nl >= 64 ? ~0ull : ((1ull << nl) - 1));
    const unsigned long long maskB = uni64(two ? (nl >= 128 ? ~0ull : ((1ull << (nl - 64)) - 1)) : 1ull);
    const char *pB                 = two ? p + 1024 : p;
    const unsigned dstB            = two ? dst + 1024 : dst;
    asm volatile("s_mov_b32 m0, %2\n\t"
                 "s_mov_b64 exec, %4\n\t"
                 "global_load_lds_dwordx4 %6, %0\n\t"
                 "s_mov_b32 m0, %3\n\t"
                 "s_mov_b64 exec, %5\n\t"
                 "global_load_lds_dwordx4 %6, %1\n\t"
                 "s_mov_b64 exec, -1"
                 :
                 : "s"(p), "s"(pB), "s"(dst), "s"(dstB), "s"(maskA), "s"(maskB), "v"(lane16)
                 : "memory", "m0", "scc");
    hoff = (hoff + 2 * C::ESZ == (unsigned)C::RING) ? 0u : hoff + 2 * C::ESZ;
  }
  __device__ __forceinline__ void issue_rec()
  {
    dma_row8(rp, rlast - rk);
    rp += rstride;
    ++rk;
  }
  __device__ __forceinline__ void issue_sweep()
  {
    dma_row8(sp, sj);
    sp -= sstride;
    --sj;
  }
  __device__ __forceinline__ void issue_axpy()
  {
    while (ab0 == 0 && aq > 0)
    {  // next lower chunk that has a multiplier
      ab0 = ab1;
      ab1 = ab2;
      ab2 = ab3;
      ab3 = 0;
      --aq;
    }
    unsigned long long b = ab0;
    const int j0         = 63 - __builtin_clzll(b);
    b &= ~(1ull << j0);
    int j1 = j0;
    if (b)
    {
      j1 = 63 - __builtin_clzll(b);
      b &= ~(1ull << j1);
    }
    ab0            = b;
    const char *q0 = ap + (long)(aq * 64 + j0) * astride;
    const char *q1 = ap + (long)(aq * 64 + j1) * astride;
    dma_pair(q0, q1, amask);
  }
  // rows that are not narrow (an entry of magnitude >= 2^24): the 8-byte arrays, one row per pair slot like a mu row
  const char *gp8;   // row c of bfT, elements [0, glen8)
  long gstride8;
  int glen8;
  const char *ap8;   // rows of b (int64), elements [0, alen8), rows with a non-zero multiplier, descending
  long astride8;
  int alen8;
  __device__ __forceinline__ void issue_gram8()
  {
    dma_row8(gp8, glen8);
    gp8 += gstride8;
  }
  __device__ __forceinline__ void issue_axpy8()
  {
    while (ab0 == 0 && aq > 0)
    {
      ab0 = ab1;
      ab1 = ab2;
      ab2 = ab3;
      ab3 = 0;
      --aq;
    }
    const int j0 = 63 - __builtin_clzll(ab0);
    ab0 &= ~(1ull << j0);
    dma_row8(ap8 + (long)(aq * 64 + j0) * astride8, alen8);
  }
  __device__ __forceinline__ void issue_dummy() { dma_pair(dummy, dummy, 1ull); }

  __device__ __forceinline__ void advance_segment()
  {
    kind  = nkind;
    left  = nleft;
    nkind = K_DUMMY;
    nleft = 0x7fffffff;
  }
  // generic (branchy) request of one pair: chain prologues only
  __device__ __forceinline__ void issue_any()
  {
    while (left == 0)
      advance_segment();
    switch (kind)
    {
    case K_GRAM: issue_gram(); break;
    case K_REC: issue_rec(); break;
    case K_SWEEP: issue_sweep(); break;
    case K_AXPY: issue_axpy(); break;
    case K_GRAM8: issue_gram8(); break;
    case K_AXPY8: issue_axpy8(); break;
    default: issue_dummy(); break;
    }
    if (kind != K_DUMMY)
      --left;
  }
  // Start a chain: whatever is still in flight is abandoned (its LDS writes land before those of
  // the requests made from now on), the ring is refilled with the first NPAIR-1 pairs.
  __device__ __forceinline__ void begin(int k0, int n0, int k1, int n1)
  {
    kind  = k0;
    left  = n0;
    nkind = k1;
    nleft = n1;
    if (left == 0)
      advance_segment();
    toff = hoff;
#pragma unroll 1
    for (int i = 0; i < C::NPAIR - 1; ++i)
      issue_any();
  }

  // wait for the oldest pair, hand out its LDS word index (per lane: + lane, + 64 q, second entry
  // at + ESZ/4) and move the tail
  __device__ __forceinline__ unsigned wait_pair()
  {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::INFL - 2) : "memory");
    const unsigned w = ((base + toff) >> 2);
    toff             = (toff + 2 * C::ESZ == (unsigned)C::RING) ? 0u : toff + 2 * C::ESZ;
    return w;
  }

  // `nsteps` steps of one consumer; ALLOWED = the segment kinds the issue side can be in meanwhile
  template <unsigned ALLOWED, class Cons> __device__ __forceinline__ void run(Cons &cons, int nsteps)
  {
    while (nsteps > 0)
    {
      if (left == 0)
        advance_segment();
      const int m = nsteps < left ? nsteps : left;
      bool done   = false;
      if constexpr ((ALLOWED & (1u << K_GRAM)) != 0)
        if (!done && kind == K_GRAM)
        {
#pragma unroll 1
          for (int i = 0; i < m; ++i)
          {
            const unsigned w = wait_pair();
            cons.load(w, lane4);
            issue_gram();
            cons.compute();
          }
          done = true;
        }
      if constexpr ((ALLOWED & (1u << K_REC)) != 0)
        if (!done && kind == K_REC)
        {
#pragma unroll 1
          for (int i = 0; i < m; ++i)
          {
            const unsigned w = wait_pair();
            cons.load(w, lane4);
            issue_rec();
            cons.compute();
          }
          done = true;
        }
      if constexpr ((ALLOWED & (1u << K_SWEEP)) != 0)
        if (!done && kind == K_SWEEP)
        {
#pragma unroll 1
          for (int i = 0; i < m; ++i)
          {
            const unsigned w = wait_pair();
            cons.load(w, lane4);
            issue_sweep();
            cons.compute();
          }
          done = true;
        }
      if constexpr ((ALLOWED & (1u << K_AXPY)) != 0)
        if (!done && kind == K_AXPY)
        {
#pragma unroll 1
          for (int i = 0; i < m; ++i)
          {
            const unsigned w = wait_pair();
            cons.load(w, lane4);
            issue_axpy();
            cons.compute();
          }
          done = true;
        }
      if constexpr ((ALLOWED & (1u << K_GRAM8)) != 0)
        if (!done && kind == K_GRAM8)
        {
#pragma unroll 1
          for (int i = 0; i < m; ++i)
          {
            const unsigned w = wait_pair();
            cons.load(w, lane4);
            issue_gram8();
            cons.compute();
          }
          done = true;
        }
      if constexpr ((ALLOWED & (1u << K_AXPY8)) != 0)
        if (!done && kind == K_AXPY8)
        {
#pragma unroll 1
          for (int i = 0; i < m; ++i)
          {
            const unsigned w = wait_pair();
            cons.load(w, lane4);
            issue_axpy8();
            cons.compute();
          }
          done = true;
        }
      if (!done)
      {  // dummies (or a kind this consumer never meets: treated as exhausted)
#pragma unroll 1
        for (int i = 0; i < m; ++i)
        {
          const unsigned w = wait_pair();
          cons.load(w, lane4);
          issue_dummy();
          cons.compute();
        }
      }
      if (kind != K_DUMMY)
        left -= m;
      nsteps -= m;
    }
  }
};

// The kernel's own copies of one lattice's data
struct Planes
{
  double *muA;     // [d][ldd]  anchored transpose of mu: muA[k*ldd + (j-k-1)] = mu(j,k), j > k
  // 2-byte mirrors: while every entry of a row is below 2^15 in magnitude the Gram pass streams the
  // INTEGER column (bT16[c][j] = b(j,c)) and scales it by 2^-row_expo(j) in registers — bf(j,c) IS
  // b(j,c) * 2^-row_expo(j) exactly (gso.cpp:27-40) — and the AXPY streams b16: half the bytes of
  // the 4-byte mirrors again
  short *bT16;   // [n][ldd]
  short *b16;    // [d][ldn]
  int *flag16;   // [d]
  int np16;      // rows 0..np16-1 all carry the flag (wave-uniform)
};

__device__ __forceinline__ double rl2(double v, int lane) { return g_rl_f64(v, lane); }

// ---------------------------------------------------------------------------------------------
// consumers.  load(w, lane4) reads this lane's words of the pair at LDS word index w; compute()
// uses them.  `i` is the wave-uniform position inside the consumer's chunk.
// ---------------------------------------------------------------------------------------------
// EB = bytes per streamed element: 4 (float mirror of bf) or 2 (int16 mirror of b, scaled by sc[q] =
// 2^-row_expo of the lane's row: the same double, exactly)
template <int NQ, int CQ, int QA, int EB> struct GramCons  // QA: chunks 0..QA-1 hold a column j <= last
{
  double (&acc)[NQ];
  const double &bkq;  // bk[CQ]
  const double (&sc)[NQ];
  int cc;             // row c = 64 CQ + cc of the first entry
  bool second;        // the second entry is a row too (false: tail of an odd count)
  double x0[NQ], x1[NQ];
  __device__ __forceinline__ void load(unsigned w, unsigned lane4)
  {
    if constexpr (EB == 4)
    {
      const unsigned a = w + (lane4 >> 2);
#pragma unroll
      for (int q = 0; q < QA; ++q)
      {
        x0[q] = (double)__uint_as_float(s2_smem[a + 64 * q]);
        x1[q] = (double)__uint_as_float(s2_smem[a + 64 * q + Cfg<NQ>::ESZ / 4]);
      }
    }
    else
    {
      const short *h   = (const short *)s2_smem;
      const unsigned a = 2 * w + (lane4 >> 2);
#pragma unroll
      for (int q = 0; q < QA; ++q)
      {
        x0[q] = (double)(int)h[a + 64 * q];
        x1[q] = (double)(int)h[a + 64 * q + Cfg<NQ>::ESZ / 2];
      }
    }
  }
  __device__ __forceinline__ void compute()
  {
    const double s0 = rl2(bkq, cc);
    const double s1 = rl2(bkq, cc + 1);
#pragma unroll
    for (int q = 0; q < QA; ++q)
    {
      const double v0 = (EB == 4) ? x0[q] : x0[q] * sc[q];
      const double p0 = s0 * v0;
      acc[q]          = acc[q] + p0;
      if (second)
      {
        const double v1 = (EB == 4) ? x1[q] : x1[q] * sc[q];
        const double p1 = s1 * v1;
        acc[q]          = acc[q] + p1;
      }
    }
    cc += 2;
  }
};

// sum over the 64 lanes (an inclusive scan with DPP row shifts and the two row broadcasts; lane 63 ends
// with the total)
__device__ __forceinline__ int wave_sum_i32(int v)
{
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
  return __builtin_amdgcn_readlane(v, 63);
}

// The integer AXPY FUSED into the Gram pass that follows it (round 4).  babai's row operation
// b_kappa += sum_j X_j b_j and the Gram row of the new b_kappa read the SAME numbers — the entries of the
// rows above kappa — once row-major (b16) and once column-major (bT16).  Column c of the column-major mirror
// holds b(j, c) for every j in lane j: the new entry b'(kappa, c) = b(kappa, c) + sum_j X_j b(j, c) is one
// wave-wide INTEGER sum (exact in any order), after which the column's Gram products
// bf'(kappa, c) bf(j, c) can be formed at once — c still ascending, as numvect.h:386-396 wants it.  The row
// exponent of the new row is only known when all of it is: the pass accumulates with the UNSCALED
// (double) b'(kappa, c), and the caller multiplies the sums by 2^-row_expo afterwards (twice for
// g(kappa, kappa)) — a power of two commutes with every rounding on the way, so these are the doubles the
// separate passes produce.  Saves the whole AXPY stream: 6 of 55 MB per 180-dimensional lattice.
// Multipliers must fit 32 bits (every |X_j b(j,c)| < 2^46; the lane sums are reduced as two 24 / 25-bit halves).
template <int NQ, int CQ, int QA> struct GramAxpyCons
{
  double (&acc)[NQ];
  long long (&bv)[NQ];      // row kappa, lane c = column c: updated in place
  const int (&lx)[NQ];      // multiplier of row j in lane j (0 for j >= kappa)
  const double (&sc)[NQ];
  const bool (&isk)[NQ];    // this (lane, chunk) is row kappa
  unsigned lane;
  int cc;
  bool second;
  int x0[NQ], x1[NQ];
  __device__ __forceinline__ void load(unsigned w, unsigned lane4)
  {
    const short *h   = (const short *)s2_smem;
    const unsigned a = 2 * w + (lane4 >> 2);
#pragma unroll
    for (int q = 0; q < QA; ++q)
    {
      x0[q] = (int)h[a + 64 * q];
      x1[q] = (int)h[a + 64 * q + Cfg<NQ>::ESZ / 2];
    }
  }
  __device__ __forceinline__ void column(const int (&x)[NQ], int c)
  {
    long long t = 0;
#pragma unroll
    for (int q = 0; q < QA; ++q)
      t += (long long)lx[q] * (long long)x[q];
    const int lo       = wave_sum_i32((int)(t & 0x7fffff));
    const int hi       = wave_sum_i32((int)(t >> 23));
    const long long nb = g_rl_i64(bv[CQ], c) + (((long long)hi << 23) + (long long)lo);
    bv[CQ]             = ((int)lane == c) ? nb : bv[CQ];
    const double f     = (double)nb;
#pragma unroll
    for (int q = 0; q < QA; ++q)
    {
      const double v = isk[q] ? f : (double)x[q] * sc[q];
      const double p = f * v;
      acc[q]         = acc[q] + p;
    }
  }
  __device__ __forceinline__ void compute()
  {
    column(x0, cc);
    if (second)
      column(x1, cc + 1);
    cc += 2;
  }
};

template <int NQ, int QA, int CQ = 0>
__device__ __forceinline__ void gram_axpy_phase(Stream<NQ> &S, double (&acc)[NQ], long long (&bv)[NQ],
                                                const int (&lx)[NQ], const double (&sc)[NQ],
                                                const bool (&isk)[NQ], unsigned lane, int n)
{
  if constexpr (CQ < NQ)
  {
    const int rows = min(n - 64 * CQ, 64);
    if (rows > 0)
    {
      GramAxpyCons<NQ, CQ, QA> g{acc, bv, lx, sc, isk, lane, 0, true};
      S.template run<(1u << K_GRAM) | (1u << K_REC)>(g, rows >> 1);
      if (rows & 1)
      {
        g.second = false;
        S.template run<(1u << K_GRAM) | (1u << K_REC)>(g, 1);
      }
      gram_axpy_phase<NQ, QA, CQ + 1>(S, acc, bv, lx, sc, isk, lane, n);
    }
  }
}

template <int NQ, int KQ, int QA> struct RecCons
{
  double (&acc)[NQ];
  int kk;  // row k = 64 KQ + kk
  unsigned lane;
  double m[NQ];
  __device__ __forceinline__ void load(unsigned w, unsigned lane4)
  {
    // position of mu(j, k) in the anchored row: j - k - 1 (the lanes j <= k read position 0: not used)
    const double *sd = (const double *)s2_smem + (w >> 1);
    const int p0     = (int)(lane4 >> 2) - kk - 1;
#pragma unroll
    for (int q = KQ; q < QA; ++q)
    {
      const int p = p0 + 64 * (q - KQ);
      m[q]        = sd[q == KQ ? max(p, 0) : p];
    }
  }
  __device__ __forceinline__ void compute()
  {
    const double rk = rl2(acc[KQ], kk);  // r(kappa,k) is final
#pragma unroll
    for (int q = KQ; q < QA; ++q)
    {
      const double t = m[q] * rk;
      const double u = acc[q] - t;
      if (q == KQ)
        acc[q] = ((int)lane > kk) ? u : acc[q];  // rows j > k only
      else
        acc[q] = u;
    }
    ++kk;
  }
};

// rnd_we, nr/nr_FP_d.inl:226-233
__device__ __forceinline__ double rnd_we(double b, int e)
{
  if (fexponent(b) + e >= 53)
    return b;
  return ldexp(rint(ldexp(b, e)), -e);
}

template <int NQ, int JQ> struct SweepCons
{
  double (&bm)[NQ];
  double (&xs)[NQ];
  const int (&e)[NQ];
  unsigned long long &nzb;  // multipliers of chunk JQ that are not zero
  int jj;                   // row j = 64 JQ + jj, descending
  int sr_start;
  unsigned lane;
  double m[NQ];
  __device__ __forceinline__ void load(unsigned w, unsigned lane4)
  {
    const double *sd = (const double *)s2_smem + (w >> 1) + (lane4 >> 2);
#pragma unroll
    for (int q = 0; q <= JQ; ++q)
      m[q] = sd[64 * q];
  }
  __device__ __forceinline__ void compute()
  {
    const double bmj = rl2(bm[JQ], jj);
    const int ej     = __builtin_amdgcn_readlane(e[JQ], jj);
    const double X   = rnd_we(bmj, ej);
    if (X != 0.0)
    {
      nzb |= 1ull << jj;
      xs[JQ] = ((int)lane == jj) ? X : xs[JQ];
#pragma unroll
      for (int q = 0; q <= JQ; ++q)
      {
        const double t = X * m[q];
        const double u = bm[q] - t;
        // chunks below JQ hold only k < j; the chunk of j itself needs the test
        const bool on = (q < JQ || (int)lane < jj) && (int)(lane + 64 * q) >= sr_start;
        bm[q]         = on ? u : bm[q];
      }
    }
    --jj;
  }
};

template <int NQ, int JQ, int EB> struct AxpyCons
{
  long long (&bv)[NQ];
  const long long &lxq;      // multipliers of chunk JQ (lane j)
  unsigned long long bits;   // multipliers of the chunk not yet applied
  bool small;                // every multiplier fits 32 bits
  int w0[NQ], w1[NQ];
  __device__ __forceinline__ void load(unsigned w, unsigned lane4)
  {
    if constexpr (EB == 4)
    {
      const unsigned a = w + (lane4 >> 2);
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        w0[q] = (int)s2_smem[a + 64 * q];
        w1[q] = (int)s2_smem[a + 64 * q + Cfg<NQ>::ESZ / 4];
      }
    }
    else
    {
      const short *h   = (const short *)s2_smem;
      const unsigned a = 2 * w + (lane4 >> 2);
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        w0[q] = (int)h[a + 64 * q];
        w1[q] = (int)h[a + 64 * q + Cfg<NQ>::ESZ / 2];
      }
    }
  }
  __device__ __forceinline__ void compute()
  {
    const int j0 = 63 - __builtin_clzll(bits);
    bits &= ~(1ull << j0);
    long long l0 = g_rl_i64(lxq, j0), l1 = 0;
    if (bits)
    {
      const int j1 = 63 - __builtin_clzll(bits);
      bits &= ~(1ull << j1);
      l1 = g_rl_i64(lxq, j1);
    }
    if (small)
    {
      const int s0 = (int)l0, s1 = (int)l1;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        bv[q] = bv[q] + (long long)w0[q] * (long long)s0;
        bv[q] = bv[q] + (long long)w1[q] * (long long)s1;
      }
    }
    else
    {
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        bv[q] = (long long)((unsigned long long)bv[q] +
                            (unsigned long long)(long long)w0[q] * (unsigned long long)l0);
        bv[q] = (long long)((unsigned long long)bv[q] +
                            (unsigned long long)(long long)w1[q] * (unsigned long long)l1);
      }
    }
  }
};

// the 8-byte twins (round 5): one row of doubles / int64 per step, the pair slot read as 64 NQ 8-byte words
template <int NQ, int CQ, int QA> struct GramCons8
{
  double (&acc)[NQ];
  const double &bkq;  // bk[CQ]
  int cc;             // row c = 64 CQ + cc
  double x[NQ];
  __device__ __forceinline__ void load(unsigned w, unsigned lane4)
  {
    const double *sd = (const double *)s2_smem + (w >> 1) + (lane4 >> 2);
#pragma unroll
    for (int q = 0; q < QA; ++q)
      x[q] = sd[64 * q];
  }
  __device__ __forceinline__ void compute()
  {
    const double s0 = rl2(bkq, cc);
#pragma unroll
    for (int q = 0; q < QA; ++q)
    {
      const double p0 = s0 * x[q];
      acc[q]          = acc[q] + p0;
    }
    ++cc;
  }
};
template <int NQ, int JQ> struct AxpyCons8
{
  long long (&bv)[NQ];
  const long long &lxq;     // multipliers of chunk JQ (lane j)
  unsigned long long bits;  // multipliers of the chunk not yet applied
  long long w0[NQ];
  __device__ __forceinline__ void load(unsigned w, unsigned lane4)
  {
    const long long *sl = (const long long *)s2_smem + (w >> 1) + (lane4 >> 2);
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      w0[q] = sl[64 * q];
  }
  __device__ __forceinline__ void compute()
  {
    const int j0 = 63 - __builtin_clzll(bits);
    bits &= ~(1ull << j0);
    const long long l0 = g_rl_i64(lxq, j0);
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      bv[q] = (long long)((unsigned long long)bv[q] + (unsigned long long)w0[q] * (unsigned long long)l0);
  }
};

// ---------------------------------------------------------------------------------------------
// phases (compile-time recursion over the chunks)
// ---------------------------------------------------------------------------------------------
template <int NQ, int QA, int CQ = 0>
__device__ __forceinline__ void gram8_phase(Stream<NQ> &S, double (&acc)[NQ], const double (&bk)[NQ], int n)
{
  if constexpr (CQ < NQ)
  {
    const int rows = min(n - 64 * CQ, 64);
    if (rows > 0)
    {
      GramCons8<NQ, CQ, QA> g{acc, bk[CQ], 0};
      S.template run<(1u << K_GRAM8) | (1u << K_REC)>(g, rows);
      gram8_phase<NQ, QA, CQ + 1>(S, acc, bk, n);
    }
  }
}
template <int NQ, int JQ = NQ - 1>
__device__ __forceinline__ void axpy8_phase(Stream<NQ> &S, long long (&bv)[NQ], const long long (&lxv)[NQ],
                                            const unsigned long long (&nz)[NQ])
{
  if constexpr (JQ >= 0)
  {
    if (nz[JQ] != 0)
    {
      AxpyCons8<NQ, JQ> a{bv, lxv[JQ], nz[JQ]};
      S.template run<(1u << K_AXPY8)>(a, __builtin_popcountll(nz[JQ]));
    }
    axpy8_phase<NQ, JQ - 1>(S, bv, lxv, nz);
  }
}

template <int NQ, int QA, int EB, int CQ = 0>
__device__ __forceinline__ void gram_phase(Stream<NQ> &S, double (&acc)[NQ], const double (&bk)[NQ],
                                           const double (&sc)[NQ], int n)
{
  if constexpr (CQ < NQ)
  {
    const int rows = min(n - 64 * CQ, 64);
    if (rows > 0)
    {
      GramCons<NQ, CQ, QA, EB> g{acc, bk[CQ], sc, 0, true};
      S.template run<(1u << K_GRAM) | (1u << K_REC)>(g, rows >> 1);
      if (rows & 1)
      {
        g.second = false;
        S.template run<(1u << K_GRAM) | (1u << K_REC)>(g, 1);
      }
      gram_phase<NQ, QA, EB, CQ + 1>(S, acc, bk, sc, n);
    }
  }
}

// rows k = 0 .. cnt-1 of the recurrence (cnt = last: row `last` has nothing to subtract from)
template <int NQ, int QA, int KQ = 0>
__device__ __forceinline__ void rec_phase(Stream<NQ> &S, double (&acc)[NQ], int cnt, unsigned lane)
{
  if constexpr (KQ < QA)
  {
    const int rows = min(cnt - 64 * KQ, 64);
    if (rows > 0)
    {
      RecCons<NQ, KQ, QA> r{acc, 0, lane};
      S.template run<(1u << K_REC)>(r, rows);
      rec_phase<NQ, QA, KQ + 1>(S, acc, cnt, lane);
    }
  }
}

// Gram pass then recurrence with the (wave-uniform) number of active chunks as a compile-time
// constant: QA = (last >> 6) + 1
template <int NQ, int QA>
__device__ __forceinline__ void gram_rec_qa(Stream<NQ> &S, double (&acc)[NQ], const double (&bk)[NQ],
                                            const double (&sc)[NQ], int n, int nrec, unsigned lane, int narrow,
                                            int kappa, double &gkk)
{
  if (narrow == 2)
    gram_phase<NQ, QA, 2>(S, acc, bk, sc, n);
  else if (narrow == 1)
    gram_phase<NQ, QA, 4>(S, acc, bk, sc, n);
  else if (narrow == 0)
    gram8_phase<NQ, QA>(S, acc, bk, n);
  gkk = g_rl_f64(acc[QA - 1], kappa & 63);  // lane kappa of chunk QA-1 = kappa >> 6
  rec_phase<NQ, QA>(S, acc, nrec, lane);
}

// rows j = kappa-1 … sr_start+1 through the ring, then j = sr_start (no mu needed)
template <int NQ, int JQ = NQ - 1>
__device__ __forceinline__ void sweep_phase(Stream<NQ> &S, double (&bm)[NQ], double (&xs)[NQ], const int (&e)[NQ],
                                            unsigned long long (&nz)[NQ], int kappa, int sr_start,
                                            unsigned lane)
{
  if constexpr (JQ >= 0)
  {
    // rows of this chunk: j in [max(64 JQ, sr_start), min(64 JQ + 63, kappa - 1)]
    const int hi = min(kappa - 1, 64 * JQ + 63);
    const int lo = max(sr_start, 64 * JQ);
    if (hi >= lo)
    {
      SweepCons<NQ, JQ> s{bm, xs, e, nz[JQ], hi - 64 * JQ, sr_start, lane};
      // every row but j = sr_start has a window of mu to stream
      const int streamed = (lo == sr_start) ? (hi - lo) : (hi - lo + 1);
      S.template run<(1u << K_SWEEP)>(s, streamed);
      if (lo == sr_start)
      {  // last row: its multiplier only
#pragma unroll
        for (int q = 0; q <= JQ; ++q)
          s.m[q] = 0.0;
        s.compute();
      }
    }
    sweep_phase<NQ, JQ - 1>(S, bm, xs, e, nz, kappa, sr_start, lane);
  }
}

template <int NQ, int EB, int JQ = NQ - 1>
__device__ __forceinline__ void axpy_phase(Stream<NQ> &S, long long (&bv)[NQ], const long long (&lxv)[NQ],
                                           const unsigned long long (&nz)[NQ], bool small)
{
  if constexpr (JQ >= 0)
  {
    if (nz[JQ] != 0)
    {
      AxpyCons<NQ, JQ, EB> a{bv, lxv[JQ], nz[JQ], small};
      S.template run<(1u << K_AXPY)>(a, (__builtin_popcountll(nz[JQ]) + 1) >> 1);
    }
    axpy_phase<NQ, EB, JQ - 1>(S, bv, lxv, nz, small);
  }
}

// ---------------------------------------------------------------------------------------------
// plain-load twins of the Gram pass and the AXPY for rows that are not narrow (same arithmetic)
// ---------------------------------------------------------------------------------------------
template <int NQ, int CQ = 0>
__device__ __forceinline__ void gram_wide(const Lattice<NQ> &T, double (&acc)[NQ], const double (&bk)[NQ], int last)
{
  if constexpr (CQ < NQ)
  {
    const int n = T.n, lane = T.lane, ldd = T.ldd;
    const int hi = min(n - 64 * CQ, 64);
    for (int cc = 0; cc < hi; ++cc)
    {
      const double bkc = g_rl_f64(bk[CQ], cc);
      const int c      = 64 * CQ + cc;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int j = lane + 64 * q;
        if (j <= last)
        {
          const double p = bkc * T.bfT[(size_t)c * ldd + j];
          acc[q]         = acc[q] + p;
        }
      }
    }
    gram_wide<NQ, CQ + 1>(T, acc, bk, last);
  }
}

template <int NQ, int JQ = NQ - 1>
__device__ __forceinline__ void axpy_wide(const Lattice<NQ> &T, long long (&bv)[NQ], const long long (&lxv)[NQ],
                                          int kappa)
{
  if constexpr (JQ >= 0)
  {
    const int n = T.n, lane = T.lane, ldn = T.ldn;
    for (int jj = min(kappa - 1 - 64 * JQ, 63); jj >= 0; --jj)
    {
      const long long lx = g_rl_i64(lxv[JQ], jj);
      const int j        = 64 * JQ + jj;
      if (lx != 0)
      {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
        {
          const int c = lane + 64 * q;
          if (c < n)
            bv[q] = (long long)((unsigned long long)bv[q] +
                                (unsigned long long)T.b[(size_t)j * ldn + c] * (unsigned long long)lx);
        }
      }
    }
    axpy_wide<NQ, JQ - 1>(T, bv, lxv, kappa);
  }
}

// ---------------------------------------------------------------------------------------------
// One GSO pass of row kappa: Gram row g(kappa, j <= kappa) then the recurrence for j < kappa.
// On return acc[] lane j < kappa = r(kappa,j); gkk = g(kappa,kappa).
// ---------------------------------------------------------------------------------------------
// sc[q] = 2^-row_expo of row lane + 64 q (rows <= kappa): the scale of the 2-byte Gram path
template <int NQ>
__device__ __forceinline__ void gso_pass(Lattice<NQ> &T, const Planes &PL, Stream<NQ> &S, int kappa,
                                         const double (&bk)[NQ], const double (&sc)[NQ], double (&acc)[NQ],
                                         double &gkk)
{
  const int n = T.n, ldd = T.ldd;
  const int qact = (kappa >> 6) + 1;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
    acc[q] = -0.0;  // -0.0 + p == p for every p: the first product starts the sum (numvect.h:389)
  // rows 0..kappa are read: 2-byte mirrors, else 4-byte mirrors, else the 8-byte arrays
  const int narrow = (PL.np16 > kappa) ? 2 : (T.np > kappa) ? 1 : 0;
  // recurrence rows k = 0..kappa-2 (row kappa-1 has nothing below it)
  const int nrec = kappa > 0 ? kappa - 1 : 0;
  S.rp      = (const char *)PL.muA;
  S.rstride = (long)ldd * 8;
  S.rk      = 0;
  S.rlast   = kappa - 1;
  if (narrow == 2)
  {
    S.gp      = (const char *)PL.bT16;
    S.gstride = (long)ldd * 2;
    S.gmask   = ~0ull >> (63 - (kappa >> 3));  // a lane fetches 8 elements
    S.begin(K_GRAM, (n + 1) >> 1, K_REC, nrec);
  }
  else if (narrow == 1)
  {
    S.gp      = (const char *)T.bfT32;
    S.gstride = (long)ldd * 4;
    S.gmask   = ~0ull >> (63 - (kappa >> 2));
    S.begin(K_GRAM, (n + 1) >> 1, K_REC, nrec);
  }
  else if (T.wide_ring)
  {  // rows that are not narrow: the 8-byte rows of bfT through the ring as well (round 5)
    S.gp8      = (const char *)T.bfT;
    S.gstride8 = (long)ldd * 8;
    S.glen8    = kappa + 1;
    S.begin(K_GRAM8, n, K_REC, nrec);
  }
  else
  {
    gram_wide<NQ>(T, acc, bk, kappa);
    S.begin(K_REC, nrec, K_DUMMY, 0x7fffffff);
  }
  const unsigned lane = (unsigned)T.lane;
  const int narrow_in = narrow;
  const int narrow_ = (narrow_in == 0 && !T.wide_ring) ? -1 : narrow_in;  // (-1: gram_wide has done the Gram pass)
#define narrow narrow_
  if constexpr (NQ == 1)
    gram_rec_qa<NQ, 1>(S, acc, bk, sc, n, nrec, lane, narrow, kappa, gkk);
  else if constexpr (NQ == 2)
  {
    if (qact == 1)
      gram_rec_qa<NQ, 1>(S, acc, bk, sc, n, nrec, lane, narrow, kappa, gkk);
    else
      gram_rec_qa<NQ, 2>(S, acc, bk, sc, n, nrec, lane, narrow, kappa, gkk);
  }
  else if constexpr (NQ == 3)
  {
    if (qact == 1)
      gram_rec_qa<NQ, 1>(S, acc, bk, sc, n, nrec, lane, narrow, kappa, gkk);
    else if (qact == 2)
      gram_rec_qa<NQ, 2>(S, acc, bk, sc, n, nrec, lane, narrow, kappa, gkk);
    else
      gram_rec_qa<NQ, 3>(S, acc, bk, sc, n, nrec, lane, narrow, kappa, gkk);
  }
  else
  {
    if (qact == 1)
      gram_rec_qa<NQ, 1>(S, acc, bk, sc, n, nrec, lane, narrow, kappa, gkk);
    else if (qact == 2)
      gram_rec_qa<NQ, 2>(S, acc, bk, sc, n, nrec, lane, narrow, kappa, gkk);
    else if (qact == 3)
      gram_rec_qa<NQ, 3>(S, acc, bk, sc, n, nrec, lane, narrow, kappa, gkk);
    else
      gram_rec_qa<NQ, 4>(S, acc, bk, sc, n, nrec, lane, narrow, kappa, gkk);
  }
#undef narrow
}

// r(kappa,kappa) = g(kappa,kappa) - sum_k mu(kappa,k) r(kappa,k), k ascending (gso_interface.cpp:147-151)
template <int NQ>
__device__ __forceinline__ double finish_diag2(const double (&mu)[NQ], const double (&rr)[NQ], double gkk,
                                               int kappa)
{
  double p[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
    p[q] = mu[q] * rr[q];
  double g = gkk;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    const int hi = min(kappa - 64 * q, 64);
    for (int kk = 0; kk < hi; ++kk)
      g = g - g_rl_f64(p[q], kk);
  }
  return g;
}

// store the confirmed row kappa: r, mu (what the sweep of the later rows streams), its column of the anchored
// transpose (what their recurrences stream), r(kappa,kappa)
template <int NQ>
__device__ __forceinline__ void store_gso_row(Lattice<NQ> &T, const Planes &PL, int kappa, const double (&mu)[NQ],
                                              const double (&rr)[NQ], double rkk)
{
  const int lane = T.lane, ldd = T.ldd;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    const int j = lane + 64 * q;
    if (j < kappa)
    {
      T.r[(size_t)kappa * ldd + j]  = rr[q];
      T.mu[(size_t)kappa * ldd + j] = mu[q];
      PL.muA[(size_t)j * ldd + (kappa - j - 1)] = mu[q];
    }
    else if (j == kappa)
    {
      T.r[(size_t)kappa * ldd + kappa] = rkk;
      T.rdg[kappa]                     = rkk;
    }
  }
}

// mu(kappa,j) = r(kappa,j) / r(j,j), gso_interface.cpp:154; false on a non-finite value
template <int NQ>
__device__ __forceinline__ bool mu_from_r(const Lattice<NQ> &T, int kappa, const double (&acc)[NQ],
                                          const double (&rd)[NQ], double (&mu)[NQ])
{
  bool ok = true;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    const int j = T.lane + 64 * q;
    mu[q]       = 0.0;
    if (j < kappa)
    {
      const double m = acc[q] / rd[q];
      if (!isfinite(m))
        ok = false;
      mu[q] = m;
    }
  }
  return __all(ok);
}

// The 2-byte mirrors of row pk (after store_row_and_refloat): b16 row, bT16 column, flag, prefix.
template <int NQ>
__device__ __forceinline__ void store_mirror16(const Lattice<NQ> &T, Planes &PL, int pk, const long long (&bv)[NQ])
{
  const int n = T.n, lane = T.lane, ldd = T.ldd, ldn = T.ldn;
  bool wide = false;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    const int c = lane + 64 * q;
    if (c < n)
    {
      PL.b16[(size_t)pk * ldn + c]  = (short)bv[q];
      PL.bT16[(size_t)c * ldd + pk] = (short)bv[q];
      wide |= (bv[q] >= (1ll << 15) || bv[q] < -(1ll << 15));
    }
  }
  const bool w = __any(wide);
  if (lane == 0)
    PL.flag16[pk] = w ? 0 : 1;
  if (w)
    PL.np16 = min(PL.np16, pk);
  else if (pk == PL.np16)
  {
    int p = pk + 1;
    while (p < T.d && __builtin_amdgcn_readfirstlane(PL.flag16[p]) != 0)
      ++p;
    PL.np16 = p;
  }
}

// sc[q] = 2^-row_expo(row lane + 64 q) for rows < kappa, 2^-rexpk for row kappa itself
template <int NQ>
__device__ __forceinline__ void scale_vector(const Lattice<NQ> &T, int kappa, const long long (&rexpj)[NQ],
                                             long long rexpk, double (&sc)[NQ])
{
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    const int j = T.lane + 64 * q;
    sc[q]       = ldexp(1.0, -(int)((j == kappa) ? rexpk : rexpj[q]));
  }
}

// bf(kappa, c) for lane c.  bfT is column-major (a Gram pass wants the columns): row kappa of it is one
// 8-byte word out of 180 different 128-byte lines.  While the row carries the 2-byte flag its mirror row is
// contiguous and bf(kappa, c) = b(kappa, c) 2^-row_expo(kappa) EXACTLY (gso.cpp:27-40; |b| < 2^15): the
// same doubles from 3 lines instead of 180 (4 MB per 180-dimensional lattice and sweep).
template <int NQ>
__device__ __forceinline__ void load_bf_row(const Lattice<NQ> &T, const Planes &PL, int kappa, double (&bk)[NQ])
{
  const int n = T.n, lane = T.lane;
  if (PL.np16 > kappa)
  {
    const long long ek = T.rexp[kappa];
    int v[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      const int c = lane + 64 * q;
      v[q]        = (c < n) ? (int)PL.b16[(size_t)kappa * T.ldn + c] : 0;
    }
    const double sck = ldexp(1.0, -(int)ek);
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      settle(v[q]);
      bk[q] = (double)v[q] * sck;
    }
  }
  else
  {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      const int c = lane + 64 * q;
      bk[q]       = (c < n) ? T.bfT[(size_t)c * T.ldd + kappa] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      settle(bk[q]);
  }
}

// update_gso_row(kappa, kappa) from column 0.  false: RED_GSO_FAILURE.
template <int NQ>
__device__ __forceinline__ bool update_full(Lattice<NQ> &T, Planes &PL, Stream<NQ> &S, int kappa)
{
  const int n = T.n, lane = T.lane, ldd = T.ldd;
  double bk[NQ], rd[NQ], acc[NQ], mu[NQ], sc[NQ];
  long long rexpj[NQ];
  load_bf_row<NQ>(T, PL, kappa, bk);
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    const int c = lane + 64 * q;
    rd[q]       = (c < kappa) ? T.rdg[c] : 1.0;
    rexpj[q]    = (c <= kappa) ? T.rexp[c] : 0;
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    settle(rd[q]);
    settle(rexpj[q]);
  }
  scale_vector<NQ>(T, -1, rexpj, 0, sc);
  double gkk = 0.0;
  gso_pass<NQ>(T, PL, S, kappa, bk, sc, acc, gkk);
  const bool ok    = mu_from_r<NQ>(T, kappa, acc, rd, mu);
  const double rkk = finish_diag2<NQ>(mu, acc, gkk, kappa);
  store_gso_row<NQ>(T, PL, kappa, mu, acc, rkk);
  return ok;
}

// LLLReduction::babai(kappa, kappa, 0) followed by update_gso_row(kappa, kappa) (lll.h:107-122).
// 1 ok, 0 GSO failure, -1 babai failure, -2 multiplier beyond 63 bits.
template <int NQ>
__device__ __forceinline__ int babai2(Lattice<NQ> &T, Planes &PL, Stream<NQ> &S, int kappa, double eta,
                                      bool P_fuse)
{
  const int n = T.n, lane = T.lane, ldd = T.ldd, ldn = T.ldn;
  const int sr_start = 0;
  double bk[NQ], rd[NQ], acc[NQ], mu[NQ];
  long long rexpj[NQ];
  load_bf_row<NQ>(T, PL, kappa, bk);
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    const int c = lane + 64 * q;
    rd[q]       = (c < kappa) ? T.rdg[c] : 1.0;
    rexpj[q]    = (c < kappa) ? T.rexp[c] : 0;
  }
  long long rexpk    = T.rexp[kappa];
  long long max_expo = LLONG_MAX;
  // ordinary loads are waited for HERE: hipcc would otherwise place its own vmcnt wait at their
  // first use, inside a ring loop, and drain the DMA pipe on every step
#pragma unroll
  for (int q = 0; q < NQ; ++q)
  {
    settle(rd[q]);
    settle(rexpj[q]);
  }
  settle(rexpk);
  double gkk         = 0.0;
  double sc[NQ];
  bool have_pass     = false;  // the fused AXPY + Gram pass of the previous iteration has left acc / gkk
  for (int iter = 0;; ++iter)
  {
    if (!have_pass)
    {
      scale_vector<NQ>(T, kappa, rexpj, rexpk, sc);
      gso_pass<NQ>(T, PL, S, kappa, bk, sc, acc, gkk);
    }
    have_pass = false;
    if (!mu_from_r<NQ>(T, kappa, acc, rd, mu))
      return 0;
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
        e[q]           = (int)(rexpk - rexpj[q]);
        const double f = fabs(ldexp(mu[q], e[q]));  // get_mu, gso_interface.h:694-702
        need |= (j >= sr_start) && (f > eta);
        const long long ex = (long long)e[q] + fexponent(mu[q]);
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
    // ---- lll.cpp:202-220: lane k owns babai_mu[k]
    double bm[NQ], xs[NQ];
    unsigned long long nz[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      bm[q] = mu[q];
      xs[q] = 0.0;
      nz[q] = 0;
    }
    S.sp      = (const char *)(T.mu + (size_t)(kappa - 1) * ldd);
    S.sstride = (long)ldd * 8;
    S.sj      = kappa - 1;
    S.begin(K_SWEEP, kappa - 1 - sr_start, K_DUMMY, 0x7fffffff);
    sweep_phase<NQ>(S, bm, xs, e, nz, kappa, sr_start, (unsigned)lane);
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
    const bool small = !__any(big32);
    // ---- integer AXPY on row kappa (row_add / row_sub / row_addmul_si, gso.cpp:84-158)
    long long bv[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      const int c = lane + 64 * q;
      bv[q]       = (c < n) ? T.b[(size_t)kappa * ldn + c] : 0;
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      settle(bv[q]);
    const int anarrow = (PL.np16 >= kappa) ? 2 : (T.np >= kappa) ? 1 : 0;  // rows below kappa are read
    if (anarrow == 2 && small && P_fuse)
    {
      // ---- AXPY fused into the Gram pass of the new row (GramAxpyCons), then the recurrence
      int lx32[NQ];
      bool isk[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        lx32[q] = (int)lxv[q];
        isk[q]  = (lane + 64 * q == kappa);
      }
      scale_vector<NQ>(T, kappa, rexpj, rexpk, sc);
      const int nrec = kappa - 1;
      S.rp      = (const char *)PL.muA;
      S.rstride = (long)ldd * 8;
      S.rk      = 0;
      S.rlast   = kappa - 1;
      S.gp      = (const char *)PL.bT16;
      S.gstride = (long)ldd * 2;
      S.gmask   = ~0ull >> (63 - (kappa >> 3));
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        acc[q] = -0.0;
      S.begin(K_GRAM, (n + 1) >> 1, K_REC, nrec);
      const int qact = (kappa >> 6) + 1;
      if (NQ == 1 || qact == 1)
        gram_axpy_phase<NQ, 1>(S, acc, bv, lx32, sc, isk, (unsigned)lane, n);
      else if (NQ == 2 || qact == 2)
        gram_axpy_phase<NQ, (NQ >= 2 ? 2 : 1)>(S, acc, bv, lx32, sc, isk, (unsigned)lane, n);
      else if (NQ == 3 || qact == 3)
        gram_axpy_phase<NQ, (NQ >= 3 ? 3 : 1)>(S, acc, bv, lx32, sc, isk, (unsigned)lane, n);
      else
        gram_axpy_phase<NQ, (NQ >= 4 ? 4 : 1)>(S, acc, bv, lx32, sc, isk, (unsigned)lane, n);
      // row_op_end: update_bf(kappa), gso.cpp:24-48 — and with the row's exponent the scale of the sums
      store_row_and_refloat<NQ, true>(T, kappa, bv, bk, rexpk);
      store_mirror16<NQ>(T, PL, kappa, bv);
      const int ek = T.row_expo_on ? (int)rexpk : 0;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        acc[q] = ldexp(acc[q], isk[q] ? -2 * ek : -ek);
      gkk = g_rl_f64(acc[NQ == 1 ? 0 : (kappa >> 6)], kappa & 63);
      if (NQ == 1 || qact == 1)
        rec_phase<NQ, 1>(S, acc, nrec, (unsigned)lane);
      else if (NQ == 2 || qact == 2)
        rec_phase<NQ, (NQ >= 2 ? 2 : 1)>(S, acc, nrec, (unsigned)lane);
      else if (NQ == 3 || qact == 3)
        rec_phase<NQ, (NQ >= 3 ? 3 : 1)>(S, acc, nrec, (unsigned)lane);
      else
        rec_phase<NQ, (NQ >= 4 ? 4 : 1)>(S, acc, nrec, (unsigned)lane);
      have_pass = true;
      continue;
    }
    if (anarrow != 0)
    {
      int pairs = 0;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        pairs += (__builtin_popcountll(nz[q]) + 1) >> 1;
      // the issue side walks the chunks downwards: ab0 = chunk NQ-1 (= aq), ab1 the one below, …
      S.aq  = NQ - 1;
      S.ab0 = nz[NQ - 1];
      S.ab1 = NQ >= 2 ? nz[NQ >= 2 ? NQ - 2 : 0] : 0;
      S.ab2 = NQ >= 3 ? nz[NQ >= 3 ? NQ - 3 : 0] : 0;
      S.ab3 = NQ >= 4 ? nz[NQ >= 4 ? NQ - 4 : 0] : 0;
      if (anarrow == 2)
      {
        S.ap      = (const char *)PL.b16;
        S.astride = (long)ldn * 2;
        S.amask   = ~0ull >> (63 - ((n - 1) >> 3));
        S.begin(K_AXPY, pairs, K_DUMMY, 0x7fffffff);
        axpy_phase<NQ, 2>(S, bv, lxv, nz, small);
      }
      else
      {
        S.ap      = (const char *)T.b32;
        S.astride = (long)ldn * 4;
        S.amask   = ~0ull >> (63 - ((n - 1) >> 2));
        S.begin(K_AXPY, pairs, K_DUMMY, 0x7fffffff);
        axpy_phase<NQ, 4>(S, bv, lxv, nz, small);
      }
    }
    else if (T.wide_ring)
    {
      int rows8 = 0;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        rows8 += __builtin_popcountll(nz[q]);
      S.aq  = NQ - 1;
      S.ab0 = nz[NQ - 1];
      S.ab1 = NQ >= 2 ? nz[NQ >= 2 ? NQ - 2 : 0] : 0;
      S.ab2 = NQ >= 3 ? nz[NQ >= 3 ? NQ - 3 : 0] : 0;
      S.ab3 = NQ >= 4 ? nz[NQ >= 4 ? NQ - 4 : 0] : 0;
      S.ap8      = (const char *)T.b;
      S.astride8 = (long)ldn * 8;
      S.alen8    = n;
      S.begin(K_AXPY8, rows8, K_DUMMY, 0x7fffffff);
      axpy8_phase<NQ>(S, bv, lxv, nz);
    }
    else
      axpy_wide<NQ>(T, bv, lxv, kappa);
    // ---- row_op_end: update_bf(kappa), gso.cpp:24-48
    store_row_and_refloat<NQ, true>(T, kappa, bv, bk, rexpk);
    store_mirror16<NQ>(T, PL, kappa, bv);
    // the requests of the next pass must see these stores
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __threadfence_block();
  }
  const double rkk = finish_diag2<NQ>(mu, acc, gkk, kappa);
  store_gso_row<NQ>(T, PL, kappa, mu, acc, rkk);
  return 1;
}

// mode 0: update_gso() (every row, no size reduction); mode 1: size_reduction(kmin,kend);
// mode 2: (re)build bfT / row_expo / the narrow mirrors from b for every row (after a basis upload)
template <int NQ>
__global__ void __launch_bounds__(256, Cfg<NQ>::WAVES_PER_SIMD)
    gso_sweep2_kernel(GsoBatch P, double *muA, short *m16, int *flag16, int kmin, int kend, double eta, int mode)
{
  using C        = Cfg<NQ>;
  const int lane = threadIdx.x & 63;
  const int wpb  = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  Stream<NQ> S;
  S.base   = (unsigned)(wave * C::RING);
  S.hoff   = 0;
  S.toff   = 0;
  S.lane4  = lane * 4;
  S.lane16 = lane * 16;
  for (int L = blockIdx.x * wpb + wave; L < P.batch; L += gridDim.x * wpb)
  {
    Lattice<NQ> T;
    T.d           = P.d;
    T.n           = P.n;
    T.ldd         = P.ldd;
    T.ldn         = P.ldn;
    T.row_expo_on = P.row_expo;
    T.lane        = lane;
    T.b           = P.b + (size_t)L * P.d * P.ldn;
    T.bfT         = P.bfT + (size_t)L * P.n * P.ldd;
    T.mu          = P.mu + (size_t)L * P.d * P.ldd;
    T.muT         = P.muT + (size_t)L * P.d * P.ldd;
    T.r           = P.r + (size_t)L * P.d * P.ldd;
    T.rdg         = P.rdg + (size_t)L * P.d;
    T.rexp        = P.rexp + (size_t)L * P.d;
    T.bfT32       = P.bfT32 + (size_t)L * P.n * P.ldd;
    T.b32         = P.b32 + (size_t)L * P.d * P.ldn;
    T.narrow_flag = P.narrow + (size_t)L * P.d;
    T.np          = 0;
    T.f32ok       = 0;
    T.wide_ring   = P.wide_ring;
    Planes PL;
    PL.muA = muA + (size_t)L * P.d * P.ldd;
    // m16: [batch][n*ldd + d*ldn] shorts — bT16 then b16 of each lattice
    PL.bT16   = m16 + (size_t)L * ((size_t)P.n * P.ldd + (size_t)P.d * P.ldn);
    PL.b16    = PL.bT16 + (size_t)P.n * P.ldd;
    PL.flag16 = flag16 + (size_t)L * P.d;
    PL.np16   = 0;
    S.dummy = (const char *)T.b32;
    if (mode != 2)
    {  // narrow prefix from the per-row flags (FPHIP_GSO_NARROW=0: P.use_narrow == 0)
      int p = 0;
      while (P.use_narrow && p < P.d && __builtin_amdgcn_readfirstlane(T.narrow_flag[p]) != 0)
        ++p;
      T.np = p;
      p    = 0;
      while (P.use_narrow > 1 && p < P.d && __builtin_amdgcn_readfirstlane(PL.flag16[p]) != 0)
        ++p;
      PL.np16 = p;  // (use_narrow: 0 = 8-byte arrays only, 1 = 4-byte mirrors, 2 = 2-byte mirrors too)
    }
    int status = 1;
    if (mode == 2)
    {
      for (int i = 0; i < P.d; ++i)
      {
        long long bv[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
        {
          const int c = lane + 64 * q;
          bv[q]       = (c < P.n) ? T.b[(size_t)i * P.ldn + c] : 0;
        }
        store_row_and_refloat<NQ, true>(T, i, bv);
        store_mirror16<NQ>(T, PL, i, bv);
      }
    }
    else
    {
      for (int kappa = kmin; kappa < kend; ++kappa)
      {
        // the requests of this row must see the stores of the previous one
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __threadfence_block();
        if (mode == 1 && kappa > 0)
        {
          const int rc = babai2<NQ>(T, PL, S, kappa, eta, P.use_narrow > 2);
          if (rc != 1)
          {
            status = rc;
            break;
          }
        }
        else if (!update_full<NQ>(T, PL, S, kappa))
        {
          status = 0;
          break;
        }
      }
    }
    if (lane == 0)
      P.status[L] = status;
  }
}

template __global__ void gso_sweep2_kernel<1>(GsoBatch, double *, short *, int *, int, int, double, int);
template __global__ void gso_sweep2_kernel<2>(GsoBatch, double *, short *, int *, int, int, double, int);
template __global__ void gso_sweep2_kernel<3>(GsoBatch, double *, short *, int *, int, int, double, int);
template __global__ void gso_sweep2_kernel<4>(GsoBatch, double *, short *, int *, int, int, double, int);

}  // namespace s2
}  // namespace fphip
