// lll_kernel.hip — batched LLL reduction for gfx950: LLLReduction<Z_NR<long>, FP_NR<double>>::lll
// over MatGSO with GSO_ROW_EXPO, one wavefront per lattice, bit-exact decisions.
//
// Reference behaviour reproduced:
//   LLLReduction::lll            fplll/lll.cpp:44-164   (kappa loop, Lovasz test, insertion index,
//                                                        zero-row handling, set_r, iteration limit)
//   LLLReduction::babai          fplll/lll.cpp:166-224  (gso_wave.h, shared with the sweep kernel)
//   MatGSO::move_row             fplll/gso.cpp:289-366  (row rotation)
//   MatGSOInterface::update_gso_row (incremental, from gso_valid_cols) gso_interface.cpp:131-164
//   MatGSO::get_gram (lazy, NaN-invalidated cache gf)  fplll/gso.h:314-331
//   row_op_end / invalidate_gram_row                   gso_interface.cpp:32-53, gso.cpp:50-54
//   Matrix::get_max_exp, Z_NR<long>::exponent          nr/matrix.cpp:127-134, nr/nr_Z_l.inl:40-48
//
// Design.  Every floating-point value of the GSO is a deterministic function of the current
// integer basis (each r/mu/g entry is produced by one fixed operation sequence), so WHEN a value is
// computed does not matter — only that stale values are never used.  The kernel therefore keeps
// the reference's two caches, in its own form:
//   * rows never move.  fplll's move_row rotates row POINTERS; here a lane-resident slot table
//     (SlotMap: lane p holds the physical slot of row position p) is rotated with two cross-lane
//     shuffles, and every row-indexed array (b, bfT columns, mu, r, muT columns, rdg, row_expo, gf,
//     valid-column counts) is addressed through it.  Rows streamed through the LDS-DMA ring are
//     read back with a per-lane gather (ds_read at slot offset) — LDS gathers are free of the
//     coalescing rules of HBM.
//   * gf[slot_i][slot_j] is a SYMMETRIC Gram cache keyed by the vectors themselves, so a row move
//     needs no Gram rotation at all (fplll's rotate_gram_left/right, matrix.cpp:65-92, is O(n)
//     element swaps); a changed vector invalidates its row and column (NaN).
//   * vc[slot] = number of valid columns of that vector's mu/r row (gso_valid_cols).  Incremental
//     update: a few new columns are computed row-wise with v_readlane chains (the sequential sums
//     of the reference), a fully invalid row with the streaming column-oriented recurrence.
// After the kernel the rows are written out in position order (b2) and the host rebuilds the
// identity-layout GSO with the sweep kernel, which yields the same values.

#include "lll_wave.h"

namespace fphip
{

#if FPHIP_LLL_PROF
// profiling build only: per phase kind {phases, rows, start-up ticks, total ticks}, then [4 * LS_KINDS] = kernel
// ticks summed over the waves, [+1] = LLL iterations, [+2] = waves
__device__ unsigned long long fphip_lll_prof_dev[4 * LS_KINDS + 4];
#endif

// info[4] per lattice: final_kappa, n_swaps, zeros, loop iterations (low 31 bits)
// EARLY: the extended kernel — LLL_EARLY_RED (lll.cpp:84-99) when P.lll_early is set, the transformation matrix u
// when P.u is — its own instantiations in their own translation unit (lll_kernel_early.hip): the early pass
// costs the plain kernels 4-8 VGPRs otherwise, and lll_kernel<4> sits at the 256 that two waves per SIMD allow
template <int NQ, bool EARLY>
__global__ void __launch_bounds__(256)
    lll_kernel(GsoBatch P, int kmin, int kstart, int kend, double delta, double eta, double logdelta)
{
  extern __shared__ __attribute__((aligned(16))) char lll_smem[];
  const int lane = threadIdx.x & 63;
  const int wpb  = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  ReduceRing<NQ> ring;
  ring.init(wave, lane);
  const int d = P.d, n = P.n, ldd = P.ldd, ldn = P.ldn;
#if FPHIP_LLL_PROF
  const unsigned long long pk0 = LStream<NQ>::now();
  const unsigned long long pc0 = __builtin_readcyclecounter();  // s_memtime: shader clock
  unsigned long long piter     = 0;
  bool pany                    = false;
#endif
  for (int L = blockIdx.x * wpb + wave; L < P.batch; L += gridDim.x * wpb)
  {
    Lattice<NQ> T;
    T.d           = d;
    T.n           = n;
    T.ldd         = ldd;
    T.ldn         = ldn;
    T.row_expo_on = P.row_expo;
    T.lane        = lane;
    T.b           = P.b + (size_t)L * d * ldn;
    T.bfT         = P.bfT + (size_t)L * n * ldd;
    T.mu          = P.mu + (size_t)L * d * ldd;
    T.muT         = P.muT + (size_t)L * d * ldd;
    T.r           = P.r + (size_t)L * d * ldd;
    T.rdg         = P.rdg + (size_t)L * d;
    T.rexp        = P.rexp + (size_t)L * d;
    // the positional narrow prefix of the sweep kernels does not apply here (rows live in slots): np = 0
    // keeps the shared code on the 8-byte rows.  The FLOAT mirror of bf is used, lattice-wide: f32ok
    T.bfT32       = P.bfT32 + (size_t)L * n * ldd;
    T.b32         = (int *)T.b;
    T.narrow_flag = (int *)T.rexp;
    T.np          = 0;
    T.f32ok       = all_rows_narrow<NQ>(P, (size_t)L, lane);
    if constexpr (EARLY)
      T.u = P.u ? P.u + (size_t)L * d * ldd : nullptr;
    LllCtx C{P.gf + (size_t)L * d * ldd, P.vc + (size_t)L * d};
    SlotMap<NQ> M;
    int final_kappa, nswaps, zeros, vp = 0;
    int last_early_red = 0;  // LLL_EARLY_RED: the LLLReduction object's member (lll.h:70), kept by a session
    long long iter;
    if (P.sess_mode == 2)
    {  // resume the session: slot table, verified prefix, narrow flag; everything else is in place
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        M.sl[q] = P.sess_slots[(size_t)L * 256 + lane + 64 * q];
      vp      = uni(P.sess_state[4 * L + 0]);
      T.f32ok = uni(P.sess_state[4 * L + 1]);
      if (uni(P.sess_state[4 * L + 2]) != P.lll_siegel)
        vp = 0;  // the prefix was verified against the other swap test
      last_early_red = uni(P.sess_state[4 * L + 3]);
    }
    else
      lll_init_state<NQ>(T, C, M);
    if (P.sess_mode != 0 && P.sess_ndirty > 0)
    {
      // the caller's row operations since the last launch: row p <- the given integers, then row_op_end(p, p + 1)
      // (update_bf, the vector's Gram row and column, the GSO rows from p on: gso_interface.cpp:32-53)
      const long long *pos = P.sess_in;
      const long long *rows = P.sess_in + P.sess_ndirty;
      const long long *urows = rows + (size_t)P.sess_ndirty * ldn;
      for (int t = 0; t < P.sess_ndirty; ++t)
      {
        const int p = uni((int)pos[t]);
        const int s = M.phys(p);
        long long bv[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
        {
          const int c = lane + 64 * q;
          bv[q]       = (c < n) ? rows[(size_t)t * ldn + c] : 0;
        }
        store_row_and_refloat<NQ, false>(T, s, bv);
        if (T.u != nullptr)
        {  // the caller's row operations act on u as well (gso.cpp:88-91, ...)
#pragma unroll
          for (int q = 0; q < NQ; ++q)
          {
            const int c = lane + 64 * q;
            if (c < ldd)
              T.u[(size_t)s * ldd + c] = urows[(size_t)t * ldd + c];
          }
        }
        after_rowop<NQ>(T, C, M, p);
        vp = min(vp, p);
        __threadfence_block();
      }
    }
    const int status = lll_run(T, C, M, ring, kmin, kstart, kend, delta, eta, logdelta,
                                        final_kappa, nswaps, zeros, iter, vp, P.lll_siegel != 0,
                                        EARLY && P.lll_early != 0, &last_early_red);
    if (P.sess_mode != 0)
    {
      // leave the state behind, and the caller's view of it: everything in position order
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        P.sess_slots[(size_t)L * 256 + lane + 64 * q] = M.sl[q];
      if (lane == 0)
      {
        P.sess_state[4 * L + 0] = (status == 1) ? vp : 0;
        P.sess_state[4 * L + 1] = T.f32ok;
        P.sess_state[4 * L + 2] = P.lll_siegel;
        P.sess_state[4 * L + 3] = last_early_red;
      }
      char *out        = P.sess_out + (size_t)L * fphip_session_out_stride(d, ldd, ldn);
      long long *ob    = (long long *)out;
      double *omu      = (double *)(out + (size_t)d * ldn * 8);
      double *orr      = omu + (size_t)d * ldd;
      long long *oexp  = (long long *)(orr + (size_t)d * ldd);
      int *ovc         = (int *)(oexp + d);
      // four rows per step, every load of the step before its stores (a row at a time would pay the latency
      // of a load per row: the compiler cannot know that the output does not alias the state)
      constexpr int G = 4;
      for (int p0 = 0; p0 < d; p0 += G)
      {
        int sr[G];
        long long bb[G][NQ];
        double mm[G][NQ], rr[G][NQ];
#pragma unroll
        for (int u = 0; u < G; ++u)
          sr[u] = M.phys(min(p0 + u, d - 1));
#pragma unroll
        for (int u = 0; u < G; ++u)
#pragma unroll
          for (int q = 0; q < NQ; ++q)
          {
            const int j = lane + 64 * q;
            bb[u][q]    = (j < ldn) ? T.b[(size_t)sr[u] * ldn + j] : 0;
            mm[u][q]    = (j < ldd) ? T.mu[(size_t)sr[u] * ldd + j] : 0.0;
            rr[u][q]    = (j < ldd) ? T.r[(size_t)sr[u] * ldd + j] : 0.0;
          }
        long long ee = 0;
        int vv       = 0;
#pragma unroll
        for (int u = 0; u < G; ++u)
          if (lane == u)
          {
            ee = T.rexp[sr[u]];
            vv = C.vc[sr[u]];
          }
#pragma unroll
        for (int u = 0; u < G; ++u)
          if (p0 + u < d)
          {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
            {
              const int j = lane + 64 * q;
              if (j < ldn)
                ob[(size_t)(p0 + u) * ldn + j] = bb[u][q];
              if (j < ldd)
              {
                omu[(size_t)(p0 + u) * ldd + j] = mm[u][q];
                orr[(size_t)(p0 + u) * ldd + j] = rr[u][q];
              }
            }
          }
        if (lane < G && p0 + lane < d)
        {
          oexp[p0 + lane] = ee;
          ovc[p0 + lane]  = vv;
        }
      }
      if (T.u != nullptr)
      {
        long long *ou = (long long *)(out + fphip_session_out_bytes(d, ldd, ldn));
        for (int p = 0; p < d; ++p)
        {
          const int s = M.phys(p);
#pragma unroll
          for (int q = 0; q < NQ; ++q)
          {
            const int c = lane + 64 * q;
            if (c < ldd)
              ou[(size_t)p * ldd + c] = T.u[(size_t)s * ldd + c];
          }
        }
      }
    }
    else
    {
      lll_write_ordered<NQ>(T, M, P.b2 + (size_t)L * d * ldn);
      if (T.u != nullptr)
      {  // u in position order as well
        long long *uo = P.u2 + (size_t)L * d * ldd;
        for (int p = 0; p < d; ++p)
        {
          const int s = M.phys(p);
#pragma unroll
          for (int q = 0; q < NQ; ++q)
          {
            const int c = lane + 64 * q;
            if (c < ldd)
              uo[(size_t)p * ldd + c] = T.u[(size_t)s * ldd + c];
          }
        }
      }
    }
    if (lane == 0)
    {
      P.status[L]           = status;
      P.lll_info[4 * L + 0] = final_kappa;
      P.lll_info[4 * L + 1] = nswaps;
      P.lll_info[4 * L + 2] = zeros;
      P.lll_info[4 * L + 3] = (int)(iter & 0x7fffffff);
    }
    __threadfence_block();
#if FPHIP_LLL_PROF
    piter += (unsigned long long)iter;
    pany = true;
#endif
  }
#if FPHIP_LLL_PROF
  if (pany && lane == 0)
  {
    for (int i = 0; i < 4 * LS_KINDS; ++i)
      atomicAdd(&fphip_lll_prof_dev[i], ring.pf[i]);
    atomicAdd(&fphip_lll_prof_dev[4 * LS_KINDS + 0], LStream<NQ>::now() - pk0);
    atomicAdd(&fphip_lll_prof_dev[4 * LS_KINDS + 1], piter);
    atomicAdd(&fphip_lll_prof_dev[4 * LS_KINDS + 2], 1ull);
    atomicAdd(&fphip_lll_prof_dev[4 * LS_KINDS + 3], (unsigned long long)__builtin_readcyclecounter() - pc0);
  }
#endif
}

#ifndef FPHIP_LLL_KERNEL_EARLY
#define FPHIP_LLL_KERNEL_EARLY 0
#endif
template __global__ void lll_kernel<1, FPHIP_LLL_KERNEL_EARLY != 0>(GsoBatch, int, int, int, double, double, double);
template __global__ void lll_kernel<2, FPHIP_LLL_KERNEL_EARLY != 0>(GsoBatch, int, int, int, double, double, double);
template __global__ void lll_kernel<3, FPHIP_LLL_KERNEL_EARLY != 0>(GsoBatch, int, int, int, double, double, double);
template __global__ void lll_kernel<4, FPHIP_LLL_KERNEL_EARLY != 0>(GsoBatch, int, int, int, double, double, double);

}  // namespace fphip

#if FPHIP_LLL_PROF
// profiling build only (not part of any header): read and clear the counters above
extern "C" int fphip_debug_lll_prof(unsigned long long *out, int count)
{
  unsigned long long h[4 * fphip::LS_KINDS + 4] = {0};
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(fphip::fphip_lll_prof_dev), sizeof(h)) != hipSuccess)
    return -1;
  for (int i = 0; i < count && i < 4 * fphip::LS_KINDS + 4; ++i)
    out[i] = h[i];
  unsigned long long z[4 * fphip::LS_KINDS + 4] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(fphip::fphip_lll_prof_dev), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
