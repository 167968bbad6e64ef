// bkz_kernel.hip — batched BKZ reduction for gfx950: BKZReduction<Z_NR<long>, FP_NR<double>>::bkz()
// (primal, no strategies: Strategy::EmptyStrategy for every block size — no pruning, no
// preprocessing, bkz_param.h:124-132; flags BKZ_DEFAULT or BKZ_MAX_LOOPS), one wavefront per
// lattice, the whole reduction in ONE launch with the GSO state of the lattice device-resident
// from the first tour to the last.
//
// Reference behaviour reproduced:
//   BKZReduction::bkz                 fplll/bkz.cpp:522-668  (tour loop, clean flag, loop limit)
//   tour / trunc_tour / hkz           bkz.cpp:360-441        (incl. the trailing size_reduction)
//   svp_reduction                     bkz.cpp:274-358        (size reduction, radius delta*r_kk,
//                                                             progress test old_first <= new_first)
//   svp_preprocessing                 bkz.cpp:100-124        (its lll(0, 0, kappa+block) call)
//   svp_postprocessing / _generic     bkz.cpp:126-272        (vector already in the basis, +-1
//                                                             coordinate, gcd tree)
//   EnumerationDyn::enumerate         enum/enumerate.cpp:58-159 (normalisation by 2^normexp)
//   enumerate_recursive + process_solution with FastEvaluator(1, BEST_N)
//                                     enum/enumerate_base.cpp:24-118, enumerate.cpp:218-239,
//                                     evaluator.h:122-134
//   LLLReduction::lll / size_reduction / babai, MatGSO::move_row, row_op_end   (lll_wave.h)
//
// Design.  A BKZ tour is a chain of ~d svp_reduction calls, each a few LLL sweeps, one small
// enumeration (10^2..10^5 nodes at beta <= 24 without pruning) and a handful of integer row
// operations; on a CPU the reference spends >90 % of it outside the enumeration.
// Launching those pieces one by one would be launch-latency bound, so the wave keeps the lattice
// and walks the whole schedule itself:
//   * GSO state = the LLL kernel's (slot table, symmetric Gram cache, valid-column counts): an
//     insertion is a slot rotation, the following lll(0, 0, kappa+beta) only recomputes what the
//     inserted vector invalidated;
//   * the block enumeration is the wave-per-subtree walk of enum_kernel.hip run on ONE tree by
//     this wave, depth-first — the reference's visiting order, so the in-kernel evaluator
//     (best solution so far, radius shrinks to its norm) ends on the reference's vector;
//   * post-processing works on lanes = columns; row swaps of the gcd tree are slot swaps.

#include "lll_wave.h"

namespace fphip
{

__device__ __forceinline__ int btri(int k) { return (k * (k - 1)) >> 1; }

// rows [first, last) changed: row_op_end(first, last), gso_interface.cpp:32-53
template <int NQ>
__device__ __forceinline__ void refloat_and_invalidate(Lattice<NQ> &T, LllCtx &C, const SlotMap<NQ> &M,
                                                       int first, int last)
{
  const int lane = T.lane, n = T.n, ldn = T.ldn;
  for (int p = first; p < last; ++p)
  {
    const int s = M.phys(p);
    long long bv[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      const int c = lane + 64 * q;
      bv[q]       = (c < n) ? T.b[(size_t)s * ldn + c] : 0;
    }
    store_row_and_refloat<NQ>(T, s, bv);
    after_rowop<NQ>(T, C, M, p);
    __threadfence_block();
  }
}

// info[4] per lattice: tours, enumeration nodes (low / high 32 bits), enumeration calls
// status: 1 RED_SUCCESS, 8 RED_BKZ_LOOPS_LIMIT, <= 0 the failing LLL status (lll_wave.h)
template <int NQ>
__global__ void __launch_bounds__(256)
    bkz_kernel(GsoBatch P, int block_size, double delta, double eta, double logdelta,
               int use_max_loops, int max_loops, int stack_doubles)
{
  constexpr int IPS = (NQ + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) char bkz_smem[];
  const int lane = threadIdx.x & 63;
  const int wpb  = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  ReduceRing<NQ> ring;
  ring.init(wave, lane);
  // enumeration stack (triangular column stack of the walk) behind the rings
  double *stk = (double *)(bkz_smem + (size_t)wpb * ReduceRing<NQ>::BYTES) +
                (size_t)wave * stack_doubles;
  const int d = P.d, n = P.n, ldd = P.ldd, ldn = P.ldn;
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
    LllCtx C{P.gf + (size_t)L * d * ldd, P.vc + (size_t)L * d};
    double *mu_blk = P.enum_mu + (size_t)L * (64 * 63 / 2);  // scaled mu rows of the current block
    SlotMap<NQ> M;
    if (P.bkz_active[L] == 0)
    {  // this lattice's reduction has ended in an earlier launch: keep its basis
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        M.sl[q] = lane + 64 * q;
      lll_write_ordered<NQ>(T, M, P.b2 + (size_t)L * d * ldn);
      continue;
    }
    lll_init_state<NQ>(T, C, M);

    int vp = 0;  // verified prefix of the LLL loop (lll_wave.h), kept across every call of the run
    auto upd   = [&](int k, int last) { return update_row_cached(T, C, M, ring, k, last); };
    auto after = [&](int k)
    {
      after_rowop<NQ>(T, C, M, k);
      vp = min(vp, k);
    };

    // trailing zero rows are not part of the lattice, bkz.cpp:35-37
    int num_rows = d;
    for (; num_rows > 0; --num_rows)
    {
      bool nz = false;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int c = lane + 64 * q;
        if (c < n)
          nz |= (T.b[(size_t)(num_rows - 1) * ldn + c] != 0);
      }
      if (__any(nz))
        break;
    }

    int status = 1, tours = 0, ncalls = 0;
    unsigned long long total_nodes = 0;
    bool stop = block_size < 2;
    for (int loop = 0; !stop; ++loop)
    {
      if (use_max_loops && loop >= max_loops)
      {
        status = 8;
        break;
      }
      bool clean = true;
      // one tour = trunc_tour blocks, hkz blocks, then the trailing size reduction (op == nops-1)
      const int n_trunc = max(num_rows - block_size, 0);
      const int hkz_lo  = max(num_rows - block_size, 0);
      const int n_hkz   = max(num_rows - 1 - hkz_lo, 0);
      const int nops    = n_trunc + n_hkz + 1;
      for (int op = 0; op < nops && status == 1; ++op)
      {
        const bool tail = op == nops - 1;
        int kappa, bs;
        if (op < n_trunc)
        {
          kappa = op;
          bs    = block_size;
        }
        else
        {
          kappa = hkz_lo + (op - n_trunc);
          bs    = num_rows - kappa;
        }
        double old_first   = 0.0;
        int old_first_expo = 0;
        for (int pass = 0; pass < 2 && status == 1; ++pass)
        {
          // ---- lll_obj.size_reduction(kmin, kend, sr_start), lll.h:107-122
          int kmin = 0, kend = kappa + 1, sr0 = 0;
          if (tail)
          {  // hkz(): lll_obj.size_reduction(max_row - 1, max_row, max_row - 2), bkz.cpp:437
            kmin = num_rows - 1;
            kend = num_rows;
            sr0  = num_rows - 2;
            if (pass == 1 || num_rows < 2)
              break;
          }
          // rows below the verified prefix are size-reduced with r(k,k) in place: babai and
          // update_gso_row are no-ops on them
          if (status == 1)
            status = size_reduce_call(T, C, M, ring, max(kmin, min(vp, kend)), kend, eta, sr0, vp);
          if (tail || status != 1)
            break;
          const int sk0 = M.phys(kappa);
          if (pass == 1)
          {  // progress test, bkz.cpp:349-357
            double new_first = T.rdg[sk0];
            new_first        = ldexp(new_first, (int)(2 * T.rexp[sk0]) - old_first_expo);
            clean            = clean && (old_first <= new_first);
            break;
          }
          old_first      = T.rdg[sk0];
          old_first_expo = (int)(2 * T.rexp[sk0]);
          // ---- svp_preprocessing: lll(0, 0, kappa + bs), bkz.cpp:107-113
          {
            int fk, ns, zs;
            long long it;
            const int rc = lll_run_call(T, C, M, ring, 0, 0, kappa + bs, delta, eta, logdelta,
                                            fk, ns, zs, it, vp);
            if (rc != 1)
            {
              status = rc;
              break;
            }
          }
          // ---- enumeration of the block, enumerate.cpp:88-141 (normalisation) ----------------
          double rd;   // lane i: rdiag[i]
          double maxdist;
          {
            int sl_blk = 0;  // lane i: slot of row kappa + i
#pragma unroll
            for (int q = 0; q < NQ; ++q)
            {
              // gather the block's slots into one register (bs <= 64)
              const int src = kappa + lane;  // position wanted by this lane
              const int v   = __shfl(M.sl[q], src & 63);
              if ((src >> 6) == q)
                sl_blk = v;
            }
            const bool in    = lane < bs;
            const double rr  = in ? T.rdg[sl_blk] : 0.0;
            const int e2     = in ? (int)(2 * T.rexp[sl_blk]) : 0;
            int ne           = in ? (int)min((long long)e2 + fexponent(rr), (long long)INT_MAX) : INT_MIN;
            ne               = max(wave_max_i32(ne), -1);  // normexp starts at -1, enumerate.cpp:88
            rd               = in ? ldexp(rr, e2 - ne) : 0.0;
            const int sk     = M.phys(kappa);
            double md        = T.rdg[sk] * delta;  // max_dist *= delta, bkz.cpp:317
            maxdist          = ldexp(md, (int)(2 * T.rexp[sk]) - ne);
            // mu rows of the block with the row exponents applied (get_mu, gso_interface.h:694-702),
            // packed like the walk's column stack
            for (int k = 1; k < bs; ++k)
            {
              const int skk      = M.phys(kappa + k);
              const long long ek = T.rexp[skk];
              if (lane < k)
              {
                const double m = T.mu[(size_t)skk * ldd + kappa + lane];
                mu_blk[btri(k) + lane] = ldexp(m, (int)(ek - T.rexp[sl_blk]));
              }
            }
            __threadfence_block();
          }
          // ---- the walk: enumerate_recursive, depth-first, FastEvaluator(1) in the kernel ------
          double best_x    = 0.0;  // lane i: coefficient of the best solution
          bool have_sol    = false;
          {
            double xs = 0.0, cs = 0.0, pds = 0.0;
            int dxs = 0, ddxs = 0;
            unsigned long long cnt = 0;
            double bnd = maxdist;  // no pruning: every level's bound is maxdist
            int k      = bs;
            double S   = 0.0;
            double nd  = 0.0;
            bool done  = false;
            while (!done)
            {
              // CHILD chain: descend while the first child survives
              for (;;)
              {
                k               = __builtin_amdgcn_readfirstlane(k);
                const int kc    = k - 1;
                const double c1 = g_rl_f64(S, kc);
                const double x1 = round(c1);  // roundto(), enumerate_base.h:33-34
                const double a1 = x1 - c1;
                const double n1 = nd + a1 * a1 * g_rl_f64(rd, kc);
                if (!(n1 <= bnd))
                {
                  done = k >= bs;
                  break;
                }
                if (lane < k)
                  stk[btri(k) + lane] = S;
                {
                  const int s1  = (c1 >= x1) ? 1 : -1;
                  const bool me = lane == kc;
                  cs            = me ? c1 : cs;
                  xs            = me ? x1 : xs;
                  pds           = me ? nd : pds;
                  dxs           = me ? s1 : dxs;
                  ddxs          = me ? s1 : ddxs;
                  cnt += me ? 1ull : 0ull;
                }
                k  = kc;
                nd = n1;
                if (k == 0)
                {
                  if (nd > 0.0)
                  {  // process_solution: keep it, shrink the radius to its norm
                    best_x   = xs;
                    have_sol = true;
                    maxdist  = nd;
                    bnd      = nd;
                  }
                  break;
                }
                const double mk = mu_blk[btri(k) + min(lane, k - 1)];
                S               = S - x1 * mk;
              }
              if (done)
                break;
              // STEP loop: next sibling at level k, climbing while they fail
              for (;;)
              {
                k                = __builtin_amdgcn_readfirstlane(k);
                const double par = stk[btri(k + 1) + min(lane, k)];  // S_{k+1}
                const double mk  = mu_blk[btri(k) + max(min(lane, k - 1), 0)];
                double xk        = g_rl_f64(xs, k);
                const double ck  = g_rl_f64(cs, k);
                const double pdk = g_rl_f64(pds, k);
                int dxk = __builtin_amdgcn_readlane(dxs, k), ddxk = __builtin_amdgcn_readlane(ddxs, k);
                if (pdk != 0.0)
                {
                  xk += (double)dxk;
                  ddxk = -ddxk;
                  dxk  = ddxk - dxk;
                }
                else
                {
                  xk += 1.0;
                }
                const bool me  = lane == k;
                xs             = me ? xk : xs;
                dxs            = me ? dxk : dxs;
                ddxs           = me ? ddxk : ddxs;
                const double a = xk - ck;
                nd             = pdk + a * a * g_rl_f64(rd, k);
                if (!(nd <= bnd))
                {
                  ++k;
                  if (k >= bs)
                  {
                    done = true;
                    break;
                  }
                  continue;
                }
                cnt += me ? 1ull : 0ull;
                if (k == 0)
                {
                  if (nd > 0.0)
                  {
                    best_x   = xs;
                    have_sol = true;
                    maxdist  = nd;
                    bnd      = nd;
                  }
                  continue;
                }
                S = par - xk * mk;
                break;
              }
            }
            // node total by the fplll rule (the initial descent is compensated, enumerate_base.cpp:181)
            unsigned long long tot = cnt;
            for (int off = 32; off > 0; off >>= 1)
              tot += (unsigned long long)__shfl_xor((long long)tot, off);
            total_nodes += tot - (unsigned long long)(bs - 1);
            ++ncalls;
          }
          // ---- svp_postprocessing, bkz.cpp:126-272 --------------------------------------------
          if (have_sol)
          {
            const bool in      = lane < bs;
            const uint64_t nzm = __ballot(in && best_x != 0.0);
            const uint64_t onm = __ballot(in && fabs(best_x) == 1.0);
            const int nz       = __popcll(nzm);
            const int iv       = onm ? 63 - __clzll((long long)onm) : -1;
            if (nz == 1)
            {
              if (iv > 0)
              {
                rotate_right<NQ>(M, kappa, kappa + iv, lane);
                clamp_valid<NQ>(T, C, M, kappa);
                vp = min(vp, kappa);
              }
            }
            else if (iv != -1)
            {
              // b[kappa+iv] += sum_i (sol_iv * sol_i) b[kappa+i]
              const double sv = g_rl_f64(best_x, iv);
              const int st    = M.phys(kappa + iv);
              long long bv[NQ];
#pragma unroll
              for (int q = 0; q < NQ; ++q)
              {
                const int c = lane + 64 * q;
                bv[q]       = (c < n) ? T.b[(size_t)st * ldn + c] : 0;
              }
              for (int i = 0; i < bs; ++i)
              {
                const double xi = g_rl_f64(best_x, i);
                if (xi == 0.0 || i == iv)
                  continue;
                const long long lx = (long long)(sv * xi);
                const int si       = M.phys(kappa + i);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                {
                  const int c = lane + 64 * q;
                  if (c < n)
                    bv[q] = (long long)((unsigned long long)bv[q] +
                                        (unsigned long long)T.b[(size_t)si * ldn + c] * (unsigned long long)lx);
                }
              }
              store_row_and_refloat<NQ>(T, st, bv);
              after_rowop<NQ>(T, C, M, kappa + iv);
              vp = min(vp, kappa);
              __threadfence_block();
              if (iv > 0)
              {
                rotate_right<NQ>(M, kappa, kappa + iv, lane);
                clamp_valid<NQ>(T, C, M, kappa);
              }
            }
            else
            {
              // generic case: gcd tree on |x| with the dual row operations (bkz.cpp:205-272)
              double x = in ? best_x : 0.0;
              for (int i = 0; i < bs; ++i)
              {
                if (g_rl_f64(x, i) < 0.0)
                {  // negate_row_of_b(i + kappa)
                  const int si = M.phys(kappa + i);
#pragma unroll
                  for (int q = 0; q < NQ; ++q)
                  {
                    const int c = lane + 64 * q;
                    if (c < n)
                      T.b[(size_t)si * ldn + c] = -T.b[(size_t)si * ldn + c];
                  }
                }
              }
              x = fabs(x);
              __threadfence_block();
              auto swap_rows = [&](int pa, int pb)
              {  // row_swap: the rows are re-floated below, so swapping their slots is equivalent
                const int sa = M.phys(pa), sb = M.phys(pb);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                {
                  const int p = lane + 64 * q;
                  M.sl[q]     = (p == pa) ? sb : ((p == pb) ? sa : M.sl[q]);
                }
              };
              for (int off = 1; off < bs; off *= 2)
              {
                for (int k = bs - 1; k - off >= 0; k -= 2 * off)
                {
                  double xk = g_rl_f64(x, k), xo = g_rl_f64(x, k - off);
                  if (xk == 0.0 && xo == 0.0)
                    continue;
                  if (xk < xo)
                  {
                    const double t = xk;
                    xk             = xo;
                    xo             = t;
                    swap_rows(kappa + k - off, kappa + k);
                  }
                  while (xo != 0.0)
                  {
                    // while (x[k-off] <= x[k]) { x[k] -= x[k-off]; row_add(k-off, k); }
                    const double qd = floor(xk / xo);
                    if (qd >= 1.0)
                    {
                      xk                 = xk - qd * xo;
                      const long long lq = (long long)qd;
                      const int sdst = M.phys(kappa + k - off), ssrc = M.phys(kappa + k);
#pragma unroll
                      for (int q = 0; q < NQ; ++q)
                      {
                        const int c = lane + 64 * q;
                        if (c < n)
                          T.b[(size_t)sdst * ldn + c] =
                              (long long)((unsigned long long)T.b[(size_t)sdst * ldn + c] +
                                          (unsigned long long)T.b[(size_t)ssrc * ldn + c] *
                                              (unsigned long long)lq);
                      }
                      __threadfence_block();
                    }
                    const double t = xk;
                    xk             = xo;
                    xo             = t;
                    swap_rows(kappa + k - off, kappa + k);
                  }
                  x = (lane == k) ? xk : ((lane == k - off) ? xo : x);
                }
              }
              refloat_and_invalidate<NQ>(T, C, M, kappa, kappa + bs);
              vp = min(vp, kappa);
              clamp_valid<NQ>(T, C, M, kappa);
              rotate_right<NQ>(M, kappa, kappa + bs - 1, lane);
              clamp_valid<NQ>(T, C, M, kappa);
            }
            __threadfence_block();
          }
        }
      }
      if (status != 1)
        break;
      ++tours;
      if (clean || block_size >= num_rows)
        break;
    }
    lll_write_ordered<NQ>(T, M, P.b2 + (size_t)L * d * ldn);
    if (lane == 0)
    {
      P.status[L]           = status;
      P.lll_info[4 * L + 0] = tours;
      P.lll_info[4 * L + 1] = (int)(unsigned)(total_nodes & 0xffffffffull);
      P.lll_info[4 * L + 2] = (int)(unsigned)(total_nodes >> 32);
      P.lll_info[4 * L + 3] = ncalls;
      P.bkz_rows[L]         = num_rows;
    }
    __threadfence_block();
  }
}

template __global__ void bkz_kernel<1>(GsoBatch, int, double, double, double, int, int, int);
template __global__ void bkz_kernel<2>(GsoBatch, int, double, double, double, int, int, int);
template __global__ void bkz_kernel<3>(GsoBatch, int, double, double, double, int, int, int);
template __global__ void bkz_kernel<4>(GsoBatch, int, double, double, double, int, int, int);

}  // namespace fphip
