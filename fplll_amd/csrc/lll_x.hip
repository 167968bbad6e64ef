// lll_x.hip — LLL over MatGSO in a SELECTABLE floating-point type (ftx.h): double-double (the device
// stand-in for FP_NR<dd_real>) or plain double.  It is the second stage of the LLL-side precision
// ladder (Wrapper::lll, fplll/wrapper.cpp:281-359: fast_lll<double>, then the wider types on the basis
// the failed attempt left): BASELINE config 5's 256-dim NTRU-like lattice makes LLLReduction<.., double>
// stop with RED_BABAI_FAILURE ("infinite loop in babai") — in the reference and in lll_kernel.hip alike —
// and reduces fine at 106 bits.
//
// What it runs is the reference's algorithm, statement for statement:
//   LLLReduction::lll             fplll/lll.cpp:44-164   (kappa loop, Lovasz test, insertion index,
//                                                          zero rows, set_r, iteration limit)
//   LLLReduction::babai           lll.cpp:166-224        (rnd_we multipliers, failure test)
//   MatGSOInterface::update_gso_row  gso_interface.cpp:131-164, get_gram gso.h:314-331
//   MatGSO::row_addmul_we (long multipliers) gso.cpp:236-262, update_bf :24-48, row_op_end
//   gso_interface.cpp:32-53, move_row gso.cpp:289-366
// but NOT the reference's summation order: Gram entries and the recurrence are accumulated per lane
// (lane j owns column j, ascending over the other index) — with 106 bits the decisions have ~50 bits of
// slack; the exact-order double kernel is lll_kernel.hip.  Like hlll_x.hip, a double-double result
// cannot be pinned bit for bit (libqd is absent, ftx.h): the acceptance test is the reference's own
// is_lll_reduced predicate on the output plus lattice equality (tests/test_dd_gpu.py).
//
// One wavefront per lattice, lane = column index (NQ per lane).  Rows never move: a slot table in LDS
// maps positions to physical rows (b, bf, mu, r, row_expo, valid-column counts are keyed by slot; the
// columns of mu / r are positions — a move invalidates every row from its lower end on, exactly the
// reference's invalidate_gso_row); the Gram cache is keyed by the pair of slots.
#include "ftx.h"
#include "gso_device.h"

namespace fphip
{

struct LllX
{
  int batch, d, n, ldn, ldd, row_expo;
  long long *b, *b2;      // [batch][d][ldn]: rows by slot; b2: output, rows in position order
  double *bf;             // [batch][d][ldn]
  double *mu_hi, *mu_lo;  // [batch][d][ldd]: row = slot, column = position
  double *r_hi, *r_lo;
  double *gf_hi, *gf_lo;  // [batch][d][ldd]: Gram cache, entry (max slot, min slot); hi NaN = unknown
  double *mu_x, *r_x, *gf_x;  // quad-double: components 2 and 3 of the three arrays (two planes back to back each)
  long long *rexp;        // [batch][d] by slot
  int *status, *info;     // info[4]: final_kappa, n_swaps, zeros, iterations
  const int *only_failed; // precision ladder: non-null = only the lattices whose entry is not 1
  int kmin, kstart, kend;
  double delta, eta;
};

template <class FT> struct PlaneX;
template <> struct PlaneX<double>
{
  double *hi, *lo, *x2, *x3;
  __device__ __forceinline__ double ld(size_t i) const { return hi[i]; }
  __device__ __forceinline__ void st(size_t i, double v) const { hi[i] = v; }
};
template <> struct PlaneX<DD>
{
  double *hi, *lo, *x2, *x3;
  __device__ __forceinline__ DD ld(size_t i) const { return DD{hi[i], lo[i]}; }
  __device__ __forceinline__ void st(size_t i, DD v) const
  {
    hi[i] = v.hi;
    lo[i] = v.lo;
  }
};
template <> struct PlaneX<QD>
{
  double *hi, *lo, *x2, *x3;
  __device__ __forceinline__ QD ld(size_t i) const { return QD{{hi[i], lo[i], x2[i], x3[i]}}; }
  __device__ __forceinline__ void st(size_t i, QD v) const
  {
    hi[i] = v.x[0];
    lo[i] = v.x[1];
    x2[i] = v.x[2];
    x3[i] = v.x[3];
  }
};

// status: 1 RED_SUCCESS, 0 RED_GSO_FAILURE, -1 RED_BABAI_FAILURE, -2 multiplier beyond 63 bits,
// -3 RED_LLL_FAILURE (iteration limit)
template <int NQ, class FT> __global__ void __launch_bounds__(64) lll_x_kernel(LllX A)
{
  __shared__ int slot[256], vcol[256];
  __shared__ double lov_hi[257], lov_lo[257], rd_hi[256], rd_lo[256];
  __shared__ double lov_x2[257], lov_x3[257], rd_x2[256], rd_x3[256];  // (quad-double)
  const int lane = threadIdx.x & 63;
  const int dT = A.d, n = A.n, ldn = A.ldn, ldd = A.ldd;
  const FT zero = f_from(FT{}, 0.0);
  const FT delta = f_from(FT{}, A.delta), eta = f_from(FT{}, A.eta);
  for (int L = blockIdx.x; L < A.batch; L += gridDim.x)
  {
    if (A.only_failed && A.only_failed[L] == 1)
      continue;
    long long *b = A.b + (size_t)L * dT * ldn;
    double *bf   = A.bf + (size_t)L * dT * ldn;
    const size_t po = (size_t)L * dT * ldd, pp = (size_t)A.batch * dT * ldd;
    const PlaneX<FT> mu{A.mu_hi + po, A.mu_lo ? A.mu_lo + po : nullptr, A.mu_x ? A.mu_x + po : nullptr, A.mu_x ? A.mu_x + pp + po : nullptr};
    const PlaneX<FT> r{A.r_hi + po, A.r_lo ? A.r_lo + po : nullptr, A.r_x ? A.r_x + po : nullptr, A.r_x ? A.r_x + pp + po : nullptr};
    const PlaneX<FT> gf{A.gf_hi + po, A.gf_lo ? A.gf_lo + po : nullptr, A.gf_x ? A.gf_x + po : nullptr, A.gf_x ? A.gf_x + pp + po : nullptr};
    long long *rexp = A.rexp + (size_t)L * dT;
    PlaneX<FT> lov, rd;  // (LDS: assigned at run time — an aggregate of shared addresses is no constant)
    lov.hi = lov_hi;
    lov.lo = lov_lo;
    rd.hi  = rd_hi;
    rd.lo  = rd_lo;
    lov.x2 = lov_x2;
    lov.x3 = lov_x3;
    rd.x2  = rd_x2;
    rd.x3  = rd_x3;

    // ---- state: identity slots, nothing valid, Gram cache empty ---------------------------------
    for (int p = lane; p < dT; p += 64)
    {
      slot[p] = p;
      vcol[p] = 0;
    }
    for (int i = 0; i < dT; ++i)
      for (int j = lane; j < dT; j += 64)
        gf.hi[(size_t)i * ldd + j] = __longlong_as_double(0x7ff8000000000000ll);
    __syncthreads();

    // MatGSO::update_bf(i) for the row in slot s, gso.cpp:24-48
    auto update_bf = [&](int s)
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
        if (c < n)
        {
          const long long v = b[(size_t)s * ldn + c];
          if (A.row_expo)
          {
            int ex;
            cm[q] = frexp((double)v, &ex);
            ce[q] = ex;
            emax  = max(emax, ex);
          }
          else
            cm[q] = (double)v;
        }
      }
      if (A.row_expo)
      {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
          emax = max(emax, __shfl_xor(emax, off));
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int c = lane + 64 * q;
        if (c < n)
          bf[(size_t)s * ldn + c] = A.row_expo ? ldexp(cm[q], ce[q] - emax) : cm[q];
      }
      if (lane == 0)
        rexp[s] = A.row_expo ? (long long)emax : 0;
      __threadfence_block();
    };
    // row_op_end(i, i + 1) for position p, gso_interface.cpp:32-53
    auto row_op_end = [&](int p)
    {
      const int s = slot[p];
      update_bf(s);
      const double qnan = __longlong_as_double(0x7ff8000000000000ll);
      for (int t = lane; t < dT; t += 64)  // invalidate_gram_row: every pair with this vector
      {
        const int a = max(s, t), c = min(s, t);
        gf.hi[(size_t)a * ldd + c] = qnan;
      }
      for (int q = lane; q < dT; q += 64)
      {
        if (q == p)
          vcol[q] = 0;
        else if (q > p && vcol[q] > p)
          vcol[q] = p;
      }
      __syncthreads();
    };
    // MatGSO::move_row(old_r, new_r), gso.cpp:289-366, on the slot table
    auto move_row = [&](int old_r, int new_r)
    {
      if (old_r == new_r)
        return;
      const int lo = min(old_r, new_r), hi = max(old_r, new_r);
      int ns[NQ], nv[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int p = lane + 64 * q;
        int src     = p;
        if (p >= lo && p <= hi)
          src = (p == new_r) ? old_r : (new_r < old_r ? p - 1 : p + 1);
        ns[q] = (p < dT) ? slot[src < dT ? src : p] : 0;
        nv[q] = (p < dT) ? vcol[src < dT ? src : p] : 0;
        if (p >= lo)
          nv[q] = min(nv[q], lo);  // invalidate_gso_row(i, lo) for every row from lo on
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int p = lane + 64 * q;
        if (p < dT)
        {
          slot[p] = ns[q];
          vcol[p] = nv[q];
        }
      }
      __syncthreads();
    };

    // update_gso_row(kp, last_j), gso_interface.cpp:131-164, for the row at position kp: lane j owns
    // column j.  Gram entries first (lane j: the dot product of rows kp and j over the columns,
    // ascending), then column-oriented forward substitution: step k broadcasts the final r(kp,k), every
    // lane j > k subtracts mu(j,k) r(kp,k) — lane j sees k = 0, 1, 2, … like the reference's inner loop.
    // Returns false on a non-finite mu.
    auto update_row = [&](int kp, int last_j) -> bool
    {
      const int sk = slot[kp];
      int vfrom    = vcol[kp];
      if (vfrom > last_j)
        return true;
      FT acc[NQ];
      // Gram row: cached entries where known
      bool need[NQ];
      bool any_need = false;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int j = lane + 64 * q;
        need[q]     = false;
        acc[q]      = zero;
        if (j <= last_j && j >= vfrom)
        {
          const int sj = slot[j];
          const int a = max(sk, sj), c = min(sk, sj);
          const FT g  = gf.ld((size_t)a * ldd + c);
          if (f_hi(g) != f_hi(g))
            need[q] = true;
          else
            acc[q] = g;
          any_need |= need[q];
        }
      }
      if (__any(any_need))
      {
        FT dot[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          dot[q] = zero;
        for (int c = 0; c < n; ++c)
        {
          const double xk = bf[(size_t)sk * ldn + c];  // (wave-uniform address: a broadcast load)
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            if (need[q])
            {
              const int j     = lane + 64 * q;
              const double xj = bf[(size_t)slot[j] * ldn + c];
              dot[q]          = f_add(dot[q], f_mul(f_from(FT{}, xk), f_from(FT{}, xj)));
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          if (need[q])
          {
            const int j  = lane + 64 * q;
            const int sj = slot[j];
            const int a = max(sk, sj), c = min(sk, sj);
            gf.st((size_t)a * ldd + c, dot[q]);
            acc[q] = dot[q];
          }
      }
      // forward substitution over k = 0 .. last_j - 1
      const bool diag = last_j == kp;  // (the diagonal entry needs mu(kp, k) of THIS row: r(kp,k) / r(k,k))
      for (int k = 0; k < last_j; ++k)
      {
        // r(kp, k): final once every k' < k has been subtracted — below vfrom it is the stored value
        FT rk = zero;
        if (k < vfrom)
          rk = r.ld((size_t)sk * ldd + k);
        else
        {
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            if ((k >> 6) == q)
              rk = f_bcast(acc[q], k & 63);
        }
        FT mself = zero;
        if (diag)
          mself = f_div(rk, rd.ld(k));
#pragma unroll
        for (int q = 0; q < NQ; ++q)
        {
          const int j = lane + 64 * q;
          if (j > k && j <= last_j && j >= vfrom)
          {
            const FT m = (j == kp) ? mself : mu.ld((size_t)slot[j] * ldd + k);  // mu(j, k)
            acc[q]     = f_sub(acc[q], f_mul(m, rk));
          }
        }
      }
      // store r(kp, j), mu(kp, j) = r(kp, j) / r(j, j) for j < kp
      bool bad = false;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int j = lane + 64 * q;
        if (j <= last_j && j >= vfrom)
        {
          r.st((size_t)sk * ldd + j, acc[q]);
          if (j < kp)
          {
            const FT rjj = rd.ld(j);
            const FT m   = f_div(acc[q], rjj);
            mu.st((size_t)sk * ldd + j, m);
            bad |= !f_finite(m);
          }
          else if (j == kp)
            rd.st(kp, acc[q]);
        }
      }
      __threadfence_block();
      __syncthreads();
      if (__any(bad))
        return false;
      if (lane == 0)
        vcol[kp] = last_j + 1;
      __syncthreads();
      return true;
    };

    // LLLReduction::babai(kappa, kappa, 0), lll.cpp:166-224
    auto babai = [&](int kp) -> int
    {
      long long max_expo = LLONG_MAX;
      for (int iter = 0;; ++iter)
      {
        if (!update_row(kp, kp - 1))
          return 0;
        const int sk = slot[kp];
        const long long ek = rexp[sk];
        FT bm[NQ];
        int be[NQ];
        bool over = false;
        long long mx = LLONG_MIN;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
        {
          const int j = lane + 64 * q;
          bm[q]       = zero;
          be[q]       = 0;
          if (j < kp)
          {
            bm[q] = mu.ld((size_t)sk * ldd + j);
            be[q] = (int)(ek - rexp[slot[j]]);
            const FT f = f_abs(f_ldexp(bm[q], be[q]));  // get_mu
            over |= f_gt(f, eta);
            mx = max(mx, (long long)be[q] + f_exponent(bm[q]));
          }
        }
        if (!__any(over))
          break;
        if (iter >= 2)
        {  // get_max_mu_exp, gso_interface.cpp:88-98; SIZE_RED_FAILURE_THRESH = 5
#pragma unroll
          for (int off = 32; off > 0; off >>= 1)
            mx = max(mx, __shfl_xor(mx, off));
          if (mx > max_expo - 5)
            return -1;
          max_expo = mx;
        }
        // the multipliers, j = kp - 1 … 0, and the integer row
        long long bk[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
        {
          const int c = lane + 64 * q;
          bk[q]       = (c < n) ? b[(size_t)sk * ldn + c] : 0;
        }
        int fail = 0;
        for (int j = kp - 1; j >= 0; --j)
        {
          FT bj = zero;
          int ej = 0;
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            if ((j >> 6) == q)
            {
              bj = f_bcast(bm[q], j & 63);
              ej = __shfl(be[q], j & 63);
            }
          const FT X = f_rnd_we(bj, ej);
          if (f_is_zero(X))
            continue;
          const int sj = slot[j];
#pragma unroll
          for (int q = 0; q < NQ; ++q)
          {
            const int k = lane + 64 * q;
            if (k < j)
              bm[q] = f_sub(bm[q], f_mul(X, mu.ld((size_t)sj * ldd + k)));
          }
          // row_addmul_we(kappa, j, -X, expo_add = ej), gso.cpp:236-262 with a long multiplier
          const FT nx = f_neg(X);
          long long expo = f_exponent(nx) + ej - 63;
          if (expo < 0)
            expo = 0;
          if (expo != 0)
          {
            fail = -2;
            break;
          }
          const long long lx = f_to_long(nx, ej);
#pragma unroll
          for (int q = 0; q < NQ; ++q)
          {
            const int c = lane + 64 * q;
            if (c < n)
              bk[q] = (long long)((unsigned long long)bk[q] +
                                  (unsigned long long)b[(size_t)sj * ldn + c] * (unsigned long long)lx);
          }
        }
        if (fail)
          return fail;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
        {
          const int c = lane + 64 * q;
          if (c < n)
            b[(size_t)sk * ldn + c] = bk[q];
        }
        __threadfence_block();
        row_op_end(kp);
      }
      return 1;
    };

    // ---- LLLReduction::lll(kappa_min, kappa_start, kappa_end), lll.cpp:44-164 -------------------
    const int kmin = A.kmin, kend = A.kend < 0 ? dT : A.kend;
    int kappa  = A.kstart + 1;
    const int dd = kend - kmin;
    int zeros = 0, n_swaps = 0, final_kappa = 0, status = 1;
    long long iter = 0;
    for (int p = 0; p < dT; ++p)
      update_bf(p);  // every row is floated once (the reference discovers them lazily: same values)
    __syncthreads();
    for (; zeros < dd; ++zeros)
    {  // b_row_is_zero(0)
      bool nz = false;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int c = lane + 64 * q;
        if (c < n)
          nz |= b[(size_t)slot[0] * ldn + c] != 0;
      }
      if (__any(nz))
        break;
      move_row(kmin, kend - 1 - zeros);
    }
    bool run = true;
    if (zeros < dd)
    {
      int rc = 1;
      // (the reference's caller holds the rows below kappa_start valid; this kernel starts from the
      // integers: compute them — the values are functions of the basis)
      for (int p = 0; p < A.kstart && rc == 1; ++p)
        if (!update_row(p, p))
          rc = 0;
      if (rc == 1 && A.kstart > 0)
        rc = babai(A.kstart);
      if (rc == 1 && !update_row(A.kstart, A.kstart))
        rc = 0;
      if (rc != 1)
      {
        status      = rc;
        final_kappa = A.kstart;
        run         = false;
      }
    }
    if (run)
    {
      // get_max_exp_of_b, nr/matrix.cpp:127-134 (Z_NR<long>::exponent, nr_Z_l.inl:30-48)
      int mexp = 0;
      for (int i = 0; i < dT; ++i)
      {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
        {
          const int c = lane + 64 * q;
          if (c < n)
          {
            const long long v = b[(size_t)i * ldn + c];
            int e;
            const double f = frexp((double)v, &e);
            if ((double)v > 0x1p53 && fabs(f) == 0.5)
            {
              unsigned long long y = (unsigned long long)(v < 0 ? -v : v);
              e                    = 64 - __clzll((long long)y);
            }
            mexp = max(mexp, e);
          }
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1)
        mexp = max(mexp, __shfl_xor(mexp, off));
      const long long max_iter =
          (long long)((double)dd - 2.0 * dd * (dd + 1) * ((double)(((long)mexp + 3)) / log(A.delta)));
      for (iter = 0; iter < max_iter && kappa < kend - zeros; ++iter)
      {
        const int rc = babai(kappa);
        if (rc != 1)
        {
          status      = rc;
          final_kappa = kappa;
          run         = false;
          break;
        }
        const int sk = slot[kappa];
        // lovasz_tests[i] = g(kappa,kappa) - sum_{t < i} mu(kappa,t) r(kappa,t), lll.cpp:110-115
        {
          const int a0 = sk;
          FT g00       = gf.ld((size_t)a0 * ldd + a0);
          if (f_hi(g00) != f_hi(g00))
          {  // get_gram(kappa, kappa)
            FT part = zero;
#pragma unroll
            for (int q = 0; q < NQ; ++q)
            {
              const int c = lane + 64 * q;
              if (c < n)
              {
                const FT x = f_from(FT{}, bf[(size_t)sk * ldn + c]);
                part       = f_add(part, f_mul(x, x));
              }
            }
            g00 = f_wave_sum(part);
            if (lane == 0)
              gf.st((size_t)a0 * ldd + a0, g00);
          }
          FT pr[NQ];
#pragma unroll
          for (int q = 0; q < NQ; ++q)
          {
            const int t = lane + 64 * q;
            pr[q]       = (t < kappa) ? f_mul(mu.ld((size_t)sk * ldd + t), r.ld((size_t)sk * ldd + t)) : zero;
          }
          // prefix differences: an inclusive scan over the lanes, then over the registers
          FT run_acc = g00;
#pragma unroll
          for (int q = 0; q < NQ; ++q)
          {
            FT v = pr[q];  // inclusive prefix sum of pr over lanes
#pragma unroll
            for (int off = 1; off < 64; off <<= 1)
            {
              const FT o = f_shfl_up(v, off);
              if (lane >= off)
                v = f_add(v, o);
            }
            const int t = lane + 64 * q;
            // lovasz[t + 1] = run_acc - prefix(t)
            if (t < kappa)
              lov.st(t + 1, f_sub(run_acc, v));
            run_acc = f_sub(run_acc, f_bcast(v, 63));
          }
          if (lane == 0)
            lov.st(0, g00);
          __syncthreads();
        }
        // Lovasz condition, lll.cpp:116-122
        auto rdiag_scaled = [&](int p, int ref_row) -> FT
        {
          FT f = f_mul(rd.ld(p), delta);
          if (A.row_expo)
            f = f_ldexp(f, (int)(2 * (rexp[slot[p]] - rexp[slot[ref_row]])));
          return f;
        };
        FT f = rdiag_scaled(kappa - 1, kappa);
        if (f_gt(f, lov.ld(kappa - 1)))
        {
          ++n_swaps;
          const int old_k = kappa;
          for (--kappa; kappa > kmin; --kappa)
          {
            f = rdiag_scaled(kappa - 1, old_k);
            if (f_gt(lov.ld(kappa - 1), f))
              break;
          }
          const FT lk = lov.ld(kappa);
          if (f_hi(lk) > 0.0)
            move_row(old_k, kappa);
          else
          {
            ++zeros;
            move_row(old_k, kend - zeros);
            kappa = old_k;
            continue;
          }
        }
        // set_r(kappa, kappa, lovasz_tests[kappa]), gso_interface.h:742-749
        {
          const FT lk  = lov.ld(kappa);
          const int s2 = slot[kappa];
          if (lane == 0)
          {
            r.st((size_t)s2 * ldd + kappa, lk);
            rd.st(kappa, lk);
            if (vcol[kappa] == kappa)
              vcol[kappa] = kappa + 1;
          }
          __threadfence_block();
          __syncthreads();
        }
        ++kappa;
      }
      if (run)
        status = (kappa < kend - zeros) ? -3 : 1;
    }
    // rows in position order
    long long *out = A.b2 + (size_t)L * dT * ldn;
    for (int p = 0; p < dT; ++p)
    {
      const int s = slot[p];
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int c = lane + 64 * q;
        if (c < ldn)
          out[(size_t)p * ldn + c] = (c < n) ? b[(size_t)s * ldn + c] : 0;
      }
    }
    if (lane == 0)
    {
      A.status[L]       = status;
      A.info[4 * L + 0] = final_kappa;
      A.info[4 * L + 1] = n_swaps;
      A.info[4 * L + 2] = zeros;
      A.info[4 * L + 3] = (int)(iter & 0x7fffffff);
    }
    __syncthreads();
  }
}

template __global__ void lll_x_kernel<1, double>(LllX);
template __global__ void lll_x_kernel<2, double>(LllX);
template __global__ void lll_x_kernel<3, double>(LllX);
template __global__ void lll_x_kernel<4, double>(LllX);
template __global__ void lll_x_kernel<1, DD>(LllX);
template __global__ void lll_x_kernel<2, DD>(LllX);
template __global__ void lll_x_kernel<3, DD>(LllX);
template __global__ void lll_x_kernel<4, DD>(LllX);
template __global__ void lll_x_kernel<1, QD>(LllX);
template __global__ void lll_x_kernel<2, QD>(LllX);
template __global__ void lll_x_kernel<3, QD>(LllX);
template __global__ void lll_x_kernel<4, QD>(LllX);

}  // namespace fphip
