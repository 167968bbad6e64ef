// hlll_x.hip — HLLL over MatHouseholder in a SELECTABLE floating-point type (ftx.h): double-double
// (the device stand-in for FP_NR<dd_real>, BASELINE config 5 as stated) or plain double.
//
// What it runs is the reference's algorithm, statement for statement:
//   HLLLReduction::hlll              fplll/hlll.cpp:26-169   (k loop, swap, norm anomaly check)
//   HLLLReduction::lovasz_test       hlll.cpp:171-224
//   HLLLReduction::size_reduction    hlll.cpp:262-351        (approx = 0.1, two-strike stop rule)
//   HLLLReduction::verify_size_reduction  hlll.cpp:455-496
//   compute_dR / compute_eR          hlll.h:148-159          (eR uses delta — sic)
//   MatHouseholder::update_R(i,false) householder.cpp:151-184, update_R_last :27-146,
//   refresh_R_bf :186-245, refresh_R :247-261, swap :372-398, size_reduce :402-451,
//   row_addmul_we :522-559
// but NOT the reference's summation order: dot products and norms are wave-level tree sums.  That is
// this kernel's contract — the exact-order double kernel is hlll_kernel.hip; a double-double result
// cannot be pinned bit for bit anyway (libqd is absent, ftx.h), and with 106 bits the decisions of
// the algorithm have 50 bits of slack: on BASELINE config 5's lattice the reference returns the same
// basis in double, long double and 106-bit MPFR, and so does this kernel.
//
// One wavefront per lattice, lane = column (NQ per lane); the working row R[k] stays in registers;
// size_reduce evaluates every candidate multiplier of the row at once (lane i: R(k,i)/R(i,i)) and
// applies the highest non-zero one, so its cost follows the number of row operations, not k.
#include "ftx.h"
#include "gso_device.h"

namespace fphip
{

// a [rows][ld] array of FT values as one plane of doubles per component (hi, lo, and two more for quad-double)
template <class FT> struct Plane;
template <> struct Plane<double>
{
  double *hi, *lo, *x2, *x3;
  __device__ __forceinline__ double ld(size_t i) const { return hi[i]; }
  __device__ __forceinline__ void st(size_t i, double v) const { hi[i] = v; }
};
template <> struct Plane<DD>
{
  double *hi, *lo, *x2, *x3;
  __device__ __forceinline__ DD ld(size_t i) const { return DD{hi[i], lo[i]}; }
  __device__ __forceinline__ void st(size_t i, DD v) const
  {
    hi[i] = v.hi;
    lo[i] = v.lo;
  }
};
template <> struct Plane<QD>
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

struct HlllX
{
  // R, V: [batch][d][ldn] in up to four planes each (P.R / P.V, Rlo / Vlo, and for quad-double Rx / Vx: two more
  // planes back to back); bf has no low part
  double *Rlo, *Vlo;
  // per lattice scalars, [batch][d] each: norm_square_b, dR, eR, prev_R, R(i,i) (four component planes each)
  double *sc;  // [batch][20][d]
  long long *prevE;
  double delta, theta;
  long long iter_cap;
  const int *only_failed;  // precision ladder: non-null = reduce only lattices whose entry is not 1
  // blocked application of the reflectors (compact WY, householder.cpp:151-184 sixteen reflectors at a time):
  // T of every block of 16 reflectors, [batch][ceil(d / 16)][16][16] (hi / lo planes); null = one by one
  double *Thi, *Tlo;
  double *Rx, *Vx;  // quad-double: components 2 and 3 of R and V ([2][batch][d][ldn] each)
};

// status: 1 RED_SUCCESS, -2 multiplier beyond 63 bits, -4 RED_HLLL_SR_FAILURE,
//         -5 RED_HLLL_NORM_FAILURE, -6 iteration cap.  info[2] per lattice: swaps, loop iterations
template <int NQ, class FT> __global__ void __launch_bounds__(64) hlll_x_kernel(HhBatch P, HlllX X)
{
  const int lane = threadIdx.x & 63;
  const int d = P.d, n = P.n, ld = P.ldn;
  const FT zero = f_from(FT{}, 0.0);
  for (int L = blockIdx.x; L < P.batch; L += gridDim.x)
  {
    if (X.only_failed && X.only_failed[L] == 1)
      continue;  // an earlier, cheaper stage of the ladder has reduced this lattice
    long long *b    = P.b + (size_t)L * d * ld;
    double *bf      = P.bf + (size_t)L * d * ld;
    double *sigma   = P.sigma + (size_t)L * d;
    long long *rexp = P.rexp + (size_t)L * d;
    const size_t pl = (size_t)P.batch * d * ld, lo0 = (size_t)L * d * ld;
    const Plane<FT> R{P.R + lo0, X.Rlo ? X.Rlo + lo0 : nullptr, X.Rx ? X.Rx + lo0 : nullptr, X.Rx ? X.Rx + pl + lo0 : nullptr};
    const Plane<FT> V{P.V + lo0, X.Vlo ? X.Vlo + lo0 : nullptr, X.Vx ? X.Vx + lo0 : nullptr, X.Vx ? X.Vx + pl + lo0 : nullptr};
    double *sc = X.sc + (size_t)L * 20 * d;
    auto scal = [&](int q) { return Plane<FT>{sc + (4 * q) * d, sc + (4 * q + 1) * d, sc + (4 * q + 2) * d, sc + (4 * q + 3) * d}; };
    const Plane<FT> nsb = scal(0), dR = scal(1), eR = scal(2), prevR = scal(3), rdg = scal(4);
    long long *prevE = X.prevE + (size_t)L * d;
    const int nblk   = (d + 15) >> 4;
    const bool blocked = X.Thi != nullptr;
    const Plane<FT> Tm{blocked ? X.Thi + (size_t)L * nblk * 256 : nullptr,
                       (blocked && X.Tlo) ? X.Tlo + (size_t)L * nblk * 256 : nullptr, nullptr, nullptr};
    const FT delta   = f_from(FT{}, X.delta), theta = f_from(FT{}, X.theta);
    FT Rk[NQ];  // the working row R[k], lane = column

    // refresh_R_bf(i) (householder.cpp:186-245): float the integer row, R[i] = bf[i] in Rk, norm
    auto refresh_R_bf = [&](int i)
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
          const long long v = b[(size_t)i * ld + c];
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
#pragma unroll
      for (int off = 32; off > 0; off >>= 1)
        emax = max(emax, __shfl_xor(emax, off));
      FT part = zero;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int c    = lane + 64 * q;
        const double f = (c < n) ? (P.row_expo ? ldexp(cm[q], ce[q] - emax) : cm[q]) : 0.0;
        if (c < n)
          bf[(size_t)i * ld + c] = f;
        Rk[q] = f_from(FT{}, f);
        part  = f_add(part, f_mul(Rk[q], Rk[q]));
      }
      const FT ns = f_wave_sum(part);
      if (lane == 0)
      {
        rexp[i] = P.row_expo ? (long long)emax : 0;
        nsb.st(i, ns);
      }
    };
    auto refresh_R = [&](int i)
    {
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int c = lane + 64 * q;
        Rk[q]       = f_from(FT{}, (c < n) ? bf[(size_t)i * ld + c] : 0.0);
      }
    };
    // update_R(i, false) (householder.cpp:151-184) on Rk
    auto apply_reflectors = [&](int i)
    {
      FT vn[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int c = lane + 64 * q;
        vn[q]       = (i > 0 && c < n) ? V.ld(c) : zero;  // row 0
      }
      for (int j = 0; j < i; ++j)
      {
        FT v[NQ];
        FT part = zero;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
        {
          const int c = lane + 64 * q;
          v[q]        = vn[q];
          if (c >= j && c < n)
            part = f_add(part, f_mul(v[q], Rk[q]));
        }
        if (j + 1 < i)
        {  // the next reflector is on its way while this one is applied
#pragma unroll
          for (int q = 0; q < NQ; ++q)
          {
            const int c = lane + 64 * q;
            vn[q]       = (c < n) ? V.ld((size_t)(j + 1) * ld + c) : zero;
          }
        }
        const FT t     = f_neg(f_wave_sum(part));
        const double s = sigma[j];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
        {
          const int c = lane + 64 * q;
          if (c >= j && c < n)
          {
            FT u = f_add(Rk[q], f_mul(v[q], t));
            if (c == j)
              u = f_mul_d(u, s);
            Rk[q] = u;
          }
        }
      }
    };
    // The totals of 16 per-lane values over the wave in ONE butterfly (17 additions instead of 16 x 6): after the
    // steps over lane bits 5..2 every lane holds one index, the steps over bits 1..0 finish it: lane l ends with
    // the total of index (l >> 2) & 15.
    auto reduce16 = [&](FT (&pp)[16]) -> FT
    {
      FT q8[8], q4[4], q2[2];
      const bool b5 = (lane & 32) != 0, b4 = (lane & 16) != 0, b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        q8[u] = f_add(b5 ? pp[u + 8] : pp[u], f_shfl_xor(b5 ? pp[u] : pp[u + 8], 32));
#pragma unroll
      for (int u = 0; u < 4; ++u)
        q4[u] = f_add(b4 ? q8[u + 4] : q8[u], f_shfl_xor(b4 ? q8[u] : q8[u + 4], 16));
#pragma unroll
      for (int u = 0; u < 2; ++u)
        q2[u] = f_add(b3 ? q4[u + 2] : q4[u], f_shfl_xor(b3 ? q4[u] : q4[u + 2], 8));
      FT q1 = f_add(b2 ? q2[1] : q2[0], f_shfl_xor(b2 ? q2[0] : q2[1], 4));
      q1    = f_add(q1, f_shfl_xor(q1, 2));
      q1    = f_add(q1, f_shfl_xor(q1, 1));
      return q1;
    };
    // update_R(i, false) in BLOCKED form: the reflectors sixteen at a time, H_j0 ... H_j0+m-1 = I - V^T T V with T
    // upper triangular (every H_j = I - v_j v_j^T, householder.cpp:168-175), so a row takes
    //     x <- x - ((x V^T) T) V :  m independent dot products (one butterfly), a 16 x 16 triangle, m AXPYs
    // instead of m dot products that each wait for the AXPY before them; then the signs of the block's columns
    // (R(i,j) = sigma_j R(i,j), :176 — no later reflector touches column j).  Other roundings than one by one:
    // this kernel's contract already (tree sums).
    auto apply_reflectors_blocked = [&](int i)
    {
      for (int j0 = 0; j0 < i; j0 += 16)
      {
        const int m = min(16, i - j0);
        const int K = j0 >> 4;
        FT pp[16];
#pragma unroll
        for (int a = 0; a < 16; ++a)
        {
          pp[a] = zero;
          if (a < m)
          {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
            {
              const int c = lane + 64 * q;
              if (c < n)
                pp[a] = f_add(pp[a], f_mul(V.ld((size_t)(j0 + a) * ld + c), Rk[q]));
            }
          }
        }
        const FT wv = reduce16(pp);
        // y_b = sum_{a <= b} T[a][b] w_a in lane b
        FT yb = zero;
#pragma unroll
        for (int a = 0; a < 16; ++a)
        {
          const FT wa = f_bcast(wv, 4 * a);
          if (a < m && lane < m && a <= lane)
            yb = f_add(yb, f_mul(Tm.ld((size_t)K * 256 + a * 16 + lane), wa));
        }
#pragma unroll
        for (int bb = 0; bb < 16; ++bb)
        {
          const FT y = f_bcast(yb, bb);
          if (bb < m)
          {
#pragma unroll
            for (int q = 0; q < NQ; ++q)
            {
              const int c = lane + 64 * q;
              if (c < n)
                Rk[q] = f_sub(Rk[q], f_mul(y, V.ld((size_t)(j0 + bb) * ld + c)));
            }
          }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q)
        {
          const int c = lane + 64 * q;
          if (c >= j0 && c < j0 + m)
            Rk[q] = f_mul_d(Rk[q], sigma[c]);
        }
      }
    };
    // column (i mod 16) of the T of reflector i's block: T[0:m, m] = -T[0:m, 0:m] (V[j0 : j0+m] v_i), T[m][m] = 1
    auto update_T = [&](int i, const FT (&vi)[NQ])
    {
      const int j0 = i & ~15, m = i & 15, K = i >> 4;
      FT pp[16];
#pragma unroll
      for (int a = 0; a < 16; ++a)
      {
        pp[a] = zero;
        if (a < m)
        {
#pragma unroll
          for (int q = 0; q < NQ; ++q)
          {
            const int c = lane + 64 * q;
            if (c < n)
              pp[a] = f_add(pp[a], f_mul(V.ld((size_t)(j0 + a) * ld + c), vi[q]));
          }
        }
      }
      const FT g = reduce16(pp);
      FT tv      = zero;
#pragma unroll
      for (int bb = 0; bb < 16; ++bb)
      {
        const FT gb = f_bcast(g, 4 * bb);
        if (bb < m && lane < m && bb >= lane)
          tv = f_add(tv, f_mul(Tm.ld((size_t)K * 256 + lane * 16 + bb), gb));
      }
      if (lane < m)
        Tm.st((size_t)K * 256 + lane * 16 + m, f_neg(tv));
      if (lane == m)
        Tm.st((size_t)K * 256 + m * 16 + m, f_from(FT{}, 1.0));
    };
    auto row_get = [&](int idx) -> FT
    {  // Rk[idx], wave-uniform idx
      FT r = zero;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if ((idx >> 6) == q)
          r = f_bcast(Rk[q], idx & 63);
      return r;
    };
    // update_R_last(i) (householder.cpp:27-146) on Rk; stores R row i, V row i, sigma, R(i,i)
    auto update_R_last = [&](int i)
    {
      FT part = zero;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int c = lane + 64 * q;
        if (c > i && c < n)
          part = f_add(part, f_mul(Rk[q], Rk[q]));
      }
      FT f3        = f_wave_sum(part);
      const FT rii = row_get(i);
      const double sgi = f_lt0(rii) ? -1.0 : 1.0;
      FT f1        = f_add(f_mul(rii, rii), f3);
      FT vii = zero, new_rii = zero, f0 = f_from(FT{}, 1.0);
      bool scale = false;
      if (!f_is_zero(f1))
      {
        const FT f2 = f_sqrt(f1);
        f0          = f_mul_d(f2, sgi);
        f1          = f_add(rii, f0);
        f3          = f_div(f_neg(f3), f1);
        if (!f_is_zero(f3))
        {
          f0      = f_sqrt(f_mul(f_neg(f0), f3));
          vii     = f_div(f3, f0);
          new_rii = f2;
          scale   = true;
        }
        else
          new_rii = f_abs(rii);
      }
      FT vi[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int c = lane + 64 * q;
        vi[q]       = zero;
        if (c < n)
        {
          FT vv = zero;
          if (c == i)
            vv = vii;
          else if (c > i && scale)
            vv = f_div(Rk[q], f0);
          vi[q] = vv;
          V.st((size_t)i * ld + c, vv);
          if (c == i)
            Rk[q] = new_rii;
          R.st((size_t)i * ld + c, Rk[q]);
        }
      }
      if (blocked)
        update_T(i, vi);
      if (lane == 0)
      {
        sigma[i] = sgi;
        rdg.st(i, new_rii);
        dR.st(i, f_mul(delta, f_mul(new_rii, new_rii)));  // compute_dR, hlll.h:148-153
        eR.st(i, f_mul(delta, new_rii));                   // compute_eR (sic: delta), hlll.h:155-159
      }
      __threadfence_block();
    };
    // MatHouseholder::size_reduce(k, k, 0) (householder.cpp:402-451).  1 reduced, 0 not, -1 overflow
    auto size_reduce = [&](int k) -> int
    {
      int reduced        = 0;
      int limit          = k;  // multipliers of rows >= limit are done
      const long long ek = rexp[k];
      for (;;)
      {
        // every candidate at once: lane i (chunk q) holds x_i = rnd_we(R(k,i) / R(i,i), e_k - e_i)
        FT xs[NQ];
        int ea[NQ];
        int top = -1;
#pragma unroll
        for (int q = NQ - 1; q >= 0; --q)
        {
          const int i = lane + 64 * q;
          xs[q]       = zero;
          ea[q]       = 0;
          bool nzq    = false;
          if (i < limit)
          {
            ea[q] = (int)(ek - rexp[i]);
            xs[q] = f_rnd_we(f_div(Rk[q], rdg.ld(i)), ea[q]);
            nzq   = !f_is_zero(xs[q]);
          }
          const unsigned long long m = __ballot(nzq);
          if (top < 0 && m != 0)
            top = 64 * q + 63 - __builtin_clzll(m);
        }
        if (top < 0)
          break;
        FT x = zero;
        int e = 0;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          if ((top >> 6) == q)
          {
            x = f_bcast(xs[q], top & 63);
            e = __shfl(ea[q], top & 63);
          }
        x = f_neg(x);
        // row_addmul_we(k, top, x, e), householder.cpp:522-559
        long long expo = f_exponent(x) + e - 63;
        if (expo > 0)
          return -1;
        const long long lx = f_to_long(x, e);
#pragma unroll
        for (int q = 0; q < NQ; ++q)
        {
          const int c = lane + 64 * q;
          if (c < n)
          {
            const size_t ik = (size_t)k * ld + c, it = (size_t)top * ld + c;
            b[ik]           = (long long)((unsigned long long)b[ik] + (unsigned long long)b[it] * (unsigned long long)lx);
            if (c < top)
              Rk[q] = f_add(Rk[q], f_mul(R.ld(it), x));
          }
        }
        limit   = top;
        reduced = 1;
      }
      return reduced;
    };

    // ------------------------------------------------------------------------------------------
    int status = 1, n_swaps = 0;
    long long iters = 0;
    refresh_R_bf(0);
    update_R_last(0);
    int k = 1, k_max = 1, prev_k = -1;
    if (d >= 2)
    {
      refresh_R_bf(1);
      for (;;)
      {
        if (++iters > X.iter_cap)
        {
          status = -6;
          break;
        }
        // ---- size_reduction(k, k, 0), hlll.cpp:262-351
        {
          bool not_stop = true, prev_not_stop = true, fail = false;
          if (blocked)
            apply_reflectors_blocked(k);
          else
            apply_reflectors(k);
          for (;;)
          {
            const int red = size_reduce(k);
            if (red < 0)
            {
              fail = true;
              break;
            }
            if (!red)
              break;
            __threadfence_block();
            FT f0                 = nsb.ld(k);
            const long long expo0 = P.row_expo ? 2 * rexp[k] : 0;
            refresh_R_bf(k);
            __threadfence_block();
            const FT f1           = nsb.ld(k);
            const long long expo1 = P.row_expo ? 2 * rexp[k] : 0;
            f0                    = f_ldexp(f_mul_d(f0, 0.1), (int)(expo0 - expo1));
            not_stop              = f_le(f1, f0);
            if (blocked)
              apply_reflectors_blocked(k);
            else
              apply_reflectors(k);
            if (prev_not_stop || not_stop)
              prev_not_stop = not_stop;
            else
              break;
          }
          if (fail)
          {
            status = -2;
            break;
          }
        }
        // ---- verify_size_reduction(k), hlll.cpp:455-496
        {
          FT part = zero;
#pragma unroll
          for (int q = 0; q < NQ; ++q)
          {
            const int c = lane + 64 * q;
            if (c >= k && c < n)
              part = f_add(part, f_mul(Rk[q], Rk[q]));
          }
          FT f1 = (n == k) ? zero : f_sqrt(f_wave_sum(part));
          f1    = f_mul(f1, theta);
          bool bad = false;
          const long long ek = rexp[k];
#pragma unroll
          for (int q = 0; q < NQ; ++q)
          {
            const int i = lane + 64 * q;
            if (i < k)
            {
              const FT f2 = f_add(f1, f_ldexp(eR.ld(i), (int)(rexp[i] - ek)));
              bad |= f_gt(f_abs(Rk[q]), f2);
            }
          }
          if (__any(bad))
          {
            status = -4;
            break;
          }
        }
        // ---- lovasz_test(k), hlll.cpp:171-224
        bool lov;
        {
          FT part = zero;
#pragma unroll
          for (int q = 0; q < NQ; ++q)
          {
            const int c = lane + 64 * q;
            if (c < k - 1)
              part = f_add(part, f_mul(Rk[q], Rk[q]));
          }
          FT f1 = f_sub(nsb.ld(k), f_wave_sum(part));
          const long long expo1 = P.row_expo ? 2 * rexp[k] : 0, expo0 = rexp[k - 1];
          f1  = f_ldexp(f1, (int)(expo1 - 2 * expo0));
          lov = f_le(dR.ld(k - 1), f1);
        }
        if (lov)
        {
          update_R_last(k);
          if (prev_k == k + 1)
          {
            const FT f0 = rdg.ld(k);
            const FT f1 = f_ldexp(prevR.ld(k), (int)(prevE[k] - rexp[k]));
            if (f_gt(f0, f1))
            {
              status = -5;
              break;
            }
          }
          prev_k = k;
          if (lane == 0)
          {
            prevR.st(k, rdg.ld(k));
            prevE[k] = rexp[k];
          }
          __threadfence_block();
          k++;
          if (k < d)
          {
            if (k > k_max)
            {
              k_max = k;
              refresh_R_bf(k);
            }
            else
              refresh_R(k);
          }
          else
            break;  // RED_SUCCESS
        }
        else
        {
          // swap(k-1, k), householder.cpp:372-398
          ++n_swaps;
#pragma unroll
          for (int q = 0; q < NQ; ++q)
          {
            const int c = lane + 64 * q;
            if (c < n)
            {
              const size_t i0 = (size_t)(k - 1) * ld + c, i1 = (size_t)k * ld + c;
              const long long tb = b[i0];
              b[i0]              = b[i1];
              b[i1]              = tb;
              const double tf    = bf[i0];
              bf[i0]             = bf[i1];
              bf[i1]             = tf;
            }
          }
          if (lane == 0)
          {
            const double ts = sigma[k - 1];
            sigma[k - 1]    = sigma[k];
            sigma[k]        = ts;
            const long long te = rexp[k - 1];
            rexp[k - 1]        = rexp[k];
            rexp[k]            = te;
            const FT tn = nsb.ld(k - 1);
            nsb.st(k - 1, nsb.ld(k));
            nsb.st(k, tn);
          }
          __threadfence_block();
          prev_k = k;
          if (k - 1 == 0)
          {
            refresh_R(0);
            update_R_last(0);
            refresh_R(1);
            k = 1;
          }
          else
          {
            k--;
            refresh_R(k);  // recover_R(k) == refresh_R(k) + the update_R of the next size_reduction
          }
        }
      }
    }
    if (lane == 0)
    {
      P.status[L]       = status;
      P.info[2 * L + 0] = n_swaps;
      P.info[2 * L + 1] = (int)(iters & 0x7fffffff);
    }
    __threadfence_block();
  }
}

template __global__ void hlll_x_kernel<1, double>(HhBatch, HlllX);
template __global__ void hlll_x_kernel<2, double>(HhBatch, HlllX);
template __global__ void hlll_x_kernel<3, double>(HhBatch, HlllX);
template __global__ void hlll_x_kernel<4, double>(HhBatch, HlllX);
template __global__ void hlll_x_kernel<1, DD>(HhBatch, HlllX);
template __global__ void hlll_x_kernel<2, DD>(HhBatch, HlllX);
template __global__ void hlll_x_kernel<3, DD>(HhBatch, HlllX);
template __global__ void hlll_x_kernel<4, DD>(HhBatch, HlllX);
template __global__ void hlll_x_kernel<1, QD>(HhBatch, HlllX);
template __global__ void hlll_x_kernel<2, QD>(HhBatch, HlllX);
template __global__ void hlll_x_kernel<3, QD>(HhBatch, HlllX);
template __global__ void hlll_x_kernel<4, QD>(HhBatch, HlllX);

// ---------------------------------------------------------------------------------------------
// Unit-test kernel for ftx.h: out[i] = a[i] (op) b[i] in double-double, one element per thread.
// op: 0 add, 1 sub, 2 mul, 3 div, 4 sqrt(a), 5 nint(a)
// ---------------------------------------------------------------------------------------------
__global__ void dd_op_kernel(const double *ahi, const double *alo, const double *bhi, const double *blo,
                             double *ohi, double *olo, int op, int count)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count)
    return;
  const DD a{ahi[i], alo[i]}, b{bhi[i], blo[i]};
  DD r{0.0, 0.0};
  switch (op)
  {
  case 0: r = f_add(a, b); break;
  case 1: r = f_sub(a, b); break;
  case 2: r = f_mul(a, b); break;
  case 3: r = f_div(a, b); break;
  case 4: r = f_sqrt(a); break;
  default: r = f_nint(a); break;
  }
  ohi[i] = r.hi;
  olo[i] = r.lo;
}

}  // namespace fphip
