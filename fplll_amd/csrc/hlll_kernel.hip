// hlll_kernel.hip — batched HLLL reduction for gfx950:
// HLLLReduction<Z_NR<long>, FP_NR<double>>::hlll() over MatHouseholder with HOUSEHOLDER_ROW_EXPO
// (the LM_FAST configuration of hlll_reduction_zf, fplll/wrapper.cpp:790-806), one wavefront per
// lattice, lane = COLUMN, bit-exact decisions.
//
// Reference behaviour reproduced:
//   HLLLReduction::hlll              fplll/hlll.cpp:26-169   (k loop, swap, norm anomaly check)
//   HLLLReduction::lovasz_test       hlll.cpp:171-224        (MSV'09 test, not the MODIFIED one)
//   HLLLReduction::size_reduction    hlll.cpp:262-351        (approx = 0.1, two-strike stop rule)
//   HLLLReduction::verify_size_reduction  hlll.cpp:455-496   (default build: eta/theta test)
//   compute_dR / compute_eR          hlll.h:148-159          (eR uses delta — sic)
//   MatHouseholder::update_R(i,false) householder.cpp:151-184, update_R_last :27-146,
//   refresh_R_bf :186-245, refresh_R :247-261, swap :372-398, size_reduce :402-451,
//   row_addmul_we :522-559
// recover_R (householder.h:597-608) restores from R_history exactly what refresh_R(i) +
// update_R(i,false) recompute (same operands, same operation order); the kernel recomputes and
// keeps no d x d x n history (13.8 MB per 120-dimensional lattice).
//
// Layout per lattice (HhBatch): b, bf, R, V are [d][ldn] row-major (lane = column: every row access
// is one coalesced read), per-row scalars (row_expo, ||b_i||^2, R(i,i), sigma, dR, eR, prev_R) live
// in registers, lane i = row i.  The working row R[k] stays in registers for a whole k-iteration;
// dot products are the reference's sequential sums (v_readlane chains, seq_sum), AXPYs are plain
// vector operations; reflectors V_j and the rows R_i / b_i of the size reduction are streamed
// through the LDS-DMA ring.

#include "gso_wave.h"

namespace fphip
{

template <int NQ> __device__ __forceinline__ double hl_get(const double (&v)[NQ], int idx)
{
  double r = 0.0;
  dispatch_chunk<NQ>(idx, [&](auto q, int ii) { r = g_rl_f64(v[decltype(q)::value], ii); });
  return r;
}
template <int NQ> __device__ __forceinline__ int hl_geti(const int (&v)[NQ], int idx)
{
  int r = 0;
  dispatch_chunk<NQ>(idx,
                     [&](auto q, int ii) { r = __builtin_amdgcn_readlane(v[decltype(q)::value], ii); });
  return r;
}
template <int NQ>
__device__ __forceinline__ void hl_set(double (&v)[NQ], int idx, double x, int lane)
{
#pragma unroll
  for (int q = 0; q < NQ; ++q)
    v[q] = (lane + 64 * q == idx) ? x : v[q];
}
template <int NQ> __device__ __forceinline__ void hl_seti(int (&v)[NQ], int idx, int x, int lane)
{
#pragma unroll
  for (int q = 0; q < NQ; ++q)
    v[q] = (lane + 64 * q == idx) ? x : v[q];
}

// status: 1 RED_SUCCESS, -2 multiplier beyond 63 bits, -4 RED_HLLL_SR_FAILURE,
//         -5 RED_HLLL_NORM_FAILURE, -6 iteration cap (safety net, not a reference status)
// info[2] per lattice: swaps, loop iterations
template <int NQ>
__global__ void __launch_bounds__(256)
    hlll_kernel(HhBatch P, double delta, double theta, long long iter_cap)
{
  constexpr int IPS = (NQ + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) char hlll_smem[];
  const int lane = threadIdx.x & 63;
  const int wpb  = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  Ring<NQ, IPS, FPHIP_RING_REDUCE> ring;
  ring.base = (unsigned)(wave * Ring<NQ, IPS, FPHIP_RING_REDUCE>::R * Ring<NQ, IPS, FPHIP_RING_REDUCE>::SLOT);
  ring.lane = lane;
  ring.head = ring.tail = 0;
  ring.ahead            = 0;
  const int d = P.d, n = P.n, ld = P.ldn;
  for (int L = blockIdx.x * wpb + wave; L < P.batch; L += gridDim.x * wpb)
  {
    long long *b    = P.b + (size_t)L * d * ld;
    double *bf      = P.bf + (size_t)L * d * ld;
    double *V       = P.V + (size_t)L * d * ld;
    double *R       = P.R + (size_t)L * d * ld;
    double *sigma   = P.sigma + (size_t)L * d;
    long long *rexp = P.rexp + (size_t)L * d;
    // per-row scalars, lane i (chunk q) = row i + 64 q
    double sg[NQ], rd[NQ], nsb[NQ], dR[NQ], eR[NQ], prevR[NQ];
    int rx[NQ], prevE[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      sg[q] = rd[q] = nsb[q] = dR[q] = eR[q] = prevR[q] = 0.0;
      rx[q] = prevE[q] = 0;
    }
    double Rk[NQ];  // the working row R[k], lane = column

    // refresh_R_bf(i) from integer values held in registers (bv) — floats the row, stores bf,
    // sets row_expo / ||b_i||^2 of lane i and leaves R[i] = bf[i] in Rk
    auto refresh_from = [&](int i, const long long(&bv)[NQ])
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
          if (P.row_expo)
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
      double sq[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int c = lane + 64 * q;
        Rk[q]       = (c < n) ? (P.row_expo ? ldexp(cm[q], ce[q] - emax) : cm[q]) : 0.0;
        sq[q]       = Rk[q] * Rk[q];
        if (c < n)
          bf[(size_t)i * ld + c] = Rk[q];
      }
      const double nb = seq_sum<NQ>(sq, 0, n);  // norm_square_b_row, householder.h:538-551
      hl_seti<NQ>(rx, i, P.row_expo ? emax : 0, lane);
      hl_set<NQ>(nsb, i, nb, lane);
    };
    auto refresh_R_bf = [&](int i)
    {
      long long bv[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int c = lane + 64 * q;
        bv[q]       = (c < n) ? b[(size_t)i * ld + c] : 0;
      }
      refresh_from(i, bv);
    };
    auto refresh_R = [&](int i)
    {  // R[i] = bf[i], householder.cpp:247-261
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int c = lane + 64 * q;
        Rk[q]       = (c < n) ? bf[(size_t)i * ld + c] : 0.0;
      }
    };
    // update_R(k, false): apply reflectors j = 0 … k-1 in order, householder.cpp:157-178
    auto apply_reflectors = [&](int k)
    {
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        settle(Rk[q]);
        settle(sg[q]);
      }
      __threadfence_block();  // V rows written earlier must be visible to the DMA reads
      ring.reset();
      ring.run(k, [&](int j) { return RowDesc{V + (size_t)j * ld, j * 8, n * 8}; },
               [&](int j, const double(&v)[NQ])
               {
                 double p[NQ];
#pragma unroll
                 for (int q = 0; q < NQ; ++q)
                 {
                   const int c = lane + 64 * q;
                   p[q]        = (c >= j && c < n) ? v[q] * Rk[q] : 0.0;
                 }
                 double s = seq_sum<NQ>(p, j, n);  // V_j . R_k over [j, n), ascending
                 s        = -s;
                 const double sj = hl_get<NQ>(sg, j);
#pragma unroll
                 for (int q = 0; q < NQ; ++q)
                 {
                   const int c = lane + 64 * q;
                   if (c >= j && c < n)
                   {
                     double t = Rk[q] + v[q] * s;  // addmul: two roundings
                     if (c == j)
                       t = sj * t;  // R(k,j) = sigma[j] * R(k,j)
                     Rk[q] = t;
                   }
                 }
               });
    };
    // update_R_last(i), householder.cpp:27-146: stores R[i], V[i], sigma[i]; R(i,i) -> rd lane i
    auto update_R_last = [&](int i)
    {
      double sq[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int c = lane + 64 * q;
        sq[q]       = (c > i && c < n) ? Rk[q] * Rk[q] : 0.0;
      }
      const double rii = hl_get<NQ>(Rk, i);
      const double sgi = (rii < 0.0) ? -1.0 : 1.0;
      double f3        = (i + 1 == n) ? 0.0 : seq_sum<NQ>(sq, i + 1, n);
      double f1        = rii * rii;
      f1               = f1 + f3;
      double vii = 0.0, new_rii = 0.0, f0 = 1.0;
      bool scale = false;
      if (f1 != 0.0)
      {
        const double f2 = sqrt(f1);
        f0              = sgi * f2;
        f1              = rii + f0;
        f3              = -f3;
        f3              = f3 / f1;
        if (f3 != 0.0)
        {
          f0      = -f0;
          f0      = f0 * f3;
          f0      = sqrt(f0);
          vii     = f3 / f0;
          new_rii = f2;
          scale   = true;
        }
        else
        {
          vii     = 0.0;
          new_rii = (rii < 0.0) ? -rii : rii;
        }
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int c = lane + 64 * q;
        if (c < n)
        {
          double vv = 0.0;
          if (c == i)
            vv = vii;
          else if (c > i && scale)
            vv = Rk[q] / f0;
          V[(size_t)i * ld + c] = vv;
          R[(size_t)i * ld + c] = (c == i) ? new_rii : Rk[q];
        }
      }
      hl_set<NQ>(sg, i, sgi, lane);
      hl_set<NQ>(rd, i, new_rii, lane);
      // compute_dR / compute_eR, hlll.h:148-159
      double t = new_rii * new_rii;
      hl_set<NQ>(dR, i, delta * t, lane);
      hl_set<NQ>(eR, i, delta * new_rii, lane);
      return new_rii;
    };

    int status = 1, n_swaps = 0;
    long long iters = 0;
    refresh_R_bf(0);
    update_R_last(0);
    int k = 1, k_max = 1, prev_k = -1;
    bool done = d < 2;
    if (!done)
      refresh_R_bf(1);
    while (!done)
    {
      if (++iters > iter_cap)
      {
        status = -6;
        break;
      }
      // ---------------- size_reduction(k, k, 0), hlll.cpp:262-351
      {
        bool prev_not_stop = true;
        apply_reflectors(k);
        for (;;)
        {
          // ---- MatHouseholder::size_reduce(k, k, 0), householder.cpp:402-451.
          // Until the first nonzero multiplier nothing changes, so the rows above it are skipped:
          // X_i^0 = rnd_we(R(k,i) / R(i,i)) evaluated for every i at once.
          const int rxk = hl_geti<NQ>(rx, k);
          int itop      = -1;
#pragma unroll
          for (int q = 0; q < NQ; ++q)
          {
            const int i = lane + 64 * q;
            bool nz     = false;
            if (i < k)
            {
              double x     = Rk[q] / rd[q];
              const int ea = rxk - rx[q];
              if (!(fexponent(x) + ea >= 53))
                x = ldexp(rint(ldexp(x, ea)), -ea);
              nz = (x != 0.0);
            }
            const uint64_t m = __ballot(nz);
            if (m)
              itop = max(itop, 64 * q + 63 - __clzll((long long)m));
          }
          itop = __builtin_amdgcn_readfirstlane(itop);
          if (itop < 0)
            break;  // not reduced: b[k] unchanged
          long long xl[NQ], bv[NQ];
          int ex[NQ];
#pragma unroll
          for (int q = 0; q < NQ; ++q)
          {
            const int c = lane + 64 * q;
            xl[q]       = 0;
            ex[q]       = rxk - rx[q];
            bv[q]       = (c < n) ? b[(size_t)k * ld + c] : 0;
          }
          bool too_big   = false;
          const int cnt  = itop + 1;  // rows itop, itop-1, …, 0
          auto r_row = [&](int s)
          {
            const int i = itop - s;
            return RowDesc{R + (size_t)i * ld, 0, i * 8};  // R(i, c) is needed for c < i
          };
          auto b_row = [&](int s) { return RowDesc{b + (size_t)(itop - s) * ld, 0, n * 8}; };
#pragma unroll
          for (int q = 0; q < NQ; ++q)
          {
            settle(Rk[q]);
            settle(rd[q]);
            settle(ex[q]);
            settle(bv[q]);
          }
          __threadfence_block();
          ring.reset();
          ring.run(
              cnt, r_row,
              [&](int s, const double(&v)[NQ])
              {
                const int i = itop - s;
                dispatch_chunk<NQ>(
                    i,
                    [&](auto iq_, int ii)
                    {
                      constexpr int iq = decltype(iq_)::value;
                      double x         = g_rl_f64(Rk[iq], ii) / g_rl_f64(rd[iq], ii);
                      const int ea     = __builtin_amdgcn_readlane(ex[iq], ii);
                      if (!(fexponent(x) + ea >= 53))  // rnd_we, nr_FP_d.inl:226-233
                        x = ldexp(rint(ldexp(x, ea)), -ea);
                      x = -x;
                      if (x != 0.0)
                      {
                        // row_addmul_we(k, i, x, ea): get_si_exp_we, nr_FP_d.inl:46-53
                        if (fexponent(x) + ea - 63 > 0)
                          too_big = true;
                        const long long lx = (long long)ldexp(x, ea);
                        xl[iq]             = (lane == ii) ? lx : xl[iq];
#pragma unroll
                        for (int q = 0; q <= iq; ++q)
                        {
                          if (q < iq || lane < ii)  // R[k][0..i) += x * R[i][0..i)
                          {
                            const double t = v[q] * x;
                            Rk[q]          = Rk[q] + t;
                          }
                        }
                      }
                    });
              },
              cnt, b_row);
          ring.run(cnt, b_row,
                   [&](int s, const double(&v)[NQ])
                   {
                     const int i = itop - s;
                     dispatch_chunk<NQ>(i,
                                        [&](auto iq_, int ii)
                                        {
                                          const long long lx = g_rl_i64(xl[decltype(iq_)::value], ii);
                                          if (lx != 0)
                                          {
#pragma unroll
                                            for (int q = 0; q < NQ; ++q)
                                              bv[q] = (long long)((unsigned long long)bv[q] +
                                                                  (unsigned long long)__double_as_longlong(v[q]) *
                                                                      (unsigned long long)lx);
                                          }
                                        });
                   });
          if (too_big)
          {
            status = -2;
            break;
          }
#pragma unroll
          for (int q = 0; q < NQ; ++q)
          {
            const int c = lane + 64 * q;
            if (c < n)
              b[(size_t)k * ld + c] = bv[q];
          }
          // ---- hlll.cpp:318-349: t = old ||b_k||^2, refresh, compare, update_R again
          const double told = hl_get<NQ>(nsb, k);
          const int e0      = P.row_expo ? 2 * rxk : 0;
          refresh_from(k, bv);
          const double tnew = hl_get<NQ>(nsb, k);
          const int e1      = P.row_expo ? 2 * hl_geti<NQ>(rx, k) : 0;
          double f0         = 0.1 * told;
          f0                = ldexp(f0, e0 - e1);
          const bool not_stop = (tnew <= f0);
          apply_reflectors(k);
          if (prev_not_stop || not_stop)
            prev_not_stop = not_stop;
          else
            break;
        }
        if (status != 1)
          break;
      }
      const int rxk = hl_geti<NQ>(rx, k);
      // ---------------- verify_size_reduction(k), hlll.cpp:455-496
      {
        double sq[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
        {
          const int c = lane + 64 * q;
          sq[q]       = (c >= k && c < n) ? Rk[q] * Rk[q] : 0.0;
        }
        double f1 = (k == n) ? 0.0 : sqrt(seq_sum<NQ>(sq, k, n));  // norm_R_row(k, k, n)
        f1        = f1 * theta;
        bool bad  = false;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
        {
          const int i = lane + 64 * q;
          if (i < k)
          {
            const double f0 = fabs(Rk[q]);
            double f2       = ldexp(eR[q], rx[q] - rxk);
            f2              = f1 + f2;
            bad |= (f0 > f2);
          }
        }
        if (__any(bad))
        {
          status = -4;
          break;
        }
      }
      // ---------------- lovasz_test(k), hlll.cpp:171-224
      bool lov;
      {
        double sq[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
        {
          const int c = lane + 64 * q;
          sq[q]       = (c < k - 1) ? Rk[q] * Rk[q] : 0.0;
        }
        double f1 = (k - 1 == 0) ? 0.0 : seq_sum<NQ>(sq, 0, k - 1);  // norm_square_R_row(k,0,k-1)
        f1        = hl_get<NQ>(nsb, k) - f1;
        const int e1 = P.row_expo ? 2 * rxk : 0;
        const int e0 = hl_geti<NQ>(rx, k - 1);
        f1           = ldexp(f1, e1 - 2 * e0);
        lov          = hl_get<NQ>(dR, k - 1) <= f1;
      }
      if (lov)
      {
        const double rkk = update_R_last(k);
        if (prev_k == k + 1)
        {  // hlll.cpp:96-108
          const double f1 = ldexp(hl_get<NQ>(prevR, k), hl_geti<NQ>(prevE, k) - rxk);
          if (rkk > f1)
          {
            status = -5;
            break;
          }
        }
        prev_k = k;
        hl_set<NQ>(prevR, k, rkk, lane);
        hl_seti<NQ>(prevE, k, rxk, lane);
        ++k;
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
          done = true;  // RED_SUCCESS
      }
      else
      {
        // ---- swap(k-1, k), householder.cpp:372-398: rows of b and bf, row_expo, ||b||^2
        ++n_swaps;
        {
          long long ba[NQ], bb[NQ];
          double fa[NQ], fb[NQ];
#pragma unroll
          for (int q = 0; q < NQ; ++q)
          {
            const int c = lane + 64 * q;
            if (c < n)
            {
              ba[q] = b[(size_t)(k - 1) * ld + c];
              bb[q] = b[(size_t)k * ld + c];
              fa[q] = bf[(size_t)(k - 1) * ld + c];
              fb[q] = bf[(size_t)k * ld + c];
            }
          }
#pragma unroll
          for (int q = 0; q < NQ; ++q)
          {
            const int c = lane + 64 * q;
            if (c < n)
            {
              b[(size_t)(k - 1) * ld + c]  = bb[q];
              b[(size_t)k * ld + c]        = ba[q];
              bf[(size_t)(k - 1) * ld + c] = fb[q];
              bf[(size_t)k * ld + c]       = fa[q];
            }
          }
          const int ea = hl_geti<NQ>(rx, k - 1), eb = hl_geti<NQ>(rx, k);
          hl_seti<NQ>(rx, k - 1, eb, lane);
          hl_seti<NQ>(rx, k, ea, lane);
          const double na = hl_get<NQ>(nsb, k - 1), nb = hl_get<NQ>(nsb, k);
          hl_set<NQ>(nsb, k - 1, nb, lane);
          hl_set<NQ>(nsb, k, na, lane);
          __threadfence_block();
        }
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
          --k;
          refresh_R(k);  // recover_R(k): see the header comment
        }
      }
    }
    // ---- outputs
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      const int i = lane + 64 * q;
      if (i < d)
      {
        rexp[i]  = rx[q];
        sigma[i] = sg[q];
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

template __global__ void hlll_kernel<1>(HhBatch, double, double, long long);
template __global__ void hlll_kernel<2>(HhBatch, double, double, long long);
template __global__ void hlll_kernel<3>(HhBatch, double, double, long long);
template __global__ void hlll_kernel<4>(HhBatch, double, double, long long);

}  // namespace fphip
