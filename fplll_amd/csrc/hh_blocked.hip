// hh_blocked.hip — Householder R-factor in BLOCKED (compact-WY) form on the MFMA matrix cores:
// the opt-in fast mode of fphip_hh_update_R (C ABI fphip_hh_update_R_blocked).
//
// The reference applies the reflectors to one row at a time, each as a sequential dot product and an
// AXPY (MatHouseholder::update_R, fplll/householder.cpp:151-184); the exact kernel (hh_update_kernel
// in gso_kernel.hip) keeps that order and is bound by the v_readlane chain of those sequential sums.
// Applying a BLOCK of 16 reflectors to a PANEL of 16 rows at once is a GEMM-shaped contraction
//     X <- X - ((X V^T) T) V            X: 16 x n panel,  V: 16 x n reflectors,  T: 16 x 16
// (H_0 H_1 … H_15 = I - V^T T V with T upper triangular, T <- [[T, -T (V v)], [0, 1]] per new
// reflector: every H_j here is I - v_j v_j^T, householder.cpp:168-175).  It changes the order of
// the floating-point sums, hence the rounding: this mode is checked to 1e-9 relative on mu / r
// against the exact kernel, never bit for bit.
//
// One wavefront per lattice.  v_mfma_f64_16x16x4_f64: lane l holds A[l&15][l>>4], B[l>>4][l&15] and
// four results D[(l>>4) + 4 r][l&15].  All three products are taken TRANSPOSED so that the panel
// stays in ONE register layout throughout — xt[t][r] of lane l = X[l&15][16 t + 4 r + (l>>4)]:
//     W^T = V X^T      B operand of 4-column chunk c = 4 t + r is xt[t][r] itself
//     Y^T = T^T W^T    B operand of chunk k is the accumulator register w[k] itself
//     X^T -= V^T Y^T   accumulates into xt[t] (the C/D layout of tile t IS that layout); B = y[k]
// Only V and T are fetched (from global memory / L2, 16 x 16 at a time).  The 16 rows of a panel are
// then factored against each other through LDS with lanes = columns (tree sums), which also yields
// the panel's own V rows and T.
#include <hip/hip_runtime.h>
#include <limits.h>

#include "gso_device.h"

namespace fphip
{

typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ int wave_max_i(int v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    v = max(v, __shfl_xor(v, off));
  return v;
}

// Tbuf: [batch][ceil(d/16)][16][16] doubles (T of every block of reflectors)
template <int NQ>
__global__ void __launch_bounds__(64) hh_blocked_kernel(HhBatch P, double *Tbuf)
{
  constexpr int NT = 4 * NQ;  // 16-column tiles of a row
  extern __shared__ __attribute__((aligned(16))) double hb_smem[];
  const int lane = threadIdx.x & 63;
  const int d = P.d, n = P.n, ld = P.ldn;
  const int ldx = ((n + 31) & ~31) + 1;  // LDS row stride of the panel (odd: rows on distinct banks)
  double *Xs    = hb_smem;               // [16][ldx]: the panel; row m becomes v_{p+m} once factored
  double *Ts    = hb_smem + 16 * ldx;    // [16][16]: T of the panel being factored
  const int m16 = lane & 15, g4 = lane >> 4;
  const int nblk = (d + 15) / 16;
  for (int L = blockIdx.x; L < P.batch; L += gridDim.x)
  {
    const long long *b = P.b + (size_t)L * d * ld;
    double *V          = P.V + (size_t)L * d * ld;
    double *R          = P.R + (size_t)L * d * ld;
    double *sigma      = P.sigma + (size_t)L * d;
    long long *rexp    = P.rexp + (size_t)L * d;
    double *T          = Tbuf + (size_t)L * nblk * 256;
    for (int p = 0; p < d; p += 16)
    {
      const int rows = min(16, d - p);
      // ---- refresh_R_bf for the 16 rows of the panel (householder.cpp:186-245), in the MFMA
      //      layout: lane l holds row p + (l&15), columns 16 t + 4 r + (l>>4)
      v4d xt[NT];
      {
        int emax = INT_MIN;
        int ce[NT][4];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)
          {
            const int c = 16 * t + 4 * r + g4;
            double m    = 0.0;
            int ex      = INT_MIN;
            if (c < n && m16 < rows)
            {
              const long long v = b[(size_t)(p + m16) * ld + c];
              if (P.row_expo)
              {
                m = frexp((double)v, &ex);
                emax = max(emax, ex);
              }
              else
              {
                m  = (double)v;
                ex = 0;
                emax = 0;
              }
            }
            xt[t][r] = m;
            ce[t][r] = ex;
          }
        // the maximum over a row: its columns sit in the four lanes l, l^16, l^32, l^48
        emax = max(emax, __shfl_xor(emax, 16));
        emax = max(emax, __shfl_xor(emax, 32));
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (P.row_expo && ce[t][r] != INT_MIN)
              xt[t][r] = ldexp(xt[t][r], ce[t][r] - emax);
        if (g4 == 0 && m16 < rows)
          rexp[p + m16] = P.row_expo ? (long long)emax : 0;
      }
      // ---- the blocks of earlier reflectors, in order: X <- X - ((X V^T) T) V on the matrix cores
      for (int K = 0; K < p / 16; ++K)
      {
        const double *VK = V + (size_t)(16 * K) * ld;
        const double *TK = T + (size_t)K * 256;
        v4d w            = {0.0, 0.0, 0.0, 0.0};
        // W^T = V_K X^T  (columns below 16 K are zero in every v_j of the block)
#pragma unroll
        for (int t = 0; t < NT; ++t)
          if (t >= K && 16 * t < n)
#pragma unroll
            for (int r = 0; r < 4; ++r)
            {
              const int c    = 16 * t + 4 * r + g4;
              const double a = (c < n) ? VK[(size_t)m16 * ld + c] : 0.0;
              w              = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xt[t][r], w, 0, 0, 0);
            }
        // Y^T = T_K^T W^T
        v4d y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
          const double a = TK[(4 * k + g4) * 16 + m16];  // T_K^T[i = l&15][4 k + (l>>4)]
          y              = __builtin_amdgcn_mfma_f64_16x16x4f64(a, w[k], y, 0, 0, 0);
        }
        // X^T -= V_K^T Y^T
#pragma unroll
        for (int t = 0; t < NT; ++t)
          if (t >= K && 16 * t < n)
#pragma unroll
            for (int k = 0; k < 4; ++k)
            {
              const int c    = 16 * t + m16;
              const double a = (c < n) ? -VK[(size_t)(4 * k + g4) * ld + c] : 0.0;
              xt[t]          = __builtin_amdgcn_mfma_f64_16x16x4f64(a, y[k], xt[t], 0, 0, 0);
            }
      }
      // ---- columns below the panel are final: R(i,j) = sigma_j * X(i,j)  (householder.cpp:176)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
        {
          const int c = 16 * t + 4 * r + g4;
          if (m16 < rows && c < n)
          {
            if (c < p)
              R[(size_t)(p + m16) * ld + c] = sigma[c] * xt[t][r];
            else
              Xs[m16 * ldx + c] = xt[t][r];
          }
        }
      __builtin_amdgcn_s_waitcnt(0);  // this wave's LDS writes before its LDS reads below
      __builtin_amdgcn_wave_barrier();
      // ---- factor the panel: rows one after the other, lanes = columns (NQ per lane)
      double sgp = 1.0;  // lane mm: sigma of row p + mm (this panel)
      for (int m = 0; m < rows; ++m)
      {
        const int i = p + m;
        double Ri[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
        {
          const int c = lane + 64 * q;
          Ri[q]       = (c >= p && c < n) ? Xs[m * ldx + c] : 0.0;
        }
        for (int mm = 0; mm < m; ++mm)
        {  // apply H_{p+mm} (and the sign of column p+mm)
          double part = 0.0, vv[NQ];
#pragma unroll
          for (int q = 0; q < NQ; ++q)
          {
            const int c = lane + 64 * q;
            vv[q]       = (c >= p + mm && c < n) ? Xs[mm * ldx + c] : 0.0;
            part += vv[q] * Ri[q];
          }
          const double s  = wave_sum_f64(part);
          const double sg = __shfl(sgp, mm);
#pragma unroll
          for (int q = 0; q < NQ; ++q)
          {
            const int c = lane + 64 * q;
            double t    = Ri[q] - s * vv[q];
            if (c == p + mm)
              t = sg * t;
            Ri[q] = t;
          }
        }
        // ---- update_R_last(i), householder.cpp:27-146 (tree sum for the tail norm)
        double sq = 0.0, rii = 0.0;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
        {
          const int c = lane + 64 * q;
          if (c > i && c < n)
            sq += Ri[q] * Ri[q];
          if (c == i)
            rii = Ri[q];
        }
        double f3 = wave_sum_f64(sq);
        rii       = wave_sum_f64(rii);  // one lane holds it, the others 0
        const double sgi = (rii < 0.0) ? -1.0 : 1.0;
        double f1        = rii * rii + f3;
        double vii = 0.0, new_rii = 0.0, f0 = 1.0;
        bool scale = false;
        if (f1 != 0.0)
        {
          const double f2 = sqrt(f1);
          f0              = sgi * f2;
          f1              = rii + f0;
          f3              = -f3 / f1;
          if (f3 != 0.0)
          {
            f0      = sqrt(-f0 * f3);
            vii     = f3 / f0;
            new_rii = f2;
            scale   = true;
          }
          else
          {
            vii     = 0.0;
            new_rii = fabs(rii);
          }
        }
        double vi[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
        {
          const int c = lane + 64 * q;
          vi[q]       = 0.0;
          if (c < n)
          {
            if (c == i)
              vi[q] = vii;
            else if (c > i && scale)
              vi[q] = Ri[q] / f0;
            V[(size_t)i * ld + c] = vi[q];
            if (c >= p)
              R[(size_t)i * ld + c] = (c == i) ? new_rii : Ri[q];
          }
        }
        if (lane == 0)
          sigma[i] = sgi;
        if (lane == m)
          sgp = sgi;
        // ---- T column m: T[0:m, m] = -T[0:m, 0:m] (V[0:m] v_i), T[m][m] = 1
        double gd = 0.0;  // lane mm: v_{p+mm} . v_i
        for (int mm = 0; mm < m; ++mm)
        {
          double part = 0.0;
#pragma unroll
          for (int q = 0; q < NQ; ++q)
          {
            const int c = lane + 64 * q;
            if (c >= p + mm && c < n)
              part += Xs[mm * ldx + c] * vi[q];
          }
          const double g = wave_sum_f64(part);
          if (lane == mm)
            gd = g;
        }
        {
          double tv = (lane == m) ? 1.0 : 0.0;
          for (int bb = 0; bb < m; ++bb)
          {
            const double gb = __shfl(gd, bb);
            if (lane < m && bb >= lane)
              tv -= Ts[lane * 16 + bb] * gb;
          }
          if (lane < 16)
            Ts[lane * 16 + m] = tv;
        }
        // row m of the LDS panel now holds v_i (read by the following rows and by T)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
        {
          const int c = lane + 64 * q;
          if (c >= p && c < n)
            Xs[m * ldx + c] = vi[q];
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
      }
      // ---- keep T of this block for the later panels (rows beyond `rows`: identity padding is
      //      never read — the last panel has no successor)
      for (int e = lane; e < 256; e += 64)
      {
        const int a = e >> 4, bb = e & 15;
        T[(size_t)(p / 16) * 256 + e] = (a < rows && bb < rows && bb >= a) ? Ts[e] : 0.0;
      }
      __threadfence_block();
    }
    if (lane == 0)
      P.status[L] = 1;
  }
}

template __global__ void hh_blocked_kernel<1>(HhBatch, double *);
template __global__ void hh_blocked_kernel<2>(HhBatch, double *);
template __global__ void hh_blocked_kernel<3>(HhBatch, double *);
template __global__ void hh_blocked_kernel<4>(HhBatch, double *);

}  // namespace fphip
