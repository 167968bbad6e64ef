// hh_rows.hip — the Householder R-factor of a BATCH of lattices in the reference's exact arithmetic, with the
// batch (not the columns) across the lanes.
//
// Reference: MatHouseholder<Z_NR<long>, FP_NR<double>>::refresh_R_bf (fplll/householder.cpp:186-245),
// update_R(i, false) (:151-184), update_R_last (:27-146); dot products are SEQUENTIAL sums over the columns
// (nr/numvect.h:386-396: ftmp.mul(first), then addmul in ascending order — two roundings per term, no FMA), the
// AXPY is element-wise (numvect.h:300-305).  Bit for bit the output of hh_update_kernel (gso_kernel.hip), which
// stays the kernel of rows wider than 192 columns and the A/B partner (FPHIP_HH_ROWS=0).
//
// Why another kernel.  hh_update_kernel puts the COLUMNS of one row in the lanes; the reference's summation
// order then forces every dot product through a chain of v_readlane + add per column: 540 vector instructions
// per reflector application for ONE row at n = 180, 131 ms for 4096 x (180 x 180) = 0.063 of the HBM roofline on
// the SURVEY 8(d) bytes — bound by instruction issue, not by bytes (profiles/r05_hh_blocked_mfma.md).  But the
// batch is embarrassingly parallel: a lane that owns a whole ROW of one lattice runs the reference's scalar loop
// as it stands — s += v[c] * r[c], c ascending — with no cross-lane traffic at all, and 64 rows advance per
// instruction.  4 (n - j) vector instructions per application for 64 rows instead of 3 (n - j) for one.
//
// Layout.  One workgroup of 4 wavefronts = 16 lattices x a PANEL of 16 consecutive rows: lane l of wave w holds
// row 16 p + 4 w + (l >> 4) of lattice 16 g + (l & 15), the whole row in registers (columns as compile-time
// indices: the loops over columns are unrolled, their bounds — the reflector index j, the row length n — are
// wave-uniform, so the boundary tiles are uniform branches).  Rows wider than the 256 architectural registers
// hold: the compiler parks the overflow in the accumulation registers (v_accvgpr_*), one wave per SIMD.
//   phase 1  reflectors 0 .. 16 p - 1 (of 16 lattices each) stream through an LDS ring by DMA
//            (global_load_lds_dwordx4, no registers), one __syncthreads per reflector; every lane applies the
//            reflector of ITS lattice to its row: ds_read_b128 of two columns (the four rows of a lattice in a
//            wave read the same address: one bank access), multiply, add.
//   phase 2  the 16 rows of the panel against each other: row 16 p + m finishes (update_R_last: tail norm,
//            sigma, v, R(i,i)), its reflector goes to LDS, the rows behind it apply it.
// A reflector is fetched once per 16 rows: the kernel moves a sixteenth of the bytes SURVEY 8(d) charges
// (V re-read for every row), so its "achieved" figure on those bytes can exceed what HBM could deliver.
#include <hip/hip_runtime.h>
#include <limits.h>

#include "gso_device.h"

namespace fphip
{

namespace
{
__device__ __forceinline__ void hr_glds16(const void *gsrc, unsigned lds_dst)
{
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
}  // namespace

// NT = 16-column tiles of a row (n <= 16 NT)
template <int NT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) hh_rows_kernel(HhBatch P)
{
  constexpr int NC = 16 * NT;   // register columns
  constexpr int SV = NC + 2;    // doubles between two lattices of an LDS buffer: 16 lattices on 16 distinct bank groups
  constexpr int NB = 3;         // ring of reflector buffers
  constexpr int BUF = 16 * SV;  // doubles per buffer
  extern __shared__ __attribute__((aligned(16))) double hr_smem[];
  double *sig_l = hr_smem + NB * BUF;  // [NB][16]: sigma_j of the 16 lattices
  const int tid  = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lat  = lane & 15;         // lattice of the group
  const int rr   = 4 * wave + (lane >> 4);  // row of the panel
  const int d = P.d, n = P.n, ld = P.ldn;
  const int ngroups = (P.batch + 15) / 16;
  const unsigned lds0 = (unsigned)(size_t)hr_smem;  // (byte address of the dynamic LDS segment)

  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x)
  {
    const int L     = 16 * grp + lat;
    const bool live = L < P.batch;
    const int Lc    = live ? L : P.batch - 1;  // (lanes beyond the batch shadow its last lattice and store nothing)
    const long long *b = P.b + (size_t)Lc * d * ld;
    double *V          = P.V + (size_t)Lc * d * ld;
    double *R          = P.R + (size_t)Lc * d * ld;
    double *sigma      = P.sigma + (size_t)Lc * d;
    long long *rexp    = P.rexp + (size_t)Lc * d;

    for (int p0 = 0; p0 < d; p0 += 16)
    {
      const int i     = p0 + rr;    // this lane's row
      const bool mine = i < d;
      const int ic    = mine ? i : d - 1;
      double Rr[NC];
      // ---- refresh_R_bf(i), householder.cpp:186-245: float the integer row; with row exponents the row is
      //      scaled by 2^-emax, emax = the largest frexp exponent of its entries (frexp(0) reports 0) — exactly
      //      ldexp(frexp mantissa, e - emax)
      {
        int emax = INT_MIN;
#pragma unroll
        for (int c = 0; c < NC; ++c)
        {
          double m = 0.0;
          if (c < n)
          {
            m = (double)b[(size_t)ic * ld + c];
            if (P.row_expo)
            {
              int ex;
              (void)frexp(m, &ex);
              emax = max(emax, ex);
            }
          }
          Rr[c] = m;
        }
        if (P.row_expo)
        {
#pragma unroll
          for (int c = 0; c < NC; ++c)
            if (c < n)
              Rr[c] = ldexp(Rr[c], -emax);
        }
        else
          emax = 0;
        if (live && mine)
          rexp[i] = (long long)emax;
      }
      // ---- apply one reflector (of this lane's lattice, in LDS at vb) to the row: householder.cpp:157-178
      //      s = sum_{c >= j} V_j[c] R_i[c] (ascending, the first term assigned: adding to -0.0 is the same),
      //      R_i[c] += V_j[c] * (-s) (two roundings), R_i[j] *= sigma_j
      auto apply = [&](const double *vb, int j, double sj) __attribute__((always_inline))
      {
        const int tj = j >> 4;
        double s     = -0.0;
#pragma unroll
        for (int t = 0; t < NT; ++t)
        {
          if (16 * t + 15 < j || 16 * t >= n)
            continue;
          if (t == tj || 16 * t + 16 > n)
          {  // a boundary tile: the columns [j, n) of it, one uniform branch per column (the empty statement keeps
             // the optimiser from turning the branches into selects — two extra instructions per column of EVERY tile)
#pragma unroll
            for (int cc = 0; cc < 16; ++cc)
            {
              const int c = 16 * t + cc;
              if (c >= j && c < n)
              {
                asm volatile("");
                s = s + vb[c] * Rr[c];
              }
            }
          }
          else
          {
#pragma unroll
            for (int cc = 0; cc < 16; cc += 2)
            {
              const int c      = 16 * t + cc;
              const double2 v2 = *(const double2 *)(vb + c);
              s                = s + v2.x * Rr[c];
              s                = s + v2.y * Rr[c + 1];
            }
          }
        }
        s = -s;
#pragma unroll
        for (int t = 0; t < NT; ++t)
        {
          if (16 * t + 15 < j || 16 * t >= n)
            continue;
          if (t == tj || 16 * t + 16 > n)
          {
#pragma unroll
            for (int cc = 0; cc < 16; ++cc)
            {
              const int c = 16 * t + cc;
              if (c >= j && c < n)
              {
                asm volatile("");
                double tv = Rr[c] + vb[c] * s;
                if (c == j)
                {
                  asm volatile("");
                  tv = sj * tv;
                }
                Rr[c] = tv;
              }
            }
          }
          else
          {
#pragma unroll
            for (int cc = 0; cc < 16; cc += 2)
            {
              const int c      = 16 * t + cc;
              const double2 v2 = *(const double2 *)(vb + c);
              Rr[c]            = Rr[c] + v2.x * s;
              Rr[c + 1]        = Rr[c + 1] + v2.y * s;
            }
          }
        }
      };
      // ---- phase 1: the reflectors of the rows above the panel, through the LDS ring
      const int J = p0;
      auto issue = [&](int j) __attribute__((always_inline))
      {  // reflector j of the 16 lattices -> buffer j % NB: columns >= 16 (j >> 4) only; wave w copies lattices 4 w ..
        const int slot      = j % NB;
        const unsigned dstb = lds0 + (unsigned)(slot * BUF) * 8u;
        const int c0        = (j >> 4) << 4;
#pragma unroll
        for (int q = 0; q < 4; ++q)
        {
          const int la = 4 * wave + q;
          const int Lg = min(16 * grp + la, P.batch - 1);
          const double *src = P.V + ((size_t)Lg * d + j) * ld;
          for (int cb = c0; cb < NC; cb += 128)
          {  // (the DMA writes lane l at its LDS base + 16 l: the base is the wave's, the column the lane's)
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(dstb + (unsigned)(la * SV + cb) * 8u));
            if (cb + 2 * lane < min(ld, NC))
              hr_glds16(src + cb + 2 * lane, dst);
          }
        }
        if (tid < 16)
          sig_l[slot * 16 + tid] = P.sigma[(size_t)min(16 * grp + tid, P.batch - 1) * d + j];
      };
      for (int j = 0; j < NB - 1 && j < J; ++j)
        issue(j);
      for (int j = 0; j < J; ++j)
      {
        // reflector j has landed for everybody; the buffer of j - 1 is free again
        if (j + NB - 1 < J)
        {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          issue(j + NB - 1);
        }
        else
        {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
        }
        const int slot = j % NB;
        apply(hr_smem + slot * BUF + lat * SV, j, sig_l[slot * 16 + lat]);
      }
      __syncthreads();
      // ---- phase 2: the rows of the panel, one after the other
      double *pb = hr_smem;  // buffer 0: the reflector of the row that just finished
      for (int m = 0; m < 16 && p0 + m < d; ++m)
      {
        const int im = p0 + m;
        if (rr == m)
        {
          // ---- update_R_last(im), householder.cpp:27-146 (sequential tail norm)
          double f3 = -0.0, rii = 0.0;
#pragma unroll
          for (int c = 0; c < NC; ++c)
          {
            if (c == im)
              rii = Rr[c];
            if (c > im && c < n)
              f3 = f3 + Rr[c] * Rr[c];
          }
          if (im + 1 >= n)
            f3 = 0.0;
          const double sgi = (rii < 0.0) ? -1.0 : 1.0;
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
          double *pv = pb + lat * SV;
#pragma unroll
          for (int c = 0; c < NC; ++c)
          {
            if (c < n)
            {
              double vv = 0.0;
              if (c == im)
                vv = vii;
              else if (c > im && scale)
                vv = Rr[c] / f0;
              pv[c] = vv;
              if (live)
              {
                V[(size_t)im * ld + c] = vv;
                R[(size_t)im * ld + c] = (c == im) ? new_rii : Rr[c];
              }
            }
            else
              pv[c] = 0.0;
          }
          sig_l[lat] = sgi;
          if (live)
            sigma[im] = sgi;
        }
        __syncthreads();
        if (rr > m && mine)
          apply(pb + lat * SV, im, sig_l[lat]);
        __syncthreads();
      }
    }
    if (live && rr == 0)
      P.status[L] = 1;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// MatHouseholder::size_reduce(k, size_reduction_end, size_reduction_start), householder.cpp:402-451, as a
// stand-alone step over the batch (C ABI fphip_hh_size_reduce; the HLLL kernel carries its own copy inside its loop):
// one wavefront per lattice, lane = column.  For i = end-1 … start: X = -rnd_we(R(k,i) / R(i,i)) (:409-426), and for
// a nonzero X row_addmul_we(k, i, X, row_expo[k] - row_expo[i]) (:522-559): b[k] += lx b[i] over the columns, and
// R[k].addmul(R[i], X, k) over ALL k leading entries — R(k,i) becomes the remainder, the entries between i and k
// pick up X times the tail update_R_last left in row i (the reference never zeroes it in a non-DEBUG build,
// :104-111; hh_rows_kernel / hh_update_kernel store the same tail).  Separate multiply and add (-ffp-contract=off).
// reduced[lattice] = the reference's return value; status -2 = a multiplier beyond 63 bits (b and R untouched).
// HBM-bound: (end - start) rows of R and of b per lattice, 16 n bytes each.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ long long hsr_fexponent(double x)
{  // FP_NR<double>::exponent(), nr_FP_d.inl:44
  return (x == 0.0) ? ((long long)INT_MIN + 1) : ((long long)ilogb(x) + 1);
}

__global__ void __launch_bounds__(256) hh_size_reduce_kernel(HhBatch P, int k, int end, int start, int *reduced)
{
  const int lane = threadIdx.x & 63;
  const int lat  = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (lat >= P.batch)
    return;
  const int n = P.n, ld = P.ldn, d = P.d;
  double *R           = P.R + (size_t)lat * d * ld;
  long long *b        = P.b + (size_t)lat * d * ld;
  const long long *rx = P.rexp + (size_t)lat * d;
  double Rk[4];
  long long bk[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
  {
    const int c = lane + 64 * q;
    Rk[q]       = (c < n) ? R[(size_t)k * ld + c] : 0.0;
    bk[q]       = (c < n) ? b[(size_t)k * ld + c] : 0;
  }
  const long long rxk = rx[k];
  int red             = 0;
  bool too_big        = false;
  for (int i = end - 1; i >= start; --i)
  {
    double mine = Rk[0];
#pragma unroll
    for (int q = 1; q < 4; ++q)
      mine = ((i >> 6) == q) ? Rk[q] : mine;
    const double rki = __shfl(mine, i & 63);
    const double rii = R[(size_t)i * ld + i];
    double x         = rki / rii;
    const int ea     = (int)(rxk - rx[i]);
    if (!(hsr_fexponent(x) + ea >= 53))  // rnd_we, nr_FP_d.inl:226-233
      x = ldexp(rint(ldexp(x, ea)), -ea);
    x = -x;
    if (x != 0.0)
    {
      if (hsr_fexponent(x) + ea - 63 > 0)  // get_si_exp_we, nr_FP_d.inl:46-53: the 2^expo path is not on the device
      {
        too_big = true;
        break;
      }
      const long long lx = (long long)ldexp(x, ea);
#pragma unroll
      for (int q = 0; q < 4; ++q)
      {
        const int c = lane + 64 * q;
        if (c < n)
        {
          bk[q] = (long long)((unsigned long long)bk[q] +
                              (unsigned long long)b[(size_t)i * ld + c] * (unsigned long long)lx);
          if (c < k)
          {
            const double t = R[(size_t)i * ld + c] * x;
            Rk[q]          = Rk[q] + t;
          }
        }
      }
      red = 1;
    }
  }
  if (too_big)
  {
    if (lane == 0)
    {
      P.status[lat] = -2;
      reduced[lat]  = 0;
    }
    return;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
  {
    const int c = lane + 64 * q;
    if (c < n && red)
    {
      b[(size_t)k * ld + c] = bk[q];
      if (c < k)
        R[(size_t)k * ld + c] = Rk[q];
    }
  }
  if (lane == 0)
  {
    P.status[lat] = 1;
    reduced[lat]  = red;
  }
}

template __global__ void hh_rows_kernel<4>(HhBatch);
template __global__ void hh_rows_kernel<8>(HhBatch);
template __global__ void hh_rows_kernel<12>(HhBatch);

}  // namespace fphip
