// gso_kernel.hip — batched floating-point Gram-Schmidt + size-reduction sweep for gfx950.
//
// Reference behaviour reproduced, bit for bit, for MatGSO<Z_NR<long>, FP_NR<double>> with
// GSO_ROW_EXPO (the BKZ fast path, fplll/bkz.cpp:816-829):
//   MatGSOInterface::update_gso_row   fplll/gso_interface.cpp:131-164
//   MatGSO::get_gram + dot_product    fplll/gso.h:314-331, fplll/nr/numvect.h:386-396
//   LLLReduction::babai               fplll/lll.cpp:166-224
//   LLLReduction::size_reduction      fplll/lll.h:107-122
//   MatGSO::row_addmul_we / update_bf fplll/gso.cpp:236-262, 24-48 ; row_op_end gso_interface.cpp:32-53
//
// Design (MI355X-first): the path is HBM-bound (≈0.25 flop/B), the working set of ONE lattice
// is cache-sized, and the algorithm is a long dependent chain (O(d^2) steps per sweep).  So the
// data-parallel axis is the BATCH of independent lattices: ONE WAVEFRONT OWNS ONE LATTICE, no
// LDS, no barriers, thousands of waves in flight hide the HBM latency of each other's chains.
// Inside a wave the 64 lanes are the vector axis of every inner loop, arranged so that each
// floating-point value is produced by exactly the reference's operation sequence:
//   * Gram row g(kappa,j): lane j walks the columns c = 0..n-1 in order (res = res + a*b, two
//     roundings) — bf is stored COLUMN-major so the 64 lanes read 512 contiguous bytes per c.
//   * GSO recurrence r(kappa,j) = g - sum_{k<j} mu(j,k) r(kappa,k): column-oriented forward
//     substitution — at step k the (now final) r(kappa,k) is broadcast with v_readlane and every
//     lane j>k subtracts mu(j,k)*r(kappa,k); each lane sees k = 0,1,2,… in the reference's
//     order.  mu is kept transposed as well (muT) so that step k reads one contiguous column.
//   * size-reduction sweep: lane k holds babai_mu[k]; j runs downwards, X_j is rounded from the
//     broadcast babai_mu[j] and lanes k<j subtract X_j*mu(j,k) (row j of mu: contiguous).
//   * integer AXPY b_kappa += sum_j lx_j b_j: lanes are columns; 64-bit wrapping arithmetic.
// d and n up to 64*NQ are handled with NQ registers per lane (template parameter).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (no FMA contraction).

#include "gso_wave.h"

namespace fphip
{

// mode 0: update_gso() (every row, no size reduction); mode 1: size_reduction(kmin,kend);
// mode 2: (re)build bfT / row_expo from b for every row (after a basis upload)
template <int NQ>
__global__ void __launch_bounds__(256)
    gso_sweep_kernel(GsoBatch P, int kmin, int kend, double eta, int mode)
{
  constexpr int IPS = (NQ + 1) / 2;  // 1 KiB DMA instructions per row (rows <= 64*NQ*8 bytes)
  extern __shared__ __attribute__((aligned(16))) char gso_smem[];
  const int lane = threadIdx.x & 63;
  const int wpb  = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  Ring<NQ, IPS> ring;
  // the kernel declares no static LDS, so the dynamic segment starts at LDS address 0
  ring.base = (unsigned)(wave * Ring<NQ, IPS>::R * Ring<NQ, IPS>::SLOT);
  ring.lane = lane;
  ring.head = ring.tail = 0;
  ring.ahead = 0;
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
    if (mode != 2)
    {  // narrow prefix from the per-row flags (FPHIP_GSO_NARROW=0: P.use_narrow == 0)
      int p = 0;
      while (P.use_narrow && p < P.d && __builtin_amdgcn_readfirstlane(T.narrow_flag[p]) != 0)
        ++p;
      T.np = p;
    }
    int status    = 1;
    if (mode == 2)
    {
      // (re)float every row from b (MatGSO::update_bf, gso.cpp:24-48) and rebuild the narrow
      // mirrors; the lattice streams 4-byte rows while all its entries stay below 2^24
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
      }
    }
    else
    {
      for (int kappa = kmin; kappa < kend; ++kappa)
      {
        if (mode == 1 && kappa > 0)
        {
          const int rc = babai(T, ring, kappa, eta);
          if (rc != 1)
          {
            status = rc;
            break;
          }
        }
        if (mode == 1 && kappa > 0)
        {
          finish_diag<NQ>(T, kappa);  // babai left the row valid up to column kappa-1
        }
        else if (!update_row(T, ring, kappa, kappa))
        {
          status = 0;
          break;
        }
        __threadfence_block();
      }
    }
    if (lane == 0)
      P.status[L] = status;
  }
}

template __global__ void gso_sweep_kernel<1>(GsoBatch, int, int, double, int);
template __global__ void gso_sweep_kernel<2>(GsoBatch, int, int, double, int);
template __global__ void gso_sweep_kernel<3>(GsoBatch, int, int, double, int);
template __global__ void gso_sweep_kernel<4>(GsoBatch, int, int, double, int);

}  // namespace fphip

// ---------------------------------------------------------------------------------------------
// Householder R-factor (MatHouseholder<Z_NR<long>, FP_NR<double>>): refresh_R_bf() + update_R()
// over all rows, fplll/householder.cpp:27-245, householder.h:532-536 — batched, one wavefront per
// lattice, lane = COLUMN.  Row i lives in registers; the reflectors V[0..i) are streamed through
// the LDS-DMA ring.  The reference's dot product is a SEQUENTIAL sum over the columns
// (nr/numvect.h:386-396), so the per-lane products are added in ascending column order with a
// v_readlane chain (bit-exact), while the AXPY R_i += (-s) V_j is a plain vector operation
// (element-wise, two roundings, numvect.h:300-305).
// ---------------------------------------------------------------------------------------------
namespace fphip
{

template <int NQ>
__global__ void __launch_bounds__(256) hh_update_kernel(HhBatch P)
{
  constexpr int IPS = (NQ + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) char hh_smem[];
  const int lane = threadIdx.x & 63;
  const int wpb  = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  Ring<NQ, IPS> ring;
  ring.base = (unsigned)(wave * Ring<NQ, IPS>::R * Ring<NQ, IPS>::SLOT);
  ring.lane = lane;
  ring.head = ring.tail = 0;
  ring.ahead            = 0;
  const int d = P.d, n = P.n, ld = P.ldn;
  for (int L = blockIdx.x * wpb + wave; L < P.batch; L += gridDim.x * wpb)
  {
    const long long *b = P.b + (size_t)L * d * ld;
    double *V          = P.V + (size_t)L * d * ld;
    double *R          = P.R + (size_t)L * d * ld;
    double *sigma      = P.sigma + (size_t)L * d;
    long long *rexp    = P.rexp + (size_t)L * d;
    double sg[NQ];  // sigma[j] in lane j
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      sg[q] = 0.0;
    for (int i = 0; i < d; ++i)
    {
      // ---- refresh_R_bf(i): float the integer row (row exponent optional), householder.cpp:186-245
      double Ri[NQ];
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
        emax = wave_max_i32(emax);
#pragma unroll
        for (int q = 0; q < NQ; ++q)
        {
          const int c = lane + 64 * q;
          Ri[q]       = (c < n) ? (P.row_expo ? ldexp(cm[q], ce[q] - emax) : cm[q]) : 0.0;
        }
        if (lane == 0)
          rexp[i] = P.row_expo ? (long long)emax : 0;
      }
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        settle(Ri[q]);
        settle(sg[q]);
      }
      __threadfence_block();  // V rows written for earlier i must be visible to the DMA reads
      ring.reset();
      // ---- update_R(i): apply reflectors j = 0 … i-1 in order, householder.cpp:157-178
      ring.run(i, [&](int j) { return RowDesc{V + (size_t)j * ld, j * 8, n * 8}; },
               [&](int j, const double(&v)[NQ])
               {
                 double p[NQ];
#pragma unroll
                 for (int q = 0; q < NQ; ++q)
                 {
                   const int c = lane + 64 * q;
                   p[q]        = (c >= j && c < n) ? v[q] * Ri[q] : 0.0;
                 }
                 double s = seq_sum<NQ>(p, j, n);  // V_j . R_i over [j, n), ascending
                 s        = -s;
                 double sj = 0.0;
                 dispatch_chunk<NQ>(j, [&](auto jq, int jj) { sj = g_rl_f64(sg[decltype(jq)::value], jj); });
#pragma unroll
                 for (int q = 0; q < NQ; ++q)
                 {
                   const int c = lane + 64 * q;
                   if (c >= j && c < n)
                   {
                     double t = Ri[q] + v[q] * s;  // addmul: two roundings
                     if (c == j)
                       t = sj * t;  // R(i,j) = sigma[j] * R(i,j)
                     Ri[q] = t;
                   }
                 }
               });
      // ---- update_R_last(i), householder.cpp:27-146
      double sq[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int c = lane + 64 * q;
        sq[q]       = (c > i && c < n) ? Ri[q] * Ri[q] : 0.0;
      }
      double rii = 0.0;
      dispatch_chunk<NQ>(i, [&](auto iq, int ii) { rii = g_rl_f64(Ri[decltype(iq)::value], ii); });
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
            vv = Ri[q] / f0;
          V[(size_t)i * ld + c] = vv;
          R[(size_t)i * ld + c] = (c == i) ? new_rii : Ri[q];
          if (c == i)
          {
            sg[q]    = sgi;
            sigma[i] = sgi;
          }
        }
      }
    }
    if (lane == 0)
      P.status[L] = 1;
  }
}

template __global__ void hh_update_kernel<1>(HhBatch);
template __global__ void hh_update_kernel<2>(HhBatch);
template __global__ void hh_update_kernel<3>(HhBatch);
template __global__ void hh_update_kernel<4>(HhBatch);

}  // namespace fphip

// ---------------------------------------------------------------------------------------------
// Calibration kernel for the rocprofv3 FETCH_SIZE counter (MI355X_MICROARCH.md §HBM: "calibrate on
// a known byte count in your own access pattern"): streams `rows` rows of `row_bytes` bytes with the
// SAME instruction the sweep uses (global_load_lds_dwordx4, 16 B per lane, windowed), no reuse.
// ---------------------------------------------------------------------------------------------
namespace fphip
{
__global__ void __launch_bounds__(256)
    gso_calib_kernel(const char *buf, size_t stride, int row_bytes, long long rows)
{
  extern __shared__ __attribute__((aligned(16))) char cal_smem[];
  const int lane  = threadIdx.x & 63;
  const int wave  = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wpb   = blockDim.x >> 6;
  const long long gw = (long long)blockIdx.x * wpb + wave;
  const long long nw = (long long)gridDim.x * wpb;
  const unsigned base = (unsigned)(wave * 8 * 2048);
  int slot = 0;
  for (long long r = gw; r < rows; r += nw)
  {
    const char *g = buf + (size_t)r * stride + lane * 16;
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(base + slot * 2048));
    for (int i = 0; i < 2; ++i)
    {
      const int off = lane * 16 + i * 1024;
      if (lane == 0 || off < row_bytes)
        glds16(g + i * 1024, dst + i * 1024);
    }
    slot = (slot + 1) & 7;
    if (slot == 0)
      wait_vmcnt<0>();
  }
  wait_vmcnt<0>();
}
}  // namespace fphip
