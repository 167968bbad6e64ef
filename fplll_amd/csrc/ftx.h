// ftx.h — the floating-point type of the extended-precision Householder / HLLL kernels (hlll_x.hip):
// plain double, or DD = double-double (an unevaluated sum hi + lo of two doubles, ~106 bits), the
// device stand-in for the reference's FP_NR<dd_real> (fplll/nr/nr_FP_dd.inl over libqd's dd_real).
//
// libqd is an un-vendored optional dependency of the reference and is absent here,
// so the algorithms below are restated from their published form (Dekker 1971, Knuth TAOCP 4.2.2,
// Hida-Li-Bailey "Library for double-double and quad-double arithmetic", 2007) in libqd's default
// configuration: "sloppy" addition (QD_IEEE_ADD undefined), the accurate three-step division,
// Karp's square root, nint by parts.  two_prod uses the hardware FMA (exact product error).  PARITY
// IS UNPINNED against libqd bit for bit; the kernels are checked against the reference run at 106
// bits of MPFR (the oracle build's FP_NR<mpfr_t>) to double-double accuracy (tests/test_dd_gpu.py).
#ifndef FPHIP_FTX_H
#define FPHIP_FTX_H

#include <hip/hip_runtime.h>
#include <limits.h>

namespace fphip
{

struct DD
{
  double hi, lo;
};

// ---- error-free transformations ---------------------------------------------------------------
__device__ __forceinline__ DD two_sum(double a, double b)
{
  const double s = a + b, bb = s - a;
  return DD{s, (a - (s - bb)) + (b - bb)};
}
__device__ __forceinline__ DD quick_two_sum(double a, double b)
{
  const double s = a + b;
  return DD{s, b - (s - a)};
}
__device__ __forceinline__ DD two_prod(double a, double b)
{
  const double p = a * b;
  return DD{p, __fma_rn(a, b, -p)};
}

// ---- one interface for both types (overloads) ---------------------------------------------------
__device__ __forceinline__ double f_from(double, double v) { return v; }
__device__ __forceinline__ DD f_from(DD, double v) { return DD{v, 0.0}; }
__device__ __forceinline__ double f_hi(double a) { return a; }
__device__ __forceinline__ double f_hi(DD a) { return a.hi; }

__device__ __forceinline__ double f_add(double a, double b) { return a + b; }
__device__ __forceinline__ DD f_add(DD a, DD b)
{  // sloppy_add
  DD s = two_sum(a.hi, b.hi);
  s.lo += (a.lo + b.lo);
  return quick_two_sum(s.hi, s.lo);
}
__device__ __forceinline__ double f_neg(double a) { return -a; }
__device__ __forceinline__ DD f_neg(DD a) { return DD{-a.hi, -a.lo}; }
__device__ __forceinline__ double f_sub(double a, double b) { return a - b; }
__device__ __forceinline__ DD f_sub(DD a, DD b) { return f_add(a, f_neg(b)); }
__device__ __forceinline__ double f_mul(double a, double b) { return a * b; }
__device__ __forceinline__ DD f_mul(DD a, DD b)
{
  DD p = two_prod(a.hi, b.hi);
  p.lo += (a.hi * b.lo + a.lo * b.hi);
  return quick_two_sum(p.hi, p.lo);
}
__device__ __forceinline__ DD f_mul_d(DD a, double b)
{
  DD p = two_prod(a.hi, b);
  p.lo += a.lo * b;
  return quick_two_sum(p.hi, p.lo);
}
__device__ __forceinline__ double f_mul_d(double a, double b) { return a * b; }
__device__ __forceinline__ double f_div(double a, double b) { return a / b; }
__device__ __forceinline__ DD f_div(DD a, DD b)
{  // accurate_div
  double q1 = a.hi / b.hi;
  DD r      = f_sub(a, f_mul_d(b, q1));
  double q2 = r.hi / b.hi;
  r         = f_sub(r, f_mul_d(b, q2));
  double q3 = r.hi / b.hi;
  DD q      = quick_two_sum(q1, q2);
  return f_add(q, DD{q3, 0.0});
}
__device__ __forceinline__ double f_sqrt(double a) { return sqrt(a); }
__device__ __forceinline__ DD f_sqrt(DD a)
{  // Karp: sqrt(a) = a*x + [a - (a*x)^2] * x / 2 with x = 1/sqrt(a) in double
  if (a.hi == 0.0)
    return DD{0.0, 0.0};
  const double x = 1.0 / sqrt(a.hi), ax = a.hi * x;
  const DD sq    = two_prod(ax, ax);
  const DD diff  = f_sub(a, sq);
  return two_sum(ax, diff.hi * (x * 0.5));
}
__device__ __forceinline__ double f_abs(double a) { return fabs(a); }
__device__ __forceinline__ DD f_abs(DD a) { return (a.hi < 0.0) ? f_neg(a) : a; }
__device__ __forceinline__ double f_ldexp(double a, int e) { return ldexp(a, e); }
__device__ __forceinline__ DD f_ldexp(DD a, int e) { return DD{ldexp(a.hi, e), ldexp(a.lo, e)}; }
__device__ __forceinline__ bool f_is_zero(double a) { return a == 0.0; }
__device__ __forceinline__ bool f_is_zero(DD a) { return a.hi == 0.0; }
__device__ __forceinline__ bool f_lt0(double a) { return a < 0.0; }
__device__ __forceinline__ bool f_lt0(DD a) { return a.hi < 0.0; }
__device__ __forceinline__ bool f_le(double a, double b) { return a <= b; }
__device__ __forceinline__ bool f_le(DD a, DD b) { return a.hi < b.hi || (a.hi == b.hi && a.lo <= b.lo); }
__device__ __forceinline__ bool f_gt(double a, double b) { return a > b; }
__device__ __forceinline__ bool f_gt(DD a, DD b) { return a.hi > b.hi || (a.hi == b.hi && a.lo > b.lo); }
__device__ __forceinline__ bool f_eq_d(double a, double v) { return a == v; }
__device__ __forceinline__ bool f_eq_d(DD a, double v) { return a.hi == v && a.lo == 0.0; }
__device__ __forceinline__ bool f_finite(double a) { return isfinite(a); }
__device__ __forceinline__ bool f_finite(DD a) { return isfinite(a.hi) && isfinite(a.lo); }

// FP_NR::exponent(): ilogb(to_double) + 1 (nr_FP_d.inl:44, nr_FP_dd.inl:55)
__device__ __forceinline__ long long f_exponent(double x)
{
  return (x == 0.0) ? ((long long)INT_MIN + 1) : ((long long)ilogb(x) + 1);
}
__device__ __forceinline__ long long f_exponent(DD x) { return f_exponent(x.hi); }

// nint
__device__ __forceinline__ double f_nint(double a) { return rint(a); }
// libqd's scalar nint: halves go UP (floor(d + 0.5)), not to even
__device__ __forceinline__ double qd_nint(double d) { return (d == floor(d)) ? d : floor(d + 0.5); }
__device__ __forceinline__ DD f_nint(DD a)
{
  double hi = qd_nint(a.hi), lo = 0.0;
  if (hi == a.hi)
  {  // the high word is an integer already: round the low word
    lo         = qd_nint(a.lo);
    const DD r = quick_two_sum(hi, lo);
    return r;
  }
  if (fabs(hi - a.hi) == 0.5 && a.lo < 0.0)
    hi -= 1.0;  // a tie of the high word that the low word breaks downwards
  return DD{hi, lo};
}
// rnd_we, nr_FP_d.inl:226-233 / nr_FP_dd.inl:234-241
template <class FT> __device__ __forceinline__ FT f_rnd_we(FT b, int e)
{
  if (f_exponent(b) + e >= 53)
    return b;
  return f_ldexp(f_nint(f_ldexp(b, e)), -e);
}
// get_si_exp_we with expo == 0 (the caller has checked): (long) of the scaled value; the reference
// truncates the HIGH word of a dd_real only (nr_FP_dd.inl:63)
__device__ __forceinline__ long long f_to_long(double a, int e) { return (long long)ldexp(a, e); }
__device__ __forceinline__ long long f_to_long(DD a, int e) { return (long long)ldexp(a.hi, e); }

// an exactly converted 64-bit integer
__device__ __forceinline__ double f_from_ll(double, long long v) { return (double)v; }
__device__ __forceinline__ DD f_from_ll(DD, long long v)
{
  const double hi = (double)v;  // round to nearest; |v| < 2^63
  // the remainder is exact in 64-bit arithmetic whenever hi is representable as a long long
  double lo = 0.0;
  if (fabs(hi) < 9.2e18)
    lo = (double)(v - (long long)hi);
  return DD{hi, lo};
}

// wave-level helpers
__device__ __forceinline__ double f_shfl_xor(double v, int m) { return __shfl_xor(v, m); }
__device__ __forceinline__ DD f_shfl_xor(DD v, int m) { return DD{__shfl_xor(v.hi, m), __shfl_xor(v.lo, m)}; }
__device__ __forceinline__ double f_bcast(double v, int lane) { return __shfl(v, lane); }
__device__ __forceinline__ DD f_bcast(DD v, int lane) { return DD{__shfl(v.hi, lane), __shfl(v.lo, lane)}; }
__device__ __forceinline__ double f_shfl_up(double v, int d) { return __shfl_up(v, d); }
__device__ __forceinline__ DD f_shfl_up(DD v, int d) { return DD{__shfl_up(v.hi, d), __shfl_up(v.lo, d)}; }
template <class FT> __device__ __forceinline__ FT f_wave_sum(FT v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    v = f_add(v, f_shfl_xor(v, off));
  return v;
}

}  // namespace fphip
#endif
