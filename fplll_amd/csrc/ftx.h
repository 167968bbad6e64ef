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

#ifndef FPHIP_FTX_HOST_TEST  // (tests/native/ftx_host.cpp compiles this header for the host: the arithmetic is plain C++)
#include <hip/hip_runtime.h>
#endif
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

// ---- QD = quad-double (an unevaluated sum of four doubles, ~212 bits): the device stand-in for the reference's
// FP_NR<qd_real> (fplll/nr/nr_FP_qd.inl over libqd's qd_real), the third stage of the wrapper's ladder
// (wrapper.cpp:630-710).  Restated from Hida-Li-Bailey (2007) in libqd's default ("sloppy") configuration: addition
// by component-wise two_sum with a three_sum carry chain, multiplication with the O(eps^3) terms accumulated in
// plain doubles, division by four quotient corrections, square root by Newton's iteration on 1 / sqrt(a), nint by
// parts; every result goes through the five-term renormalisation.  Like DD: parity with libqd is UNPINNED bit for
// bit; the arithmetic is checked against mpmath at quad-double accuracy (tests/test_dd_gpu.py).
struct QD
{
  double x[4];
};

__device__ __forceinline__ void qd_three_sum(double &a, double &b, double &c)
{
  DD t  = two_sum(a, b);
  DD u  = two_sum(c, t.hi);
  a     = u.hi;
  DD v  = two_sum(t.lo, u.lo);
  b     = v.hi;
  c     = v.lo;
}
__device__ __forceinline__ void qd_three_sum2(double &a, double &b, double c)
{
  DD t = two_sum(a, b);
  DD u = two_sum(c, t.hi);
  a    = u.hi;
  b    = t.lo + u.lo;
}
// five doubles of decreasing magnitude (roughly) -> four non-overlapping ones
__device__ __forceinline__ QD qd_renorm(double c0, double c1, double c2, double c3, double c4)
{
  // one pass of quick_two_sum from the bottom gathers the sum at the top ...
  DD t = quick_two_sum(c3, c4);
  double s = t.hi;
  c4 = t.lo;
  t  = quick_two_sum(c2, s);
  s  = t.hi;
  c3 = t.lo;
  t  = quick_two_sum(c1, s);
  s  = t.hi;
  c2 = t.lo;
  t  = quick_two_sum(c0, s);
  c0 = t.hi;
  c1 = t.lo;
  // ... and one from the top pushes the errors down, skipping zeros (branch-free form: a zero term simply leaves
  // the running sum where it is; a final pass closes the gaps a skipped zero could have left)
  double r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0;
  double acc = c0;
  int k      = 0;
#define FPHIP_QD_STEP(ci)                      \
  {                                            \
    const DD u    = quick_two_sum(acc, (ci));  \
    const bool nz = u.lo != 0.0;               \
    r0            = (nz && k == 0) ? u.hi : r0; \
    r1            = (nz && k == 1) ? u.hi : r1; \
    r2            = (nz && k == 2) ? u.hi : r2; \
    r3            = (nz && k == 3) ? u.hi : r3; \
    acc           = nz ? u.lo : u.hi;          \
    k += nz ? 1 : 0;                           \
  }
  FPHIP_QD_STEP(c1)
  FPHIP_QD_STEP(c2)
  FPHIP_QD_STEP(c3)
  FPHIP_QD_STEP(c4)
#undef FPHIP_QD_STEP
  r0 = (k == 0) ? acc : r0;
  r1 = (k == 1) ? acc : r1;
  r2 = (k == 2) ? acc : r2;
  r3 = (k == 3) ? acc : r3;
  return QD{{r0, r1, r2, r3}};
}

__device__ __forceinline__ QD f_from(QD, double v) { return QD{{v, 0.0, 0.0, 0.0}}; }
__device__ __forceinline__ double f_hi(QD a) { return a.x[0]; }
__device__ __forceinline__ QD f_neg(QD a) { return QD{{-a.x[0], -a.x[1], -a.x[2], -a.x[3]}}; }
__device__ __forceinline__ QD f_add(QD a, QD b)
{  // sloppy_add
  DD s0 = two_sum(a.x[0], b.x[0]), s1 = two_sum(a.x[1], b.x[1]), s2 = two_sum(a.x[2], b.x[2]),
     s3 = two_sum(a.x[3], b.x[3]);
  double t0 = s0.lo, t1 = s1.lo, t2 = s2.lo, t3 = s3.lo;
  DD u = two_sum(s1.hi, t0);
  double v1 = u.hi;
  t0        = u.lo;
  double v2 = s2.hi;
  qd_three_sum(v2, t0, t1);
  double v3 = s3.hi;
  qd_three_sum2(v3, t0, t2);
  t0 = t0 + t1 + t3;
  return qd_renorm(s0.hi, v1, v2, v3, t0);
}
__device__ __forceinline__ QD f_sub(QD a, QD b) { return f_add(a, f_neg(b)); }
__device__ __forceinline__ QD f_mul_d(QD a, double b)
{
  DD p0 = two_prod(a.x[0], b), p1 = two_prod(a.x[1], b), p2 = two_prod(a.x[2], b);
  double p3 = a.x[3] * b;
  double s0 = p0.hi;
  DD t      = two_sum(p0.lo, p1.hi);
  double s1 = t.hi, s2 = t.lo;
  double q1 = p1.lo, q2h = p2.hi;
  qd_three_sum(s2, q1, q2h);
  double q2 = p2.lo;
  qd_three_sum2(q1, q2, p3);
  const double s3 = q1, s4 = q2 + q2h;
  return qd_renorm(s0, s1, s2, s3, s4);
}
__device__ __forceinline__ QD f_mul(QD a, QD b)
{  // sloppy_mul
  DD p0 = two_prod(a.x[0], b.x[0]);
  DD p1 = two_prod(a.x[0], b.x[1]), p2 = two_prod(a.x[1], b.x[0]);
  DD p3 = two_prod(a.x[0], b.x[2]), p4 = two_prod(a.x[1], b.x[1]), p5 = two_prod(a.x[2], b.x[0]);
  double P1 = p1.hi, P2 = p2.hi, Q0 = p0.lo;
  qd_three_sum(P1, P2, Q0);
  double Q1 = p1.lo, Q2 = p2.lo;
  qd_three_sum(P2, Q1, Q2);
  double P3 = p3.hi, P4 = p4.hi, P5 = p5.hi;
  qd_three_sum(P3, P4, P5);
  DD s0 = two_sum(P2, P3), s1 = two_sum(Q1, P4);
  double s2 = Q2 + P5;
  DD u      = two_sum(s1.hi, s0.lo);
  s2 += (u.lo + s1.lo);
  const double s1v = u.hi + (a.x[0] * b.x[3] + a.x[1] * b.x[2] + a.x[2] * b.x[1] + a.x[3] * b.x[0] + Q0 + p3.lo +
                             p4.lo + p5.lo);
  return qd_renorm(p0.hi, P1, s0.hi, s1v, s2);
}
__device__ __forceinline__ QD f_div(QD a, QD b)
{
  const double q0 = a.x[0] / b.x[0];
  QD r            = f_sub(a, f_mul_d(b, q0));
  const double q1 = r.x[0] / b.x[0];
  r               = f_sub(r, f_mul_d(b, q1));
  const double q2 = r.x[0] / b.x[0];
  r               = f_sub(r, f_mul_d(b, q2));
  const double q3 = r.x[0] / b.x[0];
  r               = f_sub(r, f_mul_d(b, q3));
  const double q4 = r.x[0] / b.x[0];
  return qd_renorm(q0, q1, q2, q3, q4);
}
__device__ __forceinline__ QD f_ldexp(QD a, int e)
{
  return QD{{ldexp(a.x[0], e), ldexp(a.x[1], e), ldexp(a.x[2], e), ldexp(a.x[3], e)}};
}
__device__ __forceinline__ QD f_sqrt(QD a)
{  // Newton on x = 1 / sqrt(a): x <- x + x (1 - a x^2) / 2, three times from the double estimate; sqrt(a) = a x
  if (a.x[0] == 0.0)
    return QD{{0.0, 0.0, 0.0, 0.0}};
  QD x       = QD{{1.0 / sqrt(a.x[0]), 0.0, 0.0, 0.0}};
  const QD h = f_ldexp(a, -1);
  const QD half = QD{{0.5, 0.0, 0.0, 0.0}};
#pragma unroll
  for (int it = 0; it < 3; ++it)
    x = f_add(x, f_mul(x, f_sub(half, f_mul(h, f_mul(x, x)))));
  return f_mul(a, x);
}
__device__ __forceinline__ QD f_abs(QD a) { return (a.x[0] < 0.0) ? f_neg(a) : a; }
__device__ __forceinline__ bool f_is_zero(QD a) { return a.x[0] == 0.0; }
__device__ __forceinline__ bool f_lt0(QD a) { return a.x[0] < 0.0; }
__device__ __forceinline__ bool f_le(QD a, QD b)
{
  if (a.x[0] != b.x[0])
    return a.x[0] < b.x[0];
  if (a.x[1] != b.x[1])
    return a.x[1] < b.x[1];
  if (a.x[2] != b.x[2])
    return a.x[2] < b.x[2];
  return a.x[3] <= b.x[3];
}
__device__ __forceinline__ bool f_gt(QD a, QD b) { return !f_le(a, b); }
__device__ __forceinline__ bool f_eq_d(QD a, double v) { return a.x[0] == v && a.x[1] == 0.0 && a.x[2] == 0.0 && a.x[3] == 0.0; }
__device__ __forceinline__ bool f_finite(QD a) { return isfinite(a.x[0]) && isfinite(a.x[1]) && isfinite(a.x[2]) && isfinite(a.x[3]); }
__device__ __forceinline__ long long f_exponent(QD x) { return f_exponent(x.x[0]); }
__device__ __forceinline__ QD f_nint(QD a)
{  // by parts, like libqd's nint(qd_real): the first component that is not an integer decides
  double x0 = qd_nint(a.x[0]), x1 = 0.0, x2 = 0.0, x3 = 0.0;
  if (x0 == a.x[0])
  {
    x1 = qd_nint(a.x[1]);
    if (x1 == a.x[1])
    {
      x2 = qd_nint(a.x[2]);
      if (x2 == a.x[2])
        x3 = qd_nint(a.x[3]);
      else if (fabs(x2 - a.x[2]) == 0.5 && a.x[3] < 0.0)
        x2 -= 1.0;
    }
    else if (fabs(x1 - a.x[1]) == 0.5 && a.x[2] < 0.0)
      x1 -= 1.0;
  }
  else if (fabs(x0 - a.x[0]) == 0.5 && a.x[1] < 0.0)
    x0 -= 1.0;
  return qd_renorm(x0, x1, x2, x3, 0.0);
}
__device__ __forceinline__ long long f_to_long(QD a, int e) { return (long long)ldexp(a.x[0], e); }
__device__ __forceinline__ QD f_from_ll(QD, long long v)
{
  const DD t = f_from_ll(DD{}, v);
  return QD{{t.hi, t.lo, 0.0, 0.0}};
}

// wave-level helpers
__device__ __forceinline__ QD f_shfl_xor(QD v, int m)
{
  return QD{{__shfl_xor(v.x[0], m), __shfl_xor(v.x[1], m), __shfl_xor(v.x[2], m), __shfl_xor(v.x[3], m)}};
}
__device__ __forceinline__ QD f_bcast(QD v, int lane)
{
  return QD{{__shfl(v.x[0], lane), __shfl(v.x[1], lane), __shfl(v.x[2], lane), __shfl(v.x[3], lane)}};
}
__device__ __forceinline__ QD f_shfl_up(QD v, int d)
{
  return QD{{__shfl_up(v.x[0], d), __shfl_up(v.x[1], d), __shfl_up(v.x[2], d), __shfl_up(v.x[3], d)}};
}
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
