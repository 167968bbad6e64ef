// Test infrastructure: fplll_amd/csrc/ftx.h (the double-double / quad-double arithmetic of the extended-precision
// kernels) compiled FOR THE HOST, so that the CPU suite can check it against multiprecision without a GPU
// (tests/test_ftx_cpu.py).  The header's arithmetic is plain C++; only the wave-level helpers need the device.
//   stdin:  lines "op a0 a1 a2 a3 b0 b1 b2 b3" (hex doubles), op: 0 add, 1 sub, 2 mul, 3 div, 4 sqrt(a), 5 nint(a),
//           6 mul by the double b0; 10..15: the same ops in double-double on (a0,a1), (b0,b1)
//   stdout: one line of hex doubles per input line
#include <cmath>
#include <cstdio>
#include <cstdlib>
#define FPHIP_FTX_HOST_TEST 1
#define __device__
#define __forceinline__ inline
static inline double __fma_rn(double a, double b, double c) { return std::fma(a, b, c); }
template <class T> static inline T __shfl_xor(T v, int) { return v; }
template <class T> static inline T __shfl(T v, int) { return v; }
template <class T> static inline T __shfl_up(T v, int) { return v; }
using std::floor;
using std::ilogb;
using std::isfinite;
using std::ldexp;
using std::sqrt;
using std::fabs;
#include "../../fplll_amd/csrc/ftx.h"

int main()
{
  using namespace fphip;
  int op;
  double a[4], b[4];
  while (scanf("%d %la %la %la %la %la %la %la %la", &op, &a[0], &a[1], &a[2], &a[3], &b[0], &b[1], &b[2], &b[3]) == 9)
  {
    if (op >= 10)
    {
      const DD x{a[0], a[1]}, y{b[0], b[1]};
      DD r{0, 0};
      switch (op - 10)
      {
      case 0: r = f_add(x, y); break;
      case 1: r = f_sub(x, y); break;
      case 2: r = f_mul(x, y); break;
      case 3: r = f_div(x, y); break;
      case 4: r = f_sqrt(x); break;
      default: r = f_nint(x); break;
      }
      printf("%a %a\n", r.hi, r.lo);
      continue;
    }
    const QD x{{a[0], a[1], a[2], a[3]}}, y{{b[0], b[1], b[2], b[3]}};
    QD r{{0, 0, 0, 0}};
    switch (op)
    {
    case 0: r = f_add(x, y); break;
    case 1: r = f_sub(x, y); break;
    case 2: r = f_mul(x, y); break;
    case 3: r = f_div(x, y); break;
    case 4: r = f_sqrt(x); break;
    case 5: r = f_nint(x); break;
    default: r = f_mul_d(x, b[0]); break;
    }
    printf("%a %a %a %a\n", r.x[0], r.x[1], r.x[2], r.x[3]);
  }
  return 0;
}
