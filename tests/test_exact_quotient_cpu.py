"""The host volume loop divides by the small integers 1 … 256 without the divider where the CPU has FMA
(fplll_amd/csrc/pruner_volume.hip, "The division": q0 = RN(c r), e = c − q0 d by one FMA, q = RN(q0 + e r)) and
claims the divider's double for 2^-900 ≤ |c| ≤ 2^900.  The header proves it; this test hammers the same three
operations (compiled here with gcc, contraction off) against c / d: random significands over the whole exponent
range, and the dividends that sit closest to a rounding boundary — RN(q d) and its neighbours for random q, for q
with long runs of ones, for q next to a power of two."""
import os
import subprocess

import pytest

SRC = r"""
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static uint64_t st = 0x9e3779b97f4a7c15ull;
static uint64_t rnd(void) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; }
static double from_bits(uint64_t b) { double x; memcpy(&x, &b, 8); return x; }
static uint64_t bits(double x) { uint64_t b; memcpy(&b, &x, 8); return b; }
static long bad = 0, total = 0;
static void check(double c, int d, double r)
{
  const double a = fabs(c);
  if (!(a >= 0x1p-900 && a <= 0x1p900)) return;
  const double dd = (double)d;
  const double q0 = c * r;
  const double e = __builtin_fma(-q0, dd, c);
  const double q = __builtin_fma(e, r, q0);
  const double want = c / dd;
  ++total;
  if (bits(q) != bits(want) && bad++ < 5)
    printf("MISMATCH c=%a d=%d got %a want %a\n", c, d, q, want);
}
int main(int argc, char **argv)
{
  const long n = argc > 1 ? atol(argv[1]) : 200000;
  for (int d = 1; d <= 256; ++d)
  {
    const double r = 1.0 / (double)d;
    for (long i = 0; i < n; ++i)
    {
      uint64_t u = rnd();
      // random significand, exponent anywhere in the proven range, either sign
      const int ex = (int)(rnd() % 1801) - 900;
      double c = ldexp(1.0 + (double)(u >> 12) * 0x1p-52, ex);
      if (u & 1) c = -c;
      check(c, d, r);
      // dividends next to a rounding boundary of the quotient: q d rounded, and its neighbours
      double q = 1.0 + (double)(rnd() >> 12) * 0x1p-52;
      switch (i & 3)
      {
      case 1: q = from_bits(bits(q) | ((1ull << (rnd() % 52)) - 1)); break;         // a run of ones at the end
      case 2: q = 2.0 - (double)(rnd() % 64) * 0x1p-52; break;                      // just below a power of two
      case 3: q = 1.0 + (double)(rnd() % 64) * 0x1p-52; break;                      // just above
      }
      q = ldexp(q, (int)(rnd() % 1601) - 800);
      const double p = q * (double)d;
      check(p, d, r);
      check(nextafter(p, INFINITY), d, r);
      check(nextafter(p, -INFINITY), d, r);
      check(-p, d, r);
    }
  }
  printf("checked %ld mismatches %ld\n", total, bad);
  return bad != 0;
}
"""


def test_fma_quotient_by_small_integers_is_the_dividers(tmp_path):
    with open("/proc/cpuinfo") as fh:
        if " fma " not in fh.read():
            pytest.skip("no FMA on this CPU: the host loop uses the divider here")
    src = tmp_path / "q.c"
    exe = tmp_path / "q"
    src.write_text("#include <stdlib.h>\n" + SRC)
    subprocess.run(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-o", str(exe), str(src), "-lm"], check=True)
    r = subprocess.run([str(exe), "400000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "mismatches 0" in r.stdout, r.stdout[-500:]
    assert int(r.stdout.split()[1]) > 5e8
