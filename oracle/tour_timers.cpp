/* Test / measurement infrastructure ONLY (not product code; never loaded by fplll_amd/).
 *
 * libtour_timers.so — LD_PRELOADed by bench.py's `bkz60_tour` leg in front of the REAL reference
 * (oracle/_ref/libfplll.so): wall-clock timers around three members of the reference, each an explicit
 * specialisation that forwards to the reference's own definition (dlsym RTLD_NEXT) — the reference's code is
 * neither changed nor restated:
 *
 *   LLLReduction<Z_NR<long>,FP_NR<double>>::lll                (lll.h:54, lll.cpp:44-164)       host LLL
 *   EnumerationDyn<Z_NR<long>,FP_NR<double>>::enumerate        (enumerate.h:42, enumerate.cpp:58-159)
 *                                                              fplll's own enumerator: every block the
 *                                                              external enumerator declined (or none is set)
 *   ExternalEnumeration<Z_NR<long>,FP_NR<double>>::enumerate   (enumerate_ext.h:111, enumerate_ext.cpp:48-167)
 *                                                              the plugin hook incl. fplll's marshalling;
 *                                                              false = declined
 *
 * ref_driver's `bkztour` command looks `tour_timers_get` up (dlsym RTLD_DEFAULT) and, when this library is
 * preloaded, prints the split of the tour's wall time.  None of the three nests in another (lll() runs no
 * enumeration; the enumerators run no LLL), so the sums are disjoint shares of the tour. */
#include <chrono>
#include <dlfcn.h>
#include <vector>

#include <fplll.h>

namespace
{
struct Timers
{
  double lll_s = 0, enum_cpu_s = 0, ext_ok_s = 0, ext_declined_s = 0;
  unsigned long long lll_calls = 0, enum_cpu_calls = 0, ext_ok_calls = 0, ext_declined_calls = 0;
} T;
inline double now()
{
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

extern "C" void tour_timers_get(double *out8)
{
  out8[0] = T.lll_s;
  out8[1] = (double)T.lll_calls;
  out8[2] = T.enum_cpu_s;
  out8[3] = (double)T.enum_cpu_calls;
  out8[4] = T.ext_ok_s;
  out8[5] = (double)T.ext_ok_calls;
  out8[6] = T.ext_declined_s;
  out8[7] = (double)T.ext_declined_calls;
}

extern "C" void tour_timers_reset(void) { T = Timers(); }

FPLLL_BEGIN_NAMESPACE

typedef Z_NR<long> ZL;
typedef FP_NR<double> FD;

template <> bool LLLReduction<ZL, FD>::lll(int kappa_min, int kappa_start, int kappa_end, int size_reduction_start)
{
  typedef bool (*fn_t)(LLLReduction<ZL, FD> *, int, int, int, int);
  static fn_t next = (fn_t)dlsym(RTLD_NEXT, "_ZN5fplll12LLLReductionINS_4Z_NRIlEENS_5FP_NRIdEEE3lllEiiii");
  const double t0 = now();
  const bool r    = next(this, kappa_min, kappa_start, kappa_end, size_reduction_start);
  T.lll_s += now() - t0;
  T.lll_calls++;
  return r;
}

template <>
void EnumerationDyn<ZL, FD>::enumerate(int first, int last, FD &fmaxdist, long fmaxdistexpo,
                                       const vector<FD> &target_coord, const vector<enumxt> &subtree,
                                       const vector<enumf> &pruning, bool dual, bool subtree_reset)
{
  typedef void (*fn_t)(EnumerationDyn<ZL, FD> *, int, int, FD &, long, const vector<FD> &, const vector<enumxt> &,
                       const vector<enumf> &, bool, bool);
  static fn_t next = (fn_t)dlsym(
      RTLD_NEXT,
      "_ZN5fplll14EnumerationDynINS_4Z_NRIlEENS_5FP_NRIdEEE9enumerateEiiRS4_lRKSt6vectorIS4_SaIS4_EERKS7_IdSaIdEESF_bb");
  const double t0 = now();
  next(this, first, last, fmaxdist, fmaxdistexpo, target_coord, subtree, pruning, dual, subtree_reset);
  T.enum_cpu_s += now() - t0;
  T.enum_cpu_calls++;
}

template <>
bool ExternalEnumeration<ZL, FD>::enumerate(int first, int last, FD &fmaxdist, long fmaxdistexpo,
                                            const vector<enumf> &pruning, bool dual)
{
  typedef bool (*fn_t)(ExternalEnumeration<ZL, FD> *, int, int, FD &, long, const vector<enumf> &, bool);
  static fn_t next = (fn_t)dlsym(
      RTLD_NEXT, "_ZN5fplll19ExternalEnumerationINS_4Z_NRIlEENS_5FP_NRIdEEE9enumerateEiiRS4_lRKSt6vectorIdSaIdEEb");
  const double t0 = now();
  const bool r    = next(this, first, last, fmaxdist, fmaxdistexpo, pruning, dual);
  const double dt = now() - t0;
  if (r)
  {
    T.ext_ok_s += dt;
    T.ext_ok_calls++;
  }
  else
  {
    T.ext_declined_s += dt;
    T.ext_declined_calls++;
  }
  return r;
}

FPLLL_END_NAMESPACE
