/*
 * dropin_driver.cpp — TEST INFRASTRUCTURE ONLY.
 *
 * Runs the reference's UNMODIFIED LLLReduction / BKZReduction (oracle/_ref/libfplll.so, compiled
 * from /root/reference as is) against the device-backed Gram-Schmidt object of the product
 * (fplll_hip::MatGSOHip, fplll_amd/lib/libfplll_hip_gso.so — linked AHEAD of libfplll.so so that
 * its LLLReduction<Z_NR<long>,FP_NR<double>>::lll is the one bkz.cpp's calls resolve to), exactly
 * the way bkz_reduction_f does it for the host object (fplll/bkz.cpp:812-836).
 *
 *   dropin_driver bkz <basisfile> <beta> hip|cpu [max_loops] [plugin.so]
 *   dropin_driver lll <basisfile> hip|cpu
 *   dropin_driver sizered <basisfile> hip|cpu     LLLReduction::size_reduction(0, d) (lll.h:107-122; with
 *                                                 FPLLL_HIP_BABAI=1 every babai() of it runs on the device)
 *   dropin_driver hlll <basisfile> hip|cpu        HLLLReduction::hlll() on MatHouseholder(Hip), the way
 *                                                 hlll_reduction_zf does it (wrapper.cpp:790-806)
 * prints one JSON line: status, seconds, device calls / seconds, the output basis (DROPIN_U=1: the objects are
 * built with u = identity and the line carries u_out as well).
 */
#include <fplll/fplll.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <dlfcn.h>
#include <fstream>
#include <iostream>
#include <memory>

#include "../fplll_amd/csrc/dropin/matgso_hip.h"
#include "../fplll_amd/csrc/dropin/mathouseholder_hip.h"

using namespace fplll;

typedef std::array<uint64_t, FPLLL_EXTENUM_MAX_EXTENUM_DIM>(extenum_fn)(
    const int, double, std::function<extenum_cb_set_config>, std::function<extenum_cb_process_sol>,
    std::function<extenum_cb_process_subsol>, bool, bool);

int main(int argc, char **argv)
{
  if (argc < 4)
  {
    fprintf(stderr, "usage: dropin_driver bkz basisfile beta hip|cpu [max_loops] [plugin.so] | lll|sizered|hlll basisfile hip|cpu\n");
    return 2;
  }
  const std::string cmd = argv[1];
  ZZ_mat<mpz_t> A;
  {
    std::ifstream is(argv[2]);
    is >> A;
  }
  if (A.get_rows() == 0)
  {
    fprintf(stderr, "cannot read %s\n", argv[2]);
    return 2;
  }
  const bool is_bkz   = (cmd == "bkz");
  const int beta      = is_bkz ? atoi(argv[3]) : 0;
  const std::string w = is_bkz ? argv[4] : argv[3];
  const int max_loops = (is_bkz && argc > 5) ? atoi(argv[5]) : 0;
  if (is_bkz && argc > 6 && strcmp(argv[6], "none") != 0)
  {
    void *hnd = dlopen(argv[6], RTLD_NOW | RTLD_GLOBAL);
    extenum_fn *fn = hnd ? (extenum_fn *)dlsym(hnd, "fplll_hip_extenum") : nullptr;
    if (!fn)
    {
      fprintf(stderr, "cannot load plugin %s: %s\n", argv[6], dlerror());
      return 2;
    }
    set_external_enumerator(fn);
  }
  else
    set_external_enumerator(nullptr);  // fplll's own enumerator (fplll counting rule)

  ZZ_mat<long> bl, ul, ul_inv;
  if (!convert<long, mpz_t>(bl, A, 10))
  {
    fprintf(stderr, "basis does not fit long\n");
    return 2;
  }
  typedef Z_NR<long> ZT;
  typedef FP_NR<double> FT;
  if (cmd == "hlll")
  {
    // hlll_reduction_zf<long, double> with LM_FAST: MatHouseholder(ROW_EXPO) + HLLLReduction::hlll
    // (wrapper.cpp:790-806) — with the device-backed object in place of the host one
    std::unique_ptr<MatHouseholder<ZT, FT>> mh;
    fplll_hip::MatHouseholderHip *hh = nullptr;
    if (w == "hip")
    {
      hh = new fplll_hip::MatHouseholderHip(bl, ul, ul_inv, HOUSEHOLDER_ROW_EXPO);
      mh.reset(hh);
      if (!hh->on_device())
      {
        fprintf(stderr, "MatHouseholderHip has no device: %s\n", hh->last_error());
        return 3;
      }
    }
    else
      mh.reset(new MatHouseholder<ZT, FT>(bl, ul, ul_inv, HOUSEHOLDER_ROW_EXPO));
    HLLLReduction<ZT, FT> hlll_obj(*mh, LLL_DEF_DELTA, LLL_DEF_ETA, HLLL_DEF_THETA, HLLL_DEF_C, LLL_DEFAULT);
    auto t0 = std::chrono::steady_clock::now();
    hlll_obj.hlll();
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // the host members after the call: R(i,i) of the object, as a later host caller would read them
    double rsum = 0.0;
    for (int i = 0; i < bl.get_rows(); ++i)
    {
      FT f;
      long e;
      mh->get_R(f, i, i, e);
      rsum += std::log(std::fabs(f.get_d())) + e * std::log(2.0);
    }
    printf("{\"what\":\"hlll\",\"gso\":\"%s\",\"status\":%d,\"seconds\":%.3f,\"nodes\":0,\"n_swaps\":%ld,"
           "\"device_calls\":%ld,\"device_seconds\":%.3f,\"log_abs_det_R\":%.12g,\"d\":%d,\"n\":%d,\"b_out\":[",
           w.c_str(), hlll_obj.get_status(), secs, hh ? hh->n_swaps : -1L, hh ? hh->n_device_calls : 0L,
           hh ? hh->device_seconds : 0.0, rsum, bl.get_rows(), bl.get_cols());
    for (int i = 0; i < bl.get_rows(); ++i)
      for (int j = 0; j < bl.get_cols(); ++j)
        printf("%s%ld", (i || j) ? "," : "", bl(i, j).get_si());
    printf("]}\n");
    return 0;
  }
  if (getenv("DROPIN_U"))  // bkz_reduction(b, u, ...): MatGSO(b, u = identity, ...) keeps the transformation matrix
    ul.gen_identity(bl.get_rows());
  std::unique_ptr<MatGSO<ZT, FT>> gso;
  fplll_hip::MatGSOHip *hip = nullptr;
  if (w == "hip")
  {
    hip = new fplll_hip::MatGSOHip(bl, ul, ul_inv, GSO_ROW_EXPO);
    gso.reset(hip);
    if (!hip->on_device())
    {
      fprintf(stderr, "MatGSOHip has no device: %s\n", hip->last_error());
      return 3;
    }
  }
  else
    gso.reset(new MatGSO<ZT, FT>(bl, ul, ul_inv, GSO_ROW_EXPO));

  // lll basisfile hip|cpu [siegel|earlyred]: the LLL variants the device does not offer — the interposed
  // lll() must hand them to the reference's own loop (no device call), on the same object
  int lll_flags = LLL_DEFAULT;
  if (!is_bkz && argc > 4)
    lll_flags = strcmp(argv[4], "siegel") == 0 ? LLL_SIEGEL : (strcmp(argv[4], "earlyred") == 0 ? LLL_EARLY_RED : LLL_DEFAULT);
  LLLReduction<ZT, FT> lll_obj(*gso, LLL_DEF_DELTA, LLL_DEF_ETA, lll_flags);
  int status = 0;
  long nodes = 0;
  auto t0    = std::chrono::steady_clock::now();
  if (is_bkz)
  {
    vector<Strategy> strategies;
    BKZParam param(beta, strategies);
    if (max_loops > 0)
    {
      param.flags |= BKZ_MAX_LOOPS;
      param.max_loops = max_loops;
    }
    BKZReduction<ZT, FT> bkz_obj(*gso, lll_obj, param);
    bkz_obj.bkz();
    status = bkz_obj.status;
    nodes  = bkz_obj.nodes;
  }
  else if (cmd == "sizered")
  {
    lll_obj.size_reduction(0, bl.get_rows());
    status = lll_obj.status;
  }
  else
  {
    lll_obj.lll();
    status = lll_obj.status;
  }
  const double secs =
      std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("{\"what\":\"%s\",\"gso\":\"%s\",\"status\":%d,\"seconds\":%.3f,\"nodes\":%ld,\"n_swaps\":%d,"
         "\"device_calls\":%ld,\"device_seconds\":%.3f,\"kernel_seconds\":%.3f,\"session_starts\":%ld,"
         "\"dirty_rows\":%ld,\"d\":%d,\"n\":%d,\"b_out\":[",
         cmd.c_str(), w.c_str(), status, secs, nodes, lll_obj.n_swaps, hip ? hip->n_device_calls : 0L,
         hip ? hip->device_seconds : 0.0, hip ? hip->kernel_seconds : 0.0, hip ? hip->n_session_starts : 0L,
         hip ? hip->n_dirty_rows : 0L, bl.get_rows(), bl.get_cols());
  for (int i = 0; i < bl.get_rows(); ++i)
    for (int j = 0; j < bl.get_cols(); ++j)
      printf("%s%ld", (i || j) ? "," : "", bl(i, j).get_si());
  printf("]");
  if (ul.get_rows() > 0)
  {
    printf(",\"u_out\":[");
    for (int i = 0; i < ul.get_rows(); ++i)
      for (int j = 0; j < ul.get_cols(); ++j)
        printf("%s%ld", (i || j) ? "," : "", ul(i, j).get_si());
    printf("]");
  }
  printf("}\n");
  return 0;
}
