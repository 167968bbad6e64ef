// mathouseholder_hip.cpp — see mathouseholder_hip.h.  Host C++ only (g++, the reference's headers, the
// C ABI of libfplll_hip.so); part of fplll_amd/lib/libfplll_hip_gso.so.
#include "mathouseholder_hip.h"

#include <chrono>
#include <cstdlib>
#include <dlfcn.h>
#include <mutex>
#include <set>

using namespace fplll;

namespace fplll_hip
{

typedef Z_NR<long> ZT;
typedef FP_NR<double> FT;

namespace
{
std::mutex g_reg_mutex;
std::set<const MatHouseholder<ZT, FT> *> g_registry;
double now_s()
{
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

MatHouseholderHip::MatHouseholderHip(Matrix<ZT> &arg_b, Matrix<ZT> &arg_u, Matrix<ZT> &arg_uinv_t, int flags,
                                     int device)
    : MatHouseholder<ZT, FT>(arg_b, arg_u, arg_uinv_t, flags), bref_(arg_b)
{
  if ((flags & ~HOUSEHOLDER_ROW_EXPO) != 0 || arg_u.get_rows() != 0 || arg_uinv_t.get_rows() != 0)
    return;
  if (device < 0)
    device = getenv("FPLLL_HIP_DEVICE") ? atoi(getenv("FPLLL_HIP_DEVICE")) : 0;
  if (fphip_create(device, &ctx_) != FPHIP_OK)
  {
    if (ctx_)
      fphip_destroy(ctx_);
    ctx_ = nullptr;
    return;
  }
  if (fphip_hh_create(ctx_, 1, arg_b.get_rows(), arg_b.get_cols(), (flags & HOUSEHOLDER_ROW_EXPO) ? 1 : 0, &h_) !=
      FPHIP_OK)
    h_ = nullptr;  // e.g. more than 256 columns: the object works as a MatHouseholder
  hb_.resize((size_t)arg_b.get_rows() * arg_b.get_cols());
  if (h_)
  {
    std::lock_guard<std::mutex> lk(g_reg_mutex);
    g_registry.insert(static_cast<const MatHouseholder<ZT, FT> *>(this));
  }
}

MatHouseholderHip::~MatHouseholderHip()
{
  {
    std::lock_guard<std::mutex> lk(g_reg_mutex);
    g_registry.erase(static_cast<const MatHouseholder<ZT, FT> *>(this));
  }
  if (h_)
    fphip_hh_destroy(h_);
  if (ctx_)
    fphip_destroy(ctx_);
}

MatHouseholderHip *MatHouseholderHip::lookup(const MatHouseholder<ZT, FT> *m)
{
  std::lock_guard<std::mutex> lk(g_reg_mutex);
  if (g_registry.count(m) == 0)
    return nullptr;
  // (registered objects ARE MatHouseholderHip: the downcast of a non-polymorphic base is static)
  return const_cast<MatHouseholderHip *>(static_cast<const MatHouseholderHip *>(m));
}

const char *MatHouseholderHip::last_error() const { return ctx_ ? fphip_last_error(ctx_) : "no device context"; }

void MatHouseholderHip::upload_basis()
{
  const int d = bref_.get_rows(), n = bref_.get_cols();
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < n; ++j)
      hb_[(size_t)i * n + j] = bref_(i, j).get_si();
  fphip_hh_set_basis(h_, 0, 1, hb_.data());
}

// Device state -> host members: the integer basis is the device's; bf, row exponents, norms, R, V and
// sigma are then recomputed by the reference's own refresh_R_bf() / update_R() from it — the
// deterministic values a host run holds for that basis (the device's R factor is bit-identical:
// tests/test_hh_gpu.py; it is not copied because V and sigma have no getter on either side).
void MatHouseholderHip::mirror_from_device(bool basis_changed)
{
  const int d = bref_.get_rows(), n = bref_.get_cols();
  if (basis_changed)
  {
    fphip_hh_get_basis(h_, 0, 1, hb_.data());
    for (int i = 0; i < d; ++i)
      for (int j = 0; j < n; ++j)
        bref_(i, j) = (long)hb_[(size_t)i * n + j];
  }
  // (refresh_R_bf only converts the columns below n_known_cols — the widest row SHAPE of the input
  // discovered so far, householder.cpp:193, captured at construction; the reduced rows are mixtures of
  // all input rows, so every row counts as discovered first: one pass raises n_known_cols to its
  // final value, what the host loop reaches when it has visited the last row)
  invalidate_row(0);
  for (int i = 0; i < d; ++i)
    refresh_R_bf(i);
  for (int i = 0; i < d; ++i)
  {
    refresh_R_bf(i);
    update_R(i);
  }
}

bool MatHouseholderHip::update_R_device()
{
  if (!h_)
    return false;
  const double t0 = now_s();
  upload_basis();
  int st       = 0;
  const int rc = fphip_hh_update_R(h_, &st);
  if (rc == FPHIP_OK && st == 1)
    mirror_from_device(false);
  device_seconds += now_s() - t0;
  ++n_device_calls;
  return rc == FPHIP_OK && st == 1;
}

int MatHouseholderHip::hlll_device(double delta, double eta, double theta, double c, int info[2])
{
  if (!h_)
    return -100;
  const double t0 = now_s();
  upload_basis();
  int st = 0, inf[2] = {0, 0};
  const int rc = fphip_hh_hlll(h_, delta, eta, theta, c, &st, inf);
  if (rc != FPHIP_OK)
    st = -100;
  else if (st == 1 || st == -4 || st == -5)  // a finished run or a precision alarm: the device's basis is the result
    mirror_from_device(true);
  // (-2 multiplier beyond 63 bits, -6 iteration cap, a device error: the host objects stay as they were —
  //  the interposed hlll() hands the UNCHANGED input to the reference's own loop)
  if (info)
  {
    info[0] = inf[0];
    info[1] = inf[1];
  }
  n_swaps = (st == 1 || st == -4 || st == -5) ? inf[0] : 0;
  device_seconds += now_s() - t0;
  ++n_device_calls;
  return st;
}

}  // namespace fplll_hip

// ---------------------------------------------------------------------------------------------
// HLLLReduction<Z_NR<long>, FP_NR<double>>::hlll — explicit specialisation of the member the
// reference declares in fplll/hlll.h:54 and defines in fplll/hlll.cpp:26-173.  With this library
// ahead of libfplll.so in the symbol search order every call of hlll() on these types lands here.
// ---------------------------------------------------------------------------------------------
FPLLL_BEGIN_NAMESPACE

typedef bool (*hlll_member_fn)(HLLLReduction<Z_NR<long>, FP_NR<double>> *);

template <> bool HLLLReduction<Z_NR<long>, FP_NR<double>>::hlll()
{
  fplll_hip::MatHouseholderHip *h = fplll_hip::MatHouseholderHip::lookup(&m);
  if (h && h->on_device() && !verbose)
  {
    if (m.get_d() < 2)  // (the reference's loop starts at k = 1: nothing to do below two rows)
      h = nullptr;
  }
  if (h && h->on_device() && !verbose)
  {
    int info[2]  = {0, 0};
    const int st = h->hlll_device(delta.get_d(), eta.get_d(), theta.get_d(), c.get_d(), info);
    if (st == 1 || st == -4 || st == -5)
    {
      // dR / eR of every row, as the host loop leaves them (hlll.h:148-159)
      for (int i = 0; i < m.get_d(); ++i)
      {
        compute_dR(i);
        compute_eR(i);
      }
      return set_status(st == 1 ? RED_SUCCESS : (st == -4 ? RED_HLLL_SR_FAILURE : RED_HLLL_NORM_FAILURE));
    }
    // a multiplier beyond 63 bits or a device error: nothing was changed, the host path takes over
  }
  static hlll_member_fn next =
      (hlll_member_fn)dlsym(RTLD_NEXT, "_ZN5fplll13HLLLReductionINS_4Z_NRIlEENS_5FP_NRIdEEE4hlllEv");
  if (!next)
  {
    status = RED_HLLL_FAILURE;
    return false;
  }
  return next(this);
}

FPLLL_END_NAMESPACE
