// mathouseholder_hip.h — the Householder half of the drop-in: fplll's own MatHouseholder class with
// a device behind it, and HLLLReduction::hlll interposed on it.
//
// MatHouseholder<ZT, FT> (fplll/householder.h:38) has no virtual member and no hook, and
// HLLLReduction (fplll/hlll.h:27) holds a plain reference to it.  As for the Gram-Schmidt object
// (matgso_hip.h) two pieces let the reference's UNMODIFIED driver run on the device:
//
//  1. fplll_hip::MatHouseholderHip — a MatHouseholder<Z_NR<long>, FP_NR<double>> (the types
//     hlll_reduction_zf<long, double> runs on, wrapper.cpp:790-806) that owns a device-resident copy
//     (C ABI fphip_hh_*, batch of one) and offers the coarse operations update_R_device() and
//     hlll_device().  After each of them the host members (b, bf, R, V, sigma, row_expo, the norms)
//     are brought back in line through the reference's own refresh_R_bf() / update_R()
//     (householder.cpp:27-245): the values a host run leaves for that basis.
//  2. an explicit specialisation of HLLLReduction<Z_NR<long>, FP_NR<double>>::hlll (declared
//     fplll/hlll.h:54, body fplll/hlll.cpp:26-173) in libfplll_hip_gso.so: when the MatHouseholder
//     the object was built on is a MatHouseholderHip (there is no RTTI on that class: the objects
//     register themselves), the whole HLLL loop runs on the device (fphip_hh_hlll: bit-identical
//     basis, status); otherwise the call goes to the reference's definition (dlsym RTLD_NEXT).
#ifndef FPLLL_HIP_MATHOUSEHOLDER_HIP_H
#define FPLLL_HIP_MATHOUSEHOLDER_HIP_H

#include <fplll/fplll.h>

#include "../../../include/fplll_hip.h"

namespace fplll_hip
{

class MatHouseholderHip : public fplll::MatHouseholder<fplll::Z_NR<long>, fplll::FP_NR<double>>
{
public:
  typedef fplll::Z_NR<long> ZT;
  typedef fplll::FP_NR<double> FT;

  // Same arguments as MatHouseholder's constructor (householder.h:70-145).  The device takes
  // flags == HOUSEHOLDER_ROW_EXPO or 0 without transformation matrices (pass empty ones); any other
  // configuration leaves the object a plain MatHouseholder.  device < 0: FPLLL_HIP_DEVICE or 0.
  MatHouseholderHip(fplll::Matrix<ZT> &arg_b, fplll::Matrix<ZT> &arg_u, fplll::Matrix<ZT> &arg_uinv_t,
                    int flags, int device = -1);
  ~MatHouseholderHip();

  bool on_device() const { return h_ != nullptr; }
  const char *last_error() const;

  // the MatHouseholderHip behind a MatHouseholder reference, or null (HLLLReduction only keeps a
  // reference to the base class, which has no virtual member)
  static MatHouseholderHip *lookup(const fplll::MatHouseholder<ZT, FT> *m);

  // refresh_R_bf() + update_R() over all rows (householder.h:532-536, 610-614) on the device; the
  // host members hold the same R afterwards.  false: a device error (the host path is intact).
  bool update_R_device();
  // HLLLReduction(m, delta, eta, theta, c, LLL_DEFAULT).hlll() on the device.  Returns the device
  // status: 1 success, -4 / -5 the precision alarms (RED_HLLL_SR_FAILURE / RED_HLLL_NORM_FAILURE; the
  // basis is the one the reference stops on), -2 a multiplier beyond 63 bits and -100 a device
  // error (nothing was changed: the host path takes over).  info[2]: swaps, loop iterations.
  int hlll_device(double delta, double eta, double theta, double c, int info[2]);

  long n_device_calls   = 0;
  double device_seconds = 0.0;
  long n_swaps          = 0;  // of the last hlll_device()

private:
  void upload_basis();
  void mirror_from_device(bool basis_changed);

  fplll::Matrix<ZT> &bref_;  // MatHouseholder keeps its reference private
  fphip_ctx *ctx_ = nullptr;
  fphip_hh *h_    = nullptr;
  std::vector<int64_t> hb_;
};

}  // namespace fplll_hip
#endif
