// matgso_hip.h — the GSO half of the drop-in: fplll's own MatGSO class with a device behind it.
//
// fplll has no hook for the Gram-Schmidt object: LLLReduction / BKZReduction take a
// MatGSOInterface<ZT,FT>& and read mu / r through inline accessors of its HOST matrices
// (fplll/gso_interface.h:675-732).  Two pieces make the reference's UNMODIFIED drivers run on the
// device nevertheless:
//
//  1. fplll_hip::MatGSOHip — a MatGSO<Z_NR<long>, FP_NR<double>> (fplll/gso.h:33; the types BKZ
//     runs on, bkz.cpp:816-829) that owns a device-resident copy (C ABI fphip_gso_*, batch of one)
//     and offers the reference's coarse operations on it: update_gso(), size_reduction(range),
//     lll(range).  After each of them the host members (b, bf, mu, r, row_expo, gso_valid_cols, the
//     Gram cache) are brought back in line, so every inline accessor of the reference keeps
//     working — "mirror host mu/r at call boundaries".
//  2. an explicit specialisation of LLLReduction<Z_NR<long>, FP_NR<double>>::lll (fplll/lll.h:54,
//     body fplll/lll.cpp:44-164) in libfplll_hip_gso.so: when the MatGSO it was constructed with
//     is a MatGSOHip, the whole LLL loop runs on the device (fphip_gso_lll: bit-identical basis,
//     swap count, status); otherwise the call is forwarded to the reference's own definition.
//     The library only has to come before libfplll.so in the link order (or LD_PRELOAD): bkz.cpp
//     reaches lll() through the PLT, so BKZReduction::svp_preprocessing / svp_reduction / bkz()
//     (bkz.cpp:100-124, 274-358, 522-672) — compiled from the reference as is — drive the device.
//
// The enumeration half is the run-time plugin (fplll_hip_extenum, extenum_shim.cpp).
#ifndef FPLLL_HIP_MATGSO_HIP_H
#define FPLLL_HIP_MATGSO_HIP_H

#include <fplll/fplll.h>

#include "../../../include/fplll_hip.h"

namespace fplll_hip
{

class MatGSOHip : public fplll::MatGSO<fplll::Z_NR<long>, fplll::FP_NR<double>>
{
public:
  typedef fplll::Z_NR<long> ZT;
  typedef fplll::FP_NR<double> FT;

  // Same arguments as MatGSO's constructor (gso.h:56-75).  flags must be GSO_ROW_EXPO (what BKZ
  // uses).  A non-empty u (enable_transform) is kept on the device too — the LLL kernel applies its row
  // operations to u's rows as well —; u_inv_t must be empty (an object with one stays a plain MatGSO).
  // device < 0: the device of FPLLL_HIP_DEVICE or 0.
  MatGSOHip(fplll::Matrix<ZT> &arg_b, fplll::Matrix<ZT> &arg_u, fplll::Matrix<ZT> &arg_uinv_t, int flags,
            int device = -1);
  ~MatGSOHip();

  bool on_device() const { return g_ != nullptr; }
  // the matrix still has the shape the device object was created for (svp_postprocessing_generic works on d + 1
  // rows for a moment, bkz.cpp:186-219: calls made meanwhile stay on the host)
  bool shape_matches() const { return b.get_rows() == rows0_ && b.get_cols() == cols0_; }
  // the resident session (one LLLReduction object as far as LLL_EARLY_RED's last_early_red is concerned)
  bool session_active() const { return session_; }
  void end_session() { session_ = false; }
  const char *last_error() const;

  // MatGSOInterface::update_gso() (gso_interface.h:767-775) on the device
  bool update_gso_device();
  // LLLReduction::size_reduction(kappa_min, kappa_end) (lll.h:107-122) on the device; eta as in the
  // LLLReduction object.  Returns the device status: 1 ok, 0 GSO failure, -1 babai failure,
  // -2 multiplier beyond 63 bits (nothing was changed; fall back to the host path)
  int size_reduction_device(int kappa_min, int kappa_end, double eta);
  // LLLReduction(m, delta, eta, flags: LLL_DEFAULT or LLL_SIEGEL).lll(kappa_min, kappa_start, kappa_end, 0) on the
  // device.  info[4]: final_kappa, n_swaps, zeros, loop iterations.
  int lll_device(int kappa_min, int kappa_start, int kappa_end, double delta, double eta, int info[4], int flags = 0);

  // statistics: device calls made through this object and the seconds spent in them
  long n_device_calls = 0;
  double device_seconds = 0.0;

  // FPLLL_HIP_RESIDENT=0 makes every lll_device() call stateless again (upload, fresh GSO, download: the A/B)
  bool resident() const { return resident_; }
  long n_session_starts = 0, n_dirty_rows = 0;
  double kernel_seconds = 0.0;  // of the session calls: the LLL kernel alone

private:
  void upload_basis();
  void mirror_from_device(bool basis_changed);
  int lll_device_resident(int kappa_min, int kappa_start, int kappa_end, double delta, double eta, int info[4], int flags);
  void mirror_from_session();

  // Resident session (fphip_gso_session_lll): the device keeps this object's state between lll() calls, like the
  // reference's MatGSO does on the host (gso_interface.h:675-732, gso_interface.cpp:26-53).  hb_ is then the
  // basis as of the last synchronisation: rows that differ from it at the next call are the host's row
  // operations since (insertions, rerandomisation, row moves of svp_postprocessing) and go up as such.
  bool resident_ = true;
  bool session_  = false;
  std::vector<int> hvc_, dpos_;
  std::vector<int64_t> drows_, hb2_;

  fphip_ctx *ctx_ = nullptr;
  fphip_gso *g_   = nullptr;
  bool own_ctx_   = false;
  std::vector<int64_t> hb_;      // staging: integer basis
  std::vector<double> hmu_, hr_; // staging: mu, r
  std::vector<int64_t> hexp_;
  int rows0_ = 0, cols0_ = 0;
  bool track_u_ = false;         // enable_transform: u lives on the device as well
  std::vector<int64_t> hu_, hu2_; // staging: u as of the last synchronisation / as read back
  void read_back_u(bool from_session);
  // FPLLL_HIP_CHECK_U=1 (diagnostics): u b_0 == b is verified around every device call (b_0: the basis at the first upload)
  bool check_u_ = false;
  std::vector<int64_t> b0_;
  long n_calls_checked_ = 0;
  bool check_u_invariant(const char *when);
};

}  // namespace fplll_hip
#endif
