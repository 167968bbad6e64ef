// gso_sweep2.h — launch geometry of gso_sweep2_kernel, shared by the kernel (gso_sweep2.hip) and
// its host launcher (gso_host.hip).
#ifndef FPHIP_GSO_SWEEP2_H
#define FPHIP_GSO_SWEEP2_H

#include "gso_device.h"

namespace fphip
{
namespace s2
{
template <int NQ> struct Cfg
{
  // registers: NQ <= 3 fits 128 VGPRs (4 waves per SIMD, 16 per CU); NQ = 4 takes 3 waves per SIMD
  // (FPHIP_S2_NQ3_WPS / FPHIP_S2_NQ3_LDS: build-time A/B of the NQ = 3 geometry, tests/perf/runs/r4g.sh)
#ifndef FPHIP_S2_NQ3_WPS
#define FPHIP_S2_NQ3_WPS 4
#endif
#ifndef FPHIP_S2_NQ3_LDS
#define FPHIP_S2_NQ3_LDS 9984
#endif
  static constexpr int WAVES_PER_SIMD = (NQ == 4 ? 3 : NQ == 3 ? FPHIP_S2_NQ3_WPS : 4);
  static constexpr int WAVE_LDS       = (NQ == 4 ? 13312 : NQ == 3 ? FPHIP_S2_NQ3_LDS : 9984);  // LDS bytes per wave
  static constexpr int ESZ            = NQ * 256;  // bytes per ring entry (64*NQ elements of 4 bytes)
  static constexpr int NPAIR = (WAVE_LDS / (2 * ESZ)) > 16 ? 16 : (WAVE_LDS / (2 * ESZ));
  static constexpr int INFL  = 2 * (NPAIR - 1);  // entries in flight; one pair slot is always free
  static constexpr int RING  = NPAIR * 2 * ESZ;  // bytes of LDS per wave
  static_assert(NPAIR >= 3, "ring too small");
  static_assert(INFL - 2 <= 63, "vmcnt is a 6-bit counter");
};

template <int NQ>
__global__ void gso_sweep2_kernel(GsoBatch P, double *muA, short *m16, int *flag16, int kmin, int kend, double eta,
                                  int mode);
}  // namespace s2
}  // namespace fphip
#endif
