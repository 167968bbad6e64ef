// enum_wave.h — wave64 helpers shared by the enumeration kernels (enum_kernel.hip, enum_walk.hip): lane
// broadcasts (v_readlane / ds_bpermute), scalar-cache loads of the per-level (r, pruning) pairs, lane masks built
// on the scalar unit, and the empty asm statements that keep the hot loops regions of uniform branches.
#ifndef FPHIP_ENUM_WAVE_H
#define FPHIP_ENUM_WAVE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "enum_device.h"

namespace fphip
{

__device__ __forceinline__ double rl_f64(double v, int lane)
{
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int rl_i32(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
// The same value through the LDS crossbar (ds_bpermute_b32: every lane reads lane `addr4 / 4`; no LDS
// memory involved): v_readlane_b32 occupies the vector ALU for TWO issue slots (measured:
// tests/perf/micro/valu_rates.hip), and the walk is bound by exactly that port while the LDS port idles.
// The result sits in a VGPR (wave-uniform in value, lane-varying for the compiler): fine where it feeds
// vector arithmetic; conditions derived from it go through a ballot.
__device__ __forceinline__ int bp_i32(int v, int addr4) { return __builtin_amdgcn_ds_bpermute(addr4, v); }
__device__ __forceinline__ double bp_f64(double v, int addr4)
{
  return __hiloint2double(bp_i32(__double2hiint(v), addr4), bp_i32(__double2loint(v), addr4));
}
// 4 * level in a VGPR: the address operand of the bpermutes of one level
__device__ __forceinline__ int lane_addr(int level)
{
  int a = level << 2;
  asm("" : "+v"(a));
  return a;
}
// element `off8 / 8` of the row at the wave-uniform pointer `row`: scalar base + 32-bit lane offset
// (the global_load saddr form / one v_add for LDS) instead of a 64-bit address per lane
__device__ __forceinline__ double ld_off(const double *row, unsigned off8)
{
  return *(const double *)((const char *)row + off8);
}
// The row at byte offset `rowoff` of DevShared::mu_sq, element `lane`, as a raw buffer load: the
// row offset rides in the instruction's scalar offset operand, the lane offset is a loop-invariant
// register — no address arithmetic on the vector unit (a global_load needs one v_add per row, the
// packed rows three).  Reads beyond the table return zero.
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mu_rsrc(const double *tab, unsigned bytes)
{
  return __builtin_amdgcn_make_buffer_rsrc((void *)tab, 0, bytes, 0x00020000);  // (untyped 32-bit data)
}
__device__ __forceinline__ double ld_row(__amdgpu_buffer_rsrc_t tab, unsigned rowoff, unsigned lane8)
{
  const v2u v = __builtin_amdgcn_raw_buffer_load_b64(tab, lane8, rowoff, 0);
  return __hiloint2double((int)v.y, (int)v.x);
}
// (r_ii, pruning_i) of one level straight into SGPRs through the scalar cache: the table is read-only
// for the whole enumeration, the index is wave-uniform.  Issue early (rp_issue), wait right before
// the first use (rp_wait: s_waitcnt through the value, so that the compiler keeps the order).
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4i rp_issue(const double *tab, int level)
{
  v4i q;
  asm volatile("s_load_dwordx4 %0, %1, %2" : "=s"(q) : "s"(tab), "s"(level << 4));
  return q;
}
#ifdef FPHIP_RP_PLAIN
// (experiment: the same pair through a load the compiler sees — its own s_load and its own waits)
__device__ __forceinline__ v4i rp_issue2(const double *tab, unsigned off)
{
  const v4i *p = (const v4i *)((const char *)tab + off);
  v4i q        = *p;
  q.x = __builtin_amdgcn_readfirstlane(q.x);
  q.y = __builtin_amdgcn_readfirstlane(q.y);
  q.z = __builtin_amdgcn_readfirstlane(q.z);
  q.w = __builtin_amdgcn_readfirstlane(q.w);
  return q;
}
__device__ __forceinline__ void rp_wait(v4i &q) { asm volatile("" : "+s"(q)); }
#else
__device__ __forceinline__ v4i rp_issue2(const double *tab, unsigned off)
{  // (off = byte offset of the level's row in DevShared::mu_sq; tab points at the pair of row 0)
  v4i q;
  asm volatile("s_load_dwordx4 %0, %1, %2" : "=s"(q) : "s"(tab), "s"(off));
  return q;
}
__device__ __forceinline__ void rp_wait(v4i &q) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(q)); }
#endif
__device__ __forceinline__ double rp_r(const v4i &q) { return __hiloint2double(q.y, q.x); }
__device__ __forceinline__ double rp_p(const v4i &q) { return __hiloint2double(q.w, q.z); }
// v_writelane_b32: the wave-uniform `val` (an SGPR) into lane `lane` (an SGPR) of `old`; one VALU
// instruction where a select needs a v_mov of the scalar plus a v_cndmask
extern "C" __device__ int fphip_llvm_writelane(int, int, int) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ int wl_i32(int val, int lane, int old)
{
  return fphip_llvm_writelane(__builtin_amdgcn_readfirstlane(val), __builtin_amdgcn_readfirstlane(lane), old);
}
__device__ __forceinline__ double wl_f64(double val, int lane, double old)
{
  const int lo = wl_i32(__builtin_amdgcn_readfirstlane(__double2loint(val)), lane, __double2loint(old));
  const int hi = wl_i32(__builtin_amdgcn_readfirstlane(__double2hiint(val)), lane, __double2hiint(old));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ unsigned long long rfl_u64(unsigned long long v)
{
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
// The lane mask of ONE level built on the scalar unit (s_lshl_b64) and consumed as the select / carry
// operand of VOP3 instructions: `lane == k ? a : b` without the v_cmp, `cnt += (lane == k)` as one
// v_addc_co (the compiler's form is v_cmp + v_cndmask 0/1 + v_add).  The walk is VALU-issue bound.
__device__ __forceinline__ unsigned long long lane_bit(int k) { return 1ull << (k & 63); }
__device__ __forceinline__ int sel_i32(unsigned long long m, int a, int b)
{
  int r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
  return r;
}
__device__ __forceinline__ double sel_f64(unsigned long long m, double a, double b)
{
  return __hiloint2double(sel_i32(m, __double2hiint(a), __double2hiint(b)),
                          sel_i32(m, __double2loint(a), __double2loint(b)));
}
__device__ __forceinline__ unsigned add_bit(unsigned long long m, unsigned c)
{
  unsigned long long co;
  asm("v_addc_co_u32_e64 %0, %1, 0, %0, %2" : "+v"(c), "=s"(co) : "s"(m));
  return c;
}
// The lane number behind an opaque asm statement: an address built from it stays where it is used.  (Per-lane
// 64-bit addresses of the per-TASK and per-EVENT accesses — in.x + lane, out.col + lane, the ring record — are
// loop invariants the optimiser hoists to the top of the kernel, where they stay alive across the walk loops and are
// spilled: five register pairs = 40 bytes of scratch per lane at the kernel's 64-VGPR pin.)
__device__ __forceinline__ int here_lane(int lane)
{
  asm volatile("" : "+v"(lane));
  return lane;
}

__device__ __forceinline__ int tri_off(int k) { return (k * (k - 1)) >> 1; }  // slot k starts here
__device__ __forceinline__ unsigned tri8(int k)
{  // 8 * tri_off(k) on the scalar unit, opaque to the optimiser (which otherwise folds the sign of a
   // subtraction into the product: four instructions instead of three)
  unsigned t = (unsigned)(k * (k - 1)) << 2;
  asm("" : "+s"(t));
  return t;
}

__device__ __forceinline__ unsigned long long load_sys_u64(const unsigned long long *p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ONE lane's atomic without a lane-masked branch the compiler can see: EXEC is narrowed to lane 0 and widened again
// inside one asm statement (the callers sit in wave-uniform control flow: EXEC is full there).  The level registers
// of the walks (lane = level: x, centre, partial distance, sub-solution bound, node counters) are read ACROSS lanes
// — v_readlane, ds_bpermute — and the register allocator knows nothing of that: inside an `if (lane == 0)` region it
// may split the live range of such a register with a v_mov that executes for lane 0 only, and the other 63 levels
// are garbage behind the region.  A precaution, not a repair: no such copy was found in the shipped objects, and
// taking every lane-masked branch out of the subtree walk did not cure the run-to-run differences of its
// sub-solution variant with the split stack — but the hazard is real for any later build.
// Returns the pre-op value in lane 0 (the other lanes: undefined — read it with rfl_u64).
__device__ __forceinline__ unsigned long long lane0_atomic_umin_u64(unsigned long long *p, unsigned long long v)
{
  unsigned long long old;
  asm volatile("s_mov_b64 exec, 1\n\t"
               "global_atomic_umin_x2 %0, %1, %2, off sc0\n\t"
               "s_waitcnt vmcnt(0)\n\t"
               "s_mov_b64 exec, -1"
               : "=&v"(old)
               : "v"(p), "v"(v)
               : "memory");
  return old;
}
__device__ __forceinline__ unsigned long long lane0_atomic_add_u64(unsigned long long *p, unsigned long long v)
{
  unsigned long long old;
  asm volatile("s_mov_b64 exec, 1\n\t"
               "global_atomic_add_x2 %0, %1, %2, off sc0\n\t"
               "s_waitcnt vmcnt(0)\n\t"
               "s_mov_b64 exec, -1"
               : "=&v"(old)
               : "v"(p), "v"(v)
               : "memory");
  return old;
}

__device__ __forceinline__ unsigned lane0_atomic_add_u32(unsigned *p, unsigned v)
{
  unsigned old;
  asm volatile("s_mov_b64 exec, 1\n\t"
               "global_atomic_add %0, %1, %2, off sc0\n\t"
               "s_waitcnt vmcnt(0)\n\t"
               "s_mov_b64 exec, -1"
               : "=&v"(old)
               : "v"(p), "v"(v)
               : "memory");
  return old;
}
// (no value back: fire and forget)
__device__ __forceinline__ void lane0_atomic_umin_u64_noret(unsigned long long *p, unsigned long long v)
{
  asm volatile("s_mov_b64 exec, 1\n\t"
               "global_atomic_umin_x2 %0, %1, off\n\t"
               "s_mov_b64 exec, -1"
               :
               : "v"(p), "v"(v)
               : "memory");
}
__device__ __forceinline__ void lane0_atomic_or_u32_noret(unsigned *p, unsigned v)
{
  asm volatile("s_mov_b64 exec, 1\n\t"
               "global_atomic_or %0, %1, off\n\t"
               "s_mov_b64 exec, -1"
               :
               : "v"(p), "v"(v)
               : "memory");
}

// An empty statement the optimiser can neither delete nor merge: it keeps the join block of a
// lane-masked `if` separate from the joins of the UNIFORM branches around it.  Without it the CFG
// simplifier merges them, the uniformity analysis then sees "a phi at a divergent join", calls the
// walk loops' exit conditions divergent, and the structuriser rewrites them with exit codes in
// VGPRs and a copy of every level register per iteration.
#define FPHIP_JOIN() asm volatile("")
// The same on each `break` of the CHILD chain: the optimiser otherwise merges the exit tests into a
// flag computed with scalar selects.  (Not in the STEP loop: there the separate exit blocks make the
// code generator unify the exits through a hub that keeps the requested (par, mk) and the old pair
// alive side by side — a copy of each, behind a wait for the loads, in the loop latch.)
#define FPHIP_EXIT() asm volatile("")
// keeps a wave-uniform double in a VGPR pair: it is the second scalar operand of a VALU instruction
// whose first one already sits in SGPRs (one constant-bus read per instruction on gfx9).  The
// compiler treats the result as lane-varying: conditions computed from it go through a ballot.
#define FPHIP_IN_VGPR(v) asm volatile("" : "+v"(v))
// Hides a wave-uniform value from the optimiser (it stays in its SGPR): placed on the event code
// right behind a hot loop it keeps all the loop's exits on ONE successor block, so that the loop
// is a single-entry single-exit region of uniform branches the CFG structuriser leaves alone.
#define FPHIP_OPAQUE(v)                          \
  do                                             \
  {                                              \
    v = __builtin_amdgcn_readfirstlane(v);       \
    asm volatile("" : "+s"(v));                  \
  } while (0)

}  // namespace fphip
#endif
