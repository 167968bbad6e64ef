// enum_kernel.hip — wavefront-per-subtree Schnorr-Euchner / KFP enumeration for gfx950 (CDNA4).
//
// Reference behaviour reproduced (fplll v5.5.0): the tree walk of
//   EnumerationBase::enumerate_recursive   fplll/enum/enumerate_base.cpp:24-118
// with the bound handling of EnumerationDyn::set_bounds/process_solution (enumerate.cpp:218-239)
// and the subtree split of enumlib (enum-parallel/enumeration.h:311-380, 412-505) as structural
// precedent.  This is not a translation of either: the data layout is built around one wave64.
//
// Design (MI355X-first)
// ---------------------
// * One wavefront walks one subtree depth-first.  Control flow (the level k, enter/step mode) is
//   wave-uniform, so there is no divergence; the 64 lanes are used as DATA lanes:
//     - lane l of the "level registers" xs/cs/pds/dxs/ddxs/cnt holds the value for tree level l
//       (x[l], center[l], partdist[l], dx[l], ddx[l], nodes[l]); a level is read with
//       v_readlane (uniform index in an SGPR) and written with a lane-masked select;
//     - lane i of the "row registers" holds row i of the centre partial sums
//       center_partsums[i][·] (enumerate_base.h:84).  Choosing x[k] updates ALL rows i<k with one
//       vector multiply + subtract:  S_k[i] = S_{k+1}[i] - x[k]*mu(k,i).  The reference's lazy
//       center_partsum_begin bookkeeping (enumerate_base.cpp:58-68) exists to avoid exactly this
//       O(k) work on a scalar CPU; here it is one VALU instruction pair, and every value is still
//       produced by the same operation sequence (j = d-1 … k, multiply then subtract, no FMA), so
//       centres, distances, node counts and solutions are bit-identical to the reference.
// * Backtracking needs S_{k+1} again when the next sibling x[k] is tried, so each wave keeps a
//   triangular stack of columns S_1..S_L in LDS (slot k holds k doubles, lane i touches only
//   row i → conflict-free ds_read_b64/ds_write_b64, no cross-lane traffic through LDS).
//   mu rows (row k = mu(k,0..k-1)) are staged once per workgroup in LDS, same triangular packing.
// * The tree is split level-wise into phases: a phase walks every input task (a subtree root at
//   level L) down to a stop level and emits each surviving node there as a task for the next
//   phase (root column S, partial distance, coefficient prefix).  The final phase walks to the
//   leaves.  Tasks are pulled from a device-wide atomic counter (persistent waves).
// * Solutions go to a ring in pinned host memory; the host thread runs the caller's callback
//   while the kernel is running and publishes the new bound through a pinned word that waves
//   poll (system-scope loads) — the same contract enumlib implements with a mutex and an
//   atomic<double> (enum-parallel/enumeration.h:66,286-299).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off  (no FMA contraction: fplll's
// arithmetic is separate multiply and add, nr/nr_FP_d.inl:178, baseline x86-64 build).

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
__device__ __forceinline__ unsigned long long rfl_u64(unsigned long long v)
{
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
  unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ int tri_off(int k) { return (k * (k - 1)) >> 1; }  // slot k starts here

__device__ __forceinline__ unsigned long long load_sys_u64(const unsigned long long *p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One enumeration launch.  Lmax = highest root level among the input tasks (a task rooted at
// level Lt walks levels < Lt), stop = level at which surviving nodes are emitted as tasks for the
// next launch (stop < 0: walk to the leaves).  budget > 0 enables work donation: when the task
// queue of this launch has run dry (idle waves exist) or a task has run `budget` iterations, the
// wave keeps only the subtree it is currently in and hands every remaining sibling subtree
// above it to the next launch (`donate` is the lowest level whose surviving nodes are emitted
// instead of descended into).  Emission never changes which nodes are visited or how they are
// counted, only which wave visits them.
//
// Loop structure.  A node "survives" when its distance passes the level bound (the reference's
// test, enumerate_base.cpp:31/93).  The walk alternates between two states:
//   CHILD(k, S, nd): a surviving node at level k with column S = S_k and distance nd is known.
//       Peek at its first child (centre S[k-1], rounded coefficient, distance): if the child fails
//       the node has no surviving children — stay at level k (this merges the reference's
//       "descend, test, return" :53-72/:31-32 into one step and is the common case in the bulk
//       of a pruned tree).  Otherwise emit the node as a task (k == stop / donation) or descend:
//       push S on the LDS stack, record level k-1 in the level registers, count the child, and
//       loop in CHILD with the child as the current node.
//   STEP(k): the subtree below the current coefficient x[k] is exhausted — advance x[k] in
//       zig-zag order (:80-89), test (:91-94): fail → STEP(k+1), survive → CHILD.
// DUAL: the dualenum instantiation of the recursion (enumerate_base.cpp:57-61, 103-105): the centre
// partial sums are driven by alpha = x - c instead of x; the inputs are then the transformed mu / r
// EnumerationDyn::enumerate builds for a dual call (enumerate.cpp:107-123).
template <bool MU_LDS, bool SUBS, bool DUAL>
__global__ void __launch_bounds__(FPHIP_MAX_BLOCK)
    enum_phase_kernel(DevShared *__restrict__ g, HostCtl *__restrict__ h, TaskBuf in, TaskBuf out,
                      int d, int Lmax, int stop, unsigned task_lo, unsigned task_hi,
                      const unsigned *__restrict__ idxlist, int launch_idx, int count_nodes,
                      unsigned budget, const double *__restrict__ xhi_root)
{
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int triL = (Lmax * (Lmax + 1)) >> 1;  // doubles for slots 1..Lmax
  // mu rows 1..Lmax-1 (packed like the stack slots).  MU_LDS: one copy per workgroup in LDS (lowest
  // latency: split launches and tails, where few waves run).  !MU_LDS: read straight from global
  // memory — <= 16 KB, read-only, L1-resident after the first touches — so that the LDS buys
  // more resident waves (the big walk launch, where throughput matters).
  const double *mu_s;
  double *stk;
  if constexpr (MU_LDS)
  {
    double *mu_l = smem;
    stk          = smem + triL + wave * triL;
    const int nmu = (Lmax * (Lmax - 1)) >> 1;
    for (int i = threadIdx.x; i < nmu; i += blockDim.x)
      mu_l[i] = g->mu_tri[i];
    __syncthreads();
    mu_s = mu_l;
  }
  else
  {
    mu_s = g->mu_tri;
    stk  = smem + wave * triL;
  }

  const double rd = g->rdiag[lane];
  const double pr = g->pruning[lane];
  // The bound lives in two places: the pinned host word the callback thread writes, and a
  // device-memory mirror.  Waves poll the mirror (L2) often and the host word (PCIe) rarely;
  // whoever sees a smaller host value lowers the mirror for everybody.
  unsigned long long mbits = rfl_u64(load_sys_u64(&h->bound_bits));
  double maxdist           = __longlong_as_double((long long)mbits);
  double bnd               = pr * maxdist;

  // level registers (lane = level) and counters
  double xs = 0.0, cs = 0.0, pds = 0.0;
  int dxs = 0, ddxs = 0;
  unsigned long long cnt = 0;
  unsigned iter          = 0;
  // findsubsols (enumerate_base.cpp:36-40): lane = level, this wave's view of the best sub-solution
  // distance per level (subsoldists); the device-wide value in g->sub_bits is authoritative
  double sb = SUBS ? __longlong_as_double((long long)g->sub_bits[lane]) : 0.0;

#define FPHIP_REFRESH_BOUND(from_host)                                                            \
  do                                                                                              \
  {                                                                                               \
    unsigned long long nb_;                                                                       \
    if (from_host)                                                                                \
    {                                                                                             \
      nb_ = rfl_u64(load_sys_u64(&h->bound_bits));                                                \
      if (nb_ < mbits && lane == 0)                                                               \
        atomicMin(&g->bound_bits, nb_);                                                           \
    }                                                                                             \
    else                                                                                          \
    {                                                                                             \
      nb_ = rfl_u64(__hip_atomic_load(&g->bound_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); \
    }                                                                                             \
    if (nb_ < mbits)                                                                              \
    {                                                                                             \
      mbits   = nb_;                                                                              \
      maxdist = __longlong_as_double((long long)mbits);                                           \
      bnd     = pr * maxdist;                                                                     \
    }                                                                                             \
  } while (0)

  for (;;)
  {
    // ---- pull a task ------------------------------------------------------------------------
    unsigned t = 0;
    if (lane == 0)
      t = atomicAdd(&g->task_head[launch_idx], 1u);
    t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
    const unsigned long long pos = (unsigned long long)task_lo + t;
    if (pos >= task_hi)
    {  // queue empty: tell the waves still walking to shed work for the next launch
      if (budget != 0u && lane == 0)
        __hip_atomic_store(&g->drain[launch_idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }

    // multi-GPU: this rank's share of the task list is an explicit index list (built on the host
    // from a content-sorted order, see enum_host.hip), walked heaviest-first
    const unsigned long long ti =
        idxlist ? (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)idxlist[pos]) : pos;
    const int Lt      = __builtin_amdgcn_readfirstlane(in.level[ti]);  // root level of this task
    const int rid     = __builtin_amdgcn_readfirstlane(in.root[ti]);   // level-64 ancestor (d > 64)
    const double xpre = in.x[ti * 64 + lane];                          // coefficients of levels >= Lt
    const double col0 = in.col[ti * 64 + lane];  // S_Lt rows (lane < Lt)
    const double pd0  = in.pd[ti];
    int donate        = 1 << 20;
    unsigned titer    = 0;
    FPHIP_REFRESH_BOUND((t & 63u) == 0u);

    // the task root is a surviving node at level Lt whose column and distance are given
    int k     = Lt;
    double S  = col0;  // S_k of the current node (rows < k valid)
    double nd = pd0;   // its distance

    // Reports a candidate (process_solution) and waits for the host's verdict, like enumlib's
    // mutex-protected process_sol (enumeration.h:286-299).
    auto report = [&](double dist)
    {
      unsigned long long idx = 0;
      if (lane == 0)
        idx = atomicAdd(&g->sol_head, 1ull);
      idx = rfl_u64(idx);
      for (unsigned spin = 0; idx >= load_sys_u64(&h->consumed) + FPHIP_RING_CAP; ++spin)
      {  // flow control against the host consumer
        __builtin_amdgcn_s_sleep(64);
        if (spin > (1u << 24))
        {
          if (lane == 0)
            atomicOr(&g->error_flags, FPHIP_ERR_RING_TIMEOUT);
          break;
        }
      }
      SolRec *r  = &h->ring[idx % FPHIP_RING_CAP];
      double xf  = (lane < Lt) ? xs : xpre;
      r->x[lane] = (lane < d) ? xf : 0.0;
      // levels 64..127: the coefficients chosen by the top walk, stored once per level-64 ancestor
      r->x[64 + lane] = (64 + lane < d) ? xhi_root[(size_t)rid * 64 + lane] : 0.0;
      if (lane == 0)
      {
        r->dist   = dist;
        r->kind   = 0;
        r->offset = 0;
      }
      __threadfence_system();
      if (lane == 0)
        __hip_atomic_store(&r->seq, idx + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      for (unsigned spin = 0; load_sys_u64(&h->consumed) <= idx; ++spin)
      {
        __builtin_amdgcn_s_sleep(32);
        if (spin > (1u << 24))
        {
          if (lane == 0)
            atomicOr(&g->error_flags, FPHIP_ERR_RING_TIMEOUT);
          break;
        }
      }
      FPHIP_REFRESH_BOUND(true);
    };

    // process_subsolution (enumerate.cpp:241-249): a node at level lvl is shorter than every
    // sub-solution seen at that level.  Only the wave that lowers the device-wide best reports;
    // no verdict to wait for (sub-solutions never change the radius), only ring space.
    auto sub_report = [&](int lvl, double dist)
    {
      unsigned long long old = 0;
      if (lane == 0)
        old = atomicMin(&g->sub_bits[lvl], (unsigned long long)__double_as_longlong(dist));
      old               = rfl_u64(old);
      const double oldd = __longlong_as_double((long long)old);
      sb                = (lane == lvl) ? fmin(oldd, dist) : sb;
      if (!(dist < oldd))
        return;
      unsigned long long idx = 0;
      if (lane == 0)
        idx = atomicAdd(&g->sol_head, 1ull);
      idx = rfl_u64(idx);
      for (unsigned spin = 0; idx >= load_sys_u64(&h->consumed) + FPHIP_RING_CAP; ++spin)
      {
        __builtin_amdgcn_s_sleep(64);
        if (spin > (1u << 24))
        {
          if (lane == 0)
            atomicOr(&g->error_flags, FPHIP_ERR_RING_TIMEOUT);
          break;
        }
      }
      SolRec *r       = &h->ring[idx % FPHIP_RING_CAP];
      const double xf = (lane < Lt) ? xs : xpre;
      r->x[lane]      = (lane < d && lane >= lvl) ? xf : 0.0;
      r->x[64 + lane] = (64 + lane < d) ? xhi_root[(size_t)rid * 64 + lane] : 0.0;
      if (lane == 0)
      {
        r->dist   = dist;
        r->kind   = 1;
        r->offset = lvl;
      }
      __threadfence_system();
      if (lane == 0)
        __hip_atomic_store(&r->seq, idx + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    };

    bool done = false;
    bool rep  = false;  // a candidate (level 0, nd > 0) is waiting to be reported
    bool at0  = false;  // ... and it was found by the STEP loop: resume there, not in CHILD
    while (!done)
    {
      // ================= CHILD chain: descend while the first child survives ====================
      // (state: a surviving, already counted node at level k with column S = S_k, distance nd)
      // Candidates are reported OUTSIDE the two hot loops (one copy of the slow path, no live
      // ranges of it inside them).
      if (!at0)
      for (;;)
      {
        k               = __builtin_amdgcn_readfirstlane(k);
        const int kc    = k - 1;
        // speculative load for the descending case: row kc of mu is needed right after the test
        // (its latency overlaps the test); clamped (valid, unused) address when kc == 0
        const double mk1 = mu_s[tri_off(max(kc, 1)) + min(lane, max(kc, 1) - 1)];
        const double c1  = rl_f64(S, kc);  // center[kk-1] = center_partsums[kk-1][kk]
        const double x1 = round(c1);      // roundto(): half away from zero, enumerate_base.h:33-34
        const double a1 = x1 - c1;
        const double n1 = nd + a1 * a1 * rl_f64(rd, kc);  // :28-29
        if (!(n1 <= rl_f64(bnd, kc)))
        {  // :31-32 no surviving child: next sibling at level k (the root has none: task done)
          done = k >= Lt;
          break;
        }
        if ((k == stop || k >= donate) && k < Lt)
        {  // hand the subtree below this node to the next launch
          unsigned oi = 0;
          if (lane == 0)
            oi = atomicAdd(out.count, 1u);
          oi = (unsigned)__builtin_amdgcn_readfirstlane((int)oi);
          if (oi < out.cap)
          {
            out.col[(unsigned long long)oi * 64 + lane] = S;
            const double xf                             = (lane < Lt) ? xs : xpre;
            out.x[(unsigned long long)oi * 64 + lane]   = xf;
            if (lane == 0)
            {
              out.pd[oi]    = nd;
              out.level[oi] = k;
              out.root[oi]  = rid;
            }
            break;  // → next sibling at level k
          }
          // buffer full: keep walking this subtree inline (results stay exact)
          if (lane == 0)
            atomicOr(&g->error_flags, FPHIP_FLAG_TASK_OVERFLOW);
        }
        // descend: level kc becomes the current level
        if (lane < k)
          stk[tri_off(k) + lane] = S;  // needed again when x[kc] steps to its next sibling
        {
          const int s1  = (c1 >= x1) ? 1 : -1;  // :71 / :114
          const bool me = lane == kc;
          cs            = me ? c1 : cs;
          xs            = me ? x1 : xs;
          pds           = me ? nd : pds;
          dxs           = me ? s1 : dxs;
          ddxs          = me ? s1 : ddxs;
          cnt += me ? 1ull : 0ull;  // ++nodes[kk-1]
        }
        if constexpr (SUBS)
        {
          if (n1 < rl_f64(sb, kc) && n1 != 0.0)
            sub_report(kc, n1);
        }
        k  = kc;
        nd = n1;
        if (k == 0)
        {
          rep = nd > 0.0;  // process_solution, :42-46
          break;           // level 0 has no children: next sibling
        }
        S = S - (DUAL ? a1 : x1) * mk1;  // S_k = S_{k+1} - x[k]*mu(k,·), :53-58 (k >= 1 here: mk1 is row k)
      }
      if (done)
        break;
      if (rep)
      {
        report(nd);
        rep = false;
      }
      at0 = false;
      // ================= STEP loop: next sibling at level k, climbing while they fail ===========
      for (;;)
      {
        k = __builtin_amdgcn_readfirstlane(k);
        ++titer;
        if (((++iter) & 63u) == 0u)
        {
          FPHIP_REFRESH_BOUND((iter & 16383u) == 0u);
          if (budget != 0u && titer >= 256u)
          {  // work donation: once the task queue has run dry (other waves are idle), or this task
             // exceeded its budget, keep only the subtree below the current level and emit every
             // sibling subtree above it as a task for the next launch
            const unsigned dr = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(
                &g->drain[launch_idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (dr != 0u || titer >= budget)
              donate = min(donate, k + 1);
          }
        }
        // speculative loads for the surviving case (LDS latency overlaps the test); lanes beyond
        // the row read a clamped (valid, unused) address so that no exec-mask branch is needed
        const double par = stk[tri_off(k + 1) + min(lane, k)];              // S_{k+1}
        const double mk  = mu_s[tri_off(k) + max(min(lane, k - 1), 0)];
        double xk        = rl_f64(xs, k);
        const double ck  = rl_f64(cs, k);
        const double pdk = rl_f64(pds, k);
        int dxk = rl_i32(dxs, k), ddxk = rl_i32(ddxs, k);
        if (pdk != 0.0)
        {  // :80-89 (is_svp is always true here)
          xk += (double)dxk;
          ddxk = -ddxk;
          dxk  = ddxk - dxk;
        }
        else
        {
          xk += 1.0;
        }
        const bool me = lane == k;
        xs            = me ? xk : xs;
        dxs           = me ? dxk : dxs;
        ddxs          = me ? ddxk : ddxs;
        const double a = xk - ck;
        nd             = pdk + a * a * rl_f64(rd, k);  // :91-92
        if (!(nd <= rl_f64(bnd, k)))
        {  // :93-94 → the parent steps to its next sibling
          ++k;
          if (k >= Lt)
          {
            done = true;
            break;
          }
          continue;
        }
        cnt += me ? 1ull : 0ull;  // ++nodes[kk]
        if constexpr (SUBS)
        {
          if (nd < rl_f64(sb, k) && nd != 0.0)
            sub_report(k, nd);
        }
        if (k == 0)
        {
          if (nd > 0.0)
          {  // :97-101: report outside the loop, then come back to the next sibling of level 0
            rep = true;
            at0 = true;
            break;
          }
          continue;
        }
        S = par - (DUAL ? a : xk) * mk;  // :104-110
        break;              // → CHILD chain
      }
    }
  }
#undef FPHIP_REFRESH_BOUND

  if (count_nodes && cnt != 0)
    atomicAdd(&g->nodes[lane], cnt);
  if (lane == 0)
    atomicAdd(&g->iters, (unsigned long long)iter);
}

#define FPHIP_INST(M, S, D)                                                                            \
  template __global__ void enum_phase_kernel<M, S, D>(DevShared *, HostCtl *, TaskBuf, TaskBuf, int,  \
                                                      int, int, unsigned, unsigned, const unsigned *,  \
                                                      int, int, unsigned, const double *);
FPHIP_INST(true, false, false)
FPHIP_INST(false, false, false)
FPHIP_INST(true, true, false)
FPHIP_INST(false, true, false)
FPHIP_INST(true, false, true)
FPHIP_INST(false, false, true)
#undef FPHIP_INST

// ---------------------------------------------------------------------------------------------
// Blocks larger than 64 (up to 128): the levels 64..d-1.  The TOP of the tree is walked with two
// registers per lane (rows / levels 0..127) — the same CHILD / STEP walk and the same arithmetic as
// enum_phase_kernel — in one or two launches of one-wave workgroups pulling "top tasks" (column of
// all rows, coefficients of the levels >= 64 chosen so far, partial distance, root level): the
// first launch walks the root down to a cut level and emits the survivors as top tasks, the second
// walks those in parallel down to level 64, where every surviving node becomes a task for the
// wave-per-subtree kernel (column of the rows below 64, partial distance; the coefficients of
// levels >= 64 are stored once per such node in xhi_root).  No candidate can be reported up here,
// so the top runs under the initial radius (tasks that a later, smaller radius cuts die at their
// first test in the next launch: the visited set is the reference's).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double rl2(const double (&v)[2], int idx)
{
  return idx < 64 ? rl_f64(v[0], idx) : rl_f64(v[1], idx - 64);
}
__device__ __forceinline__ int rl2i(const int (&v)[2], int idx)
{
  return idx < 64 ? rl_i32(v[0], idx) : rl_i32(v[1], idx - 64);
}

template <bool SUBS, bool DUAL>
__global__ void __launch_bounds__(64)
    enum_top_kernel(DevShared *__restrict__ g, HostCtl *__restrict__ h, TopBuf in, unsigned n_in,
                    TopBuf out_top, int stop, TaskBuf out, double *__restrict__ xhi_root, int d,
                    double maxdist, int count_nodes, int launch_idx)
{
  extern __shared__ __attribute__((aligned(16))) double stk2[];  // slots 65..d (slot k: k doubles)
  const int lane   = threadIdx.x & 63;
  const int off65  = tri_off(65);
  const double *mu = g->mu_tri;
  double rd[2], bnd[2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
  {
    rd[q]  = g->rdiag[lane + 64 * q];
    bnd[q] = g->pruning[lane + 64 * q] * maxdist;
  }
  unsigned long long cnt[2] = {0, 0};
  double sb[2] = {0.0, 0.0};  // findsubsols: this wave's view of the best distance per level
  if constexpr (SUBS)
  {
    sb[0] = __longlong_as_double((long long)g->sub_bits[lane]);
    sb[1] = __longlong_as_double((long long)g->sub_bits[64 + lane]);
  }
  for (;;)
  {
    unsigned t = 0;
    if (lane == 0)
      t = atomicAdd(&g->task_head[launch_idx], 1u);
    t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
    if (t >= n_in)
      break;
    const int Lt = __builtin_amdgcn_readfirstlane(in.level[t]);
    double S[2]  = {in.col[(unsigned long long)t * 128 + lane], in.col[(unsigned long long)t * 128 + 64 + lane]};
    // xs[1]: lane = level - 64; the levels >= Lt come from the task, the walk fills the others
    double xs[2] = {0.0, in.xhi[(unsigned long long)t * 64 + lane]};
    double cs[2] = {0.0, 0.0}, pds[2] = {0.0, 0.0};
    int dxs[2] = {0, 0}, ddxs[2] = {0, 0};
    // process_subsolution for a node at level lvl >= 64
    auto sub_report = [&](int lvl, double dist)
    {
      unsigned long long old = 0;
      if (lane == 0)
        old = atomicMin(&g->sub_bits[lvl], (unsigned long long)__double_as_longlong(dist));
      old               = rfl_u64(old);
      const double oldd = __longlong_as_double((long long)old);
#pragma unroll
      for (int q = 0; q < 2; ++q)
        sb[q] = (lane + 64 * q == lvl) ? fmin(oldd, dist) : sb[q];
      if (!(dist < oldd))
        return;
      unsigned long long idx = 0;
      if (lane == 0)
        idx = atomicAdd(&g->sol_head, 1ull);
      idx = rfl_u64(idx);
      for (unsigned spin = 0; idx >= load_sys_u64(&h->consumed) + FPHIP_RING_CAP; ++spin)
      {
        __builtin_amdgcn_s_sleep(64);
        if (spin > (1u << 24))
        {
          if (lane == 0)
            atomicOr(&g->error_flags, FPHIP_ERR_RING_TIMEOUT);
          break;
        }
      }
      SolRec *r       = &h->ring[idx % FPHIP_RING_CAP];
      r->x[lane]      = 0.0;
      r->x[64 + lane] = (64 + lane < d && 64 + lane >= lvl) ? xs[1] : 0.0;
      if (lane == 0)
      {
        r->dist   = dist;
        r->kind   = 1;
        r->offset = lvl;
      }
      __threadfence_system();
      if (lane == 0)
        __hip_atomic_store(&r->seq, idx + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    int k     = Lt;
    double nd = in.pd[t];
    bool done = false;
    while (!done)
    {
      // ---- CHILD chain
      for (;;)
      {
        k               = __builtin_amdgcn_readfirstlane(k);
        const int kc    = k - 1;
        const double c1 = rl2(S, kc);
        const double x1 = round(c1);
        const double a1 = x1 - c1;
        const double n1 = nd + a1 * a1 * rl2(rd, kc);
        if (!(n1 <= rl2(bnd, kc)))
        {
          done = k >= Lt;
          break;
        }
        if (k == stop && k < Lt)
        {
          unsigned oi = 0;
          if (stop == 64)
          {  // hand the subtree below this node to the wave-per-subtree kernel
            if (lane == 0)
              oi = atomicAdd(out.count, 1u);
            oi = (unsigned)__builtin_amdgcn_readfirstlane((int)oi);
            if (oi < out.cap)
            {
              out.col[(unsigned long long)oi * 64 + lane]  = S[0];
              out.x[(unsigned long long)oi * 64 + lane]    = 0.0;
              xhi_root[(unsigned long long)oi * 64 + lane] = xs[1];
              if (lane == 0)
              {
                out.pd[oi]    = nd;
                out.level[oi] = 64;
                out.root[oi]  = (int)oi;
              }
            }
          }
          else
          {  // a top task for the next (parallel) top launch
            if (lane == 0)
              oi = atomicAdd(out_top.count, 1u);
            oi = (unsigned)__builtin_amdgcn_readfirstlane((int)oi);
            if (oi < out_top.cap)
            {
              out_top.col[(unsigned long long)oi * 128 + lane]      = S[0];
              out_top.col[(unsigned long long)oi * 128 + 64 + lane] = S[1];
              out_top.xhi[(unsigned long long)oi * 64 + lane]       = xs[1];
              if (lane == 0)
              {
                out_top.pd[oi]    = nd;
                out_top.level[oi] = k;
              }
            }
          }
          break;  // → next sibling at level k (an overfull buffer is detected by the host: count > cap)
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
          if (lane + 64 * q < k)
            stk2[tri_off(k) - off65 + lane + 64 * q] = S[q];
        {
          const int s1 = (c1 >= x1) ? 1 : -1;
#pragma unroll
          for (int q = 0; q < 2; ++q)
          {
            const bool me = lane + 64 * q == kc;
            cs[q]         = me ? c1 : cs[q];
            xs[q]         = me ? x1 : xs[q];
            pds[q]        = me ? nd : pds[q];
            dxs[q]        = me ? s1 : dxs[q];
            ddxs[q]       = me ? s1 : ddxs[q];
            cnt[q] += me ? 1ull : 0ull;
          }
        }
        if constexpr (SUBS)
        {
          if (n1 < rl2(sb, kc) && n1 != 0.0)
            sub_report(kc, n1);
        }
        k  = kc;
        nd = n1;  // k >= 64 here
#pragma unroll
        for (int q = 0; q < 2; ++q)
        {
          const double mk = mu[tri_off(k) + min(lane + 64 * q, k - 1)];
          S[q]            = S[q] - (DUAL ? a1 : x1) * mk;
        }
      }
      if (done)
        break;
      // ---- STEP loop
      for (;;)
      {
        k = __builtin_amdgcn_readfirstlane(k);
        double par[2], mk[2];
#pragma unroll
        for (int q = 0; q < 2; ++q)
        {
          par[q] = stk2[tri_off(k + 1) - off65 + min(lane + 64 * q, k)];
          mk[q]  = mu[tri_off(k) + min(lane + 64 * q, k - 1)];
        }
        double xk        = rl2(xs, k);
        const double ck  = rl2(cs, k);
        const double pdk = rl2(pds, k);
        int dxk = rl2i(dxs, k), ddxk = rl2i(ddxs, k);
        if (pdk != 0.0)
        {
          xk += (double)dxk;
          ddxk = -ddxk;
          dxk  = ddxk - dxk;
        }
        else
        {
          xk += 1.0;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
        {
          const bool me = lane + 64 * q == k;
          xs[q]         = me ? xk : xs[q];
          dxs[q]        = me ? dxk : dxs[q];
          ddxs[q]       = me ? ddxk : ddxs[q];
        }
        const double a = xk - ck;
        nd             = pdk + a * a * rl2(rd, k);
        if (!(nd <= rl2(bnd, k)))
        {
          ++k;
          if (k >= Lt)
          {
            done = true;
            break;
          }
          continue;
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
          cnt[q] += (lane + 64 * q == k) ? 1ull : 0ull;
        if constexpr (SUBS)
        {
          if (nd < rl2(sb, k) && nd != 0.0)
            sub_report(k, nd);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
          S[q] = par[q] - (DUAL ? a : xk) * mk[q];
        break;
      }
    }
  }
  if (count_nodes)
  {
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (cnt[q] != 0)
        atomicAdd(&g->nodes[lane + 64 * q], cnt[q]);
  }
}
template __global__ void enum_top_kernel<false, false>(DevShared *, HostCtl *, TopBuf, unsigned, TopBuf,
                                                       int, TaskBuf, double *, int, double, int, int);
template __global__ void enum_top_kernel<true, false>(DevShared *, HostCtl *, TopBuf, unsigned, TopBuf,
                                                      int, TaskBuf, double *, int, double, int, int);
template __global__ void enum_top_kernel<false, true>(DevShared *, HostCtl *, TopBuf, unsigned, TopBuf,
                                                      int, TaskBuf, double *, int, double, int, int);

// 64-bit content key of every task (its coefficient prefix x[Lt..d)): the task ORDER in the buffer
// is not deterministic across ranks, the content is.  One wave per task.
__global__ void __launch_bounds__(256)
    task_key_kernel(TaskBuf in, unsigned n, int d, unsigned long long *__restrict__ keys,
                    const double *__restrict__ xhi_root)
{
  const int lane = threadIdx.x & 63;
  const unsigned w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const unsigned nw = (gridDim.x * blockDim.x) >> 6;
  for (unsigned ti = w; ti < n; ti += nw)
  {
    const int Lt      = in.level[ti];
    const double xpre = in.x[(unsigned long long)ti * 64 + lane];
    const bool on     = lane >= Lt && lane < d;
    unsigned h1 = on ? (unsigned)(int)xpre * (2654435761u * (unsigned)(lane + 1)) : 0u;
    unsigned h2 = on ? ((unsigned)(int)xpre ^ 0x9e3779b9u) * (40503u * (unsigned)(2 * lane + 3) + 2246822519u) : 0u;
    if (64 + lane < d)
    {  // blocks larger than 64: the coefficients of levels >= 64 (kept once per level-64 ancestor)
      const double xh = xhi_root[(unsigned long long)in.root[ti] * 64 + lane];
      h1 += (unsigned)(int)xh * (2654435761u * (unsigned)(lane + 65));
      h2 += ((unsigned)(int)xh ^ 0x9e3779b9u) * (40503u * (unsigned)(2 * lane + 131) + 2246822519u);
    }
    for (int off = 32; off > 0; off >>= 1)
    {
      h1 += (unsigned)__shfl_xor((int)h1, off);
      h2 += (unsigned)__shfl_xor((int)h2, off);
    }
    if (lane == 0)
      keys[ti] = ((unsigned long long)h1 << 32) | h2;
  }
}

}  // namespace fphip
