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
//     - lane l of the "level registers" xs/cs/pds/dxs/cnt holds the value for tree level l
//       (x[l], center[l], partdist[l], dx[l], nodes[l]; ddx[l] is always sign(dx[l]) and is not
//       stored); a level is read with v_readlane (uniform index in an SGPR) and written with
//       v_writelane (scalar values) or a lane-masked select (vector values);
//     - (r_ii, pruning_i) of a level come through the scalar cache (one s_load_dwordx4), issued
//       at the top of a step and waited for right before their use;
//     - lane i of the "row registers" holds row i of the centre partial sums
//       center_partsums[i][·] (enumerate_base.h:84).  Choosing x[k] updates ALL rows i<k with one
//       vector multiply + subtract:  S_k[i] = S_{k+1}[i] - x[k]*mu(k,i).  The reference's lazy
//       center_partsum_begin bookkeeping (enumerate_base.cpp:58-68) exists to avoid exactly this
//       O(k) work on a scalar CPU; here it is one VALU instruction pair, and every value is still
//       produced by the same operation sequence (j = d-1 … k, multiply then subtract, no FMA), so
//       centres, distances, node counts and solutions are bit-identical to the reference.
// * Backtracking needs S_{k+1} again when the next sibling x[k] is tried, so each wave keeps a
//   triangular stack of columns S_1..S_L (slot k holds k doubles, lane i touches only row i →
//   conflict-free ds_read_b64/ds_write_b64, no cross-lane traffic through LDS).  In the big walk
//   launches the stack is SPLIT: slots below level 34 in LDS — laid out in DESCENDING level order, so
//   that a push is one unmasked 64-lane store (see the kernel) — the tall ones (a quarter of the
//   nodes of a 60-dimensional block) in a per-wave scratch in global memory: 5 KB of LDS per wave
//   instead of 10, i.e. 8 resident waves per SIMD instead of 4.  mu rows are staged per workgroup
//   in LDS in the small launches (triangular packing) and read through L1 in the big ones (one
//   padded row per level, the (r, pruning) pair behind it: a buffer load and a scalar load with
//   the same scalar offset).
// * The walk is issue-bound on TWO ports that cost the same per instruction (one issue per SIMD
//   turn of four cycles): the vector ALU, and the scalar ALU + branch unit together.  The hot
//   loops are written for both: wave-uniform branches only (-structurizecfg-skip-uniform-regions,
//   no front-end cleanup blocks: -disable-lifetime-markers), the CHILD chain and the STEP loop
//   inside ONE cycle of uniform branches, lane masks built on the scalar unit, addresses as scalar
//   offsets of buffer loads, and — v_readlane_b32 occupies the vector ALU for two issue slots
//   (tests/perf/micro/valu_rates.hip) — broadcasts that feed vector arithmetic through
//   ds_bpermute_b32 on the otherwise idle LDS port.  PMC per counted node, round 3: 40.6 VALU +
//   28.6 SALU + 10.2 branch + 8.4 LDS instructions (round 2: 58 + 38 + 14 + 1.5; round 1: 92 + 49).
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
#include "enum_wave.h"


namespace fphip
{

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
// (8 waves per SIMD = at most 64 VGPRs: the walk is issue-bound and loses a tenth of its rate at 7)
template <bool MU_LDS, bool SUBS, bool DUAL>
__global__ void __launch_bounds__(FPHIP_MAX_BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8)))
    enum_phase_kernel(DevShared *__restrict__ g, HostCtl *__restrict__ h, TaskBuf in, TaskBuf out,
                      int d, int Lmax, int stop, unsigned task_lo, unsigned task_hi,
                      const unsigned *__restrict__ idxlist, int launch_idx, int count_nodes,
                      unsigned budget, const double *__restrict__ xhi_root, double *__restrict__ gstk,
                      int Tsplit, unsigned *__restrict__ qh, const unsigned *__restrict__ rcnt, unsigned rcap,
                      unsigned long long bound_init)
{
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  constexpr unsigned MUROW8 = FPHIP_MUROW * 8u;  // bytes per row of DevShared::mu_sq
  const unsigned lane8 = (unsigned)lane << 3;  // byte offset of this lane's element in a row
  const int triL = (Lmax * (Lmax + 1)) >> 1;  // doubles for slots 1..Lmax
  // mu rows 1..Lmax-1 (packed like the stack slots).  MU_LDS: one copy per workgroup in LDS (lowest
  // latency: split launches and tails, where few waves run).  !MU_LDS: read straight from global
  // memory — <= 16 KB, read-only, L1-resident after the first touches — so that the LDS buys
  // more resident waves (the big walk launch, where throughput matters).
  const double *mu_s;
  // The stack of centre columns (slot k = S_k restricted to the rows < k, k = 1..Lmax) is split:
  // the slots below Ts — where 9 nodes out of 10 of a pruned tree live — stay in LDS, the tall,
  // rarely touched slots k >= Ts go to a per-wave scratch in global memory (L2).  The LDS per wave
  // is what bounds the resident waves of this latency-bound walk (5 KB per wave = 8 waves per SIMD
  // against 4 with the whole stack of a 50-level task in LDS).
  const int nw     = (int)(blockDim.x >> 6);
  const int Ts     = min(Tsplit, Lmax + 1);
  const int Tsm1   = Ts - 1;
  const int ldsRow = tri_off(Ts);  // doubles of LDS stack slots per wave
  // The LDS slots are laid out in DESCENDING level order: slot k (k doubles) starts
  // tri_off(Ts) - tri_off(k + 1) doubles into the wave's region, slot k - 1 right behind it.  A push
  // of S_k is then ONE unmasked 64-lane store: the lanes beyond the row land in the slots of the
  // levels below k — dead at that moment, each is rewritten by the descent that reaches it — and, for
  // k < 11, in the 64-double pad behind slot 1.  (Neither a select to a dummy slot nor an
  // exec-masked branch in the hot loop; reads need no clamp for the same reason.)
  const int ldsWave = ldsRow + FPHIP_STACK_PAD;
  double *stk;  // this wave's region
  if constexpr (MU_LDS)
  {
    double *mu_l = smem;
    stk          = smem + triL + wave * ldsWave;
    const int nmu = (Lmax * (Lmax - 1)) >> 1;
    for (int i = threadIdx.x; i < nmu; i += blockDim.x)
      mu_l[i] = g->mu_tri[i];
    __syncthreads();
    mu_s = mu_l;
  }
  else
  {
    mu_s       = &g->mu_sq[0][0];
    stk        = smem + wave * ldsWave;
  }
  const __amdgpu_buffer_rsrc_t mu_b = mu_rsrc(&g->mu_sq[0][0], (unsigned)sizeof(g->mu_sq));  // (!MU_LDS)
  // LDS slot k of this lane: stk_top - tri8(k + 1).  The offset comes out of a lane-indexed table
  // with one v_readlane (tri8tab lane j = tri8(j + 2)): computing it from k costs four scalar
  // instructions, and since the address arithmetic left the vector unit the walk is bound by the
  // scalar / branch issue port (PMC: 47 VALU against 46 SALU + 15 branch per node); an incrementally
  // kept offset ends up as a VECTOR induction variable, one v_add per update.
  char *stk_top      = (char *)stk + (((unsigned)ldsRow << 3) + lane8);
  const int tri8tab  = ((lane + 2) * (lane + 1)) << 2;
  // slot k >= Ts at gst + tri_off(k); one spare double behind the last slot (index triL)
  double *gst = gstk + (size_t)(blockIdx.x * nw + wave) * (size_t)(triL - ldsRow + 1) - ldsRow;

  const double *rptab = &g->mu_sq[0][64];  // (r_ii, pruning_i) behind each mu row, read through the scalar cache
  // The bound lives in two places: the pinned host word the callback thread writes, and a
  // device-memory mirror.  Waves poll the mirror (L2) often and the host word (PCIe) rarely;
  // whoever sees a smaller host value lowers the mirror for everybody.
  // A wave STARTS from the bound the host held when it launched the kernel (an argument) and the
  // mirror: a read of the pinned host word costs a PCIe round trip, and thousands of waves doing one
  // at their start serialise on it (measured: 0.3 ms per launch of 4096 waves, 0.6 ms of 8192).
  unsigned long long mbits =
      rfl_u64(min(bound_init, __hip_atomic_load(&g->bound_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
  double maxdist           = __longlong_as_double((long long)mbits);
  double maxdist_v         = maxdist;  // the same value held in a VGPR pair (see FPHIP_IN_VGPR)
  FPHIP_IN_VGPR(maxdist_v);

  // level registers (lane = level) and counters
  double xs = 0.0, cs = 0.0, pds = 0.0;
  int dxs = 0;  // (ddx is always sign(dx): not stored)
  unsigned long long cnt = 0;
  unsigned cnt32         = 0;
  unsigned iter          = 0;   // failed steps, in units the refresh events add (64 at a time)
  int left               = 63;  // failed steps until the next refresh event
  // findsubsols (enumerate_base.cpp:36-40): lane = level, this wave's view of the best sub-solution
  // distance per level (subsoldists); the device-wide value in g->sub_bits is authoritative
  double sb = SUBS ? __longlong_as_double((long long)g->sub_bits[lane]) : 0.0;

#define FPHIP_REFRESH_BOUND(from_host)                                                            \
  do                                                                                              \
  {                                                                                               \
    unsigned long long nb_;                                                                       \
    if (from_host)                                                                                \
    {                                                                                             \
      nb_ = load_sys_u64(&h->bound_bits);                                                         \
      nb_ = rfl_u64(nb_);                                                                         \
      if (nb_ < mbits)                                                                            \
        lane0_atomic_umin_u64_noret(&g->bound_bits, nb_); /* (no lane-masked branch: enum_wave.h) */ \
    }                                                                                             \
    else                                                                                          \
    {                                                                                             \
      nb_ = __hip_atomic_load(&g->bound_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        \
    }                                                                                             \
    nb_ = rfl_u64(nb_); /* AFTER the join: the compiler then knows the value is wave-uniform */    \
    if (nb_ < mbits)                                                                              \
    {                                                                                             \
      mbits   = nb_;                                                                              \
      maxdist = __longlong_as_double((long long)mbits);                                           \
      maxdist_v = maxdist;                                                                        \
      FPHIP_IN_VGPR(maxdist_v);                                                                   \
    }                                                                                             \
  } while (0)

  // Tasks are pulled through FPHIP_NQ ticket counters (qh[q * FPHIP_QS], one per queue, 64 bytes
  // apart): a wave draws from its home queue and moves on to a queue that still holds tasks when
  // that one is exhausted.  Two input forms.  Compact list [task_lo, task_hi) (optionally through
  // the index list of the multi-GPU partition): queue q owns the positions task_lo + q + j NQ.
  // Regioned buffer (rcnt != null; what the breadth-first stage writes): queue q owns the
  // rcnt[q * FPHIP_QS] tasks of region q, slots q * rcap + i, drawn from the END — the stage appends
  // level by level, so the deepest roots, whose ancestors all had small partial distances (the most
  // promising subtrees: a good radius early), go first.
  unsigned q = (unsigned)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * nw + wave) % FPHIP_NQ));
  const unsigned nlist = task_hi - task_lo;
  for (;;)
  {
    // ---- pull a task ------------------------------------------------------------------------
    unsigned t = 0, cq = 0;
    bool have = false;
    for (;;)
    {
      t  = lane0_atomic_add_u32(&qh[q * FPHIP_QS], 1u);  // (the node counters are live across the pull)
      t  = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
      cq = rcnt ? min((unsigned)__builtin_amdgcn_readfirstlane((int)rcnt[q * FPHIP_QS]), rcap)
                : (nlist > q ? (nlist - q + FPHIP_NQ - 1u) / FPHIP_NQ : 0u);
      if (t < cq)
      {
        have = true;
        break;
      }
      // exhausted: the lanes look at the heads of the other queues, 64 at a time
      bool any = false;
#pragma unroll
      for (unsigned hf = 0; hf < FPHIP_NQ / 64; ++hf)
      {
        const unsigned qq = (q + 1u + hf * 64u + (unsigned)lane) % FPHIP_NQ;
        const unsigned hd = __hip_atomic_load(&qh[qq * FPHIP_QS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned cn = rcnt ? min(rcnt[qq * FPHIP_QS], rcap)
                                 : (nlist > qq ? (nlist - qq + FPHIP_NQ - 1u) / FPHIP_NQ : 0u);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(hd < cn);
        if (m != 0ull)
        {
          q   = (q + 1u + hf * 64u + (unsigned)__builtin_ctzll(m)) % FPHIP_NQ;
          any = true;
          break;
        }
      }
      if (!any)
        break;
    }
    if (!have)
    {  // every queue is empty: tell the waves still walking to shed work for the next launch
      if (budget != 0u)  // (every lane: the same word)
        __hip_atomic_store(&g->drain[launch_idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
    // multi-GPU: this rank's share of the task list is an explicit index list (built on the host
    // from a content-sorted order, see enum_host.hip), walked heaviest-first
    const unsigned long long pos = (unsigned long long)task_lo + q + (unsigned long long)t * FPHIP_NQ;
    const unsigned long long ti =
        rcnt ? (unsigned long long)q * rcap + (cq - 1u - t)
             : (idxlist ? (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)idxlist[pos]) : pos);
    const int Lt      = __builtin_amdgcn_readfirstlane(in.level[ti]);  // root level of this task
    const int rid     = __builtin_amdgcn_readfirstlane(in.root[ti]);   // level-64 ancestor (d > 64)
    const int tl      = here_lane(lane);
    const double xpre = in.x[ti * 64 + tl];                            // coefficients of levels >= Lt
    const double col0 = in.col[ti * 64 + tl];  // S_Lt rows (lane < Lt)
    const double pd0  = in.pd[ti];
    int donate        = 1 << 20;
    const unsigned iter0 = iter + (unsigned)(63 - left);  // (iterations of this task = iter + (63 - left) - iter0)
    FPHIP_REFRESH_BOUND((t & 63u) == 0u);

    // the task root is a surviving node at level Lt whose column and distance are given
    int k     = Lt;
    double S  = col0;  // S_k of the current node (rows < k valid)
    double nd = pd0;   // its distance

    // Reports a candidate (process_solution) and waits for the host's verdict, like enumlib's
    // mutex-protected process_sol (enumeration.h:286-299).
    auto report = [&](double dist)
    {
      // (no lane-masked branch in here either: the level registers are live across the report — enum_wave.h)
      const unsigned long long idx = rfl_u64(lane0_atomic_add_u64(&g->sol_head, 1ull));
      for (unsigned spin = 0; idx >= rfl_u64(load_sys_u64(&h->consumed)) + FPHIP_RING_CAP; ++spin)
      {  // flow control against the host consumer
        __builtin_amdgcn_s_sleep(64);
        if (spin > (1u << 24))
        {
          lane0_atomic_or_u32_noret(&g->error_flags, FPHIP_ERR_RING_TIMEOUT);
          break;
        }
      }
      SolRec *r  = &h->ring[idx % FPHIP_RING_CAP];
      double xf  = (lane < Lt) ? xs : xpre;
      const int rl = here_lane(lane);
      r->x[rl]   = (lane < d) ? xf : 0.0;
      // levels >= 64: the coefficients chosen by the top walk, stored once per level-64 ancestor (row stride
      // of xhi_root: 64 per started chunk of levels above 64)
      {
        const int xstr = d > 64 ? 64 * ((d - 1) >> 6) : 64;
#pragma unroll
        for (int q = 1; q < 4; ++q)
          r->x[64 * q + rl] = (64 * q + lane < d) ? xhi_root[(size_t)rid * xstr + 64 * (q - 1) + rl] : 0.0;
      }
      {  // (every lane stores the same three words and, behind the fence, the same sequence number)
        const int z = here_lane(0);  // (a zero made here: as a hoisted constant pair it was spilled as well)
        r->dist     = dist;
        r->kind     = z;
        r->offset   = z;
      }
      __threadfence_system();
      __hip_atomic_store(&r->seq, idx + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      for (unsigned spin = 0; rfl_u64(load_sys_u64(&h->consumed)) <= idx; ++spin)
      {
        __builtin_amdgcn_s_sleep(32);
        if (spin > (1u << 24))
        {
          lane0_atomic_or_u32_noret(&g->error_flags, FPHIP_ERR_RING_TIMEOUT);
          break;
        }
      }
      FPHIP_REFRESH_BOUND(true);
    };

    // process_subsolution (enumerate.cpp:241-249): a node at level lvl is shorter than every
    // sub-solution seen at that level.  Only the wave that lowers the device-wide best reports;
    // no verdict to wait for (sub-solutions never change the radius), only ring space.
    // NO lane-masked branch in here (enum_wave.h, lane0_atomic_*): the report runs for every node that beats its
    // level's best — millions of times early in a call — with all the level registers live across it.
    auto sub_report = [&](int lvl, double dist)
    {
      unsigned long long old =
          rfl_u64(lane0_atomic_umin_u64(&g->sub_bits[lvl], (unsigned long long)__double_as_longlong(dist)));
      const double oldd = __longlong_as_double((long long)old);
      sb                = (lane == lvl) ? fmin(oldd, dist) : sb;
      if (__builtin_amdgcn_ballot_w64(dist < oldd) == 0ull)
        return;
      const unsigned long long idx = rfl_u64(lane0_atomic_add_u64(&g->sol_head, 1ull));
      for (unsigned spin = 0; idx >= rfl_u64(load_sys_u64(&h->consumed)) + FPHIP_RING_CAP; ++spin)
      {
        __builtin_amdgcn_s_sleep(64);
        if (spin > (1u << 24))
        {
          lane0_atomic_or_u32_noret(&g->error_flags, FPHIP_ERR_RING_TIMEOUT);
          break;
        }
      }
      SolRec *r       = &h->ring[idx % FPHIP_RING_CAP];
      const double xf = (lane < Lt) ? xs : xpre;
      const int rl    = here_lane(lane);
      r->x[rl] = (lane < d && lane >= lvl) ? xf : 0.0;
      {
        const int xstr = d > 64 ? 64 * ((d - 1) >> 6) : 64;
#pragma unroll
        for (int q = 1; q < 4; ++q)
          r->x[64 * q + rl] = (64 * q + lane < d) ? xhi_root[(size_t)rid * xstr + 64 * (q - 1) + rl] : 0.0;
      }
      // (every lane stores the same three words and, behind the fence, the same sequence number)
      r->dist   = dist;
      r->kind   = 1;
      r->offset = lvl;
      __threadfence_system();
      __hip_atomic_store(&r->seq, idx + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    };

    // The walk is two hot loops (CHILD chain, STEP loop) inside one hot cycle inside an outer event
    // loop.  The cycle holds wave-uniform branches only — everything that needs a lane-masked branch
    // (emitting a task, reporting a candidate, refreshing the bound) happens BEHIND it — and this file is
    // compiled with -structurizecfg-skip-uniform-regions: the AMDGPU backend otherwise rewrites
    // every region that holds one divergent branch with exit codes, flag registers and a copy of
    // every live register per iteration (16 of the 45 VALU and most of the 49 SALU instructions of
    // a step were that).  FPHIP_OPAQUE on the event code behind a loop keeps the loop's exits on
    // one successor block.  Scalar and branch instructions share one issue port that is as slow as
    // the vector ALU's (one instruction per SIMD turn): scalar work is counted like vector work.
    enum : int { EV_CHILD = 0, EV_EMIT = 1, EV_REPORT = 2, EV_DONE = 3, EV_RESTEP = 4, EV_OK = 5, EV_REFRESH = 6,
                 EV_SPECIAL = 7, EV_FAIL = 8, EV_LEAF = 9 };
    // Levels whose surviving first children are handed to the next launch: [elo, elo + erng].  The
    // split launches (stop >= 0, no budget) emit at level `stop` only; the walk launches (stop < 0)
    // emit at every level >= donate once the task sheds work.
    unsigned elo  = (stop >= 0 && stop < Lt) ? (unsigned)stop : (unsigned)donate;
    unsigned erng = (stop >= 0 && stop < Lt) ? 0u : 0x7fffffffu;
    // What the CHILD chain tests instead: ONE bit of a scalar mask (bit k - 1 for level k, 1 <= k <= 64)
    // that holds every level which leaves the chain for the general path behind it — the emission
    // levels and level 1.  The general path sorts out which it is.
    unsigned long long smask;
    auto hot_range = [&]()
    {
      const auto from = [](unsigned lvl) { return lvl > 64u ? 0ull : (lvl <= 1u ? ~0ull : ~0ull << (lvl - 1u)); };
      unsigned long long m = from(elo);               // levels >= elo ...
      if (erng < 64u)
        m &= ~from(elo + erng + 1u);                  // ... up to elo + erng
      // ... and level 1: the descent to the leaves (a handful of nodes in 10^10) goes through the
      // general path, so that the chain has no exit BEHIND the state updates of a descent — exits in
      // front of and behind them keep the old and the new level registers alive across the loop (a
      // copy of each per iteration)
      smask = rfl_u64(m | 1ull);
    };
    hot_range();
    bool buffer_full         = false;
    bool resume_step         = false;  // re-enter the STEP loop at level k (after a report / refresh)
    // (par, mk) = (S_{k+1}, row k of mu): what a step at level k needs to rebuild S_k.  Both are in
    // registers whenever the STEP loop is entered: left there by the CHILD descent to level k or by
    // the successful step at level k that preceded a failing first child.
    double par = 0.0, mk = 0.0;
    double xk = 0.0, a = 0.0;
    int kc = 0;
    double mk1 = 0.0, c1 = 0.0, x1 = 0.0, a1 = 0.0, n1 = 0.0;
    // S_k -> slot k (LDS: all 64 lanes store, `lds8` = 8 tri_off(k + 1), see the layout above; global
    // part: the lanes beyond the row write the spare double — a select, no lane-masked branch, and
    // its operands are computed in front of it so that it stays one)
#define FPHIP_PUSH(in_lds, kk, lds8)                                            \
  if (__builtin_expect(in_lds, 1))                                              \
    *(double *)(stk_top - (lds8)) = S;                                          \
  else                                                                          \
  { /* (the level hidden from the optimiser: no induction variable for this block in the loop) */ \
    int kt = (kk);                                                              \
    asm volatile("" : "+s"(kt));                                                \
    const unsigned gk8 = ((unsigned)(kt * (kt - 1)) << 2) + lane8;              \
    *(double *)((char *)gst + ((lane < kt) ? gk8 : (unsigned)triL << 3)) = S;   \
  }
// descent to level kc (the first child survived): PUSH stores S_k for the steps of level kc, NEXT
// moves the level variable of the caller
#define FPHIP_DESCEND(PUSH, NEXT)                                                                              \
  do                                                                                                     \
  {                                                                                                      \
    PUSH;                                                                                                \
    const int s1 = __builtin_amdgcn_ballot_w64(c1 >= x1) != 0ull ? 1 : -1; /* :71 / :114 (dx = ddx) */    \
    const unsigned long long me = lane_bit(kc);                                                          \
    cs                          = sel_f64(me, c1, cs);                                                   \
    xs                          = sel_f64(me, x1, xs);                                                   \
    pds                         = sel_f64(me, nd, pds);                                                  \
    dxs                         = wl_i32(s1, kc, dxs);                                                   \
    cnt32                       = add_bit(me, cnt32); /* ++nodes[kk-1] */                                \
    if constexpr (SUBS)                                                                                  \
    {                                                                                                    \
      if (__builtin_amdgcn_ballot_w64(n1 < rl_f64(sb, kc) && n1 != 0.0) != 0ull)                         \
        sub_report(kc, n1);                                                                              \
    }                                                                                                    \
    par = S; /* (S_{kc+1}, row kc): what a step at the new level needs */                                \
    mk  = mk1;                                                                                           \
    NEXT;                                                                                                \
    nd  = n1;                                                                                            \
    /* S_k = S_{k+1} - x[k]*mu(k,.), :53-58 (mk1 is row k; at k == 0 a dead value) */                    \
    S = S - (DUAL ? a1 : x1) * mk1;                                                                      \
  } while (0)
    for (;;)
    {
      int ev;
      if (resume_step)
      {  // (par, mk) were lost to the slow path: reload them
        const unsigned k8  = (unsigned)k << 3;
        const unsigned cl8 = min(lane8, k8 - 8u);  // (k == 0: wraps, the lanes read valid, unused data)
        if (k + 1 < Ts)
          par = *(const double *)(stk_top - tri8(k + 2));  // slot k + 1
        else
          par = ld_off(gst, tri8(k + 1) + cl8);
        if constexpr (MU_LDS)
          mk = ld_off(mu_s, tri8(k) + cl8);
        else
          mk = ld_row(mu_b, (unsigned)k * MUROW8, lane8);
      }
      bool at_step = resume_step;
      resume_step  = false;
      // ---- the hot cycle: CHILD chain -> (no surviving child) -> STEP loop -> (a sibling survives) ->
      // CHILD chain ...  Uniform branches only, like the two loops in it: the transitions between them
      // happen once per five nodes, and in a region the structuriser rewrites each one costs a copy of
      // every level register plus a cascade of flag tests.  Everything else is an event for the code
      // behind the cycle.
      for (;;)
      {
        if (!at_step)
        {
          // ================= CHILD chain: descend while the first child survives ====================
          // (state: a surviving, already counted node at level k with column S = S_k, distance nd)
          // (the loop carries kc = k - 1: one scalar register and one decrement per descent)
          kc = k - 1;
          for (;;)
          {
            asm volatile("" : "+s"(kc));  // (keeps kc itself the carried value, decremented in place)
            const unsigned kc8 = (unsigned)kc << 3;
            v4i q1             = rp_issue2(rptab, (unsigned)kc * MUROW8);
            // speculative load for the descending case: row kc of mu is needed right after the test
            // (its latency overlaps the test).  tri8(k) = 8 * tri_off(k) comes from the scalar unit: row kc
            // starts kc elements before row k.  At kc == 0 the clamp wraps and the lanes read the
            // head of the table (valid, unused).
            if constexpr (MU_LDS)
              mk1 = ld_off(mu_s, tri8(kc) + min(lane8, kc8 - 8u));
            else  // (the square copy: scalar row address, zero beyond the row)
              mk1 = ld_row(mu_b, (unsigned)kc * MUROW8, lane8);
            const int ka = lane_addr(kc);
            c1           = bp_f64(S, ka);  // center[kk-1] = center_partsums[kk-1][kk]
            // roundto() = round(): half away from zero (enumerate_base.h:33-34), as round-to-even
            // (one instruction) plus the correction of the ties that went towards zero
            x1 = rint(c1);
            a1 = x1 - c1;
            if (__builtin_amdgcn_ballot_w64(fabs(a1) == 0.5) != 0ull)
            {  // (a tie: rare; the correction itself is a pair of selects — the empty statement keeps the
               //  optimiser from flattening the block into eleven instructions executed for every node)
              asm volatile("");
              const bool fix = (a1 < 0.0) == (c1 > 0.0);
              x1             = fix ? x1 - (a1 + a1) : x1;
              a1             = fix ? -a1 : a1;
            }
            rp_wait(q1);
            n1 = nd + a1 * a1 * rp_r(q1);  // :28-29
            // :31-32 (partdistbounds = pruning * maxdist); the ballot makes the test wave-uniform for
            // the compiler although maxdist_v went through an asm statement
            if (__builtin_amdgcn_ballot_w64(n1 <= rp_p(q1) * maxdist_v) == 0ull)
            {  // no surviving child: next sibling at level k (the root has none: task done)
              ev = EV_FAIL;
              FPHIP_EXIT();
              break;
            }
            unsigned sbit = (unsigned)(smask >> kc);
            asm("" : "+s"(sbit));  // (a 32-bit test of the shifted mask: the compiler's own form adds a 64-bit compare)
            if (sbit & 1u)
            {  // an emission level (k == stop in the split launches, k >= donate once this task sheds
               // work; never the task root: donate starts beyond every level and the root is only
               // visited first) or a level above the LDS part of the stack: the general path below
              ev = EV_SPECIAL;
              FPHIP_EXIT();
              break;
            }
            // descend: level kc becomes the current level.  S is needed again when x[kc] steps to its
            // next sibling.  (The global part of the stack is no rare path: a quarter of the nodes of a
            // 60-dimensional block sit above level 33 — which is why it is handled inside the loop.)
            FPHIP_DESCEND(FPHIP_PUSH(kc < Ts - 1, kc + 1, (unsigned)bp_i32(tri8tab, ka)), --kc);
          }
          k = kc + 1;
          if (ev != EV_FAIL)
            break;  // EV_SPECIAL
          // no surviving child: next sibling at level k (the root has none: task done)
          if (k >= Lt)
          {
            ev = EV_DONE;
            break;
          }
        }
        at_step = false;
        // ================= STEP loop: next sibling at level k, climbing while they fail ===========
        int ka = lane_addr(k);  // (4 k in a VGPR, carried: the address of this level's bpermutes)
        for (;;)
        {
          v4i qk           = rp_issue2(rptab, (unsigned)k * MUROW8);
          // x[k] and its centre feed vector arithmetic only: through the LDS crossbar; the partial
          // distance above and the step select the zig-zag on the scalar unit: v_readlane
          xk               = bp_f64(xs, ka);
          const double ck  = bp_f64(cs, ka);
          const int pdlo   = __builtin_amdgcn_readlane(__double2loint(pds), k);
          const int pdhi   = __builtin_amdgcn_readlane(__double2hiint(pds), k);
          const double pdk = __hiloint2double(pdhi, pdlo);
          int dxk          = rl_i32(dxs, k);
          // :80-89 (is_svp is always true here): zig-zag around the centre unless the partial
          // distance above is exactly +0 (then x only grows).  pdk is a sum of squares, never -0:
          // the test and the selection of the step are scalar-unit work.
          int pdor = pdlo | pdhi;
          asm volatile("" : "+s"(pdor));  // (one 32-bit OR, not a 64-bit compare of a register pair)
          const bool zig = pdor != 0;
          int stepi      = zig ? dxk : 1;
          asm volatile("" : "+s"(stepi));  // (keeps the select in front of the int -> double conversion)
          xk += (double)stepi;
          dxk = zig ? ((dxk > 0 ? -1 : 1) - dxk) : dxk;  // ddx = -ddx; dx = ddx - dx, ddx == sign(dx)
          xs  = sel_f64(lane_bit(k), xk, xs);
          dxs = wl_i32(dxk, k, dxs);
          asm volatile("" : "+v"(xs), "+v"(dxs));  // (the updates stay in front of the test: sunk behind
                                                    //  it they would merge the loop's exits again)
          a             = xk - ck;
          rp_wait(qk);
          nd = pdk + a * a * rp_r(qk);  // :91-92
          if (__builtin_amdgcn_ballot_w64(nd <= rp_p(qk) * maxdist_v) != 0ull)
          {
            ev = EV_OK;
            break;
          }
          // :93-94: the parent steps to its next sibling.  The loads for the surviving case at the new
          // level are issued now: their latency (mu comes through L1 in the big launches) overlaps the
          // next test.
          ++k;
          // slot k + 1 (lanes beyond the row: a valid, unused address)
          ka += 4;
          if (__builtin_expect(k < Tsm1, 1))
            par = *(const double *)(stk_top - (unsigned)bp_i32(tri8tab, ka));
          else
          {  // (the level number hidden from the optimiser: it otherwise carries 4 k and 8 k for this
             //  block as induction variables, two scalar adds in every iteration of the loop)
            int kt = k;
            asm volatile("" : "+s"(kt));
            par = ld_off(gst, tri8(kt + 1) + min(lane8, ((unsigned)kt << 3) - 8u));
          }
          if constexpr (MU_LDS)
            mk = ld_off(mu_s, tri8(k) + min(lane8, ((unsigned)k << 3) - 8u));
          else
            mk = ld_row(mu_b, (unsigned)k * MUROW8, lane8);
          if (k >= Lt)
          {
            ev = EV_DONE;
            break;
          }
          if (--left < 0)
          {  // (every 64 failed steps; a countdown: one scalar instruction less than a masked counter)
            ev = EV_REFRESH;
            break;
          }
        }
        if (ev != EV_OK)
          break;  // EV_DONE / EV_REFRESH
        cnt32 = add_bit(lane_bit(k), cnt32);  // ++nodes[kk]
        if constexpr (SUBS)
        {
          if (__builtin_amdgcn_ballot_w64(nd < rl_f64(sb, k) && nd != 0.0) != 0ull)
            sub_report(k, nd);
        }
        if (k == 0)
        {
          ev = EV_LEAF;
          break;
        }
        S = par - (DUAL ? a : xk) * mk;  // :104-110
      }
      // ---- events
      FPHIP_OPAQUE(ev);
      if (ev == EV_DONE)
        break;
      if (ev == EV_SPECIAL)
      {  // EV_SPECIAL
        if ((unsigned)(k - elo) <= erng)
          ev = EV_EMIT;  // hand the subtree below this node to the next launch
        else
        {  // the descent to level 0 (or a level the mask holds for no reason): one iteration by hand
          kc = k - 1;
          FPHIP_DESCEND(FPHIP_PUSH(k < Ts, k, tri8(k + 1)), k = kc);
          if (k != 0)
            continue;  // → CHILD chain at the new level
          ev = (__builtin_amdgcn_ballot_w64(nd > 0.0) != 0ull) ? EV_REPORT : EV_RESTEP;
        }
      }
      if (ev == EV_EMIT)
      {
        unsigned oi = lane0_atomic_add_u32(out.count, 1u);
        oi          = (unsigned)__builtin_amdgcn_readfirstlane((int)oi);
        if (oi < out.cap)
        {
          const int el                              = here_lane(lane);
          out.col[(unsigned long long)oi * 64 + el] = S;
          const double xf                           = (lane < Lt) ? xs : xpre;
          out.x[(unsigned long long)oi * 64 + el]   = xf;
          out.pd[oi]    = nd;  // (every lane: the same three words)
          out.level[oi] = k;
          out.root[oi]  = rid;
          // → next sibling at level k
        }
        else
        {  // buffer full: walk everything inline from here on (results stay exact)
          lane0_atomic_or_u32_noret(&g->error_flags, FPHIP_FLAG_TASK_OVERFLOW);
          buffer_full = true;
          donate      = 1 << 20;
          elo         = 1u << 20;
          erng        = 0u;
          hot_range();
          continue;
        }
      }
      else if (ev == EV_REPORT)
      {
        report(nd);
        FPHIP_JOIN();
      }
      else if (ev == EV_LEAF)
      {  // level 0, :97-101: report (nd > 0), then the next sibling of level 0
        if (__builtin_amdgcn_ballot_w64(nd > 0.0) != 0ull)
        {
          report(nd);
          FPHIP_JOIN();
        }
      }
      else if (ev == EV_REFRESH)
      {  // every 64 failed steps
        cnt += cnt32;  // the per-level counters of the hot loops are 32 bits wide
        cnt32 = 0u;
        iter += 64u;
        left = 63;
        FPHIP_REFRESH_BOUND((iter & 16383u) == 0u);
        const unsigned titer = iter - iter0;
        if (budget != 0u && titer >= 256u && !buffer_full)
        {  // work donation: once the task queue has run dry (other waves are idle), or this task
           // exceeded its budget, keep only the subtree below the current level and emit every
           // sibling subtree above it as a task for the next launch
          const unsigned dr = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(
              &g->drain[launch_idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          if (dr != 0u || titer >= budget)
          {
            donate = min(donate, k + 1);
            if (stop < 0)
              elo = (unsigned)donate;
          }
        }
        FPHIP_JOIN();
        hot_range();
      }
      // next sibling at level k (the slow paths lost (par, mk): reloaded at the top)
      resume_step = true;
    }
#undef FPHIP_DESCEND
#undef FPHIP_PUSH
    cnt += cnt32;
    cnt32 = 0u;
  }
#undef FPHIP_REFRESH_BOUND

  if (count_nodes && cnt != 0)
    atomicAdd(&g->nodes[lane], cnt);
  if (lane == 0)
    atomicAdd(&g->iters, (unsigned long long)(iter + (unsigned)(63 - left)));
}

#define FPHIP_INST(M, S, D)                                                                            \
  template __global__ void enum_phase_kernel<M, S, D>(DevShared *, HostCtl *, TaskBuf, TaskBuf, int,  \
                                                      int, int, unsigned, unsigned, const unsigned *,  \
                                                      int, int, unsigned, const double *, double *, int, \
                                                      unsigned *, const unsigned *, unsigned,            \
                                                      unsigned long long);
FPHIP_INST(true, false, false)
FPHIP_INST(false, false, false)
FPHIP_INST(true, true, false)
FPHIP_INST(false, true, false)
FPHIP_INST(true, false, true)
FPHIP_INST(false, false, true)
#undef FPHIP_INST

// ---------------------------------------------------------------------------------------------
// Breadth-first expansion of the TOP of the tree.
//
// The wave-per-subtree walk needs tens of thousands of subtree roots before the chip is busy, and a
// pruned tree is thin at the top: the depth-first split launches that used to produce the roots were
// a handful of lone waves each walking one dependent chain (100 ns per step when nothing else is
// resident), and the walk launch that followed ended with its HEAVIEST root — subtree sizes under a
// level are heavy-tailed.  This kernel instead expands the tree level by level, one wavefront per
// parent node: all children of a node come from one column (S_{k+1}), so the chain per level is the
// zig-zag over ONE node's children, the shortest there is; and it decides per child whether the
// subtree below it is still HEAVY (Gaussian-heuristic size estimate from its own partial distance:
// lane k evaluates the expected number of descendants at level k, one wave sum) — heavy children
// form the next level's frontier, light ones are final tasks of the walk at once.  The final task
// list therefore holds roots of mixed levels and of comparable (estimated) size.
//
// Same arithmetic as the walk (centre, round(), zig-zag order, bound test, S_k = S_{k+1} - x mu_k;
// enumerate_base.cpp:24-118), same counting; the estimate only steers scheduling.  No candidate can
// be met up here (level 0 is never expanded), so the radius is the call's initial one.
//
// One launch expands `nlev` consecutive levels when it is a single workgroup (the thin top: a
// __syncthreads between levels), otherwise one level (the host enqueues the levels back to back,
// without waiting: frontier sizes live in g->bfs_count).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum_f32(float v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
    v += __shfl_xor(v, off);
  return v;
}

template <bool DUAL>
__global__ void __launch_bounds__(1024)
    enum_bfs_kernel(DevShared *__restrict__ g, double maxdist, QueueMem *__restrict__ qm,
                    TaskBuf f0, TaskBuf f1, TaskBuf fin, int L0, int nlev, int floor_level, float heavy,
                    int count_nodes, int compact_n, int shard_index, int shard_count)
{
  // Buffers are REGIONED: region q of a buffer holds the slots [q rcap, (q + 1) rcap), its fill count
  // sits in qm (64 bytes from the next one): emission counters are spread over FPHIP_NQ addresses
  // (see enum_device.h).  compact_n >= 0: the parents of the first level are a compact list of that
  // length instead (the level-64 tasks of the top walk): region q owns its positions q + j NQ.
  const int lane       = threadIdx.x & 63;
  const unsigned nwb   = blockDim.x >> 6;
  const unsigned gw    = blockIdx.x * nwb + (threadIdx.x >> 6);
  const unsigned nwave = gridDim.x * nwb;
  const unsigned rcap  = fin.cap / FPHIP_NQ;
  // (maxdist: the call's radius, an argument — a read of the pinned host word per wave would cost a
  // PCIe round trip each: 0.3-0.6 ms per launch)
  const unsigned rstep = nwave < FPHIP_NQ ? nwave : FPHIP_NQ;
  const unsigned t0    = gw / FPHIP_NQ;
  const unsigned tstep = nwave < FPHIP_NQ ? 1u : nwave / FPHIP_NQ;
  unsigned emitted     = gw * 7u;  // children are dealt round-robin over the regions
  for (int s = 0; s < nlev; ++s)
  {
    const int L      = L0 - s;  // level of the parents
    const int kc     = L - 1;   // level of their children
    const TaskBuf in = (s & 1) ? f1 : f0, out = (s & 1) ? f0 : f1;
    const bool compact = s == 0 && compact_n >= 0;
    const double r    = g->rdiag[kc];
    const double bnd  = g->pruning[kc] * maxdist;  // partdistbounds[kc], enumerate.cpp:218-228
    const double mk   = (lane < kc) ? g->mu_tri[tri_off(kc) + lane] : 0.0;  // row kc of mu
    // estimate of the subtree below a child at level kc: lane k < kc holds the level-k term
    const float Ak    = (lane < kc) ? g->bfs_A[kc][lane] : 0.f;
    const float hk    = 0.5f * (float)(kc - lane);
    const double R2k  = g->pruning[lane] * maxdist;
    const bool deeper = kc > floor_level;  // at the floor every child is a final task
    unsigned cnt = 0;
    for (unsigned rg = gw % FPHIP_NQ; rg < FPHIP_NQ; rg += rstep)
    {
      unsigned n;
      if (compact)
        n = (unsigned)compact_n > rg ? ((unsigned)compact_n - rg + FPHIP_NQ - 1u) / FPHIP_NQ : 0u;
      else
      {
        n = __hip_atomic_load(&qm->bfs[L][rg * FPHIP_QS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        n = n < rcap ? n : rcap;
      }
      n = (unsigned)__builtin_amdgcn_readfirstlane((int)n);
      for (unsigned t = t0; t < n; t += tstep)
      {
        const unsigned long long p = compact ? (unsigned long long)rg + (unsigned long long)t * FPHIP_NQ
                                             : (unsigned long long)rg * rcap + t;
        const double col = in.col[p * 64 + lane];
        const double xp  = in.x[p * 64 + lane];
        const double pdu = __longlong_as_double((long long)rfl_u64((unsigned long long)__double_as_longlong(in.pd[p])));
        const int ridu   = __builtin_amdgcn_readfirstlane(in.root[p]);
        const double c   = rl_f64(col, kc);  // center[kc] = center_partsums[kc][kc + 1]
        double x = rint(c);                  // roundto(): half away from zero (enumerate_base.h:33-34)
        double a = x - c;
        if (fabs(a) == 0.5 && ((a < 0.0) == (c > 0.0)))
        {
          x = x - (a + a);
          a = -a;
        }
        double nd = pdu + a * a * r;   // :28-29
        // multi-GPU (shard_count = 4 W + mode, 0 = none).  Mode 1 — the ONE launch that splits the frontier: every rank
        // expands the parents whose coefficient prefix (levels >= L: the same on every rank, whatever the order of
        // its buffers) hashes to it; the others see no surviving child (a select, no branch: this file is compiled
        // with -structurizecfg-skip-uniform-regions, and a new branch around the emission below was miscompiled).
        // Mode 2 — the replicated launches in front of the split: no child becomes a final task before the levels
        // run out (a final task emitted here would sit in every rank's list).
        const int smode = shard_count & 3;
        {
          unsigned hk = (lane >= L) ? (unsigned)(int)xp * (2654435761u * (unsigned)(lane + 1)) : 0u;
#pragma unroll
          for (int off = 32; off > 0; off >>= 1)
            hk += (unsigned)__shfl_xor((int)hk, off);
          const unsigned sW = (unsigned)max(shard_count >> 2, 1);
          const bool mine   = (smode != 1) | ((int)((hk ^ (hk >> 15)) % sW) == shard_index);
          nd                = mine ? nd : __builtin_inf();
        }
        int dx    = (c >= x) ? 1 : -1;  // :71 (ddx == sign(dx) throughout)
        const bool zig = pdu != 0.0;    // :80-89: partdist exactly 0 above: x only grows (is_svp)
        while (nd <= bnd)               // :31 / :93 (NaN fails)
        {
          ++cnt;  // ++nodes[kc]
          // subtree estimate: sum over the levels below of V_{kc-k}(sqrt(R_k^2 - nd)) / prod sqrt(r)
          bool is_heavy = false;
          if (deeper)
          {
            const float rem = (float)(R2k - nd);
            float e         = (lane < kc && rem > 0.f) ? __expf(fminf(Ak + hk * __logf(rem), 60.f)) : 0.f;
            e               = wave_sum_f32(e);
            is_heavy        = (__builtin_amdgcn_readfirstlane((int)(e > heavy)) != 0) | (smode == 2);
          }
          const TaskBuf dst  = is_heavy ? out : fin;
          const unsigned wr  = (emitted++) % FPHIP_NQ;
          unsigned *ctr      = is_heavy ? &qm->bfs[kc][wr * FPHIP_QS] : &qm->fin[wr * FPHIP_QS];
          unsigned oi        = 0;
          if (lane == 0)
            oi = atomicAdd(ctr, 1u);
          oi = (unsigned)__builtin_amdgcn_readfirstlane((int)oi);
          if (oi < rcap)
          {
            const unsigned long long o = (unsigned long long)wr * rcap + oi;
            // S_kc = S_{kc+1} - x[kc] * mu(kc, .)  (:53-58; dual: alpha instead of x, :57-61)
            dst.col[o * 64 + lane] = col - (DUAL ? a : x) * mk;
            dst.x[o * 64 + lane]   = (lane == kc) ? x : xp;
            if (lane == 0)
            {
              dst.pd[o]    = nd;
              dst.level[o] = kc;
              dst.root[o]  = ridu;
            }
          }
          else if (lane == 0)
            atomicOr(&g->error_flags, FPHIP_FLAG_BFS_OVERFLOW);
          // next sibling, :80-92
          if (zig)
          {
            x += (double)dx;
            dx = (dx > 0 ? -1 : 1) - dx;
          }
          else
            x += 1.0;
          a  = x - c;
          nd = pdu + a * a * r;
        }
      }
    }
    if (count_nodes && cnt != 0 && lane == 0)
      atomicAdd(&g->nodes[kc], (unsigned long long)cnt);
    if (s + 1 < nlev)
    {  // single-workgroup mode: the next level reads what this one wrote
      __threadfence();
      __syncthreads();
    }
  }
}
template __global__ void enum_bfs_kernel<false>(DevShared *, double, QueueMem *, TaskBuf, TaskBuf, TaskBuf, int, int,
                                                int, float, int, int, int, int);
template __global__ void enum_bfs_kernel<true>(DevShared *, double, QueueMem *, TaskBuf, TaskBuf, TaskBuf, int, int,
                                               int, float, int, int, int, int);

// ---------------------------------------------------------------------------------------------
// Blocks larger than 64 (up to 256): the levels 64..d-1.  The TOP of the tree is walked with two
// (four above 128 rows) registers per lane (rows / levels 0..127 / 0..255) — the same CHILD / STEP walk and the same arithmetic as
// enum_phase_kernel — in one or two launches of one-wave workgroups pulling "top tasks" (column of
// all rows, coefficients of the levels >= 64 chosen so far, partial distance, root level): the
// first launch walks the root down to a cut level and emits the survivors as top tasks, the second
// walks those in parallel down to level 64, where every surviving node becomes a task for the
// wave-per-subtree kernel (column of the rows below 64, partial distance; the coefficients of
// levels >= 64 are stored once per such node in xhi_root).  No candidate can be reported up here,
// so the top runs under the initial radius (tasks that a later, smaller radius cuts die at their
// first test in the next launch: the visited set is the reference's).
// ---------------------------------------------------------------------------------------------
template <int NQT> __device__ __forceinline__ double rlq(const double (&v)[NQT], int idx)
{
  double r = 0.0;
#pragma unroll
  for (int q = 0; q < NQT; ++q)
    if ((idx >> 6) == q)
      r = rl_f64(v[q], idx & 63);
  return r;
}
template <int NQT> __device__ __forceinline__ int rlqi(const int (&v)[NQT], int idx)
{
  int r = 0;
#pragma unroll
  for (int q = 0; q < NQT; ++q)
    if ((idx >> 6) == q)
      r = rl_i32(v[q], idx & 63);
  return r;
}

// NQT = registers per lane: 2 for blocks up to 128 rows (the column stack of the levels above 64 in LDS,
// 49 KB at d = 128), 4 up to 256 — FPLLL_MAX_ENUM_DIM, enumerate_base.h:59-101 — with the stack in a per-wave
// region of global memory (246 KB at d = 256: more than a CU's LDS; the top of the tree is a vanishing share of
// the nodes).  Row strides of the top task buffers: 64 NQT for the columns, 64 (NQT - 1) for the coefficients of
// the levels >= 64 (TopBuf::xhi); xhi_root's rows have the stride every reader computes from d.
template <int NQT, bool SUBS, bool DUAL>
__global__ void __launch_bounds__(64)
    enum_top_kernel(DevShared *__restrict__ g, HostCtl *__restrict__ h, TopBuf in, unsigned n_in,
                    TopBuf out_top, int stop, TaskBuf out, double *__restrict__ xhi_root, int d,
                    double maxdist, int count_nodes, int launch_idx, double *__restrict__ gtop)
{
  extern __shared__ __attribute__((aligned(16))) double stk_lds[];  // NQT == 2: slots 65..d (slot k: k doubles)
  constexpr int COLS = 64 * NQT, XS = 64 * (NQT - 1);
  const int lane   = threadIdx.x & 63;
  const int off65  = tri_off(65);
  const double *mu = g->mu_tri;
  double *stk2     = (NQT == 2) ? stk_lds : gtop + (size_t)blockIdx.x * (size_t)(tri_off(d + 1) - off65);
  double rd[NQT], bnd[NQT];
#pragma unroll
  for (int q = 0; q < NQT; ++q)
  {
    rd[q]  = g->rdiag[lane + 64 * q];
    bnd[q] = g->pruning[lane + 64 * q] * maxdist;
  }
  unsigned long long cnt[NQT];
  double sb[NQT];  // findsubsols: this wave's view of the best distance per level
#pragma unroll
  for (int q = 0; q < NQT; ++q)
  {
    cnt[q] = 0;
    sb[q]  = 0.0;
    if constexpr (SUBS)
      sb[q] = __longlong_as_double((long long)g->sub_bits[64 * q + lane]);
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
    double S[NQT], xs[NQT], cs[NQT], pds[NQT];
    int dxs[NQT], ddxs[NQT];
    // xs[q >= 1]: lane = level - 64 q; the levels >= Lt come from the task, the walk fills the others
#pragma unroll
    for (int q = 0; q < NQT; ++q)
    {
      S[q]    = in.col[(unsigned long long)t * COLS + 64 * q + lane];
      xs[q]   = q == 0 ? 0.0 : in.xhi[(unsigned long long)t * XS + 64 * (q - 1) + lane];
      cs[q]   = 0.0;
      pds[q]  = 0.0;
      dxs[q]  = 0;
      ddxs[q] = 0;
    }
    // process_subsolution for a node at level lvl >= 64
    auto sub_report = [&](int lvl, double dist)
    {
      unsigned long long old = 0;
      if (lane == 0)
        old = atomicMin(&g->sub_bits[lvl], (unsigned long long)__double_as_longlong(dist));
      old               = rfl_u64(old);
      const double oldd = __longlong_as_double((long long)old);
#pragma unroll
      for (int q = 0; q < NQT; ++q)
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
      SolRec *r  = &h->ring[idx % FPHIP_RING_CAP];
      r->x[lane] = 0.0;
#pragma unroll
      for (int q = 1; q < 4; ++q)
        r->x[64 * q + lane] = (q < NQT && 64 * q + lane < d && 64 * q + lane >= lvl) ? xs[q < NQT ? q : 0] : 0.0;
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
        const double c1 = rlq<NQT>(S, kc);
        const double x1 = round(c1);
        const double a1 = x1 - c1;
        const double n1 = nd + a1 * a1 * rlq<NQT>(rd, kc);
        if (!(n1 <= rlq<NQT>(bnd, kc)))
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
              out.col[(unsigned long long)oi * 64 + lane] = S[0];
              out.x[(unsigned long long)oi * 64 + lane]   = 0.0;
              // (row stride of xhi_root = the readers': 64 per started chunk of levels above 64 — 128 for a
              //  block of 129..192 rows, not this kernel's register count XS = 192)
              const int xstr = 64 * ((d - 1) >> 6);
#pragma unroll
              for (int q = 1; q < NQT; ++q)
                if (64 * (q - 1) < xstr)
                  xhi_root[(unsigned long long)oi * xstr + 64 * (q - 1) + lane] = xs[q];
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
#pragma unroll
              for (int q = 0; q < NQT; ++q)
              {
                out_top.col[(unsigned long long)oi * COLS + 64 * q + lane] = S[q];
                if (q > 0)
                  out_top.xhi[(unsigned long long)oi * XS + 64 * (q - 1) + lane] = xs[q];
              }
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
        for (int q = 0; q < NQT; ++q)
          if (lane + 64 * q < k)
            stk2[tri_off(k) - off65 + lane + 64 * q] = S[q];
        {
          const int s1 = (c1 >= x1) ? 1 : -1;
#pragma unroll
          for (int q = 0; q < NQT; ++q)
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
          if (n1 < rlq<NQT>(sb, kc) && n1 != 0.0)
            sub_report(kc, n1);
        }
        k  = kc;
        nd = n1;  // k >= 64 here
#pragma unroll
        for (int q = 0; q < NQT; ++q)
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
        double par[NQT], mk[NQT];
#pragma unroll
        for (int q = 0; q < NQT; ++q)
        {
          par[q] = stk2[tri_off(k + 1) - off65 + min(lane + 64 * q, k)];
          mk[q]  = mu[tri_off(k) + min(lane + 64 * q, k - 1)];
        }
        double xk        = rlq<NQT>(xs, k);
        const double ck  = rlq<NQT>(cs, k);
        const double pdk = rlq<NQT>(pds, k);
        int dxk = rlqi<NQT>(dxs, k), ddxk = rlqi<NQT>(ddxs, k);
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
        for (int q = 0; q < NQT; ++q)
        {
          const bool me = lane + 64 * q == k;
          xs[q]         = me ? xk : xs[q];
          dxs[q]        = me ? dxk : dxs[q];
          ddxs[q]       = me ? ddxk : ddxs[q];
        }
        const double a = xk - ck;
        nd             = pdk + a * a * rlq<NQT>(rd, k);
        if (!(nd <= rlq<NQT>(bnd, k)))
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
        for (int q = 0; q < NQT; ++q)
          cnt[q] += (lane + 64 * q == k) ? 1ull : 0ull;
        if constexpr (SUBS)
        {
          if (nd < rlq<NQT>(sb, k) && nd != 0.0)
            sub_report(k, nd);
        }
#pragma unroll
        for (int q = 0; q < NQT; ++q)
          S[q] = par[q] - (DUAL ? a : xk) * mk[q];
        break;
      }
    }
  }
  if (count_nodes)
  {
#pragma unroll
    for (int q = 0; q < NQT; ++q)
      if (cnt[q] != 0)
        atomicAdd(&g->nodes[lane + 64 * q], cnt[q]);
  }
}
#define FPHIP_TOP_INST(N, S_, D_)                                                                                  \
  template __global__ void enum_top_kernel<N, S_, D_>(DevShared *, HostCtl *, TopBuf, unsigned, TopBuf, int, TaskBuf, \
                                                      double *, int, double, int, int, double *);
FPHIP_TOP_INST(2, false, false)
FPHIP_TOP_INST(2, true, false)
FPHIP_TOP_INST(2, false, true)
FPHIP_TOP_INST(4, false, false)
FPHIP_TOP_INST(4, true, false)
FPHIP_TOP_INST(4, false, true)
#undef FPHIP_TOP_INST

// Work movement between ranks: tasks [lo, lo + n) of a buffer as contiguous records of FPHIP_TASK_REC (+ xstr)
// doubles — partial distance, root level, column, coefficient prefix and, for a block above 64 rows, the xstr
// coefficients of levels >= 64 of the task's level-64 ancestor (a row of the SENDER's xhi_root table: the receiver
// appends it to its own and points the task there) — and back.  One wave per task.
__global__ void __launch_bounds__(256) task_pack_kernel(TaskBuf in, unsigned lo, unsigned n, double *__restrict__ rec,
                                                        const double *__restrict__ xhi_root, int xstr)
{
  const unsigned recd = FPHIP_TASK_REC + (unsigned)xstr;
  const int lane   = threadIdx.x & 63;
  const unsigned w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const unsigned nw = (gridDim.x * blockDim.x) >> 6;
  for (unsigned t = w; t < n; t += nw)
  {
    const unsigned long long ti = lo + t;
    double *r                   = rec + (unsigned long long)t * recd;
    if (lane == 0)
    {
      r[0] = in.pd[ti];
      r[1] = (double)in.level[ti];
    }
    r[2 + lane]  = in.col[ti * 64 + lane];
    r[66 + lane] = in.x[ti * 64 + lane];
    if (xstr > 0)
    {
      const size_t row = (size_t)in.root[ti] * (size_t)xstr;
      for (int c = lane; c < xstr; c += 64)
        r[FPHIP_TASK_REC + c] = xhi_root[row + c];
    }
  }
}
__global__ void __launch_bounds__(256) task_unpack_kernel(TaskBuf out, unsigned lo, unsigned n, const double *__restrict__ rec,
                                                          double *__restrict__ xhi_root, int xstr, unsigned root_base)
{
  const unsigned recd = FPHIP_TASK_REC + (unsigned)xstr;
  const int lane   = threadIdx.x & 63;
  const unsigned w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const unsigned nw = (gridDim.x * blockDim.x) >> 6;
  for (unsigned t = w; t < n; t += nw)
  {
    const unsigned long long ti = lo + t;
    const double *r             = rec + (unsigned long long)t * recd;
    if (lane == 0)
    {
      out.pd[ti]    = r[0];
      out.level[ti] = (int)r[1];
      out.root[ti]  = xstr > 0 ? (int)(root_base + t) : 0;
    }
    out.col[ti * 64 + lane] = r[2 + lane];
    out.x[ti * 64 + lane]   = r[66 + lane];
    if (xstr > 0)
    {
      const size_t row = (size_t)(root_base + t) * (size_t)xstr;
      for (int c = lane; c < xstr; c += 64)
        xhi_root[row + c] = r[FPHIP_TASK_REC + c];
    }
  }
}

// 64-bit content key of every task (its coefficient prefix x[Lt..d)): the task ORDER in the buffer
// is not deterministic across ranks, the content is.  One wave per task.
__global__ void __launch_bounds__(256)
    task_key_kernel(TaskBuf in, unsigned n, int d, unsigned long long *__restrict__ keys,
                    const double *__restrict__ xhi_root, const unsigned *__restrict__ slots)
{
  const int lane = threadIdx.x & 63;
  const unsigned w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const unsigned nw = (gridDim.x * blockDim.x) >> 6;
  for (unsigned tp = w; tp < n; tp += nw)
  {
    const unsigned ti = slots ? slots[tp] : tp;  // (regioned buffer: the tp-th occupied slot)
    const int Lt      = in.level[ti];
    const double xpre = in.x[(unsigned long long)ti * 64 + lane];
    const bool on     = lane >= Lt && lane < d;
    unsigned h1 = on ? (unsigned)(int)xpre * (2654435761u * (unsigned)(lane + 1)) : 0u;
    unsigned h2 = on ? ((unsigned)(int)xpre ^ 0x9e3779b9u) * (40503u * (unsigned)(2 * lane + 3) + 2246822519u) : 0u;
    const int xstr = d > 64 ? 64 * ((d - 1) >> 6) : 64;
    for (int q = 1; q < 4; ++q)
      if (64 * q + lane < d)
      {  // blocks larger than 64: the coefficients of levels >= 64 (kept once per level-64 ancestor)
        const int lv    = 64 * q + lane;
        const double xh = xhi_root[(unsigned long long)in.root[ti] * xstr + 64 * (q - 1) + lane];
        h1 += (unsigned)(int)xh * (2654435761u * (unsigned)(lv + 1));
        h2 += ((unsigned)(int)xh ^ 0x9e3779b9u) * (40503u * (unsigned)(2 * lv + 3) + 2246822519u);
      }
    for (int off = 32; off > 0; off >>= 1)
    {
      h1 += (unsigned)__shfl_xor((int)h1, off);
      h2 += (unsigned)__shfl_xor((int)h2, off);
    }
    if (lane == 0)
      keys[tp] = ((unsigned long long)h1 << 32) | h2;
  }
}

}  // namespace fphip
