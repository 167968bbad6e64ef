// enum_walk.hip — the walk launches of the enumeration, second generation: ALL children of a node in one
// vector test.
//
// Reference behaviour reproduced (fplll v5.5.0): EnumerationBase::enumerate_recursive,
// fplll/enum/enumerate_base.cpp:24-118 (centre, roundto, zig-zag order :80-92, bound test :31 / :93, node
// counting :33, the dual recursion :57-61 / :103-105) with the bounds of EnumerationDyn
// (enumerate.cpp:218-239).  Same visit set, same arithmetic per node (separate multiply / add, the reference's
// operand order), same per-level counts and candidates as enum_phase_kernel (enum_kernel.hip) — which stays the
// kernel of the split launches, of sub-solution calls and the A/B partner of this one (FPHIP_WALK2=0).
//
// What changed, and why.  enum_phase_kernel visits the children of a node one test at a time: the first child in
// its CHILD chain, every later sibling in its STEP loop, and every node's sibling sequence ENDS with a failing
// test — per counted node one successful and one failing test, 40.5 vector + 38.8 scalar / branch instructions,
// 0.81 of the vector issue port (profiles/r05_enum_pmc_summary.txt).  The siblings of a node, however, are a
// function of three wave-uniform numbers (the centre c, the parent's distance, r_kk): child j of the zig-zag has
// x_j = round(c) + z_j, dist_j = pd + (x_j - c)^2 r, and the sequence of distances is non-decreasing.  This
// kernel evaluates the 64 first candidates of a node in the 64 LANES — one fma, one subtract, two multiplies, one
// add, one compare — and reads the number of surviving children off the ballot (s_ff1 of its complement): the
// failing test is gone and the later siblings need no test at all (STEP: the next index of the stored count).
// Every dist_j is produced by the reference's operation sequence, and a node is counted when it is visited, so
// the visit set and the counts are the reference's bit for bit.
//
//   EXPAND(k):  node at level k (column S_k, distance nd).  c = S_k[k-1]; x_0 = roundto(c); lanes j: x_j, dist_j;
//               m = ballot(dist_j <= pruning_{k-1} maxdist); n = ctz(~m).  n = 0: STEP(k).  Else level registers
//               (lane k-1): c, x_0, pd = nd, st = (i = 0, sign of the first step, n); push S_k; descend into child 0
//               (++nodes[k-1]; its distance = lane 0 of dist_j through the LDS crossbar).
//   STEP(k):    st_k.i + 1 < st_k.n: x = x_0 + z(i) (z from a 2 KB table through the scalar cache, as a double),
//               dist = pd + (x - c)^2 r, S_k = S_{k+1} - x mu_k, EXPAND(k).  Else climb: STEP(k + 1).
//
// Rare cases leave the hot cycle through ONE bit test (smask, as in enum_phase_kernel) or a flag in st:
//   * emission levels (work donation), the children of a level-1 node (leaves: candidates are reported one by
//     one, each may lower the bound for its next sibling, :97-101);
//   * the chain of first children below a root of distance exactly 0 (is_svp: x only grows there, :80-89 — by
//     symmetry those children are the odd indices of the zig-zag over the same centre 0: flag `grow`);
//   * a node with more than 63 surviving children (flag `more`: the 64th onwards are tested one by one).
// A bound that shrinks (a candidate was found) re-tests the not-yet-visited siblings of the active levels and
// shortens their counts (reprune): a sibling is never visited under a bound it fails, to the granularity at which
// waves learn of a new bound — enum_phase_kernel's contract as well.
//
// Build: like enum_kernel.hip (-ffp-contract=off; -structurizecfg-skip-uniform-regions, no lifetime markers).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "enum_device.h"
#include "enum_wave.h"

namespace fphip
{

// st: the sibling state of one level (lane = level)
//   hot level     bits 0-6: i, the index of the current child in zig-zag order (0..61); bits 8-14: n, the number of
//                 surviving children (1..62); the direction of the first step is not stored: it is c >= x_0 again
//   slow level    (st & 0x7fff) == ST_MARK (i = 100 < n = 127: "a sibling is left" for the hot test, which sends
//                 every i >= 64 to the general path); bits 17-31: iw, the index of the current child.  Levels of the
//                 zero chain, levels with 63+ surviving children — and lane Lt & 63, the task root's: the climb
//                 that reaches it ends the task
//   0             nothing left at this level
#define ST_I(s) ((s)&0x7f)
#define ST_N(s) (((s) >> 8) & 0x7f)
#define ST_MARK 0x7f64
#define ST_IS_SLOW(s) (((s)&0x7fff) == ST_MARK)
#define ST_IW(s) ((int)((unsigned)(s) >> 17))

// z(i): 0, +1, -1, +2, -2, ... (first step up), negated when the first step goes down (:80-92)
__device__ __forceinline__ int zig_of(int i, bool down)
{
  const int hh = (i + 1) >> 1;
  const int z  = (i & 1) ? hh : -hh;
  return down ? -z : z;
}

// r_kk of one level through the scalar cache (the bound of the level comes out of a lane register here)
__device__ __forceinline__ v2u r_issue(const double *tab, unsigned off)
{
  v2u q;
  asm volatile("s_load_dwordx2 %0, %1, %2" : "=s"(q) : "s"(tab), "s"(off));
  return q;
}
__device__ __forceinline__ void rp_wait(v2u &q) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(q)); }
__device__ __forceinline__ double rp_r2(const v2u &q) { return __hiloint2double((int)q.y, (int)q.x); }

template <bool MU_LDS, bool DUAL>
__global__ void __launch_bounds__(FPHIP_MAX_BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8)))
    enum_walk_kernel(DevShared *__restrict__ g, HostCtl *__restrict__ h, TaskBuf in, TaskBuf out,
                     int d, int Lmax, unsigned task_lo, unsigned task_hi,
                     const unsigned *__restrict__ idxlist, int launch_idx, int count_nodes,
                     unsigned budget, const double *__restrict__ xhi_root, double *__restrict__ gstk,
                     int Tsplit, unsigned *__restrict__ qh, const unsigned *__restrict__ rcnt, unsigned rcap,
                     unsigned long long bound_init)
{
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (wave-uniform: the per-wave bases in SGPRs)
  constexpr unsigned MUROW8 = FPHIP_MUROW * 8u;
  const unsigned lane8 = (unsigned)lane << 3;
  const int triL = (Lmax * (Lmax + 1)) >> 1;
  // LDS: [mu rows (MU_LDS)][per-wave column stacks] — the stack layout and the split into an LDS part (slots
  // below Ts) and a global part are enum_phase_kernel's (see there)
  const double *mu_s;
  const int nw     = (int)(blockDim.x >> 6);
  const int Ts     = min(Tsplit, Lmax + 1);
  const int Tsm1   = Ts - 1;
  const int ldsRow = tri_off(Ts);
  const int ldsWave = ldsRow + FPHIP_STACK_PAD;
  double *stk;
  if constexpr (MU_LDS)
  {
    double *mu_l = smem;
    stk          = smem + triL + wave * ldsWave;
    const int nmu = (Lmax * (Lmax - 1)) >> 1;
    for (int i = threadIdx.x; i < nmu; i += blockDim.x)
      mu_l[i] = g->mu_tri[i];
    mu_s = mu_l;
  }
  else
  {
    mu_s = &g->mu_sq[0][0];
    stk  = smem + wave * ldsWave;
  }
  if constexpr (MU_LDS)
    __syncthreads();
  const __amdgpu_buffer_rsrc_t mu_b = mu_rsrc(&g->mu_sq[0][0], (unsigned)sizeof(g->mu_sq));
  char *stk_top      = (char *)stk + (((unsigned)ldsRow << 3) + lane8);
  double *gst = gstk + (size_t)(blockIdx.x * nw + wave) * (size_t)(triL - ldsRow + 1) - ldsRow;
  const double *rptab = &g->mu_sq[0][64];
  // z_j of this lane's candidate of an expansion (first step up).  Lane 63 holds a NaN: its candidate never
  // passes, so the complement of the ballot always has a bit set and 63 surviving candidates mean "maybe more"
  double zz = lane == 63 ? __builtin_nan("") : (double)zig_of(lane, false);
  FPHIP_IN_VGPR(zz);
  int zero_a = 0;  // the address operand "lane 0" of a bpermute (opaque: a constant index becomes two v_readlane)
  asm volatile("" : "+v"(zero_a));

  unsigned long long mbits =
      rfl_u64(min(bound_init, __hip_atomic_load(&g->bound_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
  double maxdist   = __longlong_as_double((long long)mbits);
  double maxdist_v = maxdist;
  FPHIP_IN_VGPR(maxdist_v);

  // level registers (lane = level)
  double cs = 0.0, x0s = 0.0, pds = 0.0;
  int st = 0;
  unsigned long long cnt = 0;
  unsigned cnt32         = 0;
  unsigned iter          = 0;
  // steps until the next refresh event (the bound, the donation test).  A step is taken once per ~3 nodes: 24 steps
  // are the 64 failed steps of enum_phase_kernel in nodes — a wave must not run longer than that on a stale bound
  constexpr int RF = 24;
  int left               = RF - 1;

  // (bchg: the bound went down — the caller re-tests the pending siblings, see reprune)
#define FPHIP_REFRESH_BOUND(from_host, bchg)                                                      \
  do                                                                                              \
  {                                                                                               \
    unsigned long long nb_;                                                                       \
    if (from_host)                                                                                \
    {                                                                                             \
      nb_ = load_sys_u64(&h->bound_bits);                                                         \
      if (nb_ < mbits && lane == 0)                                                               \
        atomicMin(&g->bound_bits, nb_);                                                           \
      FPHIP_JOIN();                                                                               \
    }                                                                                             \
    else                                                                                          \
    {                                                                                             \
      nb_ = __hip_atomic_load(&g->bound_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        \
    }                                                                                             \
    nb_ = rfl_u64(nb_);                                                                           \
    if (nb_ < mbits)                                                                              \
    {                                                                                             \
      mbits   = nb_;                                                                              \
      maxdist = __longlong_as_double((long long)mbits);                                           \
      maxdist_v = maxdist;                                                                        \
      FPHIP_IN_VGPR(maxdist_v);                                                                   \
      bchg = true;                                                                                \
    }                                                                                             \
  } while (0)

  unsigned q = (unsigned)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * nw + wave) % FPHIP_NQ));
  const unsigned nlist = task_hi - task_lo;
  for (;;)
  {
    // ---- pull a task (the queues of enum_phase_kernel) ----------------------------------------
    unsigned t = 0, cq = 0;
    bool have = false;
    for (;;)
    {
      if (lane == 0)
        t = atomicAdd(&qh[q * FPHIP_QS], 1u);
      t  = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
      cq = rcnt ? min((unsigned)__builtin_amdgcn_readfirstlane((int)rcnt[q * FPHIP_QS]), rcap)
                : (nlist > q ? (nlist - q + FPHIP_NQ - 1u) / FPHIP_NQ : 0u);
      if (t < cq)
      {
        have = true;
        break;
      }
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
    {
      if (budget != 0u && lane == 0)
        __hip_atomic_store(&g->drain[launch_idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      FPHIP_JOIN();
      break;
    }
    const unsigned long long pos = (unsigned long long)task_lo + q + (unsigned long long)t * FPHIP_NQ;
    const unsigned long long ti =
        rcnt ? (unsigned long long)q * rcap + (cq - 1u - t)
             : (idxlist ? (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)idxlist[pos]) : pos);
    const int Lt      = __builtin_amdgcn_readfirstlane(in.level[ti]);
    const int rid     = __builtin_amdgcn_readfirstlane(in.root[ti]);
    const int tl      = here_lane(lane);
    const double xpre = in.x[ti * 64 + tl];
    const double col0 = in.col[ti * 64 + tl];
    const double pd0  = in.pd[ti];
    int donate        = 1 << 20;
    const unsigned iter0 = iter;
    {
      bool bchg = false;
      FPHIP_REFRESH_BOUND((t & 63u) == 0u, bchg);
    }

    int k     = Lt;
    double S  = col0;
    double nd = pd0;
    // the chain of first children below a root of distance exactly 0 goes through the general path (slow
    // levels) until the first step away from it: every level is "special" while zc holds
    bool zc = __builtin_amdgcn_ballot_w64(pd0 != 0.0) == 0ull;
    // the climb that reaches the root's level ends the task: a slow marker in its lane (lane 0 when Lt = 64 —
    // level 0 keeps no sibling state: its nodes are the leaf loop's)
    st = (lane == (Lt & 63)) ? ST_MARK : st;

    // the coefficients of the current path (lane = level): x_0 + z(i) of each level's sibling state
    auto xs_now = [&]() -> double
    {
      const int idx = (lane == 0) ? 0 : (ST_IS_SLOW(st) ? ST_IW(st) : ST_I(st));
      return x0s + (double)zig_of(idx, !(cs >= x0s));
    };

    auto report = [&](double dist, bool &bchg)
    {
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
      double xf  = (lane < Lt) ? xs_now() : xpre;
      const int rl = here_lane(lane);
      r->x[rl]   = (lane < d) ? xf : 0.0;
      {
        const int xstr = d > 64 ? 64 * ((d - 1) >> 6) : 64;
#pragma unroll
        for (int qq = 1; qq < 4; ++qq)
          r->x[64 * qq + rl] = (64 * qq + lane < d) ? xhi_root[(size_t)rid * xstr + 64 * (qq - 1) + rl] : 0.0;
      }
      if (lane == 0)
      {
        const int z = here_lane(0);
        r->dist     = dist;
        r->kind     = z;
        r->offset   = z;
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
      FPHIP_REFRESH_BOUND(true, bchg);
    };

    // The bound went down: the pending siblings of the hot levels in [klo, Lt) — inside the count n of their
    // expansion, not visited yet — are tested again, from the last one downwards (the distances are non-decreasing
    // along the zig-zag); n shrinks past those that fail now: the reference would meet them under the new bound and
    // turn back (:93).  Lane = level: every lane handles the state of its own level.  (Slow levels test their
    // siblings one by one anyway.)
    auto reprune = [&](int klo)
    {
      int s_           = st;
      const bool act   = lane >= klo && lane < Lt && !ST_IS_SLOW(s_);
      const double r_l = g->rdiag[lane];
      const double b_l = g->pruning[lane] * maxdist;
      const int cur    = ST_I(s_);
      const bool down  = !(cs >= x0s);
      int n            = ST_N(s_);
      for (;;)
      {
        const int j      = n - 1;
        const double xj  = x0s + (double)zig_of(j, down);
        const double aj  = xj - cs;
        const double ndj = pds + aj * aj * r_l;
        const bool fail  = act && j > cur && !(ndj <= b_l);
        if (__builtin_amdgcn_ballot_w64(fail) == 0ull)
          break;
        if (fail)
          n -= 1;
      }
      s_ = (s_ & ~(0x7f << 8)) | (n << 8);
      st = act ? s_ : st;
      FPHIP_JOIN();
    };

    enum : int { EV_EMIT = 1, EV_DONE = 3, EV_OK = 5, EV_REFRESH = 6, EV_SPECIAL = 7, EV_FAIL = 8, EV_SLOWSTEP = 10 };
    unsigned elo  = (unsigned)donate;
    unsigned erng = 0x7fffffffu;
    auto hot_range = [&]() {};  // (the special levels are recomputed from elo / zc at the top of the event loop)
    hot_range();
    bool buffer_full = false;
    bool resume_step = false;
    double par = 0.0, mk = 0.0;
    double xk = 0.0, a = 0.0;
    int kc = 0;
    double mk1 = 0.0, c1 = 0.0, x1 = 0.0, a1 = 0.0;
#define FPHIP_PUSH(in_lds, kk, lds8)                                            \
  if (__builtin_expect(in_lds, 1))                                              \
    *(double *)(stk_top - (lds8)) = S;                                          \
  else                                                                          \
  {                                                                             \
    int kt = (kk);                                                              \
    asm volatile("" : "+s"(kt));                                                \
    const unsigned gk8 = ((unsigned)(kt * (kt - 1)) << 2) + lane8;              \
    *(double *)((char *)gst + ((lane < kt) ? gk8 : (unsigned)triL << 3)) = S;   \
  }
    // (par, mk) = (S_{k+1}, row k of mu) of level k: what a step needs to build S_k
#define FPHIP_LOAD_PAR_MK()                                                     \
  do                                                                            \
  {                                                                             \
    const unsigned k8_  = (unsigned)k << 3;                                     \
    const unsigned cl8_ = min(lane8, k8_ - 8u);                                 \
    if (k + 1 < Ts)                                                             \
      par = *(const double *)(stk_top - tri8(k + 2));                           \
    else                                                                        \
      par = ld_off(gst, tri8(k + 1) + cl8_);                                    \
    if constexpr (MU_LDS)                                                       \
      mk = ld_off(mu_s, tri8(k) + cl8_);                                        \
    else                                                                        \
      mk = ld_row(mu_b, (unsigned)k * MUROW8, lane8);                           \
  } while (0)
    for (;;)
    {
      int ev;
      bool at_step = resume_step;
      resume_step  = false;
      // special levels as ONE unsigned compare of the child level: kc - 1 >= spec_t  <=>  kc = 0 (level 1: leaves
      // below) or kc + 1 >= elo (emission levels); spec_t = 0 on the zero chain (every level)
      unsigned spec_t = zc ? 0u : (elo >= 2u ? elo - 2u : 0u);
      spec_t          = (unsigned)__builtin_amdgcn_readfirstlane((int)spec_t);
      int ka = lane_addr(k);  // 4 * level in a VGPR: the address operand of a level's bpermutes, carried
      // ---- the hot cycle: EXPAND chain -> (a node without children) -> STEP loop -> (a sibling is left) ->
      // EXPAND chain ...  Uniform branches only.
      for (;;)
      {
        if (!at_step)
        {
          // ================= EXPAND chain: all children of the node at level k, descend into the first ====
          // (state: a counted node at level k with column S = S_k, distance nd; ka = 4 k)
          kc = k - 1;
          for (;;)
          {
            asm volatile("" : "+s"(kc));
            const unsigned kc8 = (unsigned)kc << 3;
            v4i q1             = rp_issue2(rptab, (unsigned)kc * MUROW8);
            if constexpr (MU_LDS)
              mk1 = ld_off(mu_s, tri8(kc) + min(lane8, kc8 - 8u));
            else
              mk1 = ld_row(mu_b, (unsigned)kc * MUROW8, lane8);
            ka -= 4;
            c1 = bp_f64(S, ka);  // center[kk-1]
            x1 = rint(c1);
            a1 = x1 - c1;
            if (__builtin_amdgcn_ballot_w64(fabs(a1) == 0.5) != 0ull)
            {  // roundto(): ties away from zero
              asm volatile("");
              const bool fix = (a1 < 0.0) == (c1 > 0.0);
              x1             = fix ? x1 - (a1 + a1) : x1;
              a1             = fix ? -a1 : a1;
            }
            // candidate of lane j: x_0 + (0, +1, -1, +2, -2, ... +31, -31) — a set symmetric about x_0, so the
            // survivors are the first n of the reference's zig-zag WHICHEVER way its first step goes (:71 / :114:
            // the step that picks a child asks again); the distance by the reference's sequence (:28-29 / :91-92)
            const double xj  = x1 + zz;
            const double aj  = xj - c1;
            rp_wait(q1);
            const double bnd = rp_p(q1) * maxdist_v;  // partdistbounds[kk-1], enumerate.cpp:218-228
            const double ndj = nd + aj * aj * rp_r(q1);
            const unsigned long long m = __builtin_amdgcn_ballot_w64(ndj <= bnd);
            if (m == 0ull)
            {  // no surviving child: next sibling at level k
              ev = EV_FAIL;
              FPHIP_EXIT();
              break;
            }
            if ((unsigned)(kc - 1) >= spec_t)
            {  // an emission level, level 1 or the zero chain: the general path
              ev = EV_SPECIAL;
              FPHIP_EXIT();
              break;
            }
            int n = __builtin_popcountll(m);  // (lane 63 never passes)
            asm("" : "+s"(n));
            if (n == 63)
            {  // 63+ children: the general path
              ev = EV_SPECIAL;
              FPHIP_EXIT();
              break;
            }
            // ---- descend into child 0 (++nodes[kk-1]): S_k is needed again when x[kc] steps to a sibling
            FPHIP_PUSH(kc < Ts - 1, kc + 1, tri8(kc + 2));
            const unsigned long long me = lane_bit(kc);
            cs    = sel_f64(me, c1, cs);
            x0s   = sel_f64(me, x1, x0s);
            pds   = sel_f64(me, nd, pds);
            st    = wl_i32(n << 8, kc, st);
            cnt32 = add_bit(me, cnt32);
            nd    = bp_f64(ndj, zero_a);  // lane 0: the first child's distance
            S     = S - (DUAL ? a1 : x1) * mk1;
            --kc;
          }
          k = kc + 1;
          ka += 4;
          if (ev != EV_FAIL)
            break;  // EV_SPECIAL
          // no surviving child: the next sibling at level k (at the task root: its marker ends the task)
        }
        at_step = false;
        // ================= STEP loop: the next child at level k, climbing while levels are exhausted ========
        int tk;
        for (;;)
        {
          tk = rl_i32(st, k) + 1;
          if (__builtin_expect((tk & 0x7f) < ((tk >> 8) & 0x7f), 1))
            break;
          ++k;
          ka += 4;
        }
        if (__builtin_expect((tk & 0x7f) >= 64, 0))
        {  // a slow level, or the task root's marker
          ev = EV_SLOWSTEP;
          break;
        }
        if (__builtin_expect(--left < 0, 0))
        {  // every RF steps
          ev = EV_REFRESH;
          break;
        }
        {
          // child tk.i of level k: x = x_0 + z(i), its distance (:91-92), the column of its node (:104-110)
          v2u qk = r_issue(rptab, (unsigned)k * MUROW8);
          if (__builtin_expect(k < Tsm1, 1))
            par = *(const double *)(stk_top - tri8(k + 2));
          else
          {
            int kt = k;
            asm volatile("" : "+s"(kt));
            par = ld_off(gst, tri8(kt + 1) + min(lane8, ((unsigned)kt << 3) - 8u));
          }
          if constexpr (MU_LDS)
            mk = ld_off(mu_s, tri8(k) + min(lane8, ((unsigned)k << 3) - 8u));
          else
            mk = ld_row(mu_b, (unsigned)k * MUROW8, lane8);
          int zi;  // z(i) of the zig-zag with the first step up: +-ceil(i / 2) on the scalar unit
          {
            const int ii = tk & 0x7f;
            const int hh = (ii + 1) >> 1;
            zi           = (ii & 1) ? hh : -hh;
            asm("" : "+s"(zi));
          }
          const double zd  = (double)zi;
          const double x0  = bp_f64(x0s, ka);
          const double ck  = bp_f64(cs, ka);
          const double pk  = bp_f64(pds, ka);
          st               = wl_i32(tk, k, st);
          cnt32            = add_bit(lane_bit(k), cnt32);  // ++nodes[kk]
          const double sgd = __hiloint2double((ck >= x0) ? 0x3ff00000 : (int)0xbff00000, 0);
          xk               = __builtin_fma(zd, sgd, x0);
          a                = xk - ck;
          rp_wait(qk);
          nd = pk + a * a * rp_r2(qk);
          S  = par - (DUAL ? a : xk) * mk;
        }
      }
      // ---- events
      FPHIP_OPAQUE(ev);
      if (ev == EV_SPECIAL)
      {
        if ((unsigned)(k - elo) <= erng)
          ev = EV_EMIT;
        else if (k == 1)
        {
          // the children of a level-1 node are leaves (:97-101): one by one, a reported candidate may lower
          // the bound its next sibling is tested against
          bool bchg       = false;
          const double c0 = rl_f64(S, 0);
          double x = rint(c0), al = x - c0;
          if (fabs(al) == 0.5 && ((al < 0.0) == (c0 > 0.0)))
          {
            x  = x - (al + al);
            al = -al;
          }
          int dx         = (c0 >= x) ? 1 : -1;
          const bool zig = __builtin_amdgcn_ballot_w64(nd != 0.0) != 0ull;
          const double r0 = g->rdiag[0], p0 = g->pruning[0];
          for (;;)
          {
            const double ndc = nd + al * al * r0;
            if (__builtin_amdgcn_ballot_w64(ndc <= p0 * maxdist) == 0ull)
              break;
            cnt += (lane == 0) ? 1ull : 0ull;
            if (__builtin_amdgcn_ballot_w64(ndc > 0.0) != 0ull)
            {
              x0s = (lane == 0) ? x : x0s;
              cs  = (lane == 0) ? x : cs;
              report(ndc, bchg);
              FPHIP_JOIN();
            }
            if (zig)
            {
              x += (double)dx;
              dx = (dx > 0 ? -1 : 1) - dx;
            }
            else
              x += 1.0;
            al = x - c0;
          }
          if (bchg)
            reprune(1);
          resume_step = true;
          continue;
        }
        else
        {
          // the expansion by hand of a slow level: the zero chain, 63+ surviving children, or a level the mask holds
          // for no reason any more.  Only the first child is established here; its siblings are the slow step's.
          const bool nz = __builtin_amdgcn_ballot_w64(nd != 0.0) != 0ull;
          if (zc && nz)
          {
            zc = false;
            hot_range();
          }
          kc              = k - 1;
          const double rk = g->rdiag[kc], pk = g->pruning[kc];
          if constexpr (MU_LDS)
            mk1 = ld_off(mu_s, tri8(kc) + min(lane8, ((unsigned)kc << 3) - 8u));
          else
            mk1 = ld_row(mu_b, (unsigned)kc * MUROW8, lane8);
          c1 = rl_f64(S, kc);
          x1 = rint(c1);
          a1 = x1 - c1;
          if (fabs(a1) == 0.5 && ((a1 < 0.0) == (c1 > 0.0)))
          {
            x1 = x1 - (a1 + a1);
            a1 = -a1;
          }
          const double nd1 = nd + a1 * a1 * rk;
          if (__builtin_amdgcn_ballot_w64(nd1 <= pk * maxdist) == 0ull)
          {  // (the bound moved between the hot test and this one: the child is gone)
            resume_step = true;
            continue;
          }
          FPHIP_PUSH(k < Ts, k, tri8(k + 1));
          cs  = (lane == kc) ? c1 : cs;
          x0s = (lane == kc) ? x1 : x0s;
          pds = (lane == kc) ? nd : pds;
          st  = (lane == kc) ? ST_MARK : st;  // (iw = 0)
          cnt += (lane == kc) ? 1ull : 0ull;  // ++nodes[kk-1]: the first child
          nd  = nd1;
          S   = S - (DUAL ? a1 : x1) * mk1;
          k   = kc;
          FPHIP_JOIN();
          continue;  // -> EXPAND at the child
        }
      }
      if (ev == EV_SLOWSTEP)
      {
        if (k >= Lt)
          break;  // the root's marker: task done
        // the next child of a slow level, one test per child as the reference has it (:80-94): only the odd
        // indices of the zig-zag (x only grows) where the parent's distance is exactly 0
        const int stk_  = rl_i32(st, k);
        const int cur   = ST_IW(stk_);
        const double x0 = rl_f64(x0s, k), ck = rl_f64(cs, k), pk = rl_f64(pds, k);
        const double rk = g->rdiag[k], pr = g->pruning[k];
        const bool grow = __builtin_amdgcn_ballot_w64(pk != 0.0) == 0ull;
        const bool down = __builtin_amdgcn_ballot_w64(ck >= x0) == 0ull;
        const int nxt   = grow ? (cur == 0 ? 1 : cur + 2) : cur + 1;
        xk              = x0 + (double)zig_of(nxt, down);
        a               = xk - ck;
        const double ndn = pk + a * a * rk;
        const bool ok    = nxt < 32760 && __builtin_amdgcn_ballot_w64(ndn <= pr * maxdist) != 0ull;
        if (!ok)
        {  // exhausted: the level above steps
          st = (lane == k) ? 0 : st;
          ++k;
          resume_step = true;
          FPHIP_JOIN();
          continue;
        }
        st = (lane == k) ? (ST_MARK | (nxt << 17)) : st;
        cnt += (lane == k) ? 1ull : 0ull;  // ++nodes[kk]
        FPHIP_LOAD_PAR_MK();
        nd = ndn;
        S  = par - (DUAL ? a : xk) * mk;
        FPHIP_JOIN();
        continue;  // -> EXPAND at level k
      }
      if (ev == EV_EMIT)
      {
        unsigned oi = 0;
        if (lane == 0)
          oi = atomicAdd(out.count, 1u);
        oi = (unsigned)__builtin_amdgcn_readfirstlane((int)oi);
        if (oi < out.cap)
        {
          const int el                              = here_lane(lane);
          out.col[(unsigned long long)oi * 64 + el] = S;
          const double xf                           = (lane < Lt) ? xs_now() : xpre;
          out.x[(unsigned long long)oi * 64 + el]   = xf;
          if (lane == 0)
          {
            out.pd[oi]    = nd;
            out.level[oi] = k;
            out.root[oi]  = rid;
          }
          FPHIP_JOIN();
        }
        else
        {
          if (lane == 0)
            atomicOr(&g->error_flags, FPHIP_FLAG_TASK_OVERFLOW);
          FPHIP_JOIN();
          buffer_full = true;
          donate      = 1 << 20;
          elo         = 1u << 20;
          erng        = 0u;
          hot_range();
          continue;
        }
      }
      else if (ev == EV_REFRESH)
      {
        cnt += cnt32;
        cnt32 = 0u;
        iter += 64u;  // (in the units of enum_phase_kernel's budgets: RF steps are about 64 of its failed steps)
        left = RF - 1;
        bool bchg = false;
        FPHIP_REFRESH_BOUND((iter & 16383u) == 0u, bchg);  // (the pinned host word every 256 refreshes)
        if (bchg)
          reprune(k);
        const unsigned titer = iter - iter0;
        // (a task may shed work from its first refresh on — RF steps, each step closing a chain of at most 64
        //  nodes: about the 256 failed steps enum_phase_kernel waits for)
        if (budget != 0u && titer >= 64u && !buffer_full)
        {
          const unsigned dr = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(
              &g->drain[launch_idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          if (dr != 0u || titer >= budget)
          {
            donate = min(donate, k + 1);
            elo    = (unsigned)donate;
          }
        }
        FPHIP_JOIN();
        hot_range();
      }
      resume_step = true;
    }
#undef FPHIP_PUSH
#undef FPHIP_LOAD_PAR_MK
    cnt += cnt32;
    cnt32 = 0u;
  }
#undef FPHIP_REFRESH_BOUND

  if (count_nodes && cnt != 0)
    atomicAdd(&g->nodes[lane], cnt);
  if (lane == 0)
    atomicAdd(&g->iters, (unsigned long long)iter);
}

#define FPHIP_INST(M, D)                                                                                \
  template __global__ void enum_walk_kernel<M, D>(DevShared *, HostCtl *, TaskBuf, TaskBuf, int, int,   \
                                                  unsigned, unsigned, const unsigned *, int, int,       \
                                                  unsigned, const double *, double *, int, unsigned *,   \
                                                  const unsigned *, unsigned, unsigned long long);
FPHIP_INST(true, false)
FPHIP_INST(false, false)
FPHIP_INST(true, true)
FPHIP_INST(false, true)
#undef FPHIP_INST

}  // namespace fphip
