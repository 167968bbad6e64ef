// bkzs_kernel.hip — BKZ with strategies on the device, and its DUAL blocks for self-dual BKZ
// (BKZ_SD_VARIANT): sd_tour = trunc_dtour + trunc_tour (fplll/bkz.cpp:401-413, 443-463), dual
// svp_reduction (:274-358: radius 1/r of the LAST row of the block, progress test reversed), the
// dual enumeration (EnumerationDyn's transformation enum/enumerate.cpp:96-123,154-158 and the
// dualenum recursion enum/enumerate_base.cpp:57-61,103-105), the dual insertions
// (bkz.cpp:148-193, 240-248), the prelude lll() (:576-577) and the closing hkz() (:627-641).
//
// ONE schedule, two kernels: bkzs_body<NQ, DUALS> is the whole of BKZ with strategies (this file was
// bkzs_kernel.hip + bkzd_kernel.hip in round 1, 80 % identical lines); bkzs_kernel = DUALS false
// (primal BKZ), bkzd_kernel = DUALS true (self-dual BKZ).  Everything the dual blocks
// add sits behind `if constexpr (DUALS)` / cur_dual().

// Reference behaviour reproduced:
//   BKZReduction::bkz / tour / trunc_tour / hkz   fplll/bkz.cpp:522-668, 360-441
//   svp_reduction (primal)                        bkz.cpp:274-358
//   svp_preprocessing (recursive tours)           bkz.cpp:100-124
//   get_pruning, GH bound                         bkz.cpp:82-98, 319-323  (decided by the HOST, below)
//   rerandomize_block                             bkz.cpp:43-80           (plan drawn by the HOST)
//   svp_postprocessing / _generic                 bkz.cpp:126-272
//   EnumerationDyn::enumerate + enumerate_recursive with pruning bounds
//                                                 enum/enumerate.cpp:58-159,218-239,
//                                                 enum/enumerate_base.cpp:24-118
//
// Two decisions of the reference cannot be taken on the device without changing results:
//   * the radius and the choice of the pruning set go through log() of every r_ii of the block and
//     one exp() (MatGSOInterface::get_root_det, gso_interface.cpp:220-242) — the HOST libm's
//     roundings; a radius that differs in the last bit changes which nodes pass the bound test;
//   * rerandomize_block draws from the caller's GMP generator (rejection sampling inside
//     gmp_urandomm_ui: data-independent, but a host-library stream).
// Both are served by the host through a per-lattice mailbox in pinned, host-coherent memory
// (BkzMail, gso_device.h): the wave writes the r_ii of the block (or the row range to
// rerandomise), bumps req_seq and waits for rsp_seq — the protocol of the enumeration kernel's
// solution ring (enum_kernel.hip `report`).  Everything else (the recursion of preprocessing
// tours, LLL, pruned enumeration, insertion) stays in the wave.
//
// The recursion tour -> svp_reduction -> svp_preprocessing -> tour is an explicit stack of frames
// in LDS (device recursion would need scratch memory, which the DMA ring's vmcnt accounting
// excludes): block sizes strictly decrease, the host rejects strategies that nest deeper than
// FPHIP_BKZS_MAX_DEPTH.

#include "lll_wave.h"

namespace fphip
{
namespace sdv
{

__device__ __forceinline__ int btri2(int k) { return (k * (k - 1)) >> 1; }
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(1))) double glb_f64;
typedef __attribute__((address_space(1))) char glb_char;
// v_writelane_b32 through the LLVM intrinsic (no clang builtin in this toolchain)
extern "C" __device__ int fphip_bk_llvm_writelane(int, int, int) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ int bk_wl_i32(int val, int lane, int old)
{
  return fphip_bk_llvm_writelane(__builtin_amdgcn_readfirstlane(val), __builtin_amdgcn_readfirstlane(lane), old);
}
__device__ __forceinline__ double bk_wl_f64(double val, int lane, double old)
{
  const int lo = bk_wl_i32(__double2loint(val), lane, __double2loint(old));
  const int hi = bk_wl_i32(__double2hiint(val), lane, __double2hiint(old));
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ unsigned long long mail_load_u64(const unsigned long long *p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ int mail_load_i32(const int *p)
{
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double mail_load_f64(const double *p)
{
  return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_SYSTEM));
}

// frame of one tour() activation (wave-uniform scalars, kept in LDS while a child tour runs)
struct BkzsFrame
{
  int bsz, flags, min_row, max_row;  // BKZParam::block_size / flags, tour(min_row, max_row)
  int op;                            // index into trunc blocks, hkz blocks, trailing size reduction
  int phase;
  int kappa, bs;                     // the svp_reduction in progress
  int pre_i;                         // next preprocessing block size (index into BkzStrat::pre)
  int rerand;
  int clean;
  int old_expo;
  double old_first, remaining;
};

enum
{
  PH_OP_BEGIN = 0,
  PH_SR,         // lll_obj.size_reduction(sr_kmin, sr_kend, sr_start), then phase = sr_next
  PH_OPENED,     // svp_reduction after its opening size reduction
  PH_LOOP_HEAD,
  PH_PRE,
  PH_ENUM,
  PH_FINISH,     // closing size reduction requested
  PH_CLOSED,     // progress test after the closing size reduction
};

// a frame comes back from LDS in vector registers: make every field wave-uniform again
__device__ __forceinline__ BkzsFrame frame_load(const BkzsFrame *p)
{
  BkzsFrame f;
  f.bsz       = uni(p->bsz);
  f.flags     = uni(p->flags);
  f.min_row   = uni(p->min_row);
  f.max_row   = uni(p->max_row);
  f.op        = uni(p->op);
  f.phase     = uni(p->phase);
  f.kappa     = uni(p->kappa);
  f.bs        = uni(p->bs);
  f.pre_i     = uni(p->pre_i);
  f.rerand    = uni(p->rerand);
  f.clean     = uni(p->clean);
  f.old_expo  = uni(p->old_expo);
  f.old_first = g_rl_f64(p->old_first, 0);
  f.remaining = g_rl_f64(p->remaining, 0);
  return f;
}

template <int NQ>
__device__ __forceinline__ void refloat_and_invalidate2(Lattice<NQ> &T, LllCtx &C, const SlotMap<NQ> &M,
                                                        int first, int last)
{
  const int lane = T.lane, n = T.n, ldn = T.ldn;
  for (int p = first; p < last; ++p)
  {
    const int s = M.phys(p);
    long long bv[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
    {
      const int c = lane + 64 * q;
      bv[q]       = (c < n) ? T.b[(size_t)s * ldn + c] : 0;
    }
    store_row_and_refloat<NQ>(T, s, bv);
    after_rowop<NQ>(T, C, M, p);
    __threadfence_block();
  }
}

// info[4] per lattice: tours, enumeration nodes (low / high 32 bits), enumeration calls
// status: 1 RED_SUCCESS, 8 RED_BKZ_LOOPS_LIMIT, <= 0 the failing LLL status, -7 the host did not
// answer a mailbox request in time, -8 schedule backstop
// DUALS = true adds self-dual BKZ (BKZ_SD_VARIANT 0x100 in top_flags: sd_tour = trunc_dtour +
// trunc_tour, bkz.cpp:401-413,443-463; dual svp_reduction :274-358, the dual enumeration
// enumerate.cpp:96-123,154-158 + enumerate_base.cpp:57-61,103-105, the dual insertions
// bkz.cpp:148-193,240-248; the prelude lll() :576-577 and the closing hkz() :627-641 selected by
// run_mode: 1 prelude, 2 tours, 4 closing hkz).  Everything it adds sits behind `if constexpr
// (DUALS)`: the DUALS = false instantiation is the kernel of §4f, instruction for instruction.
template <int NQ, bool DUALS>
__device__ __forceinline__ void
bkzs_body(GsoBatch P, BkzStrat S, BkzMail *mailbox, int *abort_flag, int block_size, int top_flags,
          double delta, double eta, double logdelta, int max_loops, int stack_doubles, int run_mode)
{
  constexpr int IPS = (NQ + 1) / 2;
  using RingT       = ReduceRing<NQ>;
  extern __shared__ __attribute__((aligned(16))) char bkzs_smem[];
  const int lane = threadIdx.x & 63;
  const int wpb  = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  RingT ring;
  ring.init(wave, lane);
  // behind the rings: per wave the enumeration stack, then the frame stack
  char *after_rings = bkzs_smem + (size_t)wpb * RingT::BYTES;
  double *stk       = (double *)after_rings + (size_t)wave * stack_doubles;
  BkzsFrame *frames = (BkzsFrame *)((double *)after_rings + (size_t)wpb * stack_doubles) +
                      (size_t)wave * FPHIP_BKZS_MAX_DEPTH;
  const int d = P.d, n = P.n, ldd = P.ldd, ldn = P.ldn;
  const bool has_strat = S.pre_off != nullptr;
  for (int L = blockIdx.x * wpb + wave; L < P.batch; L += gridDim.x * wpb)
  {
    Lattice<NQ> T;
    T.d           = d;
    T.n           = n;
    T.ldd         = ldd;
    T.ldn         = ldn;
    T.row_expo_on = P.row_expo;
    T.lane        = lane;
    T.b           = P.b + (size_t)L * d * ldn;
    T.bfT         = P.bfT + (size_t)L * n * ldd;
    T.mu          = P.mu + (size_t)L * d * ldd;
    T.muT         = P.muT + (size_t)L * d * ldd;
    T.r           = P.r + (size_t)L * d * ldd;
    T.rdg         = P.rdg + (size_t)L * d;
    T.rexp        = P.rexp + (size_t)L * d;
    T.bfT32       = P.bfT32 + (size_t)L * P.n * P.ldd;  // the float mirror of bf (see bkz_kernel.hip)
    T.b32         = (int *)T.b;
    T.narrow_flag = (int *)T.rexp;
    T.np          = 0;
    T.f32ok       = all_rows_narrow<NQ>(P, (size_t)L, lane);
    LllCtx C{P.gf + (size_t)L * d * ldd, P.vc + (size_t)L * d};
    // scaled mu rows of the block being enumerated: in LDS behind this wave's column stack when the
    // host asked for it (top_flags bit 30: few lattices per CU, where the L1 latency of every row
    // is exposed), in global memory otherwise (large batches: LDS buys resident waves)
    const bool mu_lds  = (top_flags & 0x40000000) != 0;
    const int bsm      = max(2, min(block_size, d));
    double *mu_blk_l   = stk + ((bsm * (bsm + 1)) / 2 + 2);
    double *mu_blk_g   = P.enum_mu + (size_t)L * (64 * 63 / 2);
    auto mu_blk_store = [&](int idx, double v)
    {
      if (mu_lds)
        mu_blk_l[idx] = v;
      else
        mu_blk_g[idx] = v;
    };
    BkzMail *mail  = mailbox + L;
    SlotMap<NQ> M;
    if (P.bkz_active[L] == 0)
    {  // this lattice's reduction has ended in an earlier launch (BKZ_AUTO_ABORT runs one tour per
       // launch): keep its basis
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        M.sl[q] = lane + 64 * q;
      lll_write_ordered<NQ>(T, M, P.b2 + (size_t)L * d * ldn);
      continue;
    }
    lll_init_state<NQ>(T, C, M);

    int vp = 0;
    auto upd   = [&](int k, int last) { return update_row_cached(T, C, M, ring, k, last); };
    auto after = [&](int k)
    {
      after_rowop<NQ>(T, C, M, k);
      vp = min(vp, k);
    };

    int num_rows = d;
    for (; num_rows > 0; --num_rows)
    {
      bool nz = false;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
      {
        const int c = lane + 64 * q;
        if (c < n)
          nz |= (T.b[(size_t)(num_rows - 1) * ldn + c] != 0);
      }
      if (__any(nz))
        break;
    }

    int status = 1, tours = 0, ncalls = 0;
    unsigned long long total_nodes = 0;
    unsigned long long mail_seq    = mail_load_u64(&mail->rsp_seq);  // both counters start equal

    // ---- host round trip: bump req_seq, wait for rsp_seq (wave-uniform result) -----------------
    auto mail_wait = [&]() -> bool
    {
      __threadfence_system();
      ++mail_seq;
      if (lane == 0)
        __hip_atomic_store(&mail->req_seq, mail_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      bool ok = true;
      // The wait is bounded by the host's LIVENESS, not by its speed: the service thread bumps
      // mailbox[0].heartbeat on every sweep over the mailboxes (one thread serves the whole batch,
      // a sweep can take long when thousands of lattices rerandomise at once or `rnd` is a Python
      // callable); a wave gives up only when that word has stood still for 30 s of wall clock
      // (s_memrealtime, 100 MHz) — poll counts are no measure: a loaded host stalls for seconds.
      unsigned long long hb     = mail_load_u64(&mailbox[0].heartbeat);
      unsigned long long t_live = wall_clock64();
      for (unsigned spin = 0; mail_load_u64(&mail->rsp_seq) < mail_seq; ++spin)
      {
        __builtin_amdgcn_s_sleep(32);
        bool dead = false;
        if ((spin & 63u) == 63u)
        {
          const unsigned long long h2  = mail_load_u64(&mailbox[0].heartbeat);
          const unsigned long long now = wall_clock64();
          if (h2 != hb)
            t_live = now;
          hb   = h2;
          dead = uni((int)(now - t_live > 3000000000ull)) != 0;
        }
        if (dead || (spin & 1023u) == 1023u && uni(__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
        {  // the host is gone: every wave gives up (one timeout, not one per lattice)
          ok = false;
          if (lane == 0)
            __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
      __threadfence_system();
      return ok;
    };

    // MatGSO::move_row(old_r, new_r), gso.cpp:289-366, on the slot table
    auto move_row = [&](int old_r, int new_r)
    {
      if (new_r < old_r)
      {
        rotate_right<NQ>(M, new_r, old_r, lane);
        clamp_valid<NQ>(T, C, M, new_r);
        vp = min(vp, new_r);
      }
      else if (old_r < new_r)
      {
        rotate_left<NQ>(M, old_r, new_r, lane);
        clamp_valid<NQ>(T, C, M, old_r);
        vp = min(vp, old_r);
      }
    };

    BkzsFrame F;
    int depth  = 0;
    F.bsz      = block_size;
    F.flags    = top_flags & ~0x40000000;
    F.min_row  = 0;
    F.max_row  = num_rows;
    F.op       = 0;
    F.phase    = PH_OP_BEGIN;
    F.kappa    = 0;
    F.bs       = 0;
    F.pre_i    = 0;
    F.rerand   = 0;
    F.clean    = 1;
    F.old_expo = 0;
    F.old_first = 0.0;
    F.remaining = 0.0;
    int loop    = 0;
    if constexpr (DUALS)
    {
      if ((top_flags & 0x200) && P.sld_pass == 2 && block_size >= 1)  // dual pass only: behind the checkpoint
        F.op = (num_rows + block_size - 1) / block_size + 1;
    }
    int sr_kmin = 0, sr_kend = 0, sr_start = 0, sr_next = PH_OP_BEGIN;  // the pending size reduction
    bool running = block_size >= 2;
    if (uni(__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
    {
      status  = -7;
      running = false;
    }
    if (running && (top_flags & 0x4) && loop >= max_loops)
    {
      status  = 8;
      running = false;
    }

    // the svp_reduction in progress is a dual one (SD frame, first half of its ops)
    auto cur_dual = [&]() -> bool
    {
      if constexpr (DUALS)
      {
        if (F.flags & 0x200)  // slide frame: the ops behind the checkpoint (op == p) are the dual blocks
          return F.op > (F.max_row - F.min_row + F.bsz - 1) / F.bsz;
        return (F.flags & 0x100) && F.op < max(F.max_row - F.bsz - F.min_row, 0);
      }
      else
        return false;
    };
    auto first_row = [&]() -> int { return cur_dual() ? F.kappa + F.bs - 1 : F.kappa; };
    unsigned steps = 0;  // backstop against a schedule that does not terminate (never seen)
    bool in_post = false;
    if constexpr (DUALS)
    {
      if ((top_flags & 0x100) && running && status == 1 && (run_mode & 1))
      {  // SD-BKZ starts with lll(0, 0, num_rows), bkz.cpp:576-577
        int fk, ns, zs;
        long long it;
        const int rc = lll_run_call(T, C, M, ring, 0, 0, num_rows, delta, eta, logdelta, fk, ns, zs, it, vp);
        if (rc != 1)
          status = rc;
      }
      if ((top_flags & 0x300) && !(run_mode & 2))
        running = false;  // this launch only runs the prelude and / or the closing hkz
    }
    // closing passes: SD-BKZ one (the last window), slide reduction one per block
    const int sld_p  = (num_rows + max(block_size, 1) - 1) / max(block_size, 1);
    const int nstage = DUALS ? 1 + ((top_flags & 0x200) ? sld_p : 1) : 1;
    int status_before = status;
    for (int stage = 0; stage < nstage; ++stage)
    {
    if constexpr (DUALS)
    {
      if (stage >= 1)
      {
        // closing pass of SD-BKZ: hkz(num_rows - block_size, num_rows), bkz.cpp:627-641 — it also
        // runs after RED_BKZ_LOOPS_LIMIT.  Slide reduction: hkz of every block, kappa = j bs + 1 to
        // min(num_rows, kappa + bs - 1), bkz.cpp:643-660 (the blocks are otherwise only SVP and dual
        // SVP reduced).
        if (!((top_flags & 0x300) && (run_mode & 4) && block_size >= 2 &&
              (in_post ? status == 1 : (status == 1 || status == 8))))
          break;
        if (!in_post)
          status_before = status;
        in_post     = true;
        running     = true;
        depth       = 0;
        F.flags     = top_flags & ~0x300 & ~0x40000000;
        if (top_flags & 0x200)
        {
          F.min_row = (stage - 1) * block_size + 1;
          F.max_row = min(num_rows, F.min_row + block_size - 1);
          F.bsz     = max(F.max_row - F.min_row, 0);  // hkz() takes its window from its arguments
        }
        else
        {
          F.bsz     = block_size;
          F.min_row = num_rows - block_size;
          F.max_row = num_rows;
        }
        F.op        = 0;
        F.phase     = PH_OP_BEGIN;
        F.clean     = 1;
      }
    }
    if constexpr (DUALS)
    {
      if (in_post)
        status = 1;
    }
    while (running && status == 1)
    {
      if (++steps > (1u << 27))
      {
        status = -8;
        break;
      }
      if (F.phase == PH_OP_BEGIN)
      {
        // ---- tour(): trunc_tour blocks, hkz blocks, then hkz's trailing size reduction ---------
        const int n_trunc = max(F.max_row - F.bsz - F.min_row, 0);
        const int hkz_lo  = max(F.max_row - F.bsz, 0);
        const int n_hkz   = max(F.max_row - 1 - hkz_lo, 0);
        int nops          = n_trunc + n_hkz + 1;
        bool sd_frame     = false;
        bool sld_frame    = false;
        int sld_np        = 0;
        if constexpr (DUALS)
        {
          // sd_tour: n_trunc dual blocks from the top down, then the n_trunc primal blocks; no hkz
          sd_frame = (F.flags & 0x100) != 0;
          if (sd_frame)
            nops = 2 * n_trunc;
          // slide_tour, bkz.cpp:465-520: passes of p primal blocks (stride bsz) until one leaves them
          // all unchanged — op == p is the end-of-pass checkpoint — then the p - 1 dual blocks
          sld_frame = (F.flags & 0x200) != 0;
          if (sld_frame)
          {
            sld_np = (F.max_row - F.min_row + F.bsz - 1) / F.bsz;
            nops   = 2 * sld_np;
          }
        }
        if (F.op >= nops)
        {  // the tour is over
          if (depth > 0)
          {  // back into the parent's svp_preprocessing (the clean flag of a preprocessing tour
             // is not used by svp_reduction, bkz.cpp:307)
            --depth;
            F = frame_load(&frames[depth]);
            continue;
          }
          if constexpr (DUALS)
          {
            if (in_post)
              break;
          }
          ++tours;
          if constexpr (DUALS)
          {
            if (sld_frame)
            {  // one slide tour per launch: the potential test (bkz.cpp:512-518, host libm) decides
               // on the host whether another one follows
              if (block_size < num_rows)
                status = 8;
              break;
            }
          }
          if (F.clean || block_size >= num_rows)
            break;
          ++loop;
          if ((top_flags & 0x4) && loop >= max_loops)
          {
            status = 8;
            break;
          }
          F.op    = 0;
          F.clean = 1;
          continue;
        }
        if constexpr (DUALS)
        {
          if (sld_frame && depth == 0 && P.sld_pass != 0)
          {
            // block-parallel mode: one pass per launch, only this device's blocks of it.  The end-of-pass
            // work (bounded LLL, repeat unless clean, potential) is the host's, on the merged basis.
            if (P.sld_pass == 1 && F.op == sld_np)
            {
              tours  = F.clean ? 1 : 0;  // info[0] of a pass launch: "my blocks were left unchanged"
              status = 8;
              break;
            }
            const int bit = F.op < sld_np ? F.op : F.op - sld_np - 1;
            if (!((P.sld_mask >> bit) & 1ull))
            {
              ++F.op;
              continue;
            }
          }
          if (sld_frame && F.op == sld_np)
          {  // end of a primal pass: the bounded LLL (bkz.cpp:482-493), then again unless clean
            if (F.flags & 0x10)
            {
              int fk, ns, zs;
              long long it;
              const int rc = lll_run_call(T, C, M, ring, F.min_row, F.min_row, F.max_row, delta, eta, logdelta, fk, ns,
                                     zs, it, vp);
              if (rc != 1)
              {
                status = rc;
                break;
              }
              if (ns > 0)
                F.clean = 0;
            }
            if (!F.clean)
            {
              F.op    = 0;
              F.clean = 1;
            }
            else
              ++F.op;
            continue;
          }
        }
        if (!sd_frame && !sld_frame && F.op == nops - 1)
        {  // lll_obj.size_reduction(max_row - 1, max_row, max_row - 2), bkz.cpp:437
          ++F.op;
          if (F.max_row >= 2)
          {
            sr_kmin  = F.max_row - 1;
            sr_kend  = F.max_row;
            sr_start = F.max_row - 2;
            sr_next  = PH_OP_BEGIN;
            F.phase  = PH_SR;
          }
          continue;
        }
        if (sld_frame)
        {
          if constexpr (DUALS)
          {
            if (F.op < sld_np)
            {
              F.kappa = F.min_row + F.op * F.bsz;
              F.bs    = min(F.max_row - F.kappa, F.bsz);
            }
            else
            {
              F.kappa = F.min_row + (F.op - sld_np - 1) * F.bsz + 1;
              F.bs    = F.bsz;
            }
          }
        }
        else if (sd_frame)
        {
          if constexpr (DUALS)
          {
            F.kappa = (F.op < n_trunc) ? F.max_row - F.bsz - F.op : F.min_row + (F.op - n_trunc);
            F.bs    = F.bsz;
          }
        }
        else if (F.op < n_trunc)
        {
          F.kappa = F.min_row + F.op;
          F.bs    = F.bsz;
        }
        else
        {
          F.kappa = hkz_lo + (F.op - n_trunc);
          F.bs    = F.max_row - F.kappa;
        }
        // ---- svp_reduction(kappa, bs): opening size reduction ...
        sr_kmin  = 0;
        sr_kend  = first_row() + 1;
        sr_start = 0;
        sr_next  = PH_OPENED;
        F.phase  = PH_SR;
        continue;
      }

      if (F.phase == PH_SR)
      {
        // lll_obj.size_reduction(kmin, kend, sr_start), lll.h:107-122, on the cached state: rows
        // below the verified prefix are size-reduced with r(k,k) in place (no-ops in the reference)
        if (status == 1)
          status = size_reduce_call(T, C, M, ring, max(sr_kmin, min(vp, sr_kend)), sr_kend, eta, sr_start, vp);
        F.phase = sr_next;
        continue;
      }

      if (F.phase == PH_OPENED)
      {
        // ... and the value to beat, bkz.cpp:291-293
        const int sk0 = M.phys(first_row());
        F.old_first   = T.rdg[sk0];
        F.old_expo    = (int)(2 * T.rexp[sk0]);
        F.rerand      = 0;
        F.remaining   = 1.0;
        F.phase       = PH_LOOP_HEAD;
        continue;
      }

      if (F.phase == PH_LOOP_HEAD)
      {
        // while (remaining_probability > 1. - par.min_success_probability), bkz.cpp:299
        if (!(F.remaining > 1. - 0.5))
        {
          F.phase = PH_FINISH;
          continue;
        }
        if (F.rerand)
        {
          // ---- rerandomize_block(kappa + 1, kappa + bs, density 3): the plan comes from the host
          const int lo = F.kappa + 1, hi = F.kappa + F.bs;
          if (hi - lo >= 2)
          {
            if (lane == 0)
            {
              mail->type    = 2;
              mail->lo      = lo;
              mail->hi      = hi;
              mail->density = 3;
            }
            if (!mail_wait())
            {
              status = -7;
              break;
            }
            const int n_moves = uni(mail_load_i32(&mail->n_moves));
            const int n_ops   = uni(mail_load_i32(&mail->n_ops));
            for (int i = 0; i < n_moves; ++i)
            {
              const int w = uni(mail_load_i32((const int *)&mail->plan[i]));
              move_row(w & 0xff, (w >> 8) & 0xff);  // m.move_row(b, a)
            }
            for (int i = 0; i < n_ops; ++i)
            {
              const int w   = uni(mail_load_i32((const int *)&mail->plan[n_moves + i]));
              const int a   = w & 0xff, b = (w >> 8) & 0xff;
              const bool ad = ((w >> 16) & 1) != 0;
              const int sa = M.phys(a), sb = M.phys(b);
#pragma unroll
              for (int q = 0; q < NQ; ++q)
              {
                const int c = lane + 64 * q;
                if (c < n)
                {
                  const unsigned long long va = (unsigned long long)T.b[(size_t)sa * ldn + c];
                  const unsigned long long vb = (unsigned long long)T.b[(size_t)sb * ldn + c];
                  T.b[(size_t)sa * ldn + c]   = (long long)(ad ? va + vb : va - vb);
                }
              }
              __threadfence_block();
            }
            refloat_and_invalidate2<NQ>(T, C, M, lo, hi);  // row_op_end(min_row, max_row)
            clamp_valid<NQ>(T, C, M, lo);
            vp = min(vp, lo);
          }
        }
        // ---- svp_preprocessing: lll(lll_start, lll_start, kappa + bs), bkz.cpp:107-113 ---------
        {
          const int ls = (F.flags & 0x10) ? F.kappa : 0;  // BKZ_BOUNDED_LLL
          int fk, ns, zs;
          long long it;
          const int rc = lll_run_call(T, C, M, ring, ls, ls, F.kappa + F.bs, delta, eta, logdelta, fk, ns, zs, it, vp);
          if (rc != 1)
          {
            status = rc;
            break;
          }
        }
        F.pre_i = has_strat ? uni(S.pre_off[F.bs]) : 0;
        F.phase = PH_PRE;
        continue;
      }

      if (F.phase == PH_PRE)
      {
        // ---- one tour of recursive preprocessing per block size of the strategy ----------------
        if (has_strat && F.pre_i < uni(S.pre_off[F.bs + 1]))
        {
          const int pb = uni(S.pre[F.pre_i]);
          ++F.pre_i;
          if (depth + 1 >= FPHIP_BKZS_MAX_DEPTH)
          {
            status = -7;  // (the host checks the nesting depth before the launch)
            break;
          }
          frames[depth] = F;
          __threadfence_block();
          ++depth;
          BkzsFrame G;
          G.bsz       = pb;  // BKZParam(*it, strategies, LLL_DEF_DELTA, BKZ_GH_BND), bkz.cpp:120
          G.flags     = 0x80;
          G.min_row   = F.kappa;
          G.max_row   = F.kappa + F.bs;
          G.op        = 0;
          G.phase     = PH_OP_BEGIN;
          G.kappa     = 0;
          G.bs        = 0;
          G.pre_i     = 0;
          G.rerand    = 0;
          G.clean     = 1;
          G.old_expo  = 0;
          G.old_first = 0.0;
          G.remaining = 0.0;
          F           = G;
          continue;
        }
        F.phase = PH_ENUM;
        continue;
      }

      if (F.phase == PH_ENUM)
      {
        const int kappa = F.kappa, bs = F.bs;
        // every row of the block is valid here in the reference (the preprocessing LLL / tours end
        // with all rows below kappa + bs updated); make sure the cache agrees — a recomputation
        // gives the same values, they are functions of the basis
        if (status == 1 && !update_rows_call(T, C, M, ring, kappa, kappa + bs))
          status = 0;
        if (status != 1)
          break;
        // ---- radius (bkz.cpp:309-323) and pruning set (:325) -----------------------------------
        int sl_blk = 0;  // lane i: slot of row kappa + i
#pragma unroll
        for (int q = 0; q < NQ; ++q)
        {
          const int src = kappa + lane;
          const int v   = __shfl(M.sl[q], src & 63);
          if ((src >> 6) == q)
            sl_blk = v;
        }
        const bool in   = lane < bs;
        const double rr = in ? T.rdg[sl_blk] : 0.0;
        const int e2    = in ? (int)(2 * T.rexp[sl_blk]) : 0;
        const bool dualb = cur_dual();
        const int sk    = M.phys(dualb ? kappa + bs - 1 : kappa);  // row `first`
        double md       = T.rdg[sk] * delta;  // max_dist *= delta
        if constexpr (DUALS)
        {
          if (dualb)  // max_dist.pow_si(max_dist, -1) (nr_FP_d.inl:189-192: ::pow) then *= delta —
            md = 0.0; // libm's pow: always taken from the host below
        }
        int prune       = -1;
        double expct    = 1.0;  // PruningParams(): no pruning, expectation 1
        bool hand       = false;  // the host walks this block on the multi-wave enumerator (hand-off mode)
        if (has_strat || ((F.flags & 0x80) && bs > 30) || dualb)
        {
          if (in)
          {
            mail->r[lane]  = rr;
            mail->e2[lane] = e2;
          }
          if (lane == 0)
          {
            mail->type  = 1;
            mail->bs    = bs;
            mail->flags = F.flags | (depth > 0 ? 0x10000 : 0) |  // a preprocessing tour: BKZParam defaults
                          (dualb ? 0x20000 : 0);                // dual block: radius from 1 / r(last)
            mail->delta = delta;
          }
          if (!mail_wait())
          {
            status = -7;
            break;
          }
          md    = mail_load_f64(&mail->max_dist);
          prune = uni(mail_load_i32(&mail->prune));
          expct = mail_load_f64(&mail->expectation);
          hand  = P.enum_mu_h != nullptr && uni(mail_load_i32(&mail->handoff)) != 0;
        }
        // ---- normalisation, enumerate.cpp:88-141 ------------------------------------------------
        int ne = in ? (int)min((long long)e2 + fexponent(rr), (long long)INT_MAX) : INT_MIN;
        ne     = max(wave_max_i32(ne), -1);
        double rd      = in ? ldexp(rr, e2 - ne) : 0.0;
        double maxdist = ldexp(md, (int)(2 * T.rexp[sk]) - ne);
        if (!dualb)
        {
          for (int k = 1; k < bs; ++k)
          {
            const int skk      = M.phys(kappa + k);
            const long long ek = T.rexp[skk];
            if (lane < k)
            {
              const double m = T.mu[(size_t)skk * ldd + kappa + lane];
              mu_blk_store(btri2(k) + lane, ldexp(m, (int)(ek - T.rexp[sl_blk])));
            }
          }
        }
        if constexpr (DUALS)
        {
          if (dualb)
          {
            // EnumerationDyn::enumerate for a dual call, enumerate.cpp:96-123: normexp is negated,
            // rdiag[d-1-i] = 1 / (r_i 2^(rexpo_i + normexp)), mut[d-1-j][d-1-i] = -mu(j, i);
            // the radius arrives with the exponent -(2 row_expo[last]) (bkz.cpp:312-316)
            const int nd_  = -ne;
            const int srcl = in ? bs - 1 - lane : 0;             // lane i takes row bs-1-i
            const double rsrc = __shfl(rr, srcl);
            const int esrc    = __shfl(e2, srcl);
            rd      = in ? 1.0 / ldexp(rsrc, esrc + nd_) : 0.0;
            maxdist = ldexp(md, -(int)(2 * T.rexp[sk]) - nd_);
            const int sl_rev = __shfl(sl_blk, srcl);             // slot of row kappa + bs-1-lane
            // dual row k' holds mu'(k', l) = -mu(bs-1-l, bs-1-k') for l < k'
            for (int k = 1; k < bs; ++k)
            {
              const int col       = kappa + bs - 1 - k;          // column of the primal mu
              const long long ec  = T.rexp[M.phys(col)];
              if (lane < k)
              {
                const double m = T.mu[(size_t)sl_rev * ldd + col];
                mu_blk_store(btri2(k) + lane, -ldexp(m, (int)(T.rexp[sl_rev] - ec)));
              }
            }
          }
        }
        __threadfence_block();
        // pruning coefficient of level `lane` (set_bounds, enumerate.cpp:218-228)
        double prn = 1.0;
        if (prune >= 0)
        {
          const int c0 = S.coeff_off[prune], c1 = S.coeff_off[prune + 1];
          if (c1 > c0 && in)
            prn = S.coeff[c0 + lane];
        }
        else if (prune == -2 && in)  // in-loop pruning: the host pruned THIS block (gso_host.hip, serve_radius)
          prn = mail_load_f64(&mail->prn[lane]);
        // ---- the walk (bkz_kernel.hip) with per-level bounds -----------------------------------
        double best_x = 0.0;
        bool have_sol = false;
        if (hand)
        {
          // Hand-off (opt-in, FPHIP_BKZ_HANDOFF): this block's tree is large — one wave walks
          // 3·10^6 nodes a second, the multi-wave enumerator (enum_kernel.hip) 10^9 and more — so the
          // host runs fphip_enum_run on it (a second context of this device; this wave only waits
          // on its mailbox meanwhile) and returns the vector FastEvaluator(1) would hold.  The
          // enumerator walks in another order than the reference: with a shrinking pruned radius
          // the vector, and from there the tour, may differ from the sequential one — as fplll with
          // its own multi-threaded enumlib differs from fplll alone.
          const int ntri = (bs * (bs - 1)) >> 1;
          double *muh    = P.enum_mu_h + (size_t)L * (64 * 63 / 2);
          for (int idx = lane; idx < ntri; idx += 64)
            muh[idx] = mu_lds ? mu_blk_l[idx] : mu_blk_g[idx];
          if (in)
          {
            mail->rd[lane]  = rd;
            mail->prn[lane] = prn;
          }
          if (lane == 0)
          {
            mail->type     = 3;
            mail->bs       = bs;
            mail->maxdist3 = maxdist;
            mail->dual3    = (DUALS && dualb) ? 1 : 0;
          }
          if (!mail_wait())
          {
            status = -7;
            break;
          }
          have_sol = uni(mail_load_i32(&mail->have_sol)) != 0;
          best_x   = in ? mail_load_f64(&mail->sol[lane]) : 0.0;
          total_nodes += mail_load_u64(&mail->nodes3);
          ++ncalls;
        }
        else
        {
          // The walk of enum_kernel.hip (see the comments there): two hot loops of wave-uniform
          // branches (this file is compiled with -structurizecfg-skip-uniform-regions), ddx =
          // sign(dx) not stored, round() as rint() + tie correction, scalar zig-zag bookkeeping,
          // v_writelane for the scalar level values, the step's column / mu row handed over in
          // registers, 32-bit level counters (a block's tree is far below 2^32 nodes per level:
          // flushed every 2^20 failed steps all the same).
          double xs = 0.0, cs = 0.0, pds = 0.0;
          int dxs = 0;
          unsigned long long cnt = 0;
          unsigned cnt32         = 0;
          double bnds = prn * maxdist;  // lane k: partdistbounds[k]
          const int bsu = __builtin_amdgcn_readfirstlane(bs);  // (wave-uniform, and known to be)
          int k         = bsu;
          // explicit address spaces: a select between an LDS and a global pointer would otherwise
          // be compiled into one FLAT load
          const lds_f64 *mu_l3 = (const lds_f64 *)mu_blk_l;
          const glb_f64 *mu_g1 = (const glb_f64 *)mu_blk_g;
          double Sc   = 0.0;
          double nd   = 0.0;
          double par = 0.0, mk = 0.0;  // (S_{k+1}, row k of mu) whenever the STEP loop is entered
          const unsigned lane8 = (unsigned)lane << 3;
          const int dummy      = (bsu * (bsu + 1)) >> 1;  // first spare double behind the stack rows
          unsigned fails       = 0;
          const bool dualw     = DUALS && dualb;
          enum : int { W_STEP = 0, W_DONE = 1, W_CHILD = 2 };
          for (;;)
          {
            int ev;
            // ---- CHILD chain ----------------------------------------------------------------------
            for (;;)
            {
              const int kc = k - 1;
              const int r1 = max(kc, 1);
              // speculative: row kc of mu for the descending case
              const unsigned o1 = ((unsigned)btri2(r1) << 3) + min(lane8, (unsigned)(r1 - 1) << 3);
              double mk1;
              if (mu_lds)
                mk1 = *(const lds_f64 *)((const lds_char *)mu_l3 + o1);
              else
                mk1 = *(const glb_f64 *)((const glb_char *)mu_g1 + o1);
              const double c1 = g_rl_f64(Sc, kc);
              double x1       = rint(c1);  // round(): rint + the ties that went towards zero
              double a1       = x1 - c1;
              if (fabs(a1) == 0.5 && ((a1 < 0.0) == (c1 > 0.0)))
              {
                x1 = x1 - (a1 + a1);
                a1 = -a1;
              }
              const double n1 = nd + a1 * a1 * g_rl_f64(rd, kc);
              if (!(n1 <= g_rl_f64(bnds, kc)))
              {
                ev = (k >= bsu) ? W_DONE : W_STEP;
                asm volatile("");
                break;
              }
              stk[(lane < k) ? btri2(k) + lane : dummy] = Sc;
              {
                const int s1  = (c1 >= x1) ? 1 : -1;
                const bool me = lane == kc;
                cs            = bk_wl_f64(c1, kc, cs);
                xs            = me ? x1 : xs;
                pds           = me ? nd : pds;
                dxs           = bk_wl_i32(s1, kc, dxs);
                cnt32 += me ? 1u : 0u;
              }
              par = Sc;
              mk  = mk1;
              k   = kc;
              nd  = n1;
              Sc  = Sc - (dualw ? a1 : x1) * mk1;  // dualenum: alpha[j] * mut, enumerate_base.cpp:57-61
              if (k == 0)
              {
                if (nd > 0.0)
                {
                  best_x   = xs;
                  have_sol = true;
                  maxdist  = nd;
                  bnds     = prn * nd;
                }
                ev = W_STEP;
                asm volatile("");
                break;
              }
            }
            ev = __builtin_amdgcn_readfirstlane(ev);
            asm volatile("" : "+s"(ev));
            if (ev == W_DONE)
              break;
            // ---- STEP loop -------------------------------------------------------------------------
            double xk, a;
            for (;;)
            {
              xk               = g_rl_f64(xs, k);
              const double ck  = g_rl_f64(cs, k);
              const int pdlo   = __builtin_amdgcn_readlane(__double2loint(pds), k);
              const int pdhi   = __builtin_amdgcn_readlane(__double2hiint(pds), k);
              const double pdk = __hiloint2double(pdhi, pdlo);
              int dxk          = __builtin_amdgcn_readlane(dxs, k);
              int pdor         = pdlo | pdhi;  // pdk is a sum of squares, never -0: zero <=> all bits zero
              asm volatile("" : "+s"(pdor));
              const bool zig = pdor != 0;
              int stepi      = zig ? dxk : 1;
              asm volatile("" : "+s"(stepi));
              xk += (double)stepi;
              dxk = zig ? ((dxk > 0 ? -1 : 1) - dxk) : dxk;  // ddx = -ddx; dx = ddx - dx
              const bool me = lane == k;
              xs            = me ? xk : xs;
              dxs           = bk_wl_i32(dxk, k, dxs);
              asm volatile("" : "+v"(xs), "+v"(dxs));
              a  = xk - ck;
              nd = pdk + a * a * g_rl_f64(rd, k);
              if (nd <= g_rl_f64(bnds, k))
              {
                cnt32 += me ? 1u : 0u;
                if (k != 0)
                {
                  ev = W_CHILD;
                  asm volatile("");
                  break;
                }
                if (nd > 0.0)
                {
                  best_x   = xs;
                  have_sol = true;
                  maxdist  = nd;
                  bnds     = prn * nd;
                }
                continue;  // next sibling of level 0 ((par, mk) are not used there)
              }
              ++k;
              if (k >= bsu)
              {
                ev = W_DONE;
                asm volatile("");
                break;
              }
              {  // the loads for the surviving case at the new level, a whole test ahead
                const unsigned cl8 = min(lane8, (unsigned)(k - 1) << 3);
                par                = *(const double *)((const char *)stk + (((unsigned)btri2(k + 1) << 3) + cl8));
                const unsigned o2  = ((unsigned)btri2(k) << 3) + cl8;
                if (mu_lds)
                  mk = *(const lds_f64 *)((const lds_char *)mu_l3 + o2);
                else
                  mk = *(const glb_f64 *)((const glb_char *)mu_g1 + o2);
              }
              if (((++fails) & 0xfffffu) == 0u)
              {
                cnt += cnt32;
                cnt32 = 0u;
              }
            }
            ev = __builtin_amdgcn_readfirstlane(ev);
            asm volatile("" : "+s"(ev));
            if (ev == W_DONE)
              break;
            Sc = par - (dualw ? a : xk) * mk;  // enumerate_base.cpp:103-105
          }
          cnt += cnt32;
          unsigned long long tot = cnt;
          for (int off = 32; off > 0; off >>= 1)
            tot += (unsigned long long)__shfl_xor((long long)tot, off);
          total_nodes += tot - (unsigned long long)(bs - 1);
          ++ncalls;
        }
        // ---- svp_postprocessing for a DUAL block, bkz.cpp:126-272 with dual = true ---------------
        bool handled = false;
        if constexpr (DUALS)
        {
          if (dualb && have_sol)
          {
            handled  = true;
            F.rerand = 0;
            // the evaluator's vector is index-reversed first (enumerate.cpp:154-158)
            double sx = __shfl(best_x, in ? bs - 1 - lane : 0);
            sx        = in ? sx : 0.0;
            const uint64_t nzm = __ballot(in && sx != 0.0);
            const uint64_t onm = __ballot(in && fabs(sx) == 1.0);
            const int nz       = __popcll(nzm);
            const int iv       = onm ? 63 - __clzll((long long)onm) : -1;
            const int pos      = kappa + bs - 1;
            if (nz == 1)
            {
              move_row(kappa + iv, pos);
            }
            else if (iv != -1)
            {
              // b[kappa+i] += (-sol_iv * sol_i) b[kappa+iv] for every other non-zero coordinate
              const double sv = -g_rl_f64(sx, iv);
              const int siv   = M.phys(kappa + iv);
              long long biv[NQ];
#pragma unroll
              for (int q = 0; q < NQ; ++q)
              {
                const int c = lane + 64 * q;
                biv[q]      = (c < n) ? T.b[(size_t)siv * ldn + c] : 0;
              }
              for (int i = 0; i < bs; ++i)
              {
                const double xi = g_rl_f64(sx, i);
                if (xi == 0.0 || i == iv)
                  continue;
                const long long lx = (long long)(sv * xi);
                const int si       = M.phys(kappa + i);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                {
                  const int c = lane + 64 * q;
                  if (c < n)
                    T.b[(size_t)si * ldn + c] =
                        (long long)((unsigned long long)T.b[(size_t)si * ldn + c] +
                                    (unsigned long long)biv[q] * (unsigned long long)lx);
                }
              }
              __threadfence_block();
              refloat_and_invalidate2<NQ>(T, C, M, kappa, kappa + bs);  // row_op_end(kappa, kappa + bs)
              clamp_valid<NQ>(T, C, M, kappa);
              vp = min(vp, kappa);
              move_row(kappa + iv, pos);
            }
            else
            {
              // generic case: the gcd tree with row_sub(kappa + k, kappa + k - off), no final move
              double x = sx;
              for (int i = 0; i < bs; ++i)
              {
                if (g_rl_f64(x, i) < 0.0)
                {
                  const int si = M.phys(kappa + i);
#pragma unroll
                  for (int q = 0; q < NQ; ++q)
                  {
                    const int c = lane + 64 * q;
                    if (c < n)
                      T.b[(size_t)si * ldn + c] = -T.b[(size_t)si * ldn + c];
                  }
                }
              }
              x = fabs(x);
              __threadfence_block();
              auto swap_rows = [&](int pa, int pb)
              {
                const int sa = M.phys(pa), sb = M.phys(pb);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                {
                  const int p = lane + 64 * q;
                  M.sl[q]     = (p == pa) ? sb : ((p == pb) ? sa : M.sl[q]);
                }
              };
              for (int off = 1; off < bs; off *= 2)
              {
                for (int k = bs - 1; k - off >= 0; k -= 2 * off)
                {
                  double xk = g_rl_f64(x, k), xo = g_rl_f64(x, k - off);
                  if (xk == 0.0 && xo == 0.0)
                    continue;
                  if (xk < xo)
                  {
                    const double t = xk;
                    xk             = xo;
                    xo             = t;
                    swap_rows(kappa + k - off, kappa + k);
                  }
                  while (xo != 0.0)
                  {
                    // while (x[k-off] <= x[k]) { x[k] -= x[k-off]; row_sub(k, k-off); }
                    const double qd = floor(xk / xo);
                    if (qd >= 1.0)
                    {
                      xk                 = xk - qd * xo;
                      const long long lq = (long long)qd;
                      const int sdst = M.phys(kappa + k), ssrc = M.phys(kappa + k - off);
#pragma unroll
                      for (int q = 0; q < NQ; ++q)
                      {
                        const int c = lane + 64 * q;
                        if (c < n)
                          T.b[(size_t)sdst * ldn + c] =
                              (long long)((unsigned long long)T.b[(size_t)sdst * ldn + c] -
                                          (unsigned long long)T.b[(size_t)ssrc * ldn + c] * (unsigned long long)lq);
                      }
                      __threadfence_block();
                    }
                    const double t = xk;
                    xk             = xo;
                    xo             = t;
                    swap_rows(kappa + k - off, kappa + k);
                  }
                  x = (lane == k) ? xk : ((lane == k - off) ? xo : x);
                }
              }
              refloat_and_invalidate2<NQ>(T, C, M, kappa, kappa + bs);
              vp = min(vp, kappa);
              clamp_valid<NQ>(T, C, M, kappa);
            }
            __threadfence_block();
          }
        }
        // ---- svp_postprocessing, bkz.cpp:126-272 (as in bkz_kernel.hip) -------------------------
        if (handled)
        {
        }
        else if (have_sol)
        {
          const uint64_t nzm = __ballot(in && best_x != 0.0);
          const uint64_t onm = __ballot(in && fabs(best_x) == 1.0);
          const int nz       = __popcll(nzm);
          const int iv       = onm ? 63 - __clzll((long long)onm) : -1;
          if (nz == 1)
          {
            if (iv > 0)
            {
              rotate_right<NQ>(M, kappa, kappa + iv, lane);
              clamp_valid<NQ>(T, C, M, kappa);
              vp = min(vp, kappa);
            }
          }
          else if (iv != -1)
          {
            const double sv = g_rl_f64(best_x, iv);
            const int st    = M.phys(kappa + iv);
            long long bv[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q)
            {
              const int c = lane + 64 * q;
              bv[q]       = (c < n) ? T.b[(size_t)st * ldn + c] : 0;
            }
            for (int i = 0; i < bs; ++i)
            {
              const double xi = g_rl_f64(best_x, i);
              if (xi == 0.0 || i == iv)
                continue;
              const long long lx = (long long)(sv * xi);
              const int si       = M.phys(kappa + i);
#pragma unroll
              for (int q = 0; q < NQ; ++q)
              {
                const int c = lane + 64 * q;
                if (c < n)
                  bv[q] = (long long)((unsigned long long)bv[q] +
                                      (unsigned long long)T.b[(size_t)si * ldn + c] * (unsigned long long)lx);
              }
            }
            store_row_and_refloat<NQ>(T, st, bv);
            after_rowop<NQ>(T, C, M, kappa + iv);
            vp = min(vp, kappa);
            __threadfence_block();
            if (iv > 0)
            {
              rotate_right<NQ>(M, kappa, kappa + iv, lane);
              clamp_valid<NQ>(T, C, M, kappa);
            }
          }
          else
          {
            double x = in ? best_x : 0.0;
            for (int i = 0; i < bs; ++i)
            {
              if (g_rl_f64(x, i) < 0.0)
              {
                const int si = M.phys(kappa + i);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                {
                  const int c = lane + 64 * q;
                  if (c < n)
                    T.b[(size_t)si * ldn + c] = -T.b[(size_t)si * ldn + c];
                }
              }
            }
            x = fabs(x);
            __threadfence_block();
            auto swap_rows = [&](int pa, int pb)
            {
              const int sa = M.phys(pa), sb = M.phys(pb);
#pragma unroll
              for (int q = 0; q < NQ; ++q)
              {
                const int p = lane + 64 * q;
                M.sl[q]     = (p == pa) ? sb : ((p == pb) ? sa : M.sl[q]);
              }
            };
            for (int off = 1; off < bs; off *= 2)
            {
              for (int k = bs - 1; k - off >= 0; k -= 2 * off)
              {
                double xk = g_rl_f64(x, k), xo = g_rl_f64(x, k - off);
                if (xk == 0.0 && xo == 0.0)
                  continue;
                if (xk < xo)
                {
                  const double t = xk;
                  xk             = xo;
                  xo             = t;
                  swap_rows(kappa + k - off, kappa + k);
                }
                while (xo != 0.0)
                {
                  const double qd = floor(xk / xo);
                  if (qd >= 1.0)
                  {
                    xk                 = xk - qd * xo;
                    const long long lq = (long long)qd;
                    const int sdst = M.phys(kappa + k - off), ssrc = M.phys(kappa + k);
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
                    {
                      const int c = lane + 64 * q;
                      if (c < n)
                        T.b[(size_t)sdst * ldn + c] =
                            (long long)((unsigned long long)T.b[(size_t)sdst * ldn + c] +
                                        (unsigned long long)T.b[(size_t)ssrc * ldn + c] * (unsigned long long)lq);
                    }
                    __threadfence_block();
                  }
                  const double t = xk;
                  xk             = xo;
                  xo             = t;
                  swap_rows(kappa + k - off, kappa + k);
                }
                x = (lane == k) ? xk : ((lane == k - off) ? xo : x);
              }
            }
            refloat_and_invalidate2<NQ>(T, C, M, kappa, kappa + bs);
            vp = min(vp, kappa);
            clamp_valid<NQ>(T, C, M, kappa);
            rotate_right<NQ>(M, kappa, kappa + bs - 1, lane);
            clamp_valid<NQ>(T, C, M, kappa);
          }
          __threadfence_block();
          F.rerand = 0;
        }
        else
        {
          F.rerand = 1;
        }
        F.remaining = F.remaining * (1 - expct);  // remaining_probability *= (1 - pruning.expectation)
        F.phase     = PH_LOOP_HEAD;
        continue;
      }

      if (F.phase == PH_FINISH)
      {  // closing size reduction, bkz.cpp:347-350
        sr_kmin  = 0;
        sr_kend  = first_row() + 1;
        sr_start = 0;
        sr_next  = PH_CLOSED;
        F.phase  = PH_SR;
        continue;
      }

      // PH_CLOSED: the progress test, bkz.cpp:352-357
      {
        const int sk0    = M.phys(first_row());
        double new_first = T.rdg[sk0];
        new_first        = ldexp(new_first, (int)(2 * T.rexp[sk0]) - F.old_expo);
        if (cur_dual())
          F.clean = (F.clean && __all(F.old_first >= new_first)) ? 1 : 0;
        else
          F.clean = (F.clean && __all(F.old_first <= new_first)) ? 1 : 0;
        ++F.op;
        F.phase = PH_OP_BEGIN;
      }
    }
    }  // stage
    if constexpr (DUALS)
    {
      if (in_post && status == 1)
        status = status_before;  // the closing passes keep RED_SUCCESS / RED_BKZ_LOOPS_LIMIT
    }
    lll_write_ordered<NQ>(T, M, P.b2 + (size_t)L * d * ldn);
    if (lane == 0)
    {
      P.status[L]           = status;
      P.lll_info[4 * L + 0] = tours;
      P.lll_info[4 * L + 1] = (int)(unsigned)(total_nodes & 0xffffffffull);
      P.lll_info[4 * L + 2] = (int)(unsigned)(total_nodes >> 32);
      P.lll_info[4 * L + 3] = ncalls;
      P.bkz_rows[L]         = num_rows;
      // tell the host this lattice is finished (type 0 request, no reply expected)
      mail->type = 0;
      __threadfence_system();
      __hip_atomic_store(&mail->done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_block();
  }
}

// the same schedule with the dual blocks of self-dual BKZ (the host selects it for BKZ_SD_VARIANT)
template <int NQ>
__global__ void __launch_bounds__(256)
    bkzd_kernel(GsoBatch P, BkzStrat S, BkzMail *mailbox, int *abort_flag, int block_size, int top_flags,
                double delta, double eta, double logdelta, int max_loops, int stack_doubles, int run_mode)
{
  bkzs_body<NQ, true>(P, S, mailbox, abort_flag, block_size, top_flags, delta, eta, logdelta, max_loops,
                      stack_doubles, run_mode);
}

template __global__ void bkzd_kernel<1>(GsoBatch, BkzStrat, BkzMail *, int *, int, int, double, double, double, int, int, int);
template __global__ void bkzd_kernel<2>(GsoBatch, BkzStrat, BkzMail *, int *, int, int, double, double, double, int, int, int);
template __global__ void bkzd_kernel<3>(GsoBatch, BkzStrat, BkzMail *, int *, int, int, double, double, double, int, int, int);
template __global__ void bkzd_kernel<4>(GsoBatch, BkzStrat, BkzMail *, int *, int, int, double, double, double, int, int, int);

}  // namespace sdv

// primal BKZ with strategies (fphip_gso_bkz_strategies without FPHIP_BKZ_SD_VARIANT): the same
// schedule without the dual blocks
template <int NQ>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NQ <= 1 ? 2 : 1, 8)))
    bkzs_kernel(GsoBatch P, BkzStrat S, BkzMail *mailbox, int *abort_flag, int block_size, int top_flags,
                double delta, double eta, double logdelta, int max_loops, int stack_doubles)
{
  sdv::bkzs_body<NQ, false>(P, S, mailbox, abort_flag, block_size, top_flags, delta, eta, logdelta,
                            max_loops, stack_doubles, 7);
}

template __global__ void bkzs_kernel<1>(GsoBatch, BkzStrat, BkzMail *, int *, int, int, double, double, double, int, int);
template __global__ void bkzs_kernel<2>(GsoBatch, BkzStrat, BkzMail *, int *, int, int, double, double, double, int, int);
template __global__ void bkzs_kernel<3>(GsoBatch, BkzStrat, BkzMail *, int *, int, int, double, double, double, int, int);
template __global__ void bkzs_kernel<4>(GsoBatch, BkzStrat, BkzMail *, int *, int, int, double, double, double, int, int);

}  // namespace fphip
