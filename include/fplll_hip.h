/*
 * fplll_hip.h — the drop-in boundary: a C ABI (plain pointers and sizes) over the MI355X-native
 * (gfx950, HIP) implementation of fplll's hot path.  libfplll_hip.so exports exactly these
 * symbols; everything above them (the std::function adapter for fplll's external-enumerator hook,
 * the MatGSO look-alike, the Python mirror in fplll_amd/) is host glue that only calls this ABI.
 *
 * Reference interfaces replaced (file:line under fplll/ of the reference, v5.5.0):
 *   fphip_enum_run        ← extenum_fc_enumerate                enum/enumerate_ext_api.h:88-92
 *                           (called from ExternalEnumeration::enumerate, enum/enumerate_ext.cpp:81-86;
 *                            default implementation enumlib_enumerate, enum-parallel/enumlib.cpp:94)
 *   fphip_sol_cb          ← extenum_cb_process_sol              enum/enumerate_ext_api.h:62-63
 *   fphip_subsol_cb       ← extenum_cb_process_subsol           enum/enumerate_ext_api.h:70-71
 *   (mut, rdiag, pruning) ← what extenum_cb_set_config fills    enum/enumerate_ext_api.h:52-53,
 *                            enum/enumerate_ext.cpp:91-148 (mutranspose=true layout)
 *   fphip_gso_*           ← MatGSO<Z_NR<long>,FP_NR<double>>    gso.h:33, gso_interface.h:59
 *       fphip_gso_update        ← MatGSOInterface::update_gso_row / update_gso
 *                                 gso_interface.cpp:131-164, gso_interface.h:767-775
 *       fphip_gso_size_reduce   ← LLLReduction::size_reduction → babai   lll.h:107-122, lll.cpp:166-224
 *       fphip_gso_lll           ← LLLReduction::lll (+ MatGSO::move_row)   lll.cpp:44-164, gso.cpp:289-366
 *                                 (what lll_reduction_zf<long,double> runs with LM_FAST, wrapper.cpp)
 *       fphip_gso_bkz           ← BKZReduction::bkz (empty strategies)      bkz.cpp:274-358,360-441,522-668
 *       fphip_gso_bkz_strategies ← BKZReduction::bkz with strategies       bkz.cpp:43-124 (+ the above)
 *       fphip_gso_get_*         ← get_mu_exp/get_r_exp/row_expo accessors gso_interface.h:675-732
 *   fphip_hh_*            ← MatHouseholder<Z_NR<long>,FP_NR<double>>   householder.h:70
 *       fphip_hh_update_R       ← refresh_R_bf + update_R               householder.cpp:27-245
 *       fphip_hh_hlll           ← HLLLReduction::hlll                   hlll.cpp:26-499
 *
 * Error convention: 0 = FPHIP_OK; FPHIP_UNSUPPORTED = instance declined, the caller must fall
 * back exactly as fplll does for a plugin returning ~uint64_t(0) (enum/enumerate_ext.cpp:88,
 * enum/enumerate.h:104-110); negative = hard error, message via fphip_last_error().
 * No C++ exception ever crosses this boundary.  Not thread-safe per context (fplll: "multiple
 * threads on the same object is not supported", README.md:310).
 */
#ifndef FPLLL_HIP_H
#define FPLLL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FPHIP_OK 0
#define FPHIP_UNSUPPORTED 1
#define FPHIP_ERROR (-1)

/* largest enumeration dimension handled on the device = fplll's FPLLL_MAX_ENUM_DIM (enum/enumerate_base.h:59-101).
 * Levels are lane-indexed, 64 per wavefront: blocks of 65..256 levels are walked in two stages (the levels
 * above 64 by the top walk with two — above 128 rows four — registers per lane, the rest by the wave-per-subtree
 * kernel); larger blocks are declined (FPHIP_UNSUPPORTED → fplll's own enumerator) */
#define FPHIP_ENUM_MAX_DIM 256

typedef struct fphip_ctx fphip_ctx;

/* Context bound to one HIP device (one process per GPU; device = LOCAL_RANK under torchrun). */
int fphip_create(int device, fphip_ctx **out);
/* The same with a stream priority class: -1 low, 0 normal (fphip_create), +1 high.  Streams of different
 * classes never share a hardware queue; inside a class they may, once the pool (four queues) is used up —
 * and a kernel queued behind a minutes-long launch of another context waits for it.  Create the context
 * of a long single launch (a whole BKZ run in one kernel) LOW when other contexts of the process must stay
 * responsive.  The library's own helper streams (pruner volume engines, the hand-off enumeration context
 * of a strategy-BKZ run) are HIGH: they must finish while the schedule kernel waits on its mailboxes. */
int fphip_create_ex(int device, int priority, fphip_ctx **out);
void fphip_destroy(fphip_ctx *ctx);
const char *fphip_last_error(const fphip_ctx *ctx);
int fphip_device_count(void);
/* library/ABI version, bumped when a signature changes */
int fphip_abi_version(void);

/* ------------------------------------------------------------------------------------------ */
/* Enumeration                                                                                  */
/* ------------------------------------------------------------------------------------------ */

/* Receives a candidate (squared norm `dist`, coefficients sol[0..dim)), returns the NEW
 * enumeration bound (unchanged, smaller, or 0 to stop).  Calls are serialised, made on the thread
 * that called fphip_enum_run, while the kernel is still running. */
typedef double (*fphip_sol_cb)(void *user, double dist, const double *sol);
typedef void (*fphip_subsol_cb)(void *user, double dist, const double *subsol, int offset);
/* Multi-GPU: called at every chunk / round boundary of the final phase — the SAME number of times
 * on every rank (it is a collective) — with this rank's current bound and whether it still has
 * subtree tasks; returns the bound to continue with and sets *any_active if some rank still has
 * tasks (the host glue makes it an RCCL all-reduce: MIN on the bound, MAX on the flag). */
typedef double (*fphip_exchange_cb)(void *user, double local_bound, int local_active,
                                    int *any_active);
/* Multi-GPU work movement: an all-gather of byte blocks.  Called at the round boundaries of the final phase, the
 * SAME number of times on every rank, with this rank's block; the callee writes the blocks of all ranks, in rank
 * order and back to back, into recv (recv_cap bytes) and their sizes into sizes[shard_count]; 0 = ok.  What it
 * carries: the ranks' counts of donated subtree tasks (8 bytes each), then — when they are out of balance — the
 * surplus tasks themselves (1040 bytes each), which the ranks below the average take over.  The role of enumlib's
 * shared subtree counter (enum-parallel/enumeration.h:412-505) between processes. */
typedef int (*fphip_gather_cb)(void *user, const void *send, size_t send_bytes, void *recv, size_t recv_cap,
                               size_t *sizes);

typedef struct fphip_enum_opts
{
  int dual;        /* 1 → the dualenum recursion (enumerate_base.cpp:57-61,103-105) on inputs the caller has
                      already transformed as EnumerationDyn::enumerate does (mu negated and index-reversed,
                      r inverted and reversed, enumerate.cpp:107-123); solutions arrive in that reversed
                      index order (the caller reverses them, enumerate.cpp:154-158).  Declined together
                      with findsubsols.  The extenum plugin (fplll_hip_extenum) still declines dual calls:
                      the reference's adapter hands a plugin untransformed mu/r (enumerate_ext.cpp:57-74) */
  int findsubsols; /* 1 → sub-solutions are reported through subcb (must be non-NULL) */
  /* subtree sharding across GPUs: the (cheap) top-of-tree phases are replicated on every rank, so
   * all ranks hold the same task SET; the tasks of the first walk round are sorted by content
   * (partial distance of the root, then a key of the coefficient prefix) and dealt to the ranks in
   * snake order 0..W-1, W-1..0 — disjoint, complete, nearly equal weight; this context walks the
   * share of shard_index.  Donated subtrees stay on their GPU unless `gather` (below) is set. */
  int shard_index;
  int shard_count;
  fphip_exchange_cb exchange;
  void *exchange_user;
  int exchange_chunks; /* number of chunks the final phase is cut into (>=1) */
  /* tuning; 0 = default */
  int target_tasks;    /* subtree tasks wanted for the final phase */
  int phase_growth;    /* wanted task growth per splitting phase */
  int waves_per_block; /* final-phase workgroup = waves_per_block * 64 threads */
  int min_nodes_decline; /* decline when the Gaussian-heuristic node estimate is below this */
  /* work movement between the ranks (with `exchange`; NULL: donated subtrees stay on their GPU).  After every
   * walk round the ranks compare their numbers of donated tasks; ranks above the average hand their surplus to
   * the ranks below it (by rank order).  Tasks of blocks above 64 rows move as well: their record carries the
   * coefficients of the levels >= 64 (a row of the sender's table of level-64 ancestors). */
  fphip_gather_cb gather;
  void *gather_user;
} fphip_enum_opts;

typedef struct fphip_enum_stats
{
  uint64_t total_nodes;  /* sum over levels (64-bit) */
  uint64_t solutions;    /* candidates handed to the callback */
  double wall_ms;        /* whole call */
  double kernel_ms;      /* sum of kernel durations (HIP events on the launch stream) */
  double final_kernel_ms;
  int phases;
  int final_tasks;
  int final_root_level;
  int overflowed; /* task-buffer overflow happened (handled inline, results still exact) */
  int bfs_restarts; /* the breadth-first stage overflowed a buffer and the call started over with split launches */
  uint64_t moved_tasks; /* work movement (fphip_enum_opts::gather): tasks that left or reached this rank */
} fphip_enum_stats;

/*
 * SVP enumeration of a dim-dimensional block.
 *   mut     : dim×dim row-major, mut[i*dim+j] = mu(j,i) for j>i (other entries ignored)
 *   rdiag   : r_ii · 2^-normexp ;  pruning : dim coefficients in (0,1] (NULL = all 1.0)
 *   maxdist : initial squared radius (normalised like rdiag)
 *   nodes_out[0..dim] : per-level node counts, fplll's counting rule
 *                       (enum/enumerate_base.cpp:31-33 incl. the :181-184 compensation)
 * Level bound = pruning[k]*maxdist; a node survives iff newdist <= bound (NaN-safe form).
 */
int fphip_enum_run(fphip_ctx *ctx, int dim, double maxdist, const double *mut, const double *rdiag,
                   const double *pruning, const fphip_enum_opts *opts, fphip_sol_cb cb,
                   fphip_subsol_cb subcb, void *user, uint64_t *nodes_out,
                   fphip_enum_stats *stats);

/* Lower the bound of the enumeration running on `ctx` from another host thread (never raises it; no
 * effect when nothing runs): the in-process multi-GPU mode of the fplll plugin publishes a bound
 * found on one GPU to the contexts of the others with it, between the collective exchange points. */
int fphip_enum_lower_bound(fphip_ctx *ctx, double bound);

/* ------------------------------------------------------------------------------------------ */
/* Batched, device-resident Gram-Schmidt + size reduction                                       */
/*   MatGSO<Z_NR<long>, FP_NR<double>> with GSO_ROW_EXPO (the BKZ fast path, bkz.cpp:816-829),  */
/*   for `batch` independent d×n lattices.  Entry points are SWEEPS, not single rows: a per-row  */
/*   device call would be launch-bound (4.1 M babai calls of ~0.3-1 us each in one BKZ-20 run).  */
/*   Results are bit-identical to the reference: integer basis, mu, r, row exponents.            */
/* ------------------------------------------------------------------------------------------ */
typedef struct fphip_gso fphip_gso;

/* d, n <= 256; larger → FPHIP_UNSUPPORTED (the lattice stays on fplll's CPU MatGSO). */
int fphip_gso_create(fphip_ctx *ctx, int batch, int d, int n, int row_expo, fphip_gso **out);
void fphip_gso_destroy(fphip_gso *g);
/* integer basis in/out, row-major b[lattice][row][col] (Matrix<Z_NR<long>>, nr/matrix.h:117) */
int fphip_gso_set_basis(fphip_gso *g, int first_lattice, int count, const int64_t *b);
int fphip_gso_get_basis(fphip_gso *g, int first_lattice, int count, int64_t *b);
/* copy lattice `src` into every slot (device-to-device; benchmarks) */
int fphip_gso_broadcast_basis(fphip_gso *g, int src);
/* lattices count, count+1, ... := copies of lattices 0 .. count-1, cyclically (device-side) */
int fphip_gso_tile_basis(fphip_gso *g, int count);
/* MatGSO::update_bf for every row (gso.cpp:24-48).  Done automatically by the first sweep /
 * reduction after set_basis or broadcast_basis; explicit calls are harmless. */
int fphip_gso_refresh(fphip_gso *g);
/* MatGSOInterface::update_gso() (gso_interface.h:767-775).  status[batch]: 1 ok, 0 = non-finite
 * mu (RED_GSO_FAILURE, gso_interface.cpp:156) */
int fphip_gso_update(fphip_gso *g, int *status);
/* LLLReduction::size_reduction(kappa_min, kappa_end) (lll.h:107-122 → babai lll.cpp:166-224) with
 * row_op_end bookkeeping; kappa_end = -1 means d.  status: 1 ok, 0 RED_GSO_FAILURE,
 * -1 RED_BABAI_FAILURE (lll.cpp:187-195), -2 multiplier beyond 63 bits (caller falls back) */
int fphip_gso_size_reduce(fphip_gso *g, int kappa_min, int kappa_end, double eta, int *status);
/* LLLReduction<Z_NR<long>,FP_NR<double>>(m, delta, eta, LLL_DEFAULT).lll(kappa_min, kappa_start,
 * kappa_end, 0) (lll.cpp:44-164) on a fresh MatGSO(b, GSO_ROW_EXPO) of every lattice, rows below
 * kappa_start brought up to date first; move_row (gso.cpp:289-366) included.  kappa_end = -1 means
 * d.  The basis is reduced in place (fphip_gso_get_basis), mu / r hold update_gso() of the result.
 * status[batch]: 1 RED_SUCCESS, 0 RED_GSO_FAILURE, -1 RED_BABAI_FAILURE, -2 multiplier beyond 63
 * bits, -3 RED_LLL_FAILURE (iteration limit).  info (nullable) [batch][4]: final_kappa, n_swaps,
 * zeros (rows moved to the end as linearly dependent), loop iterations. */
int fphip_gso_lll(fphip_gso *g, int kappa_min, int kappa_start, int kappa_end, double delta,
                  double eta, int *status, int *info);
/* The same with fplll's LLLFlags (defs.h:222-227; LLLReduction's constructor, lll.cpp:28-42): LLL_SIEGEL (4) —
 * swap_threshold = delta - eta^2, the tests against lovasz_tests[kappa] (lll.cpp:122,134) — and LLL_EARLY_RED (2)
 * — whenever kappa reaches a new maximum that is a power of two, every row from kappa on is size-reduced against
 * the rows below it (lll.cpp:84-99, lll.h:125-140) — run on the device; LLL_VERBOSE (1) is ignored.  One call is
 * one LLLReduction object: its last_early_red (lll.h:70) starts at 0. */
int fphip_gso_lll_flags(fphip_gso *g, int kappa_min, int kappa_start, int kappa_end, double delta, double eta,
                        int flags, int *status, int *info);
/* MatGSO(b, u, u_inv_t, flags) with a non-empty u (enable_transform, gso_interface.h:96-110): the transformation
 * matrix goes to the device — u [batch][d][d] row-major, or NULL for the identity — and fphip_gso_lll /
 * fphip_gso_lll_flags apply every row operation to its rows as well and move them with b's (gso.cpp:84-158,
 * 289-366): afterwards u_out = T u_in with b_out = T b_in.  Sessions (below) keep u too: a dirty row is then its n
 * integers of b followed by its d integers of u, fphip_gso_session_read_transform returns u in position order.
 * While u is tracked the entry points that do not update it (size_reduce, bkz*, slide, lll_ex / ladder) return
 * FPHIP_UNSUPPORTED; fphip_gso_set_basis keeps u.  u_inv_t (enable_inverse_transform) is not offered. */
int fphip_gso_enable_transform(fphip_gso *g, const int64_t *u);
int fphip_gso_get_transform(fphip_gso *g, int first, int count, int64_t *u);
/* The same lll() on a RESIDENT MatGSO: fplll's MatGSO is an object whose rows, Gram cache, mu / r and
 * gso_valid_cols persist from one lll() to the next (accessors gso_interface.h:675-732, validity tracking
 * gso_interface.cpp:26-53), and a BKZ run calls lll() thousands of times after touching a few rows.  resume = 0
 * starts a session from the basis on the device (fphip_gso_set_basis) as a fresh MatGSO; resume = 1 continues it:
 * first the caller's row operations since the last call — n_dirty rows (batch of one), dirty_pos[t] the row
 * position, dirty_rows[t][n] its new integers (followed by the d integers of its row of u when u is tracked:
 * [t][n + d]), each a row_op_end(p, p + 1) — then lll() on the state the last call
 * left (the verified prefix of rows that are a fixed point of the loop included, as fplll's own object would have
 * them valid).  While a session is active the rows live in the kernel's slots: the other fphip_gso_* entry points
 * refuse to run, fphip_gso_set_basis ends the session, a status other than 1 ends it too (the next call must be a
 * resume = 0 after fphip_gso_set_basis).  flags as fphip_gso_lll_flags (a change of LLL_SIEGEL between calls
 * forgets the verified prefix; the session is ONE LLLReduction object for LLL_EARLY_RED: last_early_red starts at
 * 0 with resume = 0 and is kept from call to call).  status / info as fphip_gso_lll. */
int fphip_gso_session_lll(fphip_gso *g, int resume, int kappa_min, int kappa_start, int kappa_end, double delta,
                          double eta, int flags, int n_dirty, const int *dirty_pos, const int64_t *dirty_rows,
                          int *status, int *info);
/* The state the last fphip_gso_session_lll left, in position order (host copy, no device call): b [d][n],
 * mu / r [d][d] row-major, valid_cols[d] = gso_valid_cols (entries mu(i,j), r(i,j) with j < valid_cols[i] are
 * meaningful; r(i,i) when valid_cols[i] == i + 1), row_expo[d].  Every pointer is nullable. */
int fphip_gso_session_read(fphip_gso *g, int lattice, int64_t *b, double *mu, double *r, int *valid_cols,
                           int64_t *row_expo);
/* ... and u [d][d] in position order (fphip_gso_enable_transform before the session started). */
int fphip_gso_session_read_transform(fphip_gso *g, int lattice, int64_t *u);
/* BKZReduction<Z_NR<long>,FP_NR<double>>(m, lll_obj, BKZParam(block_size, {}, delta, flags,
 * max_loops)).bkz() (bkz.cpp:522-668: tour / trunc_tour / hkz :360-441, svp_reduction :274-358,
 * svp_preprocessing's lll :107-113, svp_postprocessing :126-272, the block enumeration with
 * FastEvaluator(1)) on every lattice — primal BKZ with EMPTY strategies (no pruning, no
 * preprocessing: what bkz_reduction(b, beta, BKZ_DEFAULT, FT_DOUBLE) runs without a strategies
 * file, bkz_param.h:124-132), the whole reduction in one launch on device-resident GSO state.
 * flags: 0 (BKZ_DEFAULT), FPHIP_BKZ_MAX_LOOPS with max_loops, FPHIP_BKZ_AUTO_ABORT; block_size <= 64;
 * anything else
 * returns FPHIP_UNSUPPORTED (the caller keeps fplll's CPU path).  The input must be LLL-reduced, as
 * bkz_reduction guarantees (bkz.cpp:870-885) — call fphip_gso_lll first.
 * status[batch]: 1 RED_SUCCESS, 8 RED_BKZ_LOOPS_LIMIT, <= 0 the failing LLL status.
 * info (nullable) [batch][4]: tours, enumeration nodes low / high 32 bits (fplll rule), enumeration
 * calls. */
#define FPHIP_BKZ_DEFAULT 0
#define FPHIP_BKZ_MAX_LOOPS 0x4   /* fplll's BKZ_MAX_LOOPS, defs.h:262-275 */
#define FPHIP_BKZ_AUTO_ABORT 0x20 /* fplll's BKZ_AUTO_ABORT: BKZAutoAbort::test_abort(1.0, 5) between
                                     the tours (bkz.cpp:800-809); the tours are then launched one by
                                     one and the slope test runs on the host */
int fphip_gso_bkz(fphip_gso *g, int block_size, double delta, double eta, int flags, int max_loops,
                  int *status, int *info);
/* BKZ WITH strategies: BKZReduction::bkz() with BKZParam(block_size, strategies, delta, flags,
 * max_loops, ..., gh_factor) — what bkz_reduction(b, beta, flags, FT_DOUBLE) runs with a strategies
 * file (BASELINE configs 3-4): recursive preprocessing tours (svp_preprocessing, bkz.cpp:100-124),
 * the pruning set closest to radius / GH (get_pruning :82-98, bkz_param.cpp:64-80), the
 * Gaussian-heuristic radius bound (BKZ_GH_BND, gso_interface.cpp:220-276), the
 * success-probability loop and rerandomize_block (:43-80, 299-345).  One launch for the whole
 * batch; the calling thread answers the waves' requests for the two decisions that need host
 * libraries (libm for the radius, the caller's generator for the rerandomisation) until the kernel
 * has finished.  strategies = the content of a strategies JSON (load_strategies_json,
 * bkz_param.cpp:82-157), flattened: block size b has preprocessing block sizes
 * pre[pre_off[b] .. pre_off[b+1]) and pruning sets prune_off[b] .. prune_off[b+1]; set p has
 * gh_factor prune_gh[p], expectation prune_exp[p] and coefficients
 * coeff[coeff_off[p] .. coeff_off[p+1]) (empty = no pruning).  NULL = empty strategies.
 * rnd(user, lattice, n) must return gmp_urandomm_ui(<generator of that lattice>, n): fplll passes
 * RandGen::get_gmp_state() (nr/nr_rand.inl:38-43); each lattice of a batch has its own stream, the
 * one a run of the reference on that lattice alone would draw from.
 * flags: FPHIP_BKZ_MAX_LOOPS, FPHIP_BKZ_BOUNDED_LLL, FPHIP_BKZ_AUTO_ABORT (one tour per launch, the
 * slope test on the host in between, as in fphip_gso_bkz), FPHIP_BKZ_GH_BND, FPHIP_BKZ_MAX_TIME,
 * FPHIP_BKZ_DUMP_GSO (fplll's values).
 * FPHIP_BKZ_SD_VARIANT selects self-dual BKZ (sd_tour, dual svp_reduction / enumeration / insertion,
 * bkz.cpp:401-413,443-463; without MAX_LOOPS / AUTO_ABORT the auto abort is switched on, :548-554).
 * FPHIP_UNSUPPORTED: block sizes above 64, other flags, preprocessing nested deeper than 3 levels.
 * status / info as fphip_gso_bkz; status -7 = a mailbox request was not answered in time. */
/* FPHIP_BKZ_MAX_TIME / FPHIP_BKZ_DUMP_GSO (fplll's BKZ_MAX_TIME, BKZ_DUMP_GSO; bkz.cpp:563,588-592 and :373-377,
 * 536-539,667-670 with dump_gso :729-790), for fphip_gso_bkz and fphip_gso_bkz_strategies: one tour per launch;
 * between the tours the host compares the time since the call began with max_time (status 7 =
 * RED_BKZ_TIME_LIMIT, tested after the loop limit and in front of the auto-abort test like the reference) and
 * appends the reference's hand-written JSON entry — step, loop, time, norms = log r_ii + expo log 2 of the rows
 * below num_rows with 8 digits — to the dump file: "Input", one entry per tour ("End of BKZ loop" / "End of SD-BKZ
 * loop" / "End of SLD loop"), "Output".  Lattice 0 of a batch writes dump_gso_filename itself, lattice L > 0
 * dump_gso_filename.L.  fphip_gso_bkz_limits sets BKZParam::max_time (seconds; wall clock of the call — the
 * reference reads its single thread's CPU time) and BKZParam::dump_gso_filename (default "gso.json"). */
#define FPHIP_BKZ_MAX_TIME 0x8
#define FPHIP_BKZ_DUMP_GSO 0x40
int fphip_gso_bkz_limits(fphip_gso *g, double max_time, const char *dump_gso_filename);
#define FPHIP_BKZ_BOUNDED_LLL 0x10
#define FPHIP_BKZ_GH_BND 0x80
#define FPHIP_BKZ_SD_VARIANT 0x100
/* FPHIP_BKZ_SLD_RED: slide reduction (fplll's BKZ_SLD_RED; slide_tour bkz.cpp:465-520, the closing hkz of
 * every block :643-660): passes of disjoint primal blocks until one leaves them unchanged, then the dual
 * blocks shifted by one row; the slide potential (gso_interface.cpp:244-258) is evaluated on the host
 * between the tours (one tour per launch).  Not combined with FPHIP_BKZ_SD_VARIANT; declined when the last
 * block would hold a single row (d = k block_size + 1). */
#define FPHIP_BKZ_SLD_RED 0x200
/* FPHIP_BKZ_HANDOFF (not a flag of fplll): blocks whose Gaussian-heuristic tree size exceeds
 * FPHIP_BKZ_HANDOFF_NODES expected nodes (environment, default 800; the pruner's cost function,
 * fphip_pruner_enum_cost, on the block's r-profile, radius and pruning set) are enumerated by the multi-wave
 * enumerator (fphip_enum_run on a second context of the device, FastEvaluator(1) semantics) instead
 * of by the lattice's own wave: 10^9 nodes/s instead of 3·10^6.  That enumerator visits the tree in
 * another order than the reference, so with a shrinking pruned radius it may end on another vector:
 * the run is no longer bit-identical to BKZReduction::bkz() — it is to fplll what fplll with its
 * multi-threaded enumlib is to fplll alone.  Accept it by the reference's reducedness predicate
 * (tests/test_a_configs_at_size_gpu.py). */
#define FPHIP_BKZ_HANDOFF 0x1000
/* FPHIP_BKZ_PRUNE_IN_LOOP (not a flag of fplll): at the point where svp_reduction picks a
 * pruning set of the strategies (bkz.cpp:325, after the preprocessing and the radius), the PRIMAL blocks of
 * the TOP-LEVEL tour (its closing hkz included) of at least min_block_size rows get coefficients computed
 * for the block itself: prune<FP_NR<double>>(radius, preproc_cost, r_ii of the block, target,
 * PRUNER_METRIC_PROBABILITY_OF_SHORTEST, pruner_flags) on the profile the lattice's wave has just sent to
 * its mailbox, expectation = the pruner's; preprocessing tours and dual blocks keep the strategies' sets.
 * With on_device the searches' batches are scored by the volume kernel on a stream of its own while the
 * schedule kernel waits on the mailbox; otherwise by the host loop — same coefficients.  The reference has
 * no such mode; `oracle/ref_driver bkzprune` drives the reference's own svp_preprocessing / Enumeration /
 * svp_postprocessing / prune<> the same way, and the device run returns its basis, status and node count
 * (tests/test_bkzs_gpu.py).  Parameters (defaults 1e6, 0.5, 24, PRUNER_GRADIENT, 1) are per fphip_gso. */
#define FPHIP_BKZ_PRUNE_IN_LOOP 0x2000
int fphip_gso_bkz_inloop_pruning(fphip_gso *g, double preproc_cost, double target, int min_block_size,
                                 int pruner_flags, int on_device);
/* prune() calls of the service so far, and the volume jobs they evaluated by kernels / inline on the host */
int fphip_gso_bkz_inloop_stats(const fphip_gso *g, unsigned long long *prune_calls,
                               unsigned long long *device_jobs, unsigned long long *host_jobs,
                               unsigned long long *launches);
typedef struct fphip_strategies
{
  int max_block_size;
  const int *pre_off;      /* [max_block_size + 2] */
  const int *pre;
  const int *prune_off;    /* [max_block_size + 2] */
  const double *prune_gh;
  const double *prune_exp;
  const int *coeff_off;    /* [number of pruning sets + 1] */
  const double *coeff;
} fphip_strategies;
typedef unsigned long (*fphip_rand_fn)(void *user, int lattice, unsigned long n);
int fphip_gso_bkz_strategies(fphip_gso *g, int block_size, double delta, double eta, int flags,
                             int max_loops, double gh_factor, const fphip_strategies *strategies,
                             fphip_rand_fn rnd, void *rnd_user, int *status, int *info);
/* Slide reduction, BLOCK-PARALLEL.  The p primal blocks of a pass of slide_tour
 * (bkz.cpp:475-480) are disjoint, and so are its p - 1 dual blocks (:495-499); with BKZ_BOUNDED_LLL (flag
 * 0x10, required: otherwise every svp_reduction starts with an LLL from row 0) a block's svp_reduction
 * rewrites only its own rows — it reads the rows above them for the size reduction.  fphip_gso_slide_pass
 * runs ONE pass restricted to the blocks whose bit is set in block_mask (pass 1: primal block i = rows
 * [i bs, min(d, (i+1) bs)); pass 2: dual block i = rows [i bs + 1, (i+1) bs + 1); pass 3: the closing hkz of
 * every block, bkz.cpp:643-660, mask ignored) and stops; info[4 L] of a primal pass = 1 when every block
 * of the mask came out unchanged.  The tour itself — dealing the blocks over devices / ranks, gathering
 * each block's rows from the device that reduced it (the merged rows are a basis of the same lattice: a
 * block-triangular unimodular transformation of the pass's input), the bounded LLL and the repeat-until-clean
 * of the primal passes, the slide potential (gso_interface.cpp:244-258) — is host code:
 * fplll_amd.distributed.slide_reduction_blocks.  Every block is reduced from the PASS-START basis, whoever
 * reduces it, so the result does not depend on the number of devices; it is not the sequential reference's
 * (there block i sees block i - 1's new rows), and is accepted by the reference's predicates. */
int fphip_gso_slide_pass(fphip_gso *g, int block_size, double delta, double eta, int flags, double gh_factor,
                         const fphip_strategies *S, fphip_rand_fn rnd, void *rnd_user, int pass,
                         unsigned long long block_mask, int *status, int *info);
/* The whole block-parallel tour over `count` batch-of-one objects — one per context / device, all holding the
 * same LLL-reduced basis — on host threads of this process (what fplll_amd.distributed.slide_reduction_blocks
 * does with LocalGather, for callers without Python): BKZ_SLD_RED | BKZ_BOUNDED_LLL (flags must hold 0x10; 0x80
 * BKZ_GH_BND and FPHIP_BKZ_PRUNE_IN_LOOP allowed), max_loops 0 = until the potential stops falling.  On return
 * every object holds the result; *status 1 RED_SUCCESS / 8 RED_BKZ_LOOPS_LIMIT / a failure status, *nodes the
 * enumeration nodes of all blocks, *tours the number of slide tours. */
int fphip_gso_slide_reduction_blocks(fphip_gso **gs, int count, int block_size, double delta, double eta, int flags,
                                     int max_loops, double gh_factor, const fphip_strategies *S, fphip_rand_fn rnd,
                                     void *rnd_user, int *status, unsigned long long *nodes, int *tours);
/* LLLReduction::lll in a selectable floating-point type (lll_x.hip): precision 106 = double-double
 * arithmetic on the device (the stand-in for FP_NR<dd_real>: what Wrapper::lll's fast_lll<dd_real>
 * runs, wrapper.cpp:322-330; libqd's algorithms restated, csrc/ftx.h), 53 = plain double.  Same
 * algorithm (lll.cpp:44-224), statuses and info as fphip_gso_lll; sums are accumulated per lane, not in
 * the reference's order, so results are the reference's up to rounding (at 106 bits the decisions carry
 * ~50 bits of slack).  d <= 256. */
int fphip_gso_lll_ex(fphip_gso *g, int kappa_min, int kappa_start, int kappa_end, double delta, double eta,
                     int precision, int *status, int *info);
/* The LLL-side precision ladder of the reference's wrapper (Wrapper::lll, wrapper.cpp:281-359: double
 * first, then the wider types, each on the basis the failed attempt left) with both stages on the
 * device: the exact-order double kernel for the whole batch, then double-double for the lattices that
 * stopped with RED_GSO_FAILURE (0), RED_BABAI_FAILURE (-1) or RED_LLL_FAILURE (-3) — BASELINE config
 * 5's 256-dim NTRU-like lattice is such a case ("infinite loop in babai" in double, in the reference
 * and here).  stage[batch] (nullable): 53 or 106.  A lattice that fails at 106 bits too keeps its
 * status — the caller's MPFR stage is next. */
int fphip_gso_lll_ladder(fphip_gso *g, int kappa_min, int kappa_start, int kappa_end, double delta, double eta,
                         int *status, int *info, int *stage);

/* ---- the pruner (csrc/pruner_search.hip, pruner_volume.hip) ----------------------------
 * The searches are written against BATCHES of candidate coefficient vectors; a batch is scored by a
 * volume engine: the O(k^2)-per-value even-simplex volumes of every (bound vector, k) of the batch — one
 * lane per job on the device (pruner_volume_kernel), the same recurrence as a host loop without an
 * engine — and the O(n) rest with the host libm.  Both engines produce the same doubles.
 *
 * fphip_pruner_prune = prune<FP_NR<double>>(pruning, enumeration_radius, preproc_cost, gso_r, target,
 * metric, flags) (pruner/pruner.h:187-193, pruner.cpp:190-203): searches pruning coefficients that
 * minimise (cost of one enumeration x trials + preproc_cost x (trials - 1)) for the block whose squared
 * Gram-Schmidt lengths are gso_r[0..n) — greedy start, gradient descent and / or the Nelder-Mead search
 * (pruner_optimize_tc.cpp:581-825) on the even-indexed coefficients, local tuning, the same searches on
 * all of them (pruner_optimize*.cpp) — bit-identical coefficients to the reference's (every value goes
 * through the reference's sequence of IEEE operations; host libm).
 *   metric 0 = PRUNER_METRIC_PROBABILITY_OF_SHORTEST (0 < target < 1), 1 = PRUNER_METRIC_EXPECTED_SOLUTIONS
 *   flags: fplll's PRUNER_CVP 0x1, PRUNER_START_FROM_INPUT 0x2 (coefficients is then an input too),
 *          PRUNER_GRADIENT 0x4, PRUNER_NELDER_MEAD 0x8 (both = PRUNER_ZEALOUS), PRUNER_HALF 0x20,
 *          PRUNER_SINGLE 0x40; PRUNER_VERBOSE 0x10 → FPHIP_UNSUPPORTED
 *   coefficients[n] (out), expectation = the metric of the result, gh_factor = radius / Gaussian
 *   heuristic, detailed_cost[n] (nullable) = expected nodes per level — the fields of PruningParams.
 * FPHIP_ERROR where the reference throws (NaN / inf in a cost value, bad target). */
int fphip_pruner_prune(int n, const double *gso_r, double enumeration_radius, double preproc_cost, double target,
                       int metric, int flags, double *coefficients, double *expectation, double *gh_factor,
                       double *detailed_cost);
/* The overload for SEVERAL bases (pruner.h:200-221, pruner.cpp:214-227; Pruner::load_basis_shapes,
 * pruner_util.cpp:66-92): gso_rs = count profiles of n values each, row-major; the cost is averaged
 * over their shapes (normalised by the volume of the first). */
int fphip_pruner_prune_multi(int n, int count, const double *gso_rs, double enumeration_radius, double preproc_cost,
                             double target, int metric, int flags, double *coefficients, double *expectation,
                             double *gh_factor, double *detailed_cost);
/* svp_probability<FP_NR<double>>(pr) (pruner.cpp:166-176): probability that the enumeration pruned with
 * pr[0..n) still contains the shortest vector. */
int fphip_pruner_svp_probability(int n, const double *pr, double *probability);
/* Pruner(radius, ., gso_r, ., metric).single_enum_cost(pr, &detailed_cost) and .measure_metric(pr)
 * (pruner.h public overloads; pruner_cost.cpp:11-75): expected number of nodes of one enumeration of the
 * block under these coefficients, in all and per level, and its success probability / expected number of
 * solutions. */
int fphip_pruner_enum_cost(int n, const double *gso_r, double enumeration_radius, const double *pr, int metric,
                           double *cost, double *metric_value, double *detailed_cost);

/* A DEVICE volume engine: its own stream (it makes progress beside a persistent reduction kernel on the
 * context's stream), pinned staging, device buffers.  One per host thread.  fphip_pruner_prune_on is
 * fphip_pruner_prune / _prune_multi (count profiles) with the batches of the searches scored on that
 * engine (NULL: the host loop) — the same coefficients either way.  fphip_pruner_volumes exposes the
 * primitive: out[j] = V_{job_k[j]}(row job_vec[j] of bounds[nvec][m]), the volume of the even simplex
 * cut by that bound vector (Pruner::relative_volume, pruner_simplex.h:34-46), 1 <= k <= m <= 255.
 * fphip_pruner_engine_stats: jobs evaluated by kernels / inline on the host (batches below
 * FPHIP_PRUNER_MIN_DEVICE_STEPS = 16 000 polynomial steps — a lone candidate of a search — are cheaper
 * inline than a launch; the gradients, simplices and look-ahead batches go to the kernel), launches. */
typedef struct fphip_pruner_engine fphip_pruner_engine;
int fphip_pruner_engine_create(int device, fphip_pruner_engine **out);
void fphip_pruner_engine_destroy(fphip_pruner_engine *e);
int fphip_pruner_engine_stats(const fphip_pruner_engine *e, unsigned long long *device_jobs,
                              unsigned long long *host_jobs, unsigned long long *launches);
int fphip_pruner_volumes(fphip_pruner_engine *e, int m, int nvec, const double *bounds, int njobs,
                         const int *job_vec, const int *job_k, double *out);
int fphip_pruner_prune_on(fphip_pruner_engine *e, int n, int count, const double *gso_rs, double enumeration_radius,
                          double preproc_cost, double target, int metric, int flags, double *coefficients,
                          double *expectation, double *gh_factor, double *detailed_cost);


/* ---- host-side members of MatGSOInterface that BKZ callers use between reductions (csrc/gso_util_host.hip;
 * gso_interface.cpp:197-276), stateless over downloaded values: r_diag[i] = the STORED r(i,i) (the diagonal of
 * fphip_gso_get_r), row_expo = fphip_gso_get_row_expo's array (NULL: no row exponents), d = number of rows.
 * Bit for bit the reference's numbers (its operation order, the host's libm). */
double fphip_gso_util_current_slope(const double *r_diag, const int64_t *row_expo, int start_row, int stop_row);
double fphip_gso_util_log_det(const double *r_diag, const int64_t *row_expo, int d, int start_row, int end_row);
double fphip_gso_util_root_det(const double *r_diag, const int64_t *row_expo, int d, int start_row, int end_row);
double fphip_gso_util_slide_potential(const double *r_diag, const int64_t *row_expo, int d, int start_row,
                                      int end_row, int block_size);
/* is_lll_reduced<ZT, double>(m, delta, eta) (lll.cpp:226-258) on the stored mu / r matrices (d x d row-major:
 * fphip_gso_get_mu / fphip_gso_get_r of a lattice whose GSO is up to date) and the row exponents: 1 / 0. */
int fphip_gso_util_is_lll_reduced(const double *mu, const double *r, const int64_t *row_expo, int d, double delta,
                                  double eta);
/* adjust_radius_to_gh_bound(max_dist, max_dist_expo, block_size, root_det, gh_factor): returns the new
 * max_dist (unchanged when the bound is not smaller) */
double fphip_gso_util_adjust_radius_to_gh_bound(double max_dist, long max_dist_expo, int block_size,
                                                double root_det, double gh_factor);

/* raw stored values, d×d row-major; true values carry the row exponents exactly as
 * get_mu/get_r do (gso_interface.h:694-732): mu·2^(e_i-e_j), r·2^(e_i+e_j) */
int fphip_gso_get_mu(fphip_gso *g, int lattice, double *mu);
int fphip_gso_get_r(fphip_gso *g, int lattice, double *r);
int fphip_gso_get_row_expo(fphip_gso *g, int lattice, int64_t *row_expo);
/* duration of the last sweep kernel, HIP events on the launch stream */
double fphip_gso_last_kernel_ms(const fphip_gso *g);

/* ------------------------------------------------------------------------------------------ */
/* Batched Householder R-factor: MatHouseholder<Z_NR<long>, FP_NR<double>> (householder.h:38)    */
/*   fphip_hh_update_R = refresh_R_bf() + update_R() over all rows (householder.h:532-536,       */
/*   610-614; update_R householder.cpp:151-184, update_R_last :27-146, refresh_R_bf :186-245).   */
/*   R(i, j<=i) and the row exponents are bit-identical to the reference's.                      */
/* ------------------------------------------------------------------------------------------ */
typedef struct fphip_hh fphip_hh;
int fphip_hh_create(fphip_ctx *ctx, int batch, int d, int n, int row_expo, fphip_hh **out);
void fphip_hh_destroy(fphip_hh *h);
int fphip_hh_set_basis(fphip_hh *h, int first_lattice, int count, const int64_t *b);
int fphip_hh_broadcast_basis(fphip_hh *h, int src);
int fphip_hh_update_R(fphip_hh *h, int *status);
/* The same R-factor in BLOCKED compact-WY form on the MFMA matrix cores (v_mfma_f64_16x16x4_f64):
 * panels of 16 rows, R_panel -= ((R_panel V^T) T) V per block of 16 earlier reflectors.  OPT-IN
 * fast mode: the sums run in another order than the reference's (householder.cpp:157-178 is one
 * row, one reflector, one sequential dot product at a time), so R agrees with fphip_hh_update_R to
 * rounding (checked to 1e-9 relative on mu = R_ij/R_jj and r = R_ij R_jj), not bit for bit; row
 * exponents and signs are identical. */
int fphip_hh_update_R_blocked(fphip_hh *h, int *status);
/* MatHouseholder::size_reduce(kappa, size_reduction_end, size_reduction_start) (householder.cpp:402-451, with
 * row_addmul_we :522-559) as a stand-alone step over the batch, on the state fphip_hh_update_R left: for
 * i = end-1 … start, X = -rnd_we(R(kappa,i) / R(i,i)); a nonzero X adds lx·b[i] to b[kappa] and X·R[i] to the
 * kappa leading entries of R[kappa].  reduced[batch] = the reference's return value; b row kappa and R row kappa
 * are the reference's afterwards, bit for bit — the row is INVALID in the reference's sense ("not the correct
 * R[k]", :440-444): run fphip_hh_update_R before R is used again.  status (nullable) [batch]: 1, or -2 = a
 * multiplier beyond 63 bits (that lattice is left untouched). */
int fphip_hh_size_reduce(fphip_hh *h, int kappa, int size_reduction_end, int size_reduction_start, int *reduced,
                         int *status);
int fphip_hh_get_basis(fphip_hh *h, int first_lattice, int count, int64_t *b);
/* HLLLReduction<Z_NR<long>,FP_NR<double>>(m, delta, eta, theta, c, LLL_DEFAULT).hlll()
 * (hlll.cpp:26-169; size_reduction :262-351, lovasz_test :171-224, verify_size_reduction :455-496)
 * over a fresh MatHouseholder (update_R / update_R_last / refresh_R_bf / swap / size_reduce,
 * householder.cpp:27-451) of every lattice — what hlll_reduction_zf<long,double> runs with LM_FAST
 * (wrapper.cpp:790-806).  The basis is reduced in place (fphip_hh_get_basis); R / row_expo hold the
 * R factor of the result.  status[batch]: 1 RED_SUCCESS, -2 multiplier beyond 63 bits (redo on the
 * CPU), -4 RED_HLLL_SR_FAILURE, -5 RED_HLLL_NORM_FAILURE.  info (nullable) [batch][2]: swaps, loop
 * iterations. */
int fphip_hh_hlll(fphip_hh *h, double delta, double eta, double theta, double c, int *status,
                  int *info);
/* R as d×n row-major: MatHouseholder::get_R(expo) (householder.h:179); entries right of the
 * diagonal are scratch, exactly as in the reference */
/* HLLL in a selectable floating-point type: precision 106 = double-double arithmetic on the device
 * (the stand-in for FP_NR<dd_real>, fplll/nr/nr_FP_dd.inl: BASELINE config 5 as stated; libqd's
 * algorithms restated, csrc/ftx.h), precision 212 = quad-double (the stand-in for FP_NR<qd_real>,
 * fplll/nr/nr_FP_qd.inl: the third stage of the wrapper's ladder, wrapper.cpp:630-710), precision 53 = plain
 * double.  Same algorithm (hlll.cpp:26-499),
 * status and info as fphip_hh_hlll; dot products and norms are wave-level tree sums, so results
 * are the reference's up to rounding — decisions carry ~50 bits of slack at 106 bits.  After a
 * precision-106 run R(i,j) = fphip_hh_get_R + fphip_hh_get_R_lo. */
int fphip_hh_hlll_ex(fphip_hh *h, double delta, double eta, double theta, double c, int precision,
                     int *status, int *info);
int fphip_hh_get_R_lo(fphip_hh *h, int lattice, double *Rlo);
/* The precision ladder of the reference's wrapper (hlll_reduction, LM_WRAPPER: wrapper.cpp:478-529 —
 * double first, then the wider types, each stage continuing from the basis the previous one left)
 * with both stages on the device: the exact-order double kernel for the whole batch, then
 * double-double for the lattices that stopped with a precision alarm (status -4 / -5).
 * stage[batch] (nullable): 53 or 106.  A lattice that fails at 106 bits too keeps its status — the
 * caller's MPFR stage is next. */
int fphip_hh_hlll_ladder(fphip_hh *h, double delta, double eta, double theta, double c, int *status,
                         int *info, int *stage);
int fphip_hh_get_R(fphip_hh *h, int lattice, double *R);
int fphip_hh_get_row_expo(fphip_hh *h, int lattice, int64_t *row_expo);
double fphip_hh_last_kernel_ms(const fphip_hh *h);

#ifdef __cplusplus
}
#endif
#endif
