// gso_device.h — device-resident GSO state for a BATCH of independent lattices
// (ZT = long, FT = double, GSO_ROW_EXPO), shared by gso_kernel.hip and gso_host.hip.
//
// Layout in HBM (one lattice after another; every array is a dense [batch][...] slab):
//   b    [batch][d][n]   int64   integer basis, row-major  (rows are AXPY'd: coalesced over columns)
//   bfT  [batch][n][d]   double  float basis, COLUMN-major: bfT[c][j] = bf(j,c) — a Gram row is
//                                "lane j walks column c": coalesced over j
//   mu   [batch][d][d]   double  row-major  mu(j,k), k<j      (size-reduction sweep reads rows)
//   muT  [batch][d][d]   double  muT[k][j] = mu(j,k)          (GSO recurrence reads columns)
//   r    [batch][d][d]   double  row-major  r(i,j), j<=i
//   rdg  [batch][d]      double  r(j,j)
//   rexp [batch][d]      int64   row_expo (gso_interface.h:167)
//   status[batch]        int     1 ok, 0 RED_GSO_FAILURE, -1 RED_BABAI_FAILURE, -2 multiplier > 63 bits
#ifndef FPHIP_GSO_DEVICE_H
#define FPHIP_GSO_DEVICE_H

#include <stddef.h>
#include <stdint.h>

// LDS-DMA ring depths (slots per wave): sweep kernels / reduction drivers (see gso_wave.h)
#ifndef FPHIP_GSO_RING
#define FPHIP_GSO_RING 6
#endif
#ifndef FPHIP_RING_REDUCE
#define FPHIP_RING_REDUCE 8
#endif

// bytes of LDS per wave of the slot-mode reduction kernels (LLL / BKZ): the block ring of lll_stream.h
// (LStream<NQ>::BYTES), or with -DFPHIP_LLL_STREAM=0 the first generation's ring of single rows
#ifndef FPHIP_LLL_STREAM
#define FPHIP_LLL_STREAM 1
#endif
static inline size_t fphip_reduce_ring_bytes(int nq)
{
#if FPHIP_LLL_STREAM
  return nq == 3 ? 15360 : 16384;
#else
  return (size_t)FPHIP_RING_REDUCE * (size_t)((nq + 1) / 2) * 1024;
#endif
}

namespace fphip
{
struct GsoBatch
{
  int batch, d, n;
  int ldd, ldn;  // leading dimensions (d, n rounded up to even: rows start 16-byte aligned)
  int row_expo;
  long long *b;
  double *bfT;
  double *mu;
  double *muT;
  double *r;
  double *rdg;
  long long *rexp;
  int *status;
  // narrow mirrors of the sweep kernels: bf as float [batch][n][ldd], b as int32 [batch][d][ldn],
  // per-row flag [batch][d] "every entry of the row is below 2^24 in magnitude" (its mirrors are
  // then exact); a pass that reads rows 0..k streams the 4-byte mirrors when all of them are narrow
  float *bfT32;
  int *b32;
  int *narrow;
  int use_narrow;
  int wide_ring;  // sweep kernel: 1 = the 8-byte rows of lattices that are not narrow go through the ring too (round 5;
                  // FPHIP_GSO_WIDE_RING=0: the plain-load paths gram_wide / axpy_wide of rounds 2-4)
  // LLL kernel only (allocated on first use): symmetric Gram cache [batch][d][ldd], valid-column
  // counts [batch][d], output basis in position order [batch][d][ldn], info [batch][4]
  double *gf;
  int *vc;
  long long *b2;
  int *lll_info;
  double *enum_mu;  // BKZ kernel: [batch][64*63/2] scaled mu rows of the block being enumerated
  int *bkz_active;  // BKZ kernel: [batch] 0 = leave this lattice alone (its reduction has ended)
  int *bkz_rows;    // BKZ kernel: [batch] number of rows without the trailing zero rows
  double *enum_mu_h;  // strategy-BKZ kernel, hand-off mode: [batch][64*63/2] in PINNED HOST memory — the
                      // scaled mu rows of a block whose enumeration the host runs on the multi-wave
                      // enumerator (null: every block is walked by the lattice's own wave)
  // slide reduction, block-parallel mode (fphip_gso_slide_pass): this launch runs ONE pass of the slide
  // tour — sld_pass 1 = the primal blocks, 2 = the dual blocks — and of that pass only the blocks whose
  // bit is set in sld_mask (bit i = i-th block of the pass); 0 = the whole tour as usual
  int sld_pass;
  unsigned long long sld_mask;
  // Resident LLL session (lll_kernel.hip, fphip_gso_session_lll: what MatGSOHip keeps between the reference's
  // lll() calls).  sess_mode 0: every launch starts from a fresh MatGSO of the basis (the stateless calls).
  // 1: the same, and the kernel's state is left behind — the rows stay in their SLOTS; slot table, verified prefix
  // and the narrow flag go to sess_slots / sess_state; Gram cache, mu / r / rdg / rexp by slot and the valid-column
  // counts are the arrays above.  2: resume from that state.  Before lll() the launch replaces sess_ndirty rows
  // (batch of one): sess_in = the positions as int64, then the integer rows [ndirty][ldn] — a row operation each
  // (row_op_end, gso_interface.cpp:32-53).  After it the launch writes the state in POSITION order into sess_out:
  // b [d][ldn] int64, mu [d][ldd], r [d][ldd] doubles, row_expo [d] int64, valid columns [d] int32.
  int lll_siegel;            // lll_kernel: LLL_SIEGEL (the launch's delta is then the swap threshold delta - eta^2)
  int lll_early;             // lll_kernel: LLL_EARLY_RED (lll.cpp:84-99)
  // the transformation matrix u of MatGSO(b, u, ...) (gso.cpp:84-158: every row operation on b acts on u as well;
  // move_row rotates its rows with b's): [batch][d][ldd], rows in the slots of b's; nullptr = not tracked.  u2: the
  // position-ordered copy a stateless LLL call writes (swapped with u by the host, like b2 / b)
  long long *u, *u2;
  int sess_mode;
  int sess_ndirty;
  int *sess_slots;           // [batch][256]
  int *sess_state;           // [batch][4]: verified prefix, f32ok
  const long long *sess_in;  // positions [ndirty], rows of b [ndirty][ldn], rows of u [ndirty][ldd] (when u is tracked)
  char *sess_out;            // [batch][fphip_session_out_stride(d, ldd, ldn)]
};
static constexpr size_t fphip_session_out_bytes(size_t d, size_t ldd, size_t ldn)
{
  return d * ldn * 8 + 2 * d * ldd * 8 + d * 8 + ((d * 4 + 15) / 16) * 16;
}
// ... followed by the rows of u in position order ([d][ldd], written when u is tracked): a lattice's stride in sess_out
static constexpr size_t fphip_session_out_stride(size_t d, size_t ldd, size_t ldn)
{
  return fphip_session_out_bytes(d, ldd, ldn) + d * ldd * 8;
}
// ---- BKZ with strategies (bkzs_kernel.hip) ------------------------------------------------------
#define FPHIP_BKZS_MAX_DEPTH 4  /* nested tour() activations: the BKZ tour + 3 levels of preprocessing */
#define FPHIP_BKZS_PLAN_MAX 448 /* 4*63 row moves + 3*61 row additions of one rerandomize_block */
// Device copy of the strategies (bkz_param.h:22-66), flattened like fphip_strategies
// (include/fplll_hip.h); pre_off == nullptr: no strategies (EmptyStrategy for every block size).
struct BkzStrat
{
  int max_block_size;
  const int *pre_off;    // [max_block_size + 2]
  const int *pre;
  const int *coeff_off;  // [number of pruning sets + 1]
  const double *coeff;
};
// One mailbox per lattice in pinned, host-coherent memory: the wave asks the host for the two
// decisions that need host libraries (see bkzs_kernel.hip).  The wave fills the request, stores
// req_seq = n (release) and spins until rsp_seq == n.
struct BkzMail
{
  unsigned long long req_seq;  // device -> host
  unsigned long long rsp_seq;  // host -> device
  // request
  int type;     // 1: radius + pruning set of a block; 2: rerandomisation plan for rows [lo, hi)
  int bs;       // type 1: block size
  int flags;    // type 1: BKZParam::flags of the tour (GH_BND 0x80); 0x10000 = a preprocessing tour
  int lo, hi;   // type 2
  int density;  // type 2
  int done;     // the lattice's reduction has ended (diagnostics)
  int pad0;
  double delta;
  double r[64];  // type 1: r(kappa+i, kappa+i) as stored (without the row exponents) ...
  int e2[64];    //         ... and 2 * row_expo[kappa+i]
  // response
  double max_dist;     // type 1: enumeration radius, scaled by 2^-e2[0] like r[0]
  double expectation;  // type 1: success probability of the chosen pruning set
  int prune;           // type 1: index of the chosen pruning set (into coeff_off), -1 = none
  int n_moves, n_ops;  // type 2: plan[0..n_moves) = move_row(b, a) as b | a << 8, then n_ops row
  int pad1;            //         additions row a +/- row b as a | b << 8 | (add ? 1 : 0) << 16
  unsigned plan[FPHIP_BKZS_PLAN_MAX];
  // mailbox[0] only: bumped by the host on every service sweep (the device's liveness test)
  unsigned long long heartbeat;
  // ---- hand-off of a large block enumeration to the multi-wave enumerator (type 3) ------------
  int handoff;    // response of type 1: 1 = this block is large, send a type-3 request for it
  int dual3;      // type 3 request: the inputs are the dual transformation (dualenum walk)
  double rd[64];   // type 3 request: normalised r_ii of the block (the walk's rdiag)
  double prn[64];  //                 pruning coefficient per level
  double maxdist3; //                 normalised radius
  double sol[64];  // type 3 response: coefficients of the last solution delivered (FastEvaluator(1))
  int have_sol;
  int pad2;
  unsigned long long nodes3;  // nodes of the call, fplll rule
};
// Batched Householder state (MatHouseholder<Z_NR<long>, FP_NR<double>>): b, V, R are [batch][d][ldn]
// row-major (lane = column), sigma / rexp [batch][d].
struct HhBatch
{
  int batch, d, n, ldn;
  int row_expo;
  long long *b;
  double *V;
  double *R;
  double *sigma;
  long long *rexp;
  int *status;
  // HLLL kernel only (allocated on first use): float basis [batch][d][ldn], info [batch][2]
  double *bf;
  int *info;
};
}  // namespace fphip
#endif
