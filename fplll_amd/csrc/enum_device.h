// enum_device.h — structures shared by the enumeration kernel and its host driver.
#ifndef FPHIP_ENUM_DEVICE_H
#define FPHIP_ENUM_DEVICE_H

#include <stdint.h>

#define FPHIP_MAX_BLOCK 512
#define FPHIP_RING_CAP 1024u
#define FPHIP_MAX_LAUNCHES 256
#define FPHIP_TRI64 2016   /* 64*63/2 mu entries: rows below 64 (the wave-per-subtree walk) */
#define FPHIP_TRI256 32640 /* 256*255/2: rows up to 255 (the top walk of blocks larger than 64) */

/* Task queues.  One returning atomic on ONE address costs about 50 ns of that address's L2 channel
 * (measured on MI355X: a walk launch lasted 62-66 ns per task whatever the tasks held, 180 000 tickets
 * took 9 ms) — a single ticket counter caps a launch at 2·10^7 tasks/s.  Tickets and emission counters
 * are therefore spread over FPHIP_NQ queues / regions whose counters sit FPHIP_QS words apart. */
// doubles behind the LDS column stack of a wave (enum_phase_kernel): where the lanes beyond a short
// row land when a 64-lane push is not masked
#define FPHIP_STACK_PAD 64
#define FPHIP_MUROW 66 /* doubles per row of DevShared::mu_sq */
#define FPHIP_NQ 128
#define FPHIP_QS 16 /* unsigned words between two counters: 64 bytes */

#define FPHIP_TASK_REC 130 /* doubles of a task on the wire (work movement between ranks): pd, level, col[64], x[64] */

#define FPHIP_ERR_RING_TIMEOUT 1u
#define FPHIP_FLAG_TASK_OVERFLOW 2u
#define FPHIP_FLAG_BFS_OVERFLOW 4u /* a buffer of the breadth-first expansion was too small: the host starts over */

namespace fphip
{

// One solution record in pinned host memory (device → host).
struct __attribute__((aligned(16))) SolRec
{
  unsigned long long seq;  // == global index + 1 once the record is complete
  double dist;
  double x[256];  // coefficients of levels 0..255
  int kind;       // 0 = candidate solution (process_solution), 1 = sub-solution (process_subsolution)
  int offset;     // sub-solution: its level (the coefficients below it are zero)
};

// Pinned, host-coherent control block (hipHostMallocCoherent).
struct HostCtl
{
  unsigned long long bound_bits;  // host → device: current maxdist (bit pattern of a double >= 0)
  unsigned long long consumed;    // host → device: number of ring records consumed so far
  unsigned long long pad[6];
  SolRec ring[FPHIP_RING_CAP];
};

// Device-resident per-enumeration state.
struct DevShared
{
  double rdiag[256];
  double pruning[256];
  unsigned long long nodes[256];
  unsigned long long sub_bits[256];  // findsubsols: best sub-solution distance per level (bit
                                     // pattern of a positive double; starts at rdiag, only lowered)
  unsigned long long sol_head;  // monotonically increasing across calls (ring sequence)
  unsigned long long iters;     // walk-loop iterations (diagnostics)
  unsigned long long bound_bits;  // device mirror of HostCtl::bound_bits (only ever lowered)
  unsigned int error_flags;
  unsigned int pad;
  unsigned int task_head[FPHIP_MAX_LAUNCHES];
  unsigned int drain[FPHIP_MAX_LAUNCHES];  // set when a launch's task queue ran dry
  double rp[256][2];  // (rdiag[k], pruning[k]) interleaved: one 16-byte scalar load per level
  // breadth-first expansion of the top of the tree (enum_bfs_kernel): the table of the subtree-size
  // estimate (scheduling only: never affects which nodes are visited)
  float bfs_A[64][64];  // [L][k], k < L: log( V_{L-k}(1) / prod_{i=k}^{L-1} sqrt(r_ii) )
  // rows 0..63 of mu once more, one row of FPHIP_MUROW doubles per level: mu_sq[k][i] = mu(k,i) for
  // i < k, zero up to 63, then (r_kk, pruning_k).  What the walk launches read per level: the mu row
  // as a buffer load (row offset in the scalar operand, lane offset a loop-invariant register, no
  // clamp: the packed rows above cost three VALU instructions per load) and the pair as one scalar
  // load — both with the SAME byte offset, one scalar induction variable per loop.
  double mu_sq[64][FPHIP_MUROW];
  // LAST (the host uploads the struct up to the rows the block has): mu_tri[k(k-1)/2 + i] = mu(k,i), i<k
  double mu_tri[FPHIP_TRI256];
};

// Subtree tasks (structure of arrays; col/x rows are 64 doubles so that a wave loads them coalesced).
struct TaskBuf
{
  double *col;          // [cap][64]  S_L: rows i<L of the centre partial sums at the root
  double *x;            // [cap][64]  coefficients of levels >= L
  double *pd;           // [cap]      partial distance of the root node
  int *level;           // [cap]      root level L of the task (it walks levels < L)
  int *root;            // [cap]      index of the level-64 ancestor (blocks larger than 64): its
                        //            coefficients of levels >= 64 are kept once, in xhi_root
  unsigned int *count;  // number of tasks written (may exceed cap: overflow handled inline)
  unsigned int cap;
};

// Queue memory of one enumeration call (device; zeroed at the start of the call).
struct QueueMem
{
  unsigned int head[FPHIP_MAX_LAUNCHES][FPHIP_NQ * FPHIP_QS];  // ticket counters per launch and queue
  unsigned int bfs[66][FPHIP_NQ * FPHIP_QS];  // breadth-first stage: heavy nodes per level and region
  unsigned int fin[FPHIP_NQ * FPHIP_QS];      // breadth-first stage: final tasks per region
};

// Tasks of the top walk of a block larger than 64 (enum_top_kernel): subtree roots at a level > 64.
struct TopBuf
{
  double *col;          // [cap][256] S_L: rows i<L of the centre partial sums at the root (row stride 128 up to d = 128)
  double *xhi;          // [cap][192] coefficients of levels >= 64 chosen so far (row stride 64 up to d = 128)
  double *pd;           // [cap]
  int *level;           // [cap]
  unsigned int *count;  // tasks written (may exceed cap: the host declines the instance)
  unsigned int cap;
};

}  // namespace fphip
#endif
