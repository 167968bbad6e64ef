// enum_device.h — structures shared by the enumeration kernel and its host driver.
#ifndef FPHIP_ENUM_DEVICE_H
#define FPHIP_ENUM_DEVICE_H

#include <stdint.h>

#define FPHIP_MAX_BLOCK 512
#define FPHIP_RING_CAP 1024u
#define FPHIP_MAX_LAUNCHES 256
#define FPHIP_TRI64 2016   /* 64*63/2 mu entries: rows below 64 (the wave-per-subtree walk) */
#define FPHIP_TRI128 8128 /* 128*127/2: rows up to 127 (the top walk of blocks larger than 64) */

#define FPHIP_ERR_RING_TIMEOUT 1u
#define FPHIP_FLAG_TASK_OVERFLOW 2u

namespace fphip
{

// One solution record in pinned host memory (device → host).
struct __attribute__((aligned(16))) SolRec
{
  unsigned long long seq;  // == global index + 1 once the record is complete
  double dist;
  double x[128];  // coefficients of levels 0..127
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
  double rdiag[128];
  double pruning[128];
  unsigned long long nodes[128];
  unsigned long long sub_bits[128];  // findsubsols: best sub-solution distance per level (bit
                                     // pattern of a positive double; starts at rdiag, only lowered)
  unsigned long long sol_head;  // monotonically increasing across calls (ring sequence)
  unsigned long long iters;     // walk-loop iterations (diagnostics)
  unsigned long long bound_bits;  // device mirror of HostCtl::bound_bits (only ever lowered)
  unsigned int error_flags;
  unsigned int pad;
  unsigned int task_head[FPHIP_MAX_LAUNCHES];
  unsigned int drain[FPHIP_MAX_LAUNCHES];  // set when a launch's task queue ran dry
  double rp[128][2];  // (rdiag[k], pruning[k]) interleaved: one 16-byte scalar load per level
  double mu_tri[FPHIP_TRI128];  // mu_tri[k(k-1)/2 + i] = mu(k,i), i<k
};

// Subtree tasks (structure of arrays; col/x rows are 64 doubles so that a wave loads them coalesced).
struct TaskBuf
{
  double *col;          // [cap][64]  S_L: rows i<L of the centre partial sums at the root
  double *x;            // [cap][64]  coefficients of levels >= L
  double *pd;           // [cap]      partial distance of the root node
  int *level;           // [cap]      root level L of the task (it walks levels < L)
  int *root;            // [cap]      index of the level-64 ancestor (blocks larger than 64): its
                        //            coefficients of levels 64..127 are kept once, in xhi_root
  unsigned int *count;  // number of tasks written (may exceed cap: overflow handled inline)
  unsigned int cap;
};

// Tasks of the top walk of a block larger than 64 (enum_top_kernel): subtree roots at a level > 64.
struct TopBuf
{
  double *col;          // [cap][128] S_L: rows i<L of the centre partial sums at the root
  double *xhi;          // [cap][64]  coefficients of levels 64..127 chosen so far (lane = level-64)
  double *pd;           // [cap]
  int *level;           // [cap]
  unsigned int *count;  // tasks written (may exceed cap: the host declines the instance)
  unsigned int cap;
};

}  // namespace fphip
#endif
