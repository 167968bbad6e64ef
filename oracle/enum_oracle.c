/*
 * enum_oracle.c — TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Plain-C restatement of fplll's Schnorr-Euchner / Kannan-Fincke-Pohst SVP enumeration, pinned
 * against the real reference build (oracle/_ref) by tests/test_oracle_vs_ref.py.
 *
 * The reference walks the tree with a per-level template recursion
 * (fplll/enum/enumerate_base.cpp:24-118).  The recursion is restated here as an explicit loop
 * with two entry modes per level ("ENTER" = lines 28-72, "STEP" = lines 80-116); the visit order,
 * every floating-point operation and its order, and the node-count rule are the reference's.
 * Compile with -ffp-contract=off: the reference is built for baseline x86-64 (no FMA), so
 * `partdist + alphak*alphak*rdiag` is two multiplies and one add, each rounded.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct
{
  int d;
  const double *mut; /* mut[i*d+j] */
  double rdiag[ORACLE_MAX_DIM];
  double pruning[ORACLE_MAX_DIM];
  double partdistbounds[ORACLE_MAX_DIM];
  double maxdist;
  /* enumerate_base.h:84-93 */
  double *center_partsums; /* (d+1)×(d+1) */
  int center_partsum_begin[ORACLE_MAX_DIM + 1];
  double partdist[ORACLE_MAX_DIM + 1], center[ORACLE_MAX_DIM], alpha[ORACLE_MAX_DIM];
  double x[ORACLE_MAX_DIM], dx[ORACLE_MAX_DIM], ddx[ORACLE_MAX_DIM];
  double subsoldists[ORACLE_MAX_DIM];
  int is_svp;
  uint64_t *nodes;
  /* sinks */
  oracle_sol_cb cb;
  oracle_subsol_cb subcb;
  void *user;
  int64_t nsols;
  double *best_sol;
  double *best_dist;
  double fx[ORACLE_MAX_DIM];
} enum_t;

#define CPS(e, i, j) ((e)->center_partsums[(size_t)(i) * ((e)->d + 1) + (j)])
#define MUT(e, i, j) ((e)->mut[(size_t)(i) * (e)->d + (j)])

/* EnumerationDyn::set_bounds, enumerate.cpp:218-229 */
static void set_bounds(enum_t *e)
{
  for (int i = 0; i < e->d; ++i)
    e->partdistbounds[i] = e->pruning[i] * e->maxdist;
}

/* EnumerationDyn::process_solution, enumerate.cpp:231-239 */
static void process_solution(enum_t *e, double newmaxdist)
{
  for (int j = 0; j < e->d; ++j)
    e->fx[j] = e->x[j];
  e->nsols++;
  if (e->cb)
  {
    e->maxdist = e->cb(e->user, newmaxdist, e->fx);
  }
  else
  {
    /* FastEvaluator(max_sols=1, BEST_N): evaluator.h:127-134 with calc_enum_bound exact */
    e->maxdist = newmaxdist;
    if (e->best_sol)
      memcpy(e->best_sol, e->fx, sizeof(double) * e->d);
    if (e->best_dist)
      *e->best_dist = newmaxdist;
  }
  set_bounds(e);
}

/* EnumerationDyn::process_subsolution, enumerate.cpp:241-249 */
static void process_subsolution(enum_t *e, int offset, double newdist)
{
  if (!e->subcb)
    return;
  for (int j = 0; j < offset; ++j)
    e->fx[j] = 0.0;
  for (int j = offset; j < e->d; ++j)
    e->fx[j] = e->x[j];
  e->subcb(e->user, newdist, e->fx, offset);
}

/* dual != 0: the dualenum instantiation of the recursion (enumerate_base.cpp:57-61, 103-105) and of
 * the initial descent (enumerate.cpp:185-189): the centre partial sums are driven by alpha = x - c
 * instead of x.  mut / rdiag are then the transformed inputs EnumerationDyn::enumerate builds for a
 * dual call (enumerate.cpp:107-123) and the caller reverses the solution (:154-158). */
static int64_t enumerate_impl(int d, const double *mut, const double *rdiag, const double *pruning,
                              double maxdist, int findsubsols, oracle_sol_cb cb,
                              oracle_subsol_cb subcb, void *user, uint64_t *nodes, double *best_sol,
                              double *best_dist, int dual)
{
  if (d <= 0 || d >= ORACLE_MAX_DIM)
    return -1;
  enum_t *e = (enum_t *)calloc(1, sizeof(enum_t));
  e->d      = d;
  e->mut    = mut;
  e->center_partsums = (double *)calloc((size_t)(d + 1) * (d + 1), sizeof(double));
  for (int i = 0; i < d; ++i)
  {
    e->rdiag[i]   = rdiag[i];
    e->pruning[i] = pruning ? pruning[i] : 1.0;
  }
  e->maxdist   = maxdist;
  e->nodes     = nodes;
  e->cb        = cb;
  e->subcb     = subcb;
  e->user      = user;
  e->best_sol  = best_sol;
  e->best_dist = best_dist;
  memset(nodes, 0, sizeof(uint64_t) * (d + 1));
  /* subsoldists = rdiag, enumerate.cpp:143 */
  for (int i = 0; i < d; ++i)
    e->subsoldists[i] = rdiag[i];

  /* ---- prepare_enumeration (SVP, no subtree, primal), enumerate.cpp:161-216 ---- */
  const int k_end = d;
  int k;
  e->is_svp = 1;
  {
    double newdist = 0.0;
    for (k = d - 1; k >= 0 && newdist <= e->maxdist; --k)
    {
      double newcenter = 0.0; /* center_partsum[k] = 0 for SVP, enumerate.cpp:82-86 */
      for (int j = k + 1; j < k_end; ++j)
        newcenter -= (dual ? e->alpha[j] : e->x[j]) * MUT(e, k, j);
      e->x[k]        = round(newcenter);
      e->center[k]   = newcenter;
      e->partdist[k] = newdist;
      e->dx[k] = e->ddx[k] = (newcenter >= e->x[k]) ? 1.0 : -1.0;
      e->alpha[k]          = e->x[k] - newcenter;
      newdist += e->alpha[k] * e->alpha[k] * e->rdiag[k];
    }
    e->x[0] = 1; /* enumerate.cpp:212, overwritten again by the recursive descent below */
    ++k;
  }

  /* ---- do_enumerate + enumerate_loop prologue, enumerate.cpp:251-257, enumerate_base.cpp:152-195 */
  set_bounds(e);
  if (k >= k_end)
    goto done;
  e->center_partsum_begin[0] = 0;
  for (int i = 0; i < k_end; ++i)
  {
    e->center_partsum_begin[i + 1] = k_end - 1;
    CPS(e, i, k_end)               = 0.0; /* center_partsum[i] */
  }
  e->partdist[k_end] = 0.0;
  for (int i = k + 1; i < k_end; i++)
    nodes[i]--; /* :181-184, wraps exactly like the reference's uint64_t */
  k = k_end - 1;

  /* ---- enumerate_recursive<kk>, flattened.  mode 0 = ENTER (first visit of x[k]),
   *      mode 1 = STEP (child subtree returned: next sibling at level k). */
  int mode = 0;
  while (k < k_end)
  {
    double newdist;
    if (mode == 0)
    {
      double alphak = e->x[k] - e->center[k];
      newdist       = e->partdist[k] + alphak * alphak * e->rdiag[k]; /* :28-29 */
      if (!(newdist <= e->partdistbounds[k]))
      { /* :31-32 return → parent continues at its STEP */
        ++k;
        mode = 1;
        continue;
      }
      ++nodes[k];
      e->alpha[k] = alphak;
      if (findsubsols && newdist < e->subsoldists[k] && newdist != 0.0)
      { /* :36-40 */
        e->subsoldists[k] = newdist;
        process_subsolution(e, k, newdist);
      }
      if (k == 0)
      {
        if (newdist > 0.0 || !e->is_svp) /* :42-46 */
          process_solution(e, newdist);
        mode = 1; /* enumerate_recursive<-1> is empty: go straight to the sibling step */
        continue;
      }
      /* :53-71 */
      e->partdist[k - 1] = newdist;
      for (int j = e->center_partsum_begin[k]; j > k - 1; --j)
        CPS(e, k - 1, j) = CPS(e, k - 1, j + 1) - (dual ? e->alpha[j] : e->x[j]) * MUT(e, k - 1, j);
      if (e->center_partsum_begin[k] > e->center_partsum_begin[k - 1])
        e->center_partsum_begin[k - 1] = e->center_partsum_begin[k];
      e->center_partsum_begin[k] = k;
      e->center[k - 1]           = CPS(e, k - 1, k);
      e->x[k - 1]                = round(e->center[k - 1]);
      e->dx[k - 1] = e->ddx[k - 1] = (e->center[k - 1] >= e->x[k - 1]) ? 1.0 : -1.0;
      --k;
      mode = 0;
    }
    else
    {
      /* :80-89 next sibling */
      if (!e->is_svp || e->partdist[k] != 0.0)
      {
        e->x[k] += e->dx[k];
        e->ddx[k] = -e->ddx[k];
        e->dx[k]  = e->ddx[k] - e->dx[k];
      }
      else
      {
        ++e->x[k];
      }
      double alphak2 = e->x[k] - e->center[k];
      newdist        = e->partdist[k] + alphak2 * alphak2 * e->rdiag[k]; /* :91-92 */
      if (!(newdist <= e->partdistbounds[k]))
      { /* :93-94 */
        ++k;
        mode = 1;
        continue;
      }
      ++nodes[k];
      e->alpha[k] = alphak2;
      if (k == 0)
      {
        if (newdist > 0.0 || !e->is_svp) /* :97-101 */
          process_solution(e, newdist);
        mode = 1;
        continue;
      }
      /* :103-115 */
      e->partdist[k - 1] = newdist;
      CPS(e, k - 1, k)   = CPS(e, k - 1, k + 1) - (dual ? e->alpha[k] : e->x[k]) * MUT(e, k - 1, k);
      if (k > e->center_partsum_begin[k - 1])
        e->center_partsum_begin[k - 1] = k;
      e->center[k - 1] = CPS(e, k - 1, k);
      e->x[k - 1]      = round(e->center[k - 1]);
      e->dx[k - 1] = e->ddx[k - 1] = (e->center[k - 1] >= e->x[k - 1]) ? 1.0 : -1.0;
      --k;
      mode = 0;
    }
  }

done:;
  int64_t ns = e->nsols;
  free(e->center_partsums);
  free(e);
  return ns;
}

int64_t oracle_enumerate(int d, const double *mut, const double *rdiag, const double *pruning,
                         double maxdist, int findsubsols, oracle_sol_cb cb, oracle_subsol_cb subcb,
                         void *user, uint64_t *nodes, double *best_sol, double *best_dist)
{
  return enumerate_impl(d, mut, rdiag, pruning, maxdist, findsubsols, cb, subcb, user, nodes, best_sol,
                        best_dist, 0);
}

int64_t oracle_enumerate_dual(int d, const double *mut, const double *rdiag, const double *pruning,
                              double maxdist, uint64_t *nodes, double *best_sol, double *best_dist)
{
  return enumerate_impl(d, mut, rdiag, pruning, maxdist, 0, NULL, NULL, NULL, nodes, best_sol, best_dist, 1);
}

/* the same with a caller-supplied evaluator callback (the tests' Python evaluators) */
int64_t oracle_enumerate_dual_cb(int d, const double *mut, const double *rdiag, const double *pruning,
                                 double maxdist, oracle_sol_cb cb, void *user, uint64_t *nodes)
{
  return enumerate_impl(d, mut, rdiag, pruning, maxdist, 0, cb, NULL, user, nodes, NULL, NULL, 1);
}
