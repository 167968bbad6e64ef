/*
 * gmp_rng.c — TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * The reference draws the row permutation and the ±1 transformation of rerandomize_block
 * (fplll/bkz.cpp:43-80) from RandGen (fplll/nr/nr_rand.inl:12-48): gmp_randinit_default +
 * gmp_randseed_ui, then gmp_urandomm_ui.  This file gives the C oracle the same stream from the
 * same libgmp the reference build links (oracle/Makefile: $(CONDA)/lib/libgmp.so); it is a separate
 * shared library so that liboracle.so itself needs nothing beyond libm.
 */
#include <gmp.h>

static gmp_randstate_t state;
static int initialised;

/* RandGen::init_with_seed */
void oracle_gmp_rng_seed(unsigned long seed)
{
  if (!initialised)
  {
    gmp_randinit_default(state);
    initialised = 1;
  }
  gmp_randseed_ui(state, seed);
}

/* gmp_urandomm_ui(RandGen::get_gmp_state(), n): matches oracle_rand_fn */
unsigned long oracle_gmp_rng_next(void *user, unsigned long n)
{
  (void)user;
  if (!initialised)
    oracle_gmp_rng_seed(0);
  return gmp_urandomm_ui(state, n);
}

/* ---- one generator per lattice of a batch (the device strategy-BKZ tests) ------------------------
 * matches fphip_rand_fn (include/fplll_hip.h): rnd(user, lattice, n).  Each stream is
 * RandGen::init_with_seed(seed) — what a run of the reference on that lattice alone starts from. */
#include <stdlib.h>
/* Instance-based: several runs (threads of one test process) each own their streams. */
typedef struct
{
  gmp_randstate_t *streams;
  int n_streams;
  unsigned long long draws;
} oracle_gmp_streams;

void *oracle_gmp_streams_create(int batch, unsigned long seed)
{
  oracle_gmp_streams *s = (oracle_gmp_streams *)malloc(sizeof *s);
  s->streams            = (gmp_randstate_t *)malloc(sizeof(gmp_randstate_t) * (size_t)batch);
  s->n_streams          = batch;
  s->draws              = 0;
  for (int i = 0; i < batch; ++i)
  {
    gmp_randinit_default(s->streams[i]);
    gmp_randseed_ui(s->streams[i], seed);
  }
  return s;
}

void oracle_gmp_streams_destroy(void *h)
{
  oracle_gmp_streams *s = (oracle_gmp_streams *)h;
  if (!s)
    return;
  for (int i = 0; i < s->n_streams; ++i)
    gmp_randclear(s->streams[i]);
  free(s->streams);
  free(s);
}

unsigned long oracle_gmp_streams_next(void *user, int lattice, unsigned long n)
{
  oracle_gmp_streams *s = (oracle_gmp_streams *)user;
  ++s->draws;
  return gmp_urandomm_ui(s->streams[lattice], n);
}

unsigned long long oracle_gmp_streams_draws(void *h) { return ((oracle_gmp_streams *)h)->draws; }
