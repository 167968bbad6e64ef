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
