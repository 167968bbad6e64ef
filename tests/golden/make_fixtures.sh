#!/bin/bash
# Regenerates tests/golden/enum_*.json with the REAL reference (oracle/_ref/ref_driver, built by
# `make -C oracle ref` from /root/reference).  Inputs: seeded q-ary lattices (gen_qary_prime),
# LLL-reduced (+ optional BKZ-20), GSO with row exponents; the fixture holds exactly what fplll
# hands an external enumerator (mu^T, rdiag, pruning, maxdist) and what fplll's own enumerator
# returned (per-level node counts, every eval_sol call, final bound).
set -e
cd "$(dirname "$0")/../.."
D=oracle/_ref/ref_driver
R=oracle/_ref
G=tests/golden
#            n  k bits seed bkz first d  pruning   max_sols strategy rfac
$D enumfix  40 20 10  3   0   0   12 none       1         0 0.99 > $G/enum_d12_best1.json
$D enumfix  60 30 12  1   0   0   24 none       1         0 0.99 > $G/enum_d24_best1.json
$D enumfix  80 40 12  1   0   0   32 none       1         0 0.99 > $G/enum_d32_best1.json
$D enumfix  80 40 12  1   0   0   32 none       100000000 0 0.99 > $G/enum_d32_fixed.json
$D enumfix  80 40 12  1   0   0   32 none       1         2 0.99 > $G/enum_d32_first1.json
$D enumfix  80 40 12  1   0   4   32 none       5         0 1.30 > $G/enum_d32_best5.json
$D enumfix  80 40 12  1   0   4   32 none       3         1 1.30 > $G/enum_d32_opp3.json
$D enumfix  80 40 12  1   0   2   36 linear:18  1         0 0.99 > $G/enum_d36_lin18_best1.json
$D enumfix 100 50 14  2  20   0   40 linear:20  100000000 0 0.99 > $G/enum_d40_lin20_fixed.json
$D enumfix 100 50 14  2  20   0   40 linear:20  1         0 0.99 > $G/enum_d40_lin20_best1.json
$D enumfix 120 60 16  5  20  10   48 linear:30  100000000 0 0.99 > $G/enum_d48_lin30_fixed.json
$D enumfix 120 60 16  5  20  10   48 linear:30  1         0 0.99 > $G/enum_d48_lin30_best1.json
# blocks larger than 64 (two-stage walk on the device); REFDRV_RADIUS_SCALE shrinks the radius
REFDRV_RADIUS_SCALE=0.50 $D enumfix 100 50 14 5 20 0 72 linear:60 100000000 0 0.99 > $G/enum_d72_lin60_fixed.json
REFDRV_RADIUS_SCALE=0.45 $D enumfix 100 50 14 5 20 0 80 linear:70 100000000 0 0.99 > $G/enum_d80_lin70_fixed.json
REFDRV_RADIUS_SCALE=0.45 $D enumfix 100 50 14 5 20 0 80 linear:70 1         0 0.99 > $G/enum_d80_lin70_best1.json
REFDRV_RADIUS_SCALE=0.36 $D enumfix 110 55 14 6 20 0 96 linear:90 100000000 0 0.99 > $G/enum_d96_lin90_fixed.json

# --- DUAL enumeration (REFDRV_DUAL=1: enumerate(..., dual = true) with the radius of svp_reduction's
#     dual branch; the fixture holds the TRANSFORMED inputs of EnumerationDyn::enumerate, the
#     reference's per-level counts and every eval_sol call, coefficients in enumeration order)
REFDRV_DUAL=1 $D enumfix  60 30 10 4 20 10 36 none      100000000 0 0.99 > $G/dualenum_d36_fixed.json
REFDRV_DUAL=1 $D enumfix  60 30 10 3 20  0 30 linear:15 1         0 0.99 > $G/dualenum_d30_lin15_best1.json
REFDRV_DUAL=1 REFDRV_RADIUS_SCALE=0.4 $D enumfix 100 50 14 5 20 0 72 linear:60 100000000 0 0.99 > $G/dualenum_d72_lin60_fixed.json

# --- GSO / size-reduction (MatGSO<long,double>, GSO_ROW_EXPO): n k bits seed perturb
$D gsofix 30 15 10 1 0 > $G/gso_q30_p0.json
$D gsofix 48 24 12 2 3 > $G/gso_q48_p3.json
$D gsofix 64 32 14 3 5 > $G/gso_q64_p5.json
# --- Householder R factor: n k bits seed perturb row_expo
$D hhfix 40 20 11 4 2 0 > $G/hh_q40_p2_e0.json
$D hhfix 64 32 14 3 3 1 > $G/hh_q64_p3_e1.json
# MatHouseholder::size_reduce(kappa, end, start) on the state update_R() left (whole range, a sub-range, row
# exponents on, a row that needs nothing)
$D hhsr 40 20 11 4 2 0 25 25 0 > $G/hhsr_q40_k25.json
$D hhsr 40 20 11 4 3 0 35 30 5 > $G/hhsr_q40_k35_r5_30.json
$D hhsr 64 32 14 3 3 1 40 40 0 > $G/hhsr_q64_k40_e1.json
$D hhsr 40 20 11 4 0 0 30 30 0 > $G/hhsr_q40_k30_clean.json
# --- LLL (LLLReduction<long,double>::lll): type d k bits seed kmin kstart kend zero_rows dup_rows
$D lllfix q  40 20 20 1  0  0 -1 0 0 > $G/lll_q40.json
$D lllfix q  72 36 16 2  0  0 -1 0 0 > $G/lll_q72.json
$D lllfix q 130 65 12 3  0  0 -1 0 0 > $G/lll_q130.json
$D lllfix r  30  0 40 4  0  0 -1 0 0 > $G/lll_r30.json
$D lllfix u  24  0 30 5  0  0 -1 0 0 > $G/lll_u24.json
# LLL_SIEGEL (LLLFIX_FLAGS = fplll's LLLFlags of the run; recorded in the fixture)
LLLFIX_FLAGS=4 $D lllfix q  40 20 20 1  0  0 -1 0 0 > $G/lll_q40_siegel.json
LLLFIX_FLAGS=4 $D lllfix q  72 36 16 2  0  0 -1 0 0 > $G/lll_q72_siegel.json
LLLFIX_FLAGS=4 $D lllfix u  24  0 30 5  0  0 -1 0 0 > $G/lll_u24_siegel.json
LLLFIX_FLAGS=4 $D lllfix q 130 65 12 3  0 40 100 0 0 > $G/lll_q130_siegel_sub.json
# LLL_EARLY_RED (flags 2, lll.cpp:84-99): the swap counts differ from the plain runs' (another path)
LLLFIX_FLAGS=2 $D lllfix q  40 20 20 1  0  0 -1 0 0 > $G/lll_q40_earlyred.json
LLLFIX_FLAGS=2 $D lllfix q  72 36 16 2  0  0 -1 0 0 > $G/lll_q72_earlyred.json
LLLFIX_FLAGS=2 $D lllfix u  24  0 30 5  0  0 -1 0 0 > $G/lll_u24_earlyred.json
LLLFIX_FLAGS=2 $D lllfix q 130 65 12 3  0 40 100 0 0 > $G/lll_q130_earlyred_sub.json
LLLFIX_FLAGS=2 $D lllfix q  72 36 16 7  0  0 -1 2 1 > $G/lll_q72_earlyred_zeros.json
# MatGSO(b, u = identity, ...): the run keeps the transformation matrix (u_out in the fixture)
LLLFIX_U=1 $D lllfix q  40 20 20 1  0  0 -1 1 2 > $G/lll_q40_zero1_dup2_u.json
LLLFIX_U=1 $D lllfix q  72 36 16 2  0  0 -1 0 0 > $G/lll_q72_u.json
LLLFIX_U=1 LLLFIX_FLAGS=2 $D lllfix r 30 0 40 3 0 0 -1 0 0 > $G/lll_r30_earlyred_u.json
# ... and with the inverse transformation tracked as well (u_inv_t: enable_inverse_transform, gso.cpp:84-158)
LLLFIX_U=1 LLLFIX_UINV=1 $D lllfix q 40 20 20 1 0 0 -1 0 0 > $G/lll_q40_u_uinv.json
LLLFIX_U=1 LLLFIX_UINV=1 $D lllfix r 30 0 40 3 0 0 -1 0 0 > $G/lll_r30_u_uinv.json
# both flags at once; and on a sub-range with u
LLLFIX_FLAGS=6 $D lllfix q  40 20 20 4  0  0 -1 0 0 > $G/lll_q40_siegel_earlyred.json
LLLFIX_U=1 LLLFIX_FLAGS=6 $D lllfix q  72 36 16 9  0  5 60 0 0 > $G/lll_q72_siegel_earlyred_sub_u.json
$D lllfix q  40 20 20 6  0  0 -1 2 0 > $G/lll_q40_zero2.json
$D lllfix q  40 20 20 7  0  0 -1 0 2 > $G/lll_q40_dup2.json
$D lllfix q  40 20 20 8  0 10 30 0 0 > $G/lll_q40_range10_30.json
$D lllfix q  40 20 20 9  5  5 35 0 0 > $G/lll_q40_kmin5.json
# --- HLLL (HLLLReduction<long,double>::hlll, LM_FAST Householder flags): type d k bits seed
$D hlllfix q 40 20 20 1 > $G/hlll_q40.json
$D hlllfix q 72 36 16 2 > $G/hlll_q72.json
$D hlllfix r 30  0 40 4 > $G/hlll_r30.json
$D hlllfix u 24  0 30 5 > $G/hlll_u24.json
$D hlllfix n 64  0 10 6 > $G/hlll_n64.json
# config 5's lattice at full size, FT = double (12 s)
$D hlllfix n 256 0 10 6 | gzip -9 > $G/c5_hlll_n256_double.json.gz
# --- BKZ (BKZReduction<long,double>::bkz, empty strategies): type d k bits seed block_size max_loops
$D bkzfix q 40 20 20 1 10 0 > $G/bkz_q40_b10.json
$D bkzfix q 40 20 20 1 20 0 > $G/bkz_q40_b20.json
$D bkzfix q 72 36 16 2 12 2 > $G/bkz_q72_b12_loops2.json
$D bkzfix r 30  0 40 4  8 0 > $G/bkz_r30_b8.json
$D bkzfix u 24  0 30 5 24 0 > $G/bkz_u24_hkz.json
$D bkzfix q 60 30 12 7 16 0 > $G/bkz_q60_b16.json
# --- BKZ with strategies (preprocessing tours, pruning selection, GH bound, rerandomisation):
#     pruning coefficients from the reference's pruner on an 80-dim q-ary profile (genstrat),
#     preprocessing / expectations set by make_strategies.py; REFDRV_RNG_SEED fixes RandGen
T=$(mktemp -d)
$D dumpbasis 80 40 20 3 20 > $T/b80.txt
$D genstrat $T/b80.txt 40 > $T/gen.json
python3 $G/make_strategies.py $T/gen.json $T/stratA.json 40 14 34 1.0
python3 $G/make_strategies.py $T/gen.json $T/stratB.json 40 14 34 0.6
REFDRV_STRATEGIES=$T/stratA.json REFDRV_BKZ_FLAGS=0x80 REFDRV_RNG_SEED=5 $D bkzfix q 64 32 14 3 40 2 > $G/bkzs_q64_b40_pre_gh.json
REFDRV_STRATEGIES=$T/stratB.json REFDRV_BKZ_FLAGS=0x80 REFDRV_RNG_SEED=5 $D bkzfix q 64 32 14 3 40 2 > $G/bkzs_q64_b40_rerand.json
REFDRV_STRATEGIES=$T/stratA.json REFDRV_BKZ_FLAGS=0x80 REFDRV_BKZ_AUTO_ABORT=1 REFDRV_RNG_SEED=9 $D bkzfix q 56 28 12 4 36 0 > $G/bkzs_q56_b36_autoabort.json
REFDRV_STRATEGIES=$T/stratB.json REFDRV_BKZ_FLAGS=0x10 REFDRV_RNG_SEED=11 $D bkzfix q 64 32 14 6 34 1 > $G/bkzs_q64_b34_bounded_lll.json
REFDRV_STRATEGIES=$T/stratB.json REFDRV_BKZ_FLAGS=0x80 REFDRV_RNG_SEED=13 $D bkzfix r 40 0 40 4 32 2 > $G/bkzs_r40_b32_rerand.json
# in-loop pruning (REFDRV_INLOOP="preproc_cost target min_block pruner_flags": ref_driver's InloopBKZ drives the
# reference's public members, prune<>() per top-level block)
REFDRV_INLOOP="1e5 0.5 24 4" REFDRV_STRATEGIES=$T/stratA.json REFDRV_BKZ_FLAGS=0x80 REFDRV_RNG_SEED=5 $D bkzfix q 64 32 14 3 40 2 > $G/bkzp_q64_b40_inloop.json
REFDRV_INLOOP="2e4 0.6 20 4" REFDRV_STRATEGIES=$T/stratB.json REFDRV_BKZ_FLAGS=0x80 REFDRV_RNG_SEED=13 $D bkzfix r 40 0 40 4 32 2 > $G/bkzp_r40_b32_inloop_rerand.json
REFDRV_INLOOP="1e6 0.3 30 36" REFDRV_STRATEGIES=$T/stratA.json REFDRV_BKZ_FLAGS=0x80 REFDRV_BKZ_AUTO_ABORT=1 REFDRV_RNG_SEED=9 $D bkzfix q 56 28 12 4 36 0 > $G/bkzp_q56_b36_inloop_half_autoabort.json
# --- self-dual BKZ (0x100) and slide reduction (0x200): dual svp_reduction / dual enumeration
REFDRV_BKZ_FLAGS=0x100 $D bkzfix q 40 20 20 1 10 3 > $G/bkzd_q40_b10_sd_loops3.json
REFDRV_BKZ_FLAGS=0x200 $D bkzfix q 40 20 20 1 10 0 > $G/bkzd_q40_b10_slide.json
REFDRV_BKZ_FLAGS=0x100 $D bkzfix q 60 30 12 7 16 0 > $G/bkzd_q60_b16_sd_autoabort.json
REFDRV_BKZ_FLAGS=0x210 $D bkzfix q 64 32 14 6 16 0 > $G/bkzd_q64_b16_slide_bounded_lll.json
REFDRV_STRATEGIES=$T/stratB.json REFDRV_BKZ_FLAGS=0x180 REFDRV_RNG_SEED=7 $D bkzfix q 64 32 14 3 40 1 > $G/bkzd_q64_b40_sd_strategies.json
REFDRV_STRATEGIES=$T/stratA.json REFDRV_BKZ_FLAGS=0x280 REFDRV_RNG_SEED=8 $D bkzfix q 70 35 14 9 32 2 > $G/bkzd_q70_b32_slide_strategies.json
REFDRV_BKZ_FLAGS=0x200 $D bkzfix r 30 0 40 4 8 0 > $G/bkzd_r30_b8_slide.json
# BKZ_DUMP_GSO (REFDRV_DUMP_GSO=<file>: the reference's dump goes into the fixture as "gso_dump") and BKZ_MAX_TIME
# with max_time = 0 (flag 0x8: RED_BKZ_TIME_LIMIT in front of the first tour)
REFDRV_DUMP_GSO=$T/d1.json $D bkzfix q 40 20 20 1 10 0 > $G/bkzx_q40_b10_dump.json
REFDRV_BKZ_FLAGS=0x8 $D bkzfix q 40 20 20 1 10 0 > $G/bkzx_q40_b10_time0.json
REFDRV_DUMP_GSO=$T/d2.json REFDRV_BKZ_FLAGS=0x100 $D bkzfix q 40 20 20 1 10 3 > $G/bkzx_q40_b10_sd_loops3_dump.json
REFDRV_DUMP_GSO=$T/d3.json REFDRV_BKZ_FLAGS=0x200 $D bkzfix q 40 20 20 1 10 0 > $G/bkzx_q40_b10_slide_dump.json
REFDRV_DUMP_GSO=$T/d4.json REFDRV_STRATEGIES=$T/stratA.json REFDRV_BKZ_FLAGS=0x80 REFDRV_RNG_SEED=5 $D bkzfix q 64 32 14 3 40 2 > $G/bkzx_q64_b40_pre_gh_dump.json
# three nested tours (40 -> 30 -> 20) with expectations scaled to 0.8x
python3 $G/make_strategies.py $T/gen.json $T/stratC.json 40 10 30 0.8
REFDRV_STRATEGIES=$T/stratC.json REFDRV_BKZ_FLAGS=0x80 REFDRV_RNG_SEED=21 $D bkzfix q 56 28 12 5 40 1 > $G/bkzs_q56_b40_nested3.json
# the strategies the reference's own tests/test_bkz.cpp builds (test_bkz_param, _linear_pruning); its
# test lattices themselves (dim55_in, example_in, 1000-bit intrel) do not fit ZT = long
$D teststrat 20 0 > $T/p.json
$D teststrat 20 1 > $T/l.json
REFDRV_STRATEGIES=$T/p.json REFDRV_RNG_SEED=3 $D bkzfix q 50 25 12 2 20 0 > $G/bkzs_q50_b20_teststrat_param.json
REFDRV_STRATEGIES=$T/l.json REFDRV_RNG_SEED=3 $D bkzfix q 50 25 12 2 20 0 > $G/bkzs_q50_b20_teststrat_linear.json
REFDRV_STRATEGIES=$T/l.json REFDRV_RNG_SEED=3 REFDRV_BKZ_AUTO_ABORT=1 $D bkzfix r 40 0 45 5 20 0 > $G/bkzs_r40_b20_teststrat_linear_autoabort.json
rm -rf $T
# config 2 at full size (4 s)
$D bkzfix q 120 60 20 0 20 0 | gzip -9 > $G/c2_bkz20_q120.json.gz
# --- C3 (BASELINE configs[2]): the 180-dim q-ary lattice, LLL + BKZ-20 by the reference, its
#     beta=60 blocks as the plugin sees them, and pruner-generated strategies for the tour bench
$D dumpbasis 180 90 20 0 20 > $G/basis_q180_seed0_lll_bkz20.txt
for k in 0 1 2; do
  f=$((k*1))
  $D enumfix 0 $G/basis_q180_seed0_lll_bkz20.txt 0 0 0 $f 60 prune:0.5 1 0 0.99 > $G/c3_b60_k${k}_pruner.json
done
$D genstrat $G/basis_q180_seed0_lll_bkz20.txt 60 > $G/strategies_q180_b60.json
# config 3 at full size: one BKZ-60 tour of that lattice with those strategies (49 s)
REFDRV_STRATEGIES=$G/strategies_q180_b60.json REFDRV_BKZ_FLAGS=0x80 REFDRV_RNG_SEED=1 $D bkzfix f:$G/basis_q180_seed0_lll_bkz20.txt 180 0 0 0 60 1 | gzip -9 > $G/c3_bkz60_tour_strategies.json.gz
# sub-solutions (findsubsols): the evaluator's final table is added to the fixture
REFDRV_SUBSOLS=1 $D enumfix  80 40 12 1  0 4 32 none      5 0 1.30 > $G/enum_d32_best5_subsols.json
REFDRV_SUBSOLS=1 $D enumfix 100 50 14 2 20 0 40 linear:20 1 0 0.99 > $G/enum_d40_lin20_best1_subsols.json
md5sum $G/enum_*.json > $G/MD5SUMS
# a 200-dimensional LLL-reduced q-ary basis (8-bit q) for the NQ = 4 (more than 192 columns) cases of
# the sweep / LLL / Householder device tests: reference latticegen + `fplll -a lll -m fast -f double`
$R/latticegen -randseed 7 q 200 100 8 p > /tmp/q200.txt && $R/fplll -a lll -m fast -f double /tmp/q200.txt | gzip -9 > $G/basis_q200_seed7_lll.txt.gz
# the bench's blocks with the reference's results (5-10e9 nodes each, ~12 CPU-minutes in total)
for k in 0 1 2; do
  $D enumfix 0 $G/basis_q180_seed0_lll_bkz20.txt 0 0 0 $k 60 linear:30 1 0 0.99 > $G/c3_b60_k${k}_linear30.json
done
# R-factors at 106 bits of MPFR (the precision of dd_real, defs.h:140) of the HLLL-reduced bases of
# hlll_{q40,n64,q72}.json: golden values for the double-double device path (ref_driver hhmp; the
# python wrapper that adds the basis to the JSON is in the commit that introduced these files)
#   $D hhmp <basis.txt> 106  ->  tests/golden/hhmp106_{q40,n64,q72}.json.gz

# ---- pruner fixtures (round 3): prune<FP_NR<double>> of the real reference on blocks of the C3 basis ----
# ref_driver prunefix basisfile first d gh_factor preproc_cost target metric flags
B=basis_q180_seed0_lll_bkz20.txt
$DRV prunefix $B 60 60 1.1 1e7 0.5 0 4   > prune_q180_k60_b60_p05.json             # PRUNER_GRADIENT
$DRV prunefix $B 10 45 1.2 1e7 0.3 0 4   > prune_q180_k10_b45_p03.json             # odd block size
$DRV prunefix $B 60 50 1.1 1e7 1.0 1 68  > prune_q180_k60_b50_single_expsol.json   # PRUNER_SINGLE, expected solutions
$DRV prunefix $B 100 31 1.0 1e5 0.9 0 36 > prune_q180_k100_b31_half.json           # PRUNER_HALF
$DRV prunefix $B 60 40 1.1 1e7 0.5 0 12  > prune_q180_k60_b40_zealous.json         # PRUNER_ZEALOUS = GRADIENT | NELDER_MEAD
$DRV prunefix $B 20 30 1.0 1e6 0.7 0 8   > prune_q180_k20_b30_neldermead.json      # PRUNER_NELDER_MEAD alone
$DRV prunefix $B 90 37 1.2 1e6 1.5 1 12  > prune_q180_k90_b37_zealous_expsol.json  # odd size, expected solutions
# ref_driver prunemulti basisfile first d count stride gh_factor preproc_cost target metric flags
$DRV prunemulti $B 30 40 3 20 1.1 1e7 0.5 0 4  > prunemulti_q180_k30_b40_x3.json   # three bases, gradient
$DRV prunemulti $B 10 33 2 50 1.0 1e6 0.6 0 12 > prunemulti_q180_k10_b33_x2_zealous.json

# ---- strategies loader (round 3): what load_strategies_json holds after reading a file ----
$DRV stratdump strategies_q180_b60.json > stratdump_q180_b60.json
$DRV stratdump strategies_handmade.json > stratdump_handmade.json   # (strategies_handmade.json is written by hand)

# ---- GSO host utilities (round 3): get_current_slope / get_log_det / get_root_det / get_slide_potential /
# adjust_radius_to_gh_bound of the reference on the stored r diagonal of MatGSO<long,double>
$DRV gsoutil basis_q180_seed0_lll_bkz20.txt > gsoutil_q180.json
# (40-dim bases written from bkz_q40_b10.json's b_in / reversed rows / b_out: basis_q40_*.txt; their dumps carry
#  the stored mu / r matrices and the reference's is_lll_reduced verdicts)
for f in q40_lll q40_lll_rows_reversed q40_bkz10; do $DRV gsoutil basis_$f.txt > gsoutil_$f.json; done
