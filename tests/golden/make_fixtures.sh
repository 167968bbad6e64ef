#!/bin/bash
# Regenerates tests/golden/enum_*.json with the REAL reference (oracle/_ref/ref_driver, built by
# `make -C oracle ref` from /root/reference).  Inputs: seeded q-ary lattices (gen_qary_prime),
# LLL-reduced (+ optional BKZ-20), GSO with row exponents; the fixture holds exactly what fplll
# hands an external enumerator (mu^T, rdiag, pruning, maxdist) and what fplll's own enumerator
# returned (per-level node counts, every eval_sol call, final bound).
set -e
cd "$(dirname "$0")/../.."
D=oracle/_ref/ref_driver
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
md5sum $G/enum_*.json > $G/MD5SUMS
