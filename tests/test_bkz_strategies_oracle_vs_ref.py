"""Pins oracle/gso_oracle.c::oracle_gso_bkz_param — BKZ WITH strategies — against the REAL reference:
BKZReduction<Z_NR<long>,FP_NR<double>>::bkz() (fplll/bkz.cpp:522-668) driven by a strategies file:
recursive preprocessing tours (svp_preprocessing :100-124), pruning selection by the radius / GH
ratio (get_pruning :82-98, Strategy::get_pruning bkz_param.cpp:64-80), the Gaussian-heuristic
radius bound (BKZ_GH_BND, gso_interface.cpp:220-276), the success-probability loop and
rerandomize_block (:43-80, random numbers from the reference's RandGen via the same libgmp).
Fixtures: tests/golden/bkzs_*.json (oracle/ref_driver.cpp `bkzfix` with REFDRV_STRATEGIES).
The output basis, status and total enumeration node count must be identical.

This is the oracle for the next row of the scope table (BASELINE configs
3-4 run BKZ-60 with strategies); the device path for it does not exist yet."""
import os

import numpy as np
import pytest

import conftest as C


@pytest.mark.parametrize("path", C.bkz_strategy_fixtures(), ids=lambda p: os.path.basename(p)[:-5])
def test_bkz_with_strategies_matches_reference(path):
    f = C.load_bkz_fixture(path)
    g = C.OracleGSO(f["b_in"])
    st, info = g.bkz_param(f["block_size"], f["delta"], f["eta"], f["flags"], f["max_loops"],
                           f["gh_factor"], f["strategies"], f["rng_seed"])
    assert st == f["status"]
    nodes = (int(info[1]) & 0xffffffff) | ((int(info[2]) & 0xffffffff) << 32)
    assert nodes == f["nodes"]
    assert np.array_equal(g.b, f["b_out"])
    assert not np.array_equal(f["b_in"], f["b_out"])
    if "rerand" in f["name"]:
        assert info[4] > 0  # the fixture does exercise rerandomize_block
    g.close()


def test_fixtures_cover_the_strategy_features():
    names = [os.path.basename(p) for p in C.bkz_strategy_fixtures()]
    assert len(names) >= 5
    feats = set()
    for p in C.bkz_strategy_fixtures():
        f = C.load_bkz_fixture(p)
        S = f["strategies"]
        if any(S["pre_off"][b + 1] > S["pre_off"][b] for b in range(f["block_size"] + 1)):
            feats.add("preprocessing")
        if f["flags"] & 0x80:
            feats.add("gh_bnd")
        if f["flags"] & 0x10:
            feats.add("bounded_lll")
        if f["flags"] & 0x20:
            feats.add("auto_abort")
        if len(S["coeff"]) > 0:
            feats.add("pruning")
    assert feats == {"preprocessing", "gh_bnd", "bounded_lll", "auto_abort", "pruning"}


def test_config3_bkz60_tour_at_full_size():
    """BASELINE config 3 at its full size: ONE BKZ-60 tour (BKZ_MAX_LOOPS 1, BKZ_GH_BND 1.1) of the
    180-dimensional q-ary lattice (LLL + BKZ-20 by the reference first) with the pruner-generated
    strategies of tests/golden/strategies_q180_b60.json — 15 160 enumerations, 1.22e9 nodes, 11
    rerandomisations, 49 s in the reference.  The oracle must end on the reference's basis and node
    count (~55 s on one core; the device run of the same golden is tests/perf/bkzs_c3.py)."""
    f = C.load_bkz_fixture(os.path.join(C.GOLDEN, "c3_bkz60_tour_strategies.json.gz"))
    assert (f["d"], f["block_size"], f["max_loops"], f["flags"]) == (180, 60, 1, 0x84)
    g = C.OracleGSO(f["b_in"])
    st, info = g.bkz_param(f["block_size"], f["delta"], f["eta"], f["flags"], f["max_loops"],
                           f["gh_factor"], f["strategies"], f["rng_seed"])
    nodes = (int(info[1]) & 0xffffffff) | ((int(info[2]) & 0xffffffff) << 32)
    assert st == f["status"] == 8
    assert nodes == f["nodes"] == 1224293770
    assert info[3] == 15160 and info[4] == 11
    assert np.array_equal(g.b, f["b_out"])
    g.close()
