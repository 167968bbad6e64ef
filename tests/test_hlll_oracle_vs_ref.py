"""Pins oracle/hh_oracle.c::oracle_hlll against the REAL reference:
HLLLReduction<Z_NR<long>,FP_NR<double>>::hlll() over MatHouseholder(HOUSEHOLDER_ROW_EXPO |
HOUSEHOLDER_OP_FORCE_LONG) — fplll/hlll.cpp:26-499, householder.cpp — on tests/golden/hlll_*.json
(oracle/ref_driver.cpp `hlllfix`): q-ary, knapsack, uniform and NTRU-like bases.  The reduced basis
and the status must be identical."""
import os

import numpy as np
import pytest

import conftest as C


@pytest.mark.parametrize("path", C.hlll_fixtures(), ids=lambda p: os.path.basename(p)[:-5])
def test_hlll_oracle_matches_reference(path):
    f = C.load_hlll_fixture(path)
    st, out, info = C.oracle_hlll(f["b_in"], f["delta"], f["eta"], f["theta"], f["c"])
    assert st == f["status"] == 1
    assert info[0] > 0
    assert np.array_equal(out, f["b_out"])
    assert not np.array_equal(f["b_in"], f["b_out"])


def test_config5_size_ntru256_double():
    """BASELINE config 5's lattice family at its full size — a 256-dimensional NTRU-like basis
    (latticegen n 256 10: [[I, Rot(h)], [0, qI]]) — with FT = double, which is what this repo's HLLL
    path computes in (the reference's dd_real needs libqd, absent here: SURVEY.md §8(c)); the
    reference's double HLLL succeeds on it (146 491 swaps, 12 s).  ~11 s on one core."""
    f = C.load_hlll_fixture(os.path.join(C.GOLDEN, "c5_hlll_n256_double.json.gz"))
    assert (f["d"], f["n"]) == (256, 256)
    st, out, info = C.oracle_hlll(f["b_in"], f["delta"], f["eta"], f["theta"], f["c"])
    assert st == f["status"] == 1
    assert info[0] == 146491
    assert np.array_equal(out, f["b_out"])
