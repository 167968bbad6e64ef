"""Pins oracle/gso_oracle.c against the REAL reference: MatGSO<Z_NR<long>,FP_NR<double>> with
GSO_ROW_EXPO — update_gso() and LLLReduction::size_reduction(0,d) — on seeded q-ary bases that were
LLL-reduced and then un-size-reduced by random row operations (tests/golden/gso_*.json, generated
by oracle/ref_driver.cpp `gsofix`).  Everything must be bit-identical: the integer basis, mu, r
(stored, scaled values) and the row exponents."""
import os

import numpy as np
import pytest

import conftest as C


@pytest.mark.parametrize("path", C.gso_fixtures(), ids=lambda p: os.path.basename(p)[:-5])
def test_gso_oracle_matches_reference(path):
    f = C.load_gso_fixture(path)
    g = C.OracleGSO(f["b_in"])
    assert g.update_all() == 1
    assert np.array_equal(g.row_expo, f["row_expo0"])
    assert np.array_equal(g.mu, f["mu0"])
    assert np.array_equal(g.r, f["r0"])
    rc = g.size_reduction(0, f["d"])
    assert rc == f["status"] == 1
    assert np.array_equal(g.b, f["b_out"])
    assert np.array_equal(g.row_expo, f["row_expo1"])
    assert np.array_equal(g.mu, f["mu1"])
    assert np.array_equal(g.r, f["r1"])
    if "p0" not in f["name"]:
        assert not np.array_equal(f["b_in"], f["b_out"]), "fixture should exercise the integer AXPY"
    g.close()
