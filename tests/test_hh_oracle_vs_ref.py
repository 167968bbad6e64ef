"""Pins oracle/hh_oracle.c against the REAL reference: MatHouseholder<Z_NR<long>,FP_NR<double>>
refresh_R_bf() + update_R() (tests/golden/hh_*.json from oracle/ref_driver.cpp `hhfix`): the lower
triangle of R (incl. diagonal) and the row exponents must be bit-identical."""
import os

import numpy as np
import pytest

import conftest as C


@pytest.mark.parametrize("path", C.hh_fixtures(), ids=lambda p: os.path.basename(p)[:-5])
def test_hh_oracle_matches_reference(path):
    f = C.load_hh_fixture(path)
    R, V, sigma, rexp = C.oracle_hh_update_all(f["b_in"], f["row_expo_on"])
    assert np.array_equal(rexp, f["row_expo"])
    assert np.array_equal(np.tril(R[:, :f["d"]]), f["R"])
    assert np.all(np.diag(R) >= 0)  # R_ii > 0 (tests/test_gso.cpp:82-152 checks the same)


@pytest.mark.parametrize("path", C.hhsr_fixtures(), ids=lambda p: os.path.basename(p)[:-5])
def test_hh_size_reduce_oracle_matches_reference(path):
    """MatHouseholder::size_reduce(kappa, end, start) (householder.cpp:402-451) on the state update_R() left: the
    flag, row kappa of the basis and R(kappa, 0..kappa) afterwards are the reference's, bit for bit (`hhsr`
    fixtures: the whole range, a sub-range, row exponents on, a row that needs nothing)."""
    f = C.load_hhsr_fixture(path)
    flag, b, R, rexp = C.oracle_hh_size_reduce(f["b_in"], f["row_expo_on"], f["kappa"], f["end"], f["start"])
    k = f["kappa"]
    assert flag == f["reduced"]
    assert np.array_equal(b[k], f["b_row"])
    others = np.arange(f["d"]) != k
    assert np.array_equal(b[others], f["b_in"][others])
    assert np.array_equal(R[k, :k + 1], f["R_row"][:k + 1])
    assert np.array_equal(rexp, f["row_expo"])


def test_hhsr_fixtures_cover_the_cases():
    fs = [C.load_hhsr_fixture(p) for p in C.hhsr_fixtures()]
    assert len(fs) >= 4
    assert any(f["reduced"] == 0 for f in fs) and any(f["reduced"] == 1 for f in fs)
    assert any(f["row_expo_on"] for f in fs) and any(f["start"] > 0 and f["end"] < f["kappa"] for f in fs)
